"""The second restatement, widened (round 5): everything AROUND mcts_single, transliterated from the Julia text alone and run WITHOUT the
C oracle under it — the game plugins come from tests/ref_games.py, the search from tests/ref_transliteration.py (RefTree over the
RefGames adapter), and this file adds

  * snetwork2's forward, GPU method (DenseNet.jl:294-304: relu.(base*x); b .= relu.(b .+ relu.(w*b)); policy*b .+ policy_bias;
    σ.(value*b .+ value_bias)) + softmax! (mcts_gpu.jl:417), in numpy Float32;
  * the self-play loop mcts(actor, visits, ngames, buffer) (mcts_gpu.jl:477-579) with decode (:464-474) and the PoolSample of
    mainGobang.jl:34-82 (Sample, push_buffer, update_buffer);
  * the two-actor loop mcts(actor1, actor2, visits, ngames) (:581-651) and duelnetwork (:653-668).

tests/test_ref_selfplay.py requires bit-equality with oracle/agz_oracle.c (agzo_selfplay, agzo_duel, agzo_forward) on the golden
self-play fixture, a golden duel fixture and further generations of every game: with this, no part of the oracle rests on a single
reading of the Julia.

What is DEFINED here rather than transliterated — the reference leaves it to third-party code whose arithmetic is unspecified or
unseeded, and oracle and product fix the same definitions:
  * a dot product is the k-ordered chain y = fma(w[k], x[k], y) from 0 (CUBLAS sgemm's order is unspecified); `fma32` below is the
    exact single-rounding fma, built from float64 products (exact for Float32 factors) and an error-free sum;
  * exp (NNlib's softmax! and σ) is the polynomial `expf_spec`; σ(x) = (t = exp(-|x|); x >= 0 ? 1 / (1 + t) : t / (1 + t)) as NNlib writes it;
  * sum(Weights) and the cumulative walk of StatsBase.sample run in Float32 in source order, t = u * sum with the move uniform
    u(seed; game id, ply) = (23 Philox bits + 1/2) 2^-23 (Julia draws an unseeded Float64 rand());
  * the search uniforms (Philox4x32-10, written out below — Random123's published algorithm; checked against its known answers).
"""
import numpy as np

import ref_games as RG
import ref_transliteration as RT

F32 = np.float32
F64 = np.float64
M32 = 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------- Philox4x32-10 (Random123)
def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = [int(x) & M32 for x in ctr]
    k0, k1 = [int(x) & M32 for x in key]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def uniform_search(seed, game_id, step, rollout, depth):
    """(0, 1]: 24 bits + 1; one block serves four consecutive depths"""
    o = philox4x32_10((game_id, step, rollout, depth >> 2), (seed & M32, (seed >> 32) & M32))
    return F32((o[depth & 3] >> 8) + 1) * F32(2.0 ** -24)


def uniform_move(seed, game_id, step):
    """(0, 1): the odd multiples of 2^-24"""
    o = philox4x32_10((game_id, step, 0, 0x80000000), (seed & M32, (seed >> 32) & M32))
    return F32(2 * (o[0] >> 9) + 1) * F32(2.0 ** -24)


# ---------------------------------------------------------------------------------------------------- Float32 arithmetic
def fma32(a, b, c):
    """round_to_Float32(a * b + c) with ONE rounding, elementwise (Float32 arrays).  a * b is exact in float64 (24 + 24 bits); the
    float64 sum s = p + c is corrected with its exact error e (TwoSum) where s falls exactly half way between two Float32 numbers."""
    p = a.astype(F64) * b.astype(F64)
    c = c.astype(F64)
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)                                   # s + e == p + c exactly
    r = s.astype(F32)                                               # round-to-nearest-even of s
    # s + e rounds like s unless s lies EXACTLY half way between two neighbouring Float32 numbers lo < s < hi (both are float64 numbers,
    # and no float64 lies between s and that point otherwise): then the sign of e decides
    rd = r.astype(F64)
    lo = np.where(rd <= s, r, np.nextafter(r, F32(-np.inf)))
    hi = np.nextafter(lo, F32(np.inf))
    tie = np.isfinite(r) & (e != 0) & (rd != s) & ((s - lo.astype(F64)) == (hi.astype(F64) - s))
    if tie.any():
        r = np.where(tie, np.where(e > 0, hi, lo), r)
    return r.astype(F32)


def expf_spec(x):
    """the exp of this build (oracle agzo_expf, device exp_spec): range reduction by ln 2 in two pieces, degree-7 polynomial, exact scaling"""
    x = np.asarray(x, F32)
    kf = np.rint(x * F32(1.44269504088896341)).astype(F32)
    r = fma32(kf, np.full_like(kf, F32(-0.693145751953125)), x)
    r = fma32(kf, np.full_like(kf, F32(-1.42860682030941723212e-6)), r)
    z = r * r
    p = np.full_like(r, F32(1.9875691500e-4))
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        p = fma32(p, r, np.full_like(r, F32(c)))
    y = fma32(p, z, r)
    y = y + F32(1)
    k = kf.astype(np.int64)
    big = k >= -126
    s1 = np.where(big, k + 127, k + 127 + 64).astype(np.uint32) << np.uint32(23)
    out = y * s1.view(F32)
    out = np.where(big, out, out * F32(5.42101086242752217e-20))
    out = np.where(x < F32(-104.0), F32(0), out)
    return np.where(x > F32(88.5), F32(np.inf), out).astype(F32)


def sigmoid(x):                                                     # NNlib σ
    t = expf_spec(-np.abs(x))
    return np.where(x >= 0, F32(1) / (F32(1) + t), t / (F32(1) + t)).astype(F32)


def softmax_(x):
    """softmax!(prior) (mcts_gpu.jl:417) per column: exp(x - max) / sum, the sum in source order"""
    m = x.max(axis=1, keepdims=True)
    ex = expf_spec(x - m)
    s = np.zeros(x.shape[0], F32)
    for j in range(x.shape[1]):
        s = s + ex[:, j]
    return (ex / s[:, None]).astype(F32)


class snetwork2:
    """mutable struct snetwork2 (DenseNet.jl:279-286); weights as Flux stores them: W of Dense(in, out) is (out, in), column-major"""

    def __init__(self, base, res, policy, policy_bias, value, value_bias):
        self.base, self.res, self.policy, self.policy_bias, self.value, self.value_bias = base, res, policy, policy_bias, value, value_bias

    @staticmethod
    def from_flat(inp, H, T, A, W0, Wres, Wp, bp, Wv, bv):
        col = lambda w, o, i: np.asarray(w, F32)[: o * i].reshape(i, o).T            # noqa: E731  (out, in) from column-major memory
        return snetwork2(col(W0, H, inp), [col(np.asarray(Wres, F32)[t * H * H:], H, H) for t in range(T)], col(Wp, A, H), np.asarray(bp, F32),
                         col(Wv, 1, H), np.asarray(bv, F32))

    @staticmethod
    def matmul(W, x):
        """W * x with the defined accumulation order: y = fma(W[:, k], x[k, :], y), k ascending, from 0.  x: (in, L) -> (out, L)"""
        y = np.zeros((W.shape[0], x.shape[1]), F32)
        for k in range(W.shape[1]):
            y = fma32(np.broadcast_to(W[:, k:k + 1], y.shape), np.broadcast_to(x[k:k + 1, :], y.shape), y)
        return y

    def __call__(self, x):
        """(m::snetwork2)(x::CuArray) :294-304.  x: (in, L) Float32 -> (policy (A, L) logits, value (1, L))"""
        relu = lambda a: np.where(a > 0, a, F32(0)).astype(F32)                       # noqa: E731
        b = relu(self.matmul(self.base, x))
        for w in self.res:
            b = relu(b + relu(self.matmul(w, b)))
        policy = self.matmul(self.policy, b) + self.policy_bias[:, None]
        value = sigmoid(self.matmul(self.value, b) + self.value_bias[:, None])
        return policy.astype(F32), value.astype(F32)

    def actor(self, planes):
        """what mcts_single does with the actor (:414-417): planes [L][2 VS] -> (softmaxed priors [L][A], v [L])"""
        policy, value = self(np.ascontiguousarray(np.asarray(planes, F32).T))
        return softmax_(policy.T.copy()), value[0].copy()


# ---------------------------------------------------------------------------------------------------- mainGobang.jl:34-82
class Sample:
    def __init__(self, VS, A, FS):
        self.state, self.policy, self.player, self.value, self.fstate = np.zeros(2 * VS, np.int8), np.zeros(A, F32), 1, F32(0), np.zeros(FS, np.int8)


class PoolSample:
    def __init__(self, N, VS, A, FS):
        self.length, self.currentIndex, self.pool, self.full = N, 1, [Sample(VS, A, FS) for _ in range(N)], False

    def push_buffer(self, state, policy, player, i):                # :54-68   state (2VS, L), policy (A, L): column i (1-based)
        index = self.currentIndex
        self.pool[index - 1].state[:] = state[:, i - 1].astype(np.int8)
        self.pool[index - 1].policy[:] = policy[:, i - 1]
        self.pool[index - 1].player = player
        newindex = 1 if index == self.length else index + 1
        if newindex == 1:
            self.full = True
        self.currentIndex = newindex
        return index

    def update_buffer(self, index, result, fstate):                 # :70-80
        for id_ in index:
            player = self.pool[id_ - 1].player
            self.pool[id_ - 1].value = F32((1 + result * player) / 2)
            self.pool[id_ - 1].fstate[:] = (fstate * player).astype(np.int8)


def decode(rg, pos):                                                # mcts_gpu.jl:464-474
    fstate = np.zeros(rg.VectorizedState, np.int8)
    for j in range(1, rg.VectorizedState + 1):
        fstate[j - 1] = pos.player if RG.getindex(pos.bplayer, j) else -pos.player
    return fstate


def sample_weights(items, w, u):
    """StatsBase.sample(items, Weights(w)): t = rand() * sum(w); i = 1; cw = w[1]; while cw < t && i < n: i += 1; cw += w[i]"""
    total = F32(0)
    for x in w:
        total = total + x
    t = u * total
    i, cw = 1, w[0]
    while cw < t and i < len(w):
        i += 1
        cw = cw + w[i - 1]
    return items[i - 1]


def argmax1(v):                                                     # Julia argmax: the first maximum, 1-based
    return int(np.argmax(v)) + 1


# ---------------------------------------------------------------------------------------------------- mcts_gpu.jl:477-579
def mcts(actor, visits, ngames, buffer, rg, cpuct=2.0, seed=1, game_id_base=0):
    """-> dict(valid, v, n, d, tot_length, order): `order` = for every pushed sample (index into the buffer, game id, round, move)"""
    gm = RT.RefGames(rg)
    positions = [rg.start() for _ in range(ngames)]
    ids = [game_id_base + k for k in range(ngames)]
    rtemp = [[] for _ in range(ngames)]
    round_, v, d, n, tot_length = 0, 0, 0, 0, 0
    order = []
    while positions:
        tree = RT.RefTree(gm, positions, visits, ids)               # (init once + re_init per round in the reference: all statistics are reset by mcts_single)
        tree.mcts_single(actor, visits, training=True, cpuct=cpuct, seed=seed, step=round_)
        policy, batch = tree.policy_final[1:, 1:], tree.root_batch.T     # Array(vnodesStats.policy_final) (A, L), Array(vnodesStats.batch) (2VS, L)
        finished = []
        for i in range(1, len(positions) + 1):
            index = buffer.push_buffer(batch, policy, positions[i - 1].player, i)
            rtemp[i - 1].append(index)
            pol = buffer.pool[index - 1].policy
            if round_ < 25:
                lp = [c for c in range(1, rg.maxActions + 1) if pol[c - 1] != 0]
                c = sample_weights(lp, [pol[k - 1] for k in lp], uniform_move(seed, ids[i - 1], round_))
            else:
                c = argmax1(pol)
            order.append((index, ids[i - 1], round_, c - 1))
            if not rg.canPlay(positions[i - 1], c):
                return dict(valid=False, order=order)
            positions[i - 1] = rg.play(positions[i - 1], c)
            f, res = rg.isOver(positions[i - 1])
            if f:
                fstate = decode(rg, positions[i - 1])
                finished.append(i)
                tot_length += round_
                buffer.update_buffer(rtemp[i - 1], res, fstate)
                if res == 1:
                    v += 1
                elif res == 0:
                    n += 1
                else:
                    d += 1
        for k, c in enumerate(finished, start=1):
            del rtemp[c - k]
            del positions[c - k]
            del ids[c - k]
        round_ += 1
    return dict(valid=True, v=v, n=n, d=d, tot_length=tot_length, order=order)


# ---------------------------------------------------------------------------------------------------- mcts_gpu.jl:581-651, :653-668
def mcts_duel(actor1, actor2, visits, ngames, rg, cpuct=2.0, seed=1, game_id_base=0, tau_rounds=15):
    """mcts(actor1, actor2, visits, ngames; cpuct = 2f0) -> ([v, n, d], moves per game)"""
    gm = RT.RefGames(rg)
    positions = [rg.start() for _ in range(ngames)]
    ids = [game_id_base + k for k in range(ngames)]
    moves = {g: [] for g in ids}
    round_, v, d, n = 0, 0, 0, 0
    while positions:
        actor = actor1 if round_ % 2 == 0 else actor2
        finished = []
        tree = RT.RefTree(gm, positions, visits, ids)
        tree.mcts_single(actor, visits, training=False, cpuct=cpuct, seed=seed, step=round_)
        policy = tree.policy_final[1:, 1:]
        for i in range(1, len(positions) + 1):
            if round_ < tau_rounds:
                c = sample_weights(list(range(1, rg.maxActions + 1)), list(policy[:, i - 1]), uniform_move(seed, ids[i - 1], round_))
            else:
                c = argmax1(policy[:, i - 1])
            moves[ids[i - 1]].append(c - 1)
            if not rg.canPlay(positions[i - 1], c):
                return None, moves
            positions[i - 1] = rg.play(positions[i - 1], c)
            f, res = rg.isOver(positions[i - 1])
            if f:
                finished.append(i)
                if res == 1:
                    v += 1
                elif res == 0:
                    n += 1
                else:
                    d += 1
        for k, c in enumerate(finished, start=1):
            del positions[c - k]
            del ids[c - k]
        round_ += 1
    return [v, n, d], moves


def duelnetwork(actor1, actor2, visits, ngames, rg, seed=1):
    hngames = ngames // 2
    (v1, n1, d1), _ = mcts_duel(actor1, actor2, visits, hngames, rg, seed=seed)
    (d2, n2, v2), _ = mcts_duel(actor2, actor1, visits, hngames, rg, seed=seed + 1)
    return v1 + v2, n1 + n2, d1 + d2


# the search of tests/ref_transliteration.py draws its uniforms from THIS file's Philox when it runs under the loops above (no oracle call)
RT.uniform = uniform_search
