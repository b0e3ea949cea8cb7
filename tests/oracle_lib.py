"""ctypes binding of the CPU oracle (oracle/libagz_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (alphagpu_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

GOBANG, CONNECT4, HEX, REVERSI8, REVERSI6 = 0, 1, 2, 3, 4
KIND = {"gobang": GOBANG, "connect4": CONNECT4, "hex": HEX, "reversi8": REVERSI8, "reversi6": REVERSI6}


class BB(C.Structure):
    _fields_ = [("c", C.c_uint64 * 3)]


class Pos(C.Structure):
    _fields_ = [("bplayer", BB), ("bopponent", BB), ("legalplay", BB),
                ("player", C.c_int8), ("aux", C.c_int8), ("pad", C.c_int8 * 6)]


class Game(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("kind", "n", "nvict", "d1", "d2", "len", "A", "VS", "FS", "ML")]


class Net(C.Structure):
    _fields_ = [("inp", C.c_int), ("H", C.c_int), ("T", C.c_int), ("A", C.c_int),
                ("W0", C.c_void_p), ("Wres", C.c_void_p), ("Wp", C.c_void_p), ("bp", C.c_void_p),
                ("Wv", C.c_void_p), ("bv", C.c_void_p), ("bf16", C.c_void_p)]


class Samples(C.Structure):
    _fields_ = [("nsamples", C.c_long), ("capacity", C.c_long),
                ("A", C.c_int), ("VS", C.c_int), ("FS", C.c_int),
                ("state", C.c_void_p), ("policy", C.c_void_p), ("player", C.c_void_p),
                ("value", C.c_void_p), ("fstate", C.c_void_p), ("game_id", C.c_void_p),
                ("ply", C.c_void_p), ("move", C.c_void_p),
                ("wins", C.c_long), ("draws", C.c_long), ("losses", C.c_long),
                ("total_plies", C.c_long), ("faults", C.c_long)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libagz_oracle.so")
        src = os.path.join(ORACLE_DIR, "agz_oracle.c")
        if os.environ.get("AGZ_ORACLE_SAN"):
            # the ASan + UBSan build of the oracle (make -C oracle SAN=1; run the CPU suite with
            # LD_PRELOAD=$(gcc -print-file-name=libasan.so) AGZ_ORACLE_SAN=1, see oracle/Makefile)
            path = os.path.join(ORACLE_DIR, "libagz_oracle_san.so")
            if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "SAN=1"])
        elif not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        L.agzo_uniform_search.restype = C.c_float
        L.agzo_uniform_search.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.agzo_uniform_move.restype = C.c_float
        L.agzo_uniform_move.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.agzo_expf.restype = C.c_float
        L.agzo_expf.argtypes = [C.c_float]
        L.agzo_tree_create.restype = C.c_void_p
        L.agzo_tree_create.argtypes = [C.POINTER(Game), C.c_int, C.c_int]
        L.agzo_tree_destroy.argtypes = [C.c_void_p]
        L.agzo_tree_set_roots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.agzo_search_reset.argtypes = [C.c_void_p]
        L.agzo_select.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float]
        L.agzo_set_reference_keying.argtypes = [C.c_int]
        L.agzo_encode_leaves.argtypes = [C.c_void_p, C.c_void_p]
        L.agzo_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32]
        L.agzo_backup.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.agzo_search.argtypes = [C.c_void_p, C.POINTER(Net), C.c_int, C.c_float, C.c_int, C.c_uint64,
                                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for n in ("policy", "root_planes", "root_visits", "root_q", "leaf", "newindex", "root_policy_row"):
            getattr(L, "agzo_get_" + n).argtypes = [C.c_void_p, C.c_void_p]
        L.agzo_get_counters.restype = C.c_long
        L.agzo_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.agzo_samples_create.restype = C.POINTER(Samples)
        L.agzo_samples_create.argtypes = [C.POINTER(Game), C.c_long]
        L.agzo_samples_destroy.argtypes = [C.POINTER(Samples)]
        L.agzo_selfplay.argtypes = [C.POINTER(Game), C.POINTER(Net), C.c_int, C.c_int, C.c_float, C.c_int,
                                    C.c_uint64, C.c_uint32, C.POINTER(Samples)]
        L.agzo_duel.argtypes = [C.POINTER(Game), C.POINTER(Net), C.POINTER(Net), C.c_int, C.c_int, C.c_float, C.c_int,
                                C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.agzo_fmcts.argtypes = [C.POINTER(Game), C.POINTER(Net), C.POINTER(Pos), C.c_int, C.c_float,
                                 C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.agzo_fmcts_selfplay.restype = C.c_long
        L.agzo_fmcts_selfplay.argtypes = [C.POINTER(Game), C.POINTER(Net), C.c_int, C.c_int, C.c_float,
                                          C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.agzo_init_weights.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.agzo_forward.argtypes = [C.POINTER(Net), C.c_void_p, C.c_void_p, C.c_void_p]
        L.agzo_softmax.argtypes = [C.c_void_p, C.c_int]
        L.agzo_mfma_dot.restype = C.c_float
        L.agzo_mfma_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.agzo_net_bf16_create.restype = C.c_void_p
        L.agzo_net_bf16_create.argtypes = [C.POINTER(Net)]
        L.agzo_net_bf16_destroy.argtypes = [C.c_void_p]
        L.agzo_forward_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.agzo_exp2_spec.restype = C.c_float
        L.agzo_exp2_spec.argtypes = [C.c_float]
        L.agzo_softmax_bf16mode.argtypes = [C.c_void_p, C.c_int]
        L.agzo_encode.argtypes = [C.POINTER(Game), C.POINTER(Pos), C.c_void_p]
        L.agzo_pos_to_image.argtypes = [C.POINTER(Game), C.POINTER(Pos), C.c_void_p]
        L.agzo_pos_from_image.argtypes = [C.POINTER(Game), C.c_void_p, C.POINTER(Pos)]
        L.agzo_pos_image_bytes.argtypes = [C.POINTER(Game)]
        L.agzo_bb_shift.argtypes = [C.POINTER(Game), C.POINTER(BB), C.c_int, C.POINTER(BB)]
        L.agzo_perft.restype = C.c_long
        L.agzo_perft.argtypes = [C.POINTER(Game), C.POINTER(Pos), C.c_int, C.c_void_p]
        L.agzo_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_game(kind, n=0, nvict=0):
    g = Game()
    if isinstance(kind, str):
        kind = KIND[kind]
    rc = lib().agzo_game_init(C.byref(g), kind, n, nvict)
    if rc != 0:
        raise ValueError("bad game parameters")
    return g


def pos_init(g):
    p = Pos()
    lib().agzo_pos_init(C.byref(g), C.byref(p))
    return p


def can_play(g, p, a):
    return bool(lib().agzo_can_play(C.byref(g), C.byref(p), a))


def play(g, p, a):
    o = Pos()
    lib().agzo_play(C.byref(g), C.byref(p), a, C.byref(o))
    return o


def is_over(g, p):
    r = C.c_int(0)
    f = lib().agzo_is_over(C.byref(g), C.byref(p), C.byref(r))
    return bool(f), r.value


def perft(g, p, depth):
    term = np.zeros(3, np.int64)
    n = lib().agzo_perft(C.byref(g), C.byref(p), depth, _p(term))
    return n, term


def bb_bits(bb, n):
    return [(bb.c[i >> 6] >> (i & 63)) & 1 for i in range(n)]


def pos_image(g, p):
    n = lib().agzo_pos_image_bytes(C.byref(g))
    buf = np.zeros(n, dtype=np.uint8)
    lib().agzo_pos_to_image(C.byref(g), C.byref(p), _p(buf))
    return buf


class OracleNet:
    """Random-init snetwork2 (Flux glorot_uniform, zero bias) in Flux (out,in) column-major layout."""

    def __init__(self, g, H, T, seed=0x5EED):
        self.inp, self.H, self.T, self.A = 2 * g.VS, H, T, g.A
        self.W0 = np.zeros(H * self.inp, np.float32)
        self.Wres = np.zeros(max(T, 1) * H * H, np.float32)
        self.Wp = np.zeros(self.A * H, np.float32)
        self.bp = np.zeros(self.A, np.float32)
        self.Wv = np.zeros(H, np.float32)
        self.bv = np.zeros(1, np.float32)
        lib().agzo_init_weights(seed, self.inp, H, T, self.A, _p(self.W0), _p(self.Wres), _p(self.Wp),
                                _p(self.bp), _p(self.Wv), _p(self.bv))
        self._sync()

    def _sync(self):
        self.c = Net(self.inp, self.H, self.T, self.A, _p(self.W0).value, _p(self.Wres).value,
                     _p(self.Wp).value, _p(self.bp).value, _p(self.Wv).value, _p(self.bv).value, None)
        self._prep = None

    def bf16(self):
        """The same network evaluated as the product's bf16 MFMA mode does (bit-level model): a view whose searches,
        self-play and duels use agzo_forward_bf16 + the bf16-mode softmax."""
        o = OracleNet.__new__(OracleNet)
        o.__dict__.update(self.__dict__)
        o._prep = lib().agzo_net_bf16_create(C.byref(self.c))          # (leaked at exit: test infrastructure)
        o.c = Net(self.inp, self.H, self.T, self.A, _p(self.W0).value, _p(self.Wres).value,
                  _p(self.Wp).value, _p(self.bp).value, _p(self.Wv).value, _p(self.bv).value, o._prep)
        return o

    def logits_bf16(self, planes):
        """planes [n][in] -> logits [n][A], v [n] of the bf16 MFMA forward (bit-level model)."""
        prep = self._prep or lib().agzo_net_bf16_create(C.byref(self.c))
        planes = np.ascontiguousarray(planes, np.float32)
        n = planes.shape[0]
        lg = np.zeros((n, self.A), np.float32)
        v = np.zeros(n, np.float32)
        for i in range(n):
            lib().agzo_forward_bf16(prep, _p(planes[i]), _p(lg[i]), C.c_void_p(v.ctypes.data + 4 * i))
        if self._prep is None:
            lib().agzo_net_bf16_destroy(prep)
        return lg, v

    def forward(self, planes):
        """planes [n][in] -> softmaxed priors [n][A], v [n]"""
        planes = np.ascontiguousarray(planes, np.float32)
        n = planes.shape[0]
        pr = np.zeros((n, self.A), np.float32)
        v = np.zeros(n, np.float32)
        for i in range(n):
            lib().agzo_forward(C.byref(self.c), _p(planes[i]), _p(pr[i]), C.c_void_p(v.ctypes.data + 4 * i))
            lib().agzo_softmax(_p(pr[i]), self.A)
        return pr, v

    def logits(self, planes):
        planes = np.ascontiguousarray(planes, np.float32)
        n = planes.shape[0]
        lg = np.zeros((n, self.A), np.float32)
        v = np.zeros(n, np.float32)
        for i in range(n):
            lib().agzo_forward(C.byref(self.c), _p(planes[i]), _p(lg[i]), C.c_void_p(v.ctypes.data + 4 * i))
        return lg, v


class OracleTree:
    """mcts_gpu.jl batched search on the CPU."""

    def __init__(self, g, Lmax, V):
        self.g, self.Lmax, self.V = g, Lmax, V
        self.h = lib().agzo_tree_create(C.byref(g), Lmax, V)
        self.L = 0

    def close(self):
        if self.h:
            lib().agzo_tree_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_roots(self, positions, game_ids=None):
        L = len(positions)
        arr = (Pos * L)(*positions)
        ids = None if game_ids is None else np.ascontiguousarray(game_ids, np.uint32)
        lib().agzo_tree_set_roots(self.h, C.cast(arr, C.c_void_p), _p(ids), L)
        self.L = L

    def search(self, net, V, cpuct, training, seed, step, prior_inject=None, v_inject=None, capture=False):
        A, L = self.g.A, self.L
        pc = np.zeros((V, L, A), np.float32) if capture else None
        vc = np.zeros((V, L), np.float32) if capture else None
        if prior_inject is not None:
            prior_inject = np.ascontiguousarray(prior_inject, np.float32)
            v_inject = np.ascontiguousarray(v_inject, np.float32)
            assert prior_inject.shape == (V, L, A) and v_inject.shape == (V, L)
        lib().agzo_search(self.h, C.byref(net.c) if net is not None else None, V, cpuct, int(training), seed, step,
                          _p(prior_inject), _p(v_inject), _p(pc), _p(vc))
        return pc, vc

    # stepwise
    def reset(self):
        lib().agzo_search_reset(self.h)

    def select(self, seed, step, rollout, cpuct):
        lib().agzo_select(self.h, seed, step, rollout, cpuct)

    def encode_leaves(self):
        out = np.zeros((self.L, 2 * self.g.VS), np.float32)
        lib().agzo_encode_leaves(self.h, _p(out))
        return out

    def expand(self, prior, training, seed, step, rollout):
        prior = np.ascontiguousarray(prior, np.float32)
        lib().agzo_expand(self.h, _p(prior), int(training), seed, step, rollout)

    def backup(self, v, seed, step, rollout):
        v = np.ascontiguousarray(v, np.float32)
        lib().agzo_backup(self.h, _p(v), seed, step, rollout)

    def _get(self, name, shape, dtype=np.float32):
        out = np.zeros(shape, dtype)
        getattr(lib(), "agzo_get_" + name)(self.h, _p(out))
        return out

    def policy(self):
        return self._get("policy", (self.L, self.g.A))

    def root_policy_row(self):
        return self._get("root_policy_row", (self.L, self.g.A))

    def root_planes(self):
        return self._get("root_planes", (self.L, 2 * self.g.VS))

    def root_visits(self):
        return self._get("root_visits", (self.L, self.g.A))

    def root_q(self):
        return self._get("root_q", (self.L, self.g.A))

    def leaf(self):
        return self._get("leaf", (self.L,), np.int32)

    def newindex(self):
        return self._get("newindex", (self.L,), np.int32)

    def counters(self):
        p, n = C.c_long(0), C.c_long(0)
        f = lib().agzo_get_counters(self.h, C.byref(p), C.byref(n))
        return p.value, n.value, f


def selfplay(g, net, ngames, V, cpuct, tau_plies, seed, game_id_base=0, nets=None, tags=None):
    """nets / tags: the actor per (game, ply) — tags[k][p] = index into nets of the network that searches ply p of game k
    (agzo_selfplay_tagged: a chain of calls whose network changes between calls)."""
    cap = ngames * (2 * g.len + 8)
    s = lib().agzo_samples_create(C.byref(g), cap)
    if nets is None:
        rc = lib().agzo_selfplay(C.byref(g), C.byref(net.c), ngames, V, cpuct, tau_plies, seed, game_id_base, s)
    else:
        tags = np.ascontiguousarray(tags, np.uint8)
        assert tags.shape[0] == ngames
        arrp = (C.c_void_p * len(nets))(*[C.addressof(n.c) for n in nets])
        f = lib().agzo_selfplay_tagged
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]
        f.restype = C.c_int
        rc = f(C.byref(g), arrp, len(nets), tags.ctypes.data_as(C.c_void_p), tags.shape[1], ngames, V, cpuct, tau_plies, seed, game_id_base, s)
    sc = s.contents
    n = sc.nsamples

    def arr(ptr, dtype, shape):
        if n == 0:
            return np.zeros(shape, dtype)
        cnt = int(np.prod(shape))
        buf = (C.c_char * (cnt * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=cnt).reshape(shape).copy()

    out = dict(
        rc=rc, n=n,
        state=arr(sc.state, np.int8, (n, 2 * g.VS)), policy=arr(sc.policy, np.float32, (n, g.A)),
        player=arr(sc.player, np.int8, (n,)), value=arr(sc.value, np.float32, (n,)),
        fstate=arr(sc.fstate, np.int8, (n, g.FS)), game_id=arr(sc.game_id, np.uint32, (n,)),
        ply=arr(sc.ply, np.int32, (n,)), move=arr(sc.move, np.int32, (n,)),
        wins=sc.wins, draws=sc.draws, losses=sc.losses, total_plies=sc.total_plies, faults=sc.faults)
    lib().agzo_samples_destroy(s)
    return out


def duel(g, net1, net2, ngames, V, cpuct, tau_plies, seed, game_id_base=0, first=0):
    """mcts(actor1, actor2, visits, ngames; cpuct) (mcts_gpu.jl:581-651) -> dict(rc, wdl [v, n, d], moves [ngames][max_plies]
    (-1 past the end), nplies [ngames])."""
    max_plies = 2 * g.len + 8
    wdl = np.zeros(3, np.int64)
    moves = np.zeros((ngames, max_plies), np.int32)
    nplies = np.zeros(ngames, np.int32)
    rc = lib().agzo_duel(C.byref(g), C.byref(net1.c), C.byref(net2.c), ngames, V, cpuct, tau_plies, seed, game_id_base,
                         first, _p(wdl), _p(moves), max_plies, _p(nplies))
    return dict(rc=rc, wdl=[int(x) for x in wdl], moves=moves, nplies=nplies)


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, as v_cvt_pk_bf16_f32 and the host weight tiling do."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def forward_bf16_model(net, planes):
    """The snetwork2 forward (DenseNet.jl:294-304) with the ROUNDING POINTS of the bf16 MFMA path (agz_nn_wave.hpp /
    agz_nn_big.hpp): weights and layer outputs rounded to bf16, products and sums exact-ish (float64 accumulate, the
    MFMA's fp32 accumulation order is not modelled), residual add and ReLU in fp32, heads in fp32 + bias.
    planes [n][in] (0/1) -> logits [n][A] float64, value pre-activation [n] float64."""
    H, T, A, inp = net.H, net.T, net.A, net.inp
    W0 = bf16_round(net.W0).reshape(inp, H).astype(np.float64)            # Flux (out,in) column-major: W[o + H*i]
    x = np.asarray(planes, np.float64)
    b = np.maximum(x @ W0, 0.0)
    b = bf16_round(b.astype(np.float32)).astype(np.float64)
    for t in range(T):
        W = bf16_round(net.Wres[t * H * H:(t + 1) * H * H]).reshape(H, H).astype(np.float64)
        y = np.maximum((b @ W).astype(np.float32), np.float32(0))           # relu(W b) in fp32
        y = np.maximum(y + b.astype(np.float32), np.float32(0))             # relu(b + .)
        b = bf16_round(y).astype(np.float64)
    Wp = bf16_round(net.Wp).reshape(H, A).astype(np.float64)
    Wv = bf16_round(net.Wv).reshape(H, 1).astype(np.float64)
    logits = b @ Wp + net.bp.astype(np.float64)
    vpre = (b @ Wv)[:, 0] + float(net.bv[0])
    return logits, vpre


def fmcts(g, net, pos, readout, c, seed, game_id=0):
    pol = np.zeros(g.A, np.float32)
    val = np.zeros(1, np.float32)
    lib().agzo_fmcts(C.byref(g), C.byref(net.c), C.byref(pos), readout, c, seed, game_id, _p(pol), _p(val))
    return pol, float(val[0])


def fmcts_selfplay(g, net, ngames, readout, c, tau_plies, seed, threads, max_plies=0):
    return lib().agzo_fmcts_selfplay(C.byref(g), C.byref(net.c), ngames, readout, c, tau_plies, seed, threads,
                                     max_plies)


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().agzo_philox4x32_10(_p(c), _p(k), _p(o))
    return o
