"""A SECOND, independent restatement of the reference's search — test infrastructure, the only pin the oracle can get here.

/root/reference is Julia + CUDA.jl and cannot run in this image (no Julia), so `oracle/agz_oracle.c` cannot be checked against
reference outputs.  This file transliterates kdescendTree! (mcts_gpu.jl:100-199), expand (:250-302), backUp (:306-328),
copy_pol (:330-339), re_init (:359-373) and mcts_single (:376-462) from the Julia text ALONE — same array names, same 1-based
indices, same loop order, Julia's promotion rules spelled out with numpy scalar types (Float32 = np.float32, the Float64
terminal value of backUp = np.float64, Int literals as Python ints) — without looking at the C oracle's data structures.
tests/test_ref_transliteration.py requires that it reproduces the C oracle bit for bit on the golden searches: a transcription
error in either restatement shows up as a difference.

The game plugins and the actor are parameters: the tests of THIS file run the search over the oracle's game functions and forward
(ctypes; `OracleGames`), tests/ref_selfplay.py runs it over their own transliterations (tests/ref_games.py, snetwork2 in numpy;
`RefGames`) — then nothing of the oracle is under it.  The ONE deliberate definition is the randomness: the
reference draws prob = CUDA.rand(maxLengthGame, L) per rollout (:397) and reads prob[cpt, i] at a visit (:178); here the visit
reads the uniform U(seed; game id, step, rollout, depth) that was fixed by the event that last changed the node's row — its
expansion, or the latest backUp through it (`unext`) — which is what oracle and product define.
"""
import numpy as np

import oracle_lib as O

F32 = np.float32
F64 = np.float64


def uniform(seed, game_id, step, rollout, depth):
    """one Philox4x32-10 block serves four consecutive depths; 24 bits + 1 -> (0, 1]"""
    o = O.philox((int(game_id), int(step), int(rollout), int(depth) >> 2), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return F32((int(o[depth & 3]) >> 8) + 1) * F32(5.9604644775390625e-8)


class OracleGames:
    """the game plugins as the C oracle restates them (ctypes): what RefTree used before tests/ref_games.py existed; 1-based actions"""

    def __init__(self, g):
        self.g, self.A, self.VS = g, g.A, g.VS

    def play(self, st, a):
        return O.play(self.g, st, a - 1)

    def isOver(self, st):
        return O.is_over(self.g, st)

    def canPlay(self, st, a):
        return O.can_play(self.g, st, a - 1)

    def planes(self, st):
        return O.bb_bits(st.bplayer, self.VS) + O.bb_bits(st.bopponent, self.VS)


class RefGames:
    """the game plugins as tests/ref_games.py transliterates them from the Julia text: no oracle code under the search at all"""

    def __init__(self, rg):
        import ref_games as RG
        self.RG, self.rg, self.A, self.VS = RG, rg, rg.maxActions, rg.VectorizedState

    def play(self, st, a):
        return self.rg.play(st, a)

    def isOver(self, st):
        return self.rg.isOver(st)

    def canPlay(self, st, a):
        return self.rg.canPlay(st, a)

    def planes(self, st):                                        # decoder (:202-223): bplayer[j], bopponent[j], j = 1 .. VectorizedState
        return [1 if self.RG.getindex(st.bplayer, j) else 0 for j in range(1, self.VS + 1)] + \
               [1 if self.RG.getindex(st.bopponent, j) else 0 for j in range(1, self.VS + 1)]


class RefTree:
    """init(positions, visits) (:350-357): create_cunodes_stats (:35-39) + create_roots (:42-53); every array 1-based like the
    Julia ones (index 0 unused), the game index last as in the column-major originals."""

    def __init__(self, g, positions, visits, game_ids=None):
        """g: an oracle Game (the plugins come from the C oracle) or a RefGames / OracleGames adapter"""
        self.gm = g if hasattr(g, "planes") else OracleGames(g)
        self.maxActions, self.visits_cap = self.gm.A, visits
        L = self.L = len(positions)
        A, V = self.gm.A, visits
        z = lambda *s: np.zeros(s, F32)                          # noqa: E731
        self.q, self.prior, self.policy, self.nvisits = z(A + 1, V + 1, L + 1), z(A + 1, V + 1, L + 1), z(A + 1, V + 1, L + 1), z(A + 1, V + 1, L + 1)
        self.Achild = np.zeros((A + 1, V + 1, L + 1), np.int64)
        self.childID = np.zeros((V + 1, V + 1, L + 1), np.int64)
        self.childnbr = np.zeros((V + 1, L + 1), np.int64)
        self.policy_final = z(A + 1, L + 1)
        self.parent = np.zeros((V + 1, L + 1), np.int64)         # parent[1, i] == 0 (:48)
        self.actionFromParent = np.zeros((V + 1, L + 1), np.int64)
        self.expanded = np.zeros((V + 1, L + 1), np.int8)
        self.uptodate = np.ones((V + 1, L + 1), np.int8)
        self.state = [[None] * (L + 1) for _ in range(V + 1)]
        self.leaf = np.zeros(L + 1, np.int64)
        self.newindex = np.ones(L + 1, np.int64)
        self.game_id = np.arange(L + 1, dtype=np.int64) - 1 if game_ids is None else np.concatenate([[0], np.asarray(game_ids, np.int64)])
        # the randomness definition: uniform of a node's next visit, depth of a node (root 0)
        self.unext = z(V + 1, L + 1)
        self.depth = np.zeros((V + 1, L + 1), np.int64)
        self.re_init(positions)

    def re_init(self, positions):                                # :359-373
        for i in range(1, self.L + 1):
            self.state[1][i] = positions[i - 1]
        self.expanded[:] = 0
        self.uptodate[:] = 1

    # ------------------------------------------------------------------------------------------------ :100-199
    def kdescendTree(self, cpuct):
        maxActions, gm = self.maxActions, self.gm
        cpuct = F32(cpuct)
        for i in range(1, self.L + 1):
            nindex = 1
            cpt = 1
            while self.expanded[nindex, i] == 1:
                bestmove = -1
                pr = 0                                           # Int 0; becomes Float32 at the first += (:112, :174)
                if self.uptodate[nindex, i] != 1:                # :114
                    A = F32(0)
                    n = F32(1)
                    prior_rem = F32(0)
                    childnbr = int(self.childnbr[nindex, i])
                    for k in range(1, maxActions + 1):           # :120-131
                        n = n + self.nvisits[k, nindex, i]
                        if self.Achild[k, nindex, i] == 0:
                            prior_rem = prior_rem + self.prior[k, nindex, i]
                        if self.prior[k, nindex, i] > 0:
                            A = A + F32(1)
                    lam = cpuct * np.sqrt(n) / (A + n)           # :132  (all Float32)
                    alpha = F32(0)
                    prior_rem = prior_rem * lam                  # :134
                    for k in range(1, maxActions + 1):           # :135-138
                        gap = max(lam * self.prior[k, nindex, i], F32(1e-4))
                        alpha = max(alpha, self.q[k, nindex, i] + gap)
                    err = F32(np.inf)
                    for _j in range(1, 101):                     # :141-162
                        S = prior_rem / alpha
                        gg = -prior_rem / (alpha * alpha)
                        for k in range(1, childnbr + 1):
                            CID = int(self.childID[k, nindex, i])
                            action = int(self.actionFromParent[CID, i])
                            top = lam * self.prior[action, nindex, i]
                            bot = alpha - self.q[action, nindex, i]
                            S = S + top / bot
                            gg = gg + (-top / (bot * bot))
                        newerr = S - F32(1)
                        if newerr < F32(0.001) or newerr == err:
                            break
                        alpha = alpha - newerr / gg
                        err = newerr
                    for k in range(1, maxActions + 1):           # :165-169
                        self.policy[k, nindex, i] = lam * self.prior[k, nindex, i] / (alpha - self.q[k, nindex, i])
                u = self.unext[nindex, i]                        # stands for prob[cpt, i] (:178) — see the module docstring
                for k in range(1, maxActions + 1):               # :172-182
                    delta = self.policy[k, nindex, i]
                    pr = pr + delta
                    if delta > 0:
                        bestmove = k
                    if pr >= u:
                        break
                if self.Achild[bestmove, nindex, i] == 0:        # :183-191
                    self.newindex[i] += 1
                    self.childnbr[nindex, i] += 1
                    self.childID[self.childnbr[nindex, i], nindex, i] = self.newindex[i]
                    self.Achild[bestmove, nindex, i] = self.childnbr[nindex, i]
                    self.parent[self.newindex[i], i] = nindex
                    self.actionFromParent[self.newindex[i], i] = bestmove
                    self.state[self.newindex[i]][i] = gm.play(self.state[nindex][i], bestmove)
                    self.depth[self.newindex[i], i] = cpt        # the child of a node at depth cpt - 1
                nindex = int(self.childID[self.Achild[bestmove, nindex, i], nindex, i])
                cpt += 1
            self.leaf[i] = nindex

    # ------------------------------------------------------------------------------------------------ :250-302
    def expand(self, prior, training, seed, step, rollout):
        """prior[j, i] (1-based rows): the actor's softmaxed output for the leaf of game i"""
        maxActions, gm = self.maxActions, self.gm
        for i in range(1, self.L + 1):
            nindex = int(self.leaf[i])
            self.unext[nindex, i] = uniform(seed, self.game_id[i], step, rollout, int(self.depth[nindex, i]))
            st = self.state[nindex][i]
            f, r = gm.isOver(st)
            self.expanded[nindex, i] = np.int8(1) - np.int8(1 if f else 0)
            if not f:
                if nindex == 1:
                    normalize = 0
                    A = F32(0)
                    for j in range(1, maxActions + 1):
                        if gm.canPlay(st, j):
                            self.prior[j, nindex, i] = prior[j, i]
                            normalize = normalize + self.prior[j, nindex, i]
                            A = A + F32(1)
                    if training:
                        for j in range(1, maxActions + 1):
                            if gm.canPlay(st, j):
                                self.prior[j, nindex, i] = F32(0.75) * self.prior[j, nindex, i] / normalize + F32(0.25) / A
                    else:
                        for j in range(1, maxActions + 1):
                            self.prior[j, nindex, i] = self.prior[j, nindex, i] / normalize
                else:
                    normalize = 0
                    for j in range(1, maxActions + 1):
                        if gm.canPlay(st, j):
                            self.prior[j, nindex, i] = prior[j, i]
                            normalize = normalize + self.prior[j, nindex, i]
                    for j in range(1, maxActions + 1):
                        self.prior[j, nindex, i] = self.prior[j, nindex, i] / normalize
            for k in range(1, maxActions + 1):
                self.policy[k, nindex, i] = self.prior[k, nindex, i]

    # ------------------------------------------------------------------------------------------------ :306-328
    def backUp(self, v, seed, step, rollout):
        gm = self.gm
        for i in range(1, self.L + 1):
            lf = int(self.leaf[i])
            a = int(self.parent[lf, i])
            while a != 0:                                        # rows this backup makes stale (:321): the uniform of their next visit
                self.unext[a, i] = uniform(seed, self.game_id[i], step, rollout, int(self.depth[a, i]))
                a = int(self.parent[a, i])
            nindex = int(self.parent[lf, i])
            move = int(self.actionFromParent[lf, i])
            f, r = gm.isOver(self.state[lf][i])
            if f:
                # (1 + player * r) / 2: Int8 * Int8 -> Int8, 1 + Int8 -> Int64, / 2 -> Float64 (:314)
                value = F64(1 + int(np.int8(np.int8(self.state[lf][i].player) * np.int8(r)))) / F64(2)
            else:
                value = F32(v[i])
            while nindex != 0:
                vis = self.nvisits[move, nindex, i]
                # Float32 * Float32, then + (1 - value) and / (visits + 1) in the type of `value`; the store rounds to Float32
                one_minus = (F64(1) - value) if isinstance(value, np.float64) else (F32(1) - value)
                num = vis * self.q[move, nindex, i]
                if isinstance(value, np.float64):
                    self.q[move, nindex, i] = F32((F64(num) + one_minus) / F64(vis + F32(1)))
                else:
                    self.q[move, nindex, i] = (num + one_minus) / (vis + F32(1))
                self.nvisits[move, nindex, i] = vis + F32(1)
                self.uptodate[nindex, i] = 0
                move = int(self.actionFromParent[nindex, i])
                nindex = int(self.parent[nindex, i])
                value = one_minus                                # value = 1 - value (:324)

    def copy_pol(self):                                          # :330-339
        for i in range(1, self.L + 1):
            for k in range(1, self.maxActions + 1):
                self.policy_final[k, i] = self.policy[k, 1, i]

    def decoder(self, nodes):                                    # :202-246 (planes of state[nodes[i], i]), rows [i][2 VS]
        out = np.zeros((self.L, 2 * self.gm.VS), F32)
        for i in range(1, self.L + 1):
            out[i - 1, :] = self.gm.planes(self.state[int(nodes[i])][i])
        return out

    # ------------------------------------------------------------------------------------------------ :376-462
    def mcts_single(self, actor, visits, training=True, cpuct=2.0, seed=1, step=0):
        """actor(planes [L][2 VS]) -> (softmaxed priors [L][A], v [L])  (the actor call :414 + softmax! :417)"""
        self.q[:] = 0; self.Achild[:] = 0; self.childID[:] = 0; self.nvisits[:] = 0     # noqa: E702  (:380-387)
        self.prior[:] = 0; self.policy[:] = 0; self.childnbr[:] = 0; self.newindex[:] = 1   # noqa: E702
        for k in range(1, visits + 1):
            self.kdescendTree(cpuct)
            batch = self.decoder(self.leaf)
            pr, v = actor(batch)
            prior = np.zeros((self.maxActions + 1, self.L + 1), F32)
            prior[1:, 1:] = np.asarray(pr, F32).T
            vv = np.concatenate([[F32(0)], np.asarray(v, F32)])
            self.expand(prior, training, seed, step, k - 1)
            self.backUp(vv, seed, step, k - 1)
        self.root_batch = self.decoder(np.ones(self.L + 1, np.int64))                   # decoder_roots :441
        self.copy_pol()
