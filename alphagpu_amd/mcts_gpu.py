"""Host-side mirror of the reference's `module mcts_gpu` (mcts_gpu.jl) over libagz.

Same entry points and argument meaning as the Julia module:

    init(positions_or_L, visits, game, ...)          mcts_gpu.jl:342-357
    re_init(positions, engine)                       :368-373
    mcts_single(actor, visits, engine; training, cpuct)   :376-462
    mcts(actor, visits, ngames, buffer; cpuct)       :477-579   -> (data, valid)
    mcts(actor1, actor2, ...) == mcts_duel           :581-651   -> [v, n, d]
    duelnetwork(actor1, actor2, visits, ngames)      :653-668   -> (v, n, d)

The CUDA arrays the reference passes around (vnodes, vnodesStats, leaf, newindex) live inside the
engine handle; `Engine` is what `init` returns.  All compute happens in HIP kernels; there is no CPU path.
"""
import ctypes as C
import secrets

import numpy as np

from . import lib as _lib
from .game import GameSpec
from .net import SNetwork2

NN_BF16, NN_EXACT = 0, 1
POS_JULIA, POS_COMPACT = 0, 1


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fresh_seed():
    """A new Philox key per generation / duel call: the reference draws unseeded CUDA.rand / StatsBase randomness on
    every call (mcts_gpu.jl:397,520), so two generations with an unchanged network must not replay the same games."""
    return secrets.randbits(63) | 1


class Engine:
    """Owns the device-side trees for up to `max_games` games of `visits` rollouts (create_cunodes_stats /
    create_roots, mcts_gpu.jl:35-53)."""

    def __init__(self, game, max_games, visits, device=0, seed=1, game_id_base=0, nn_mode=NN_BF16,
                 sample_capacity_games=0):
        self.game = game
        self.L = _lib.load_library()
        cfg = _lib.Config(game=game.kind, n=game.n, nvict=game.nvict, max_games=int(max_games), max_visits=int(visits),
                          device=int(device), seed=int(seed), game_id_base=int(game_id_base), nn_mode=int(nn_mode),
                          sample_capacity_games=int(sample_capacity_games))
        self.h = C.c_void_p()
        rc = self.L.agz_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            msg = self.L.agz_last_error(None)
            raise _lib.AgzError(rc, msg.decode() if msg else "agz_create failed")
        self.max_games, self.visits, self.nn_mode, self.device = int(max_games), int(visits), int(nn_mode), int(device)
        self.nslots = 0
        self._nets = {}

    # -- plumbing ------------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self.L.agz_last_error(self.h)
            raise _lib.AgzError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "h", None):
            self.L.agz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_seed(self, seed):
        """Philox key of every later search / selfplay / duel of this engine (agz_set_seed)."""
        self._chk(self.L.agz_set_seed(self.h, int(seed)))

    # -- network ----------------------------------------------------------------------------------------
    def set_network(self, net, which=0):
        """actor = convert_back(net) (DenseNet.jl:331-333)."""
        self._chk(self.L.agz_set_network_slot(self.h, which, net.H, net.T, *net.pointers()))
        self._nets[which] = net

    # -- roots ------------------------------------------------------------------------------------------
    def set_roots(self, positions=None, L=None, game_ids=None, fmt=POS_COMPACT):
        """re_init(positions, ...) (mcts_gpu.jl:368-373).  positions: None -> Position() x L; bytes-like/ndarray of
        Julia images (104/152 B) or compact 80-B records."""
        if positions is None:
            n = int(L)
            buf = None
        else:
            buf = np.ascontiguousarray(np.frombuffer(positions, np.uint8) if not isinstance(positions, np.ndarray)
                                       else positions.view(np.uint8).reshape(-1))
            rec = 80 if fmt == POS_COMPACT else self.game.pos_image_bytes
            n = buf.size // rec if L is None else int(L)
        ids = None if game_ids is None else np.ascontiguousarray(game_ids, np.uint32)
        self._chk(self.L.agz_set_roots(self.h, _p(buf), fmt, _p(ids), n))
        self.nslots = n

    # -- search -----------------------------------------------------------------------------------------
    def search(self, visits, cpuct=2.0, training=True, step=0, which=0):
        self._chk(self.L.agz_search_actor(self.h, which, int(visits), float(cpuct), int(bool(training)), int(step)))

    def search_begin(self, cpuct, training, step):
        self._chk(self.L.agz_search_begin(self.h, float(cpuct), int(bool(training)), int(step)))

    def rollout_select(self, rollout, last=False):
        self._chk(self.L.agz_rollout_select(self.h, int(rollout), int(bool(last))))

    def rollout_eval(self):
        self._chk(self.L.agz_rollout_eval(self.h))

    def get_eval(self):
        pr = np.zeros((self.nslots, self.game.A), np.float32)
        v = np.zeros(self.nslots, np.float32)
        self._chk(self.L.agz_get_eval(self.h, _p(pr), _p(v)))
        return pr, v

    def get_logits(self):
        """Raw actor output of the last network launch: logits [L][A] (before softmax!), v [L]."""
        lg = np.zeros((self.nslots, self.game.A), np.float32)
        v = np.zeros(self.nslots, np.float32)
        self._chk(self.L.agz_get_logits(self.h, _p(lg), _p(v)))
        return lg, v

    def inject_eval(self, prior, v):
        prior = np.ascontiguousarray(prior, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        assert prior.shape == (self.nslots, self.game.A) and v.shape == (self.nslots,)
        self._chk(self.L.agz_inject_eval(self.h, _p(prior), _p(v)))

    def rollout_expand_backup(self):
        self._chk(self.L.agz_rollout_expand_backup(self.h))

    def search_end(self):
        self._chk(self.L.agz_search_end(self.h))

    def _get(self, name, shape, dtype=np.float32):
        out = np.zeros(shape, dtype)
        self._chk(getattr(self.L, "agz_get_" + name)(self.h, _p(out)))
        return out

    def policy(self):          # Array(vnodesStats.policy_final)  (A,L) column-major == [L][A]
        return self._get("policy", (self.nslots, self.game.A))

    def batch(self):           # Array(vnodesStats.batch) after decoder_roots
        return self._get("batch", (self.nslots, 2 * self.game.VS))

    def leaf_batch(self):
        return self._get("leaf_batch", (self.nslots, 2 * self.game.VS))

    def root_visits(self):
        return self._get("root_visits", (self.nslots, self.game.A))

    def root_q(self):
        return self._get("root_q", (self.nslots, self.game.A))

    def leaf(self):
        return self._get("leaf", (self.nslots,), np.int32)

    def node_count(self):
        return self._get("node_count", (self.nslots,), np.int32)

    def counters(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self.L.agz_get_counters(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def search_form(self):
        """(tree kernel, network kernel) the last search ran."""
        a, b = C.create_string_buffer(256), C.create_string_buffer(256)
        self._chk(self.L.agz_get_search_form(self.h, a, b, 256))
        return a.value.decode(), b.value.decode()

    def samples_packed_host(self):
        """Packed sample records of the last selfplay (agz.h layout) copied to host memory: uint8 array [n, rec_bytes].
        The device staging buffer and the pinned host buffer are kept and reused by the next call (a host loop pays for their
        allocation once, not per generation): the returned array is a view that stays valid until then."""
        import torch
        n = self.num_samples()
        rb = self.game.rec_bytes
        need = max(n, 1) * rb
        if getattr(self, "_pk_dev", None) is None or self._pk_dev.numel() < need:
            cap = need + need // 8
            self._pk_dev = torch.empty(cap, dtype=torch.uint8, device=f"cuda:{self.device}")
            self._pk_host = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        self.samples_packed_into(self._pk_dev.data_ptr(), n)
        self._pk_host[: n * rb].copy_(self._pk_dev[: n * rb])
        return self._pk_host[: n * rb].numpy().reshape(n, rb)

    def set_profiling(self, on):
        self._chk(self.L.agz_set_profiling(self.h, 3 if on is True else int(on)))

    def kernel_times(self, reset=False):
        t, n, k = C.c_double(0), C.c_double(0), C.c_int64(0)
        self._chk(self.L.agz_get_kernel_times(self.h, C.byref(t), C.byref(n), C.byref(k), int(reset)))
        return t.value, n.value, k.value

    def tree_busy_ms(self):
        """Time during which at least one tree-kernel launch was running (union of the launch intervals) since the last
        kernel_times(reset=True); equals the summed tree time unless sub-batch chains overlap launches."""
        t = C.c_double(0)
        self._chk(self.L.agz_get_tree_busy_ms(self.h, C.byref(t)))
        return t.value

    def nn_leaves(self):
        """Leaves evaluated by the stand-alone network launches that kernel_times()'s nn_ms covers."""
        n = C.c_uint64(0)
        self._chk(self.L.agz_get_nn_leaves(self.h, C.byref(n)))
        return n.value

    def synchronize(self):
        self._chk(self.L.agz_synchronize(self.h))

    def set_network_tag(self, tag):
        """The tag (0..255) stored with every sample that later self-play searches produce (agz_set_network_tag): a host loop that changes
        the network between the calls of a chain numbers its networks with it."""
        self._chk(self.L.agz_set_network_tag(self.h, int(tag)))

    def age_stats(self):
        """Persistent self-play kernels since the last kernel_times(reset=True): (searches of a game, those run with node rows by the
        root's legal rank, games that changed workgroup through the migration queue)."""
        out = (C.c_uint64 * 3)()
        self._chk(self.L.agz_get_age_stats(self.h, C.byref(out)))
        return int(out[0]), int(out[1]), int(out[2])

    # -- generation ---------------------------------------------------------------------------------------
    def selfplay(self, ngames, visits, cpuct=2.0, tau_plies=25):
        st = _lib.SelfplayStats()
        rc = self.L.agz_selfplay(self.h, int(ngames), int(visits), float(cpuct), int(tau_plies), C.byref(st))
        self.nslots = 0
        stats = {f: getattr(st, f) for f, _ in _lib.SelfplayStats._fields_}
        if rc == -5:            # AGZ_ERR_ILLEGAL_MOVE == the reference's "faute" (valid=false)
            stats["valid"] = False
            return stats
        self._chk(rc)
        stats["valid"] = True
        return stats

    def selfplay_chain(self, ngames, next_ngames, visits, cpuct=2.0, tau_plies=25):
        """agz_selfplay_chain: one call of a chain of self-play calls — returns when its own `ngames` games are over, having started up to
        `next_ngames` games of the next call in the slots that came free (they stay in flight; the last call of a chain passes 0)."""
        st = _lib.SelfplayStats()
        rc = self.L.agz_selfplay_chain(self.h, int(ngames), int(next_ngames), int(visits), float(cpuct), int(tau_plies), C.byref(st))
        self.nslots = 0
        stats = {f: getattr(st, f) for f, _ in _lib.SelfplayStats._fields_}
        if rc == -5:
            stats["valid"] = False
            return stats
        self._chk(rc)
        stats["valid"] = True
        return stats

    def duel(self, ngames, visits, cpuct=2.0, tau_plies=15, first=0):
        wdl = (C.c_int64 * 3)()
        self._chk(self.L.agz_duel(self.h, int(ngames), int(visits), float(cpuct), int(tau_plies), int(first), C.byref(wdl)))
        self.nslots = 0
        return [int(wdl[0]), int(wdl[1]), int(wdl[2])]

    def samples(self):
        """Samples of the last selfplay in PoolSample order (ply-major, then game id)."""
        n = C.c_int64(0)
        self._chk(self.L.agz_get_samples_packed(self.h, None, 0, C.byref(n)))
        n = n.value
        g = self.game
        out = dict(state=np.zeros((n, 2 * g.VS), np.int8), policy=np.zeros((n, g.A), np.float32),
                   player=np.zeros(n, np.int8), value=np.zeros(n, np.float32), fstate=np.zeros((n, g.FS), np.int8),
                   game_id=np.zeros(n, np.uint32), ply=np.zeros(n, np.int32), move=np.zeros(n, np.int32))
        self._chk(self.L.agz_get_samples(self.h, _p(out["state"]), _p(out["policy"]), _p(out["player"]), _p(out["value"]),
                                         _p(out["fstate"]), _p(out["game_id"]), _p(out["ply"]), _p(out["move"])))
        return out

    def samples_into(self, state, policy, player, value, fstate):
        """Samples of the last selfplay written into caller-owned C-contiguous arrays (PoolSample layout and dtypes); any may be None."""
        g, n = self.game, self.num_samples()
        for a, shape, dt in ((state, (n, 2 * g.VS), np.int8), (policy, (n, g.A), np.float32), (player, (n,), np.int8),
                             (value, (n,), np.float32), (fstate, (n, g.FS), np.int8)):
            if a is not None and (a.shape != shape or a.dtype != dt or not a.flags["C_CONTIGUOUS"]):
                raise ValueError(f"samples_into: expected a C-contiguous {dt.__name__}{shape} array, got {a.dtype}{a.shape}")
        self._chk(self.L.agz_get_samples(self.h, _p(state), _p(policy), _p(player), _p(value), _p(fstate), None, None, None))
        return n

    def samples_packed_into(self, dev_ptr, capacity_records):
        """Write packed sample records to DEVICE memory (for the RCCL all-gather); returns the record count."""
        n = C.c_int64(0)
        self._chk(self.L.agz_get_samples_packed(self.h, C.c_void_p(dev_ptr), int(capacity_records), C.byref(n)))
        return n.value

    def num_samples(self):
        n = C.c_int64(0)
        self._chk(self.L.agz_get_samples_packed(self.h, None, 0, C.byref(n)))
        return n.value


# ---------------------------------------------------------------------------------------------------------
# module-level functions with the reference's names
# ---------------------------------------------------------------------------------------------------------
def init(positions_or_L, visits, game, **kw):
    """init(positions::Vector{Position}, visits) / init(visits, L) — mcts_gpu.jl:342-357."""
    if isinstance(positions_or_L, (int, np.integer)):
        eng = Engine(game, int(positions_or_L), visits, **kw)
        eng.set_roots(None, L=int(positions_or_L))
    else:
        fmt = kw.pop("fmt", POS_COMPACT)
        buf = np.frombuffer(positions_or_L, np.uint8) if not isinstance(positions_or_L, np.ndarray) else positions_or_L.view(np.uint8).reshape(-1)
        rec = 80 if fmt == POS_COMPACT else game.pos_image_bytes
        eng = Engine(game, buf.size // rec, visits, **kw)
        eng.set_roots(buf, fmt=fmt)
    return eng


def re_init(positions, engine, fmt=POS_COMPACT, game_ids=None):
    """re_init(positions, vnodes, L, nthreads, numblocks) — mcts_gpu.jl:368-373."""
    engine.set_roots(positions, fmt=fmt, game_ids=game_ids)


def mcts_single(actor, visits, engine, training=True, cpuct=2.0, step=0):
    """mcts_single(actor, visits, nthreads, vnodes, vnodesStats, leaf, newindex, L; training, cpuct) — :376-462.
    Results: engine.policy() (policy_final), engine.batch() (root planes)."""
    if actor is not None and engine._nets.get(0) is not actor:
        engine.set_network(actor, 0)
    engine.search(visits, cpuct=cpuct, training=training, step=step)


def mcts(actor, visits, ngames, buffer, game=None, cpuct=2.0, engine=None, tau_plies=25, seed=None, slots=None, **kw):
    """mcts(actor, visits, ngames, buffer::PoolSample; cpuct) — mcts_gpu.jl:477-579.
    Plays `ngames` self-play games to the end, pushes every (state, policy, player, value, fstate) sample into
    `buffer` in the reference's order and returns (data, valid) like the reference's named tuple.
    seed=None draws a fresh Philox key per call (the reference's randomness is unseeded); pass a seed to reproduce.
    slots < ngames: only `slots` games are in flight at a time and a slot whose game has ended takes the next game that has not
    started (every search on a full batch); the samples are those of the lock-step run over ngames slots."""
    own = engine is None
    seed = fresh_seed() if seed is None else int(seed)
    if own:
        nslots = ngames if slots is None else min(int(slots), int(ngames))
        engine = Engine(game if game is not None else buffer.game, nslots, visits, seed=seed, sample_capacity_games=ngames, **kw)
    else:
        engine.set_seed(seed)
    try:
        if engine._nets.get(0) is not actor:
            engine.set_network(actor, 0)
        stats = engine.selfplay(ngames, visits, cpuct=cpuct, tau_plies=tau_plies)
        if stats["valid"] and buffer is not None:
            buffer.push_from_engine(engine)
        return stats, stats["valid"]
    finally:
        if own:
            engine.close()


def mcts_duel(actor1, actor2, visits, ngames, game, cpuct=2.0, engine=None, tau_plies=15, seed=None, **kw):
    """mcts(actor1, actor2, visits, ngames; cpuct) — mcts_gpu.jl:581-651 -> [v, n, d] (actor1 moves first).
    seed=None: fresh Philox key per call."""
    own = engine is None
    seed = fresh_seed() if seed is None else int(seed)
    if own:
        engine = Engine(game, ngames, visits, seed=seed, **kw)
    else:
        engine.set_seed(seed)
    try:
        engine.set_network(actor1, 0)
        engine.set_network(actor2, 1)
        return engine.duel(ngames, visits, cpuct=cpuct, tau_plies=tau_plies, first=0)
    finally:
        if own:
            engine.close()


def duelnetwork(actor1, actor2, visits, ngames, game, seed=None, **kw):
    """duelnetwork(actor1, actor2, visits, ngames) — mcts_gpu.jl:653-668 -> (v, n, d) from actor1's side.
    The two halves use different Philox keys (seed, seed + 1); seed=None draws a fresh pair."""
    hn = ngames // 2
    seed = fresh_seed() if seed is None else int(seed)
    v1, n1, d1 = mcts_duel(actor1, actor2, visits, hn, game, seed=seed, **kw)
    d2, n2, v2 = mcts_duel(actor2, actor1, visits, hn, game, seed=seed + 1, **kw)
    return v1 + v2, n1 + n2, d1 + d2


__all__ = ["Engine", "GameSpec", "SNetwork2", "init", "re_init", "mcts_single", "mcts", "mcts_duel", "duelnetwork",
           "NN_BF16", "NN_EXACT", "POS_JULIA", "POS_COMPACT"]
