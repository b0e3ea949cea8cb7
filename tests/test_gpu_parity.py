"""GPU parity (-m gpu): the HIP path, called through the libagz C ABI, against the CPU oracle.

EXACT mode  : whole search / whole self-play generation bit-identical to the oracle (visits, leaves, moves,
              policy and Q bits; |dQ| <= 1e-4 is asserted as well, as BASELINE.json states it).
BF16 mode   : the benchmarked mode.  The oracle carries a bit-level model of the bf16 MFMA forward (agzo_forward_bf16, pinned
              by instruction outputs captured on the hardware) and of the bf16-mode softmax, so whole searches and whole
              generations are compared bit for bit as well; the teacher-forced form (the GPU's own softmaxed priors and
              values handed to the oracle rollout by rollout) is kept for the stepwise API, and the logits are also
              bounded against the oracle's fp32 forward.
"""
import glob
import os

import numpy as np
import pytest

import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common
import oracle_lib as O
import parity

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# bf16 network vs the oracle:
#  * BIT FOR BIT against the oracle's model of the bf16 MFMA forward (agzo_forward_bf16: weights / layer outputs rounded to
#    bf16, the matrix instruction's block arithmetic as captured on the hardware, tests/golden/mfma_kat.npz) — logits and values;
#  * against the oracle's fp32 forward (agzo_forward, DenseNet.jl:294-304) in LOGIT space, which bounds the bf16 rounding
#    itself: |dlogit| <= 2^-6 * max(1, |row|_inf), |dv| <= 2^-6 (priors are softmaxed logits: an absolute prior tolerance
#    could not fail).
BF16_LOGIT_REL_FP32 = 2.0 ** -6
BF16_VALUE_TOL_FP32 = 2.0 ** -6


def check_network_outputs(e, onet, planes, oracle_rows=None, what=""):
    """GPU logits / values of the last network launch against both oracle forwards (rows: the leaves sent through the C
    forwards); returns the worst normalised errors vs the fp32 forward."""
    lg, v = e.get_logits()
    rows = np.arange(planes.shape[0]) if oracle_rows is None else np.asarray(oracle_rows)
    blg, bv = onet.logits_bf16(planes[rows])
    assert_same_bits(lg[rows], blg, f"{what}: logits vs the bf16 MFMA model")
    assert_same_bits(v[rows], bv, f"{what}: values vs the bf16 MFMA model")
    olg, ov = onet.logits(planes[rows])
    oscale = np.maximum(1.0, np.abs(olg).max(axis=1, keepdims=True))
    wf = float((np.abs(lg[rows] - olg) / oscale).max())
    wvf = float(np.abs(v[rows] - ov).max())
    assert wf <= BF16_LOGIT_REL_FP32 and wvf <= BF16_VALUE_TOL_FP32, f"{what}: fp32-oracle logit err {wf:.3e} value err {wvf:.3e}"
    return wf, wvf


def spec(name):
    kind, n, k = common.GAMES[name]
    return ag.GameSpec(kind, n, k), O.make_game(kind, n, k)


def nets(g, og, H, T, seed=0x5EED):
    return ag.SNetwork2.random(g, H, T, seed), O.OracleNet(og, H, T, seed)


def assert_same_bits(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype == np.float32:
        bad = common.bits(a) != common.bits(b)
        assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} differ; max|d|={np.abs(a - b).max()}"
    else:
        assert np.array_equal(a, b), what


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "search_*.npz"))))
def test_exact_search_matches_golden(path):
    z = np.load(path)
    name = os.path.basename(path)[len("search_"):-4]
    g, og = spec(name)
    net, _ = nets(g, og, int(z["H"]), int(z["T"]), int(z["netseed"]))
    L, V = int(z["L"]), int(z["V"])
    with M.Engine(g, L, V, seed=int(z["seed"]), nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        e.set_roots(z["roots"], game_ids=z["game_ids"])
        e.search(V, cpuct=float(z["cpuct"]), training=bool(z["training"]), step=int(z["step"]))
        assert_same_bits(e.leaf(), z["leaf"], "leaf")
        assert_same_bits(e.node_count(), z["newindex"], "newindex")
        assert_same_bits(e.root_visits(), z["visits"], "visits")
        assert np.abs(e.root_q() - z["q"]).max() <= 1e-4
        assert_same_bits(e.root_q(), z["q"], "q")
        assert_same_bits(e.policy(), z["policy"], "policy_final")
        assert_same_bits(e.batch(), z["root_planes"], "root planes")
        p, n, r = e.counters()
        assert [p, n] == z["counters"][:2].tolist() and r == L * V


@pytest.mark.parametrize("name,L,V,H,T", [
    ("tictactoe", 64, 16, 128, 6), ("gobang9", 48, 64, 128, 6), ("connect4", 64, 64, 128, 6),
    ("hex9", 24, 128, 64, 2), ("reversi8", 32, 64, 64, 3), ("reversi6", 32, 48, 32, 2),
    ("gobang13", 12, 32, 32, 1), ("hex11", 8, 24, 32, 1),
    ("gobang9", 6, 200, 32, 1), ("connect4", 6, 256, 32, 1)])        # trees of more than 128 nodes (rank / child ids above 127)
def test_exact_search_matches_oracle(name, L, V, H, T):
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    roots = common.diverse_roots(og, L, seed=3)
    ids = (1000 + 7 * np.arange(L)).astype(np.uint32)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots, ids)
    t.search(onet, V, 1.5, True, 42, 5)
    with M.Engine(g, L, V, seed=42, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        e.set_roots(common.pos_bytes(roots), game_ids=ids)
        e.search(V, cpuct=1.5, training=True, step=5)
        assert_same_bits(e.leaf(), t.leaf(), "leaf")
        assert_same_bits(e.node_count(), t.newindex(), "newindex")
        assert_same_bits(e.root_visits(), t.root_visits(), "visits")
        assert np.abs(e.root_q() - t.root_q()).max() <= 1e-4
        assert_same_bits(e.root_q(), t.root_q(), "q")
        assert_same_bits(e.policy(), t.policy(), "policy_final")
        p, n, f = t.counters()
        assert e.counters()[:2] == (p, n)


@pytest.mark.parametrize("name,L,V,H,T", [
    ("gobang9", 48, 64, 128, 6), ("connect4", 64, 64, 128, 6), ("tictactoe", 64, 16, 128, 6), ("hex9", 24, 128, 128, 2), ("reversi8", 40, 64, 64, 3),
    ("reversi8", 40, 32, 128, 2), ("reversi6", 40, 24, 128, 2),          # (whole-search kernel with passes)
    # BASELINE configs 3-5 with the trunk the reference ships (512 wide, 8 towers): k_rollout_eager + k_mlp_big
    ("gobang9", 136, 32, 512, 8), ("hex9", 40, 128, 512, 8), ("reversi8", 136, 24, 512, 8), ("gobang9", 40, 16, 256, 3)])
def test_bf16_search_matches_oracle_bitwise(name, L, V, H, T):
    """The BENCHMARKED mode (bf16 MFMA network, fp32 tree arithmetic), whole mcts_single, no teacher forcing: the oracle evaluates
    the network with its bit-level model of the MFMA forward and the bf16-mode softmax, after which leaves, node counts,
    visits, Q, policy_final must be bit-identical (and |dQ| <= 1e-4 as BASELINE.json states it)."""
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    roots = common.diverse_roots(og, L, seed=3)
    ids = (500 + 3 * np.arange(L)).astype(np.uint32)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots, ids)
    t.search(onet.bf16(), V, 1.5, True, 42, 7)
    with M.Engine(g, L, V, seed=42, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(common.pos_bytes(roots), game_ids=ids)
        e.search(V, cpuct=1.5, training=True, step=7)
        assert_same_bits(e.leaf(), t.leaf(), "leaf")
        assert_same_bits(e.node_count(), t.newindex(), "newindex")
        assert_same_bits(e.root_visits(), t.root_visits(), "visits")
        assert np.abs(e.root_q() - t.root_q()).max() <= 1e-4
        assert_same_bits(e.root_q(), t.root_q(), "q")
        assert_same_bits(e.policy(), t.policy(), "policy_final")
        p, n, f = t.counters()
        assert e.counters()[:2] == (p, n)


@pytest.mark.parametrize("name,n,V,H,T", [("gobang9", 40, 32, 128, 1), ("hex9", 20, 32, 128, 1), ("gobang9", 300, 36, 128, 2), ("gobang9", 32, 16, 128, 1), ("gobang9", 24, 32, 512, 1), ("hex9", 12, 32, 512, 1),
                                          ("gobang13", 6, 64, 128, 1), ("gobang11", 8, 48, 128, 1), ("hex11", 6, 48, 128, 1)])
def test_bf16_generation_with_rows_by_legal_rank_matches_oracle_bitwise(name, n, V, H, T):
    """From ply 17 of a 9x9 game of Gobang / Hex the ply loop searches with node rows indexed by the ROOT's legal rank (8 instead of
    12 entries per lane: agz_tree_eager.hpp KPR_, `policy_final` spread back over the actions by k_spread_policy): the generation is
    the oracle's, sample for sample, and the same with the rows kept by action (AGZ_NO_COMPACT=1).  The third case has enough games
    for 32-game workgroups in the first plies; the fourth has trees of 16 nodes: too small for the compaction buffer of the 8-entry
    level (it lives in the edge table, 2 V floats), so only the 4-entry level from ply 49 on is used; the next two run the wide-trunk search k_search_big; 13x13 / 11x11 boards have levels of 16 / 8 / 4 and 12 / 8 / 4 entries per lane
    (24 and 16 by action)."""
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    ref = O.selfplay(og, onet.bf16(), n, V, 1.5, 25, 91, 700)
    assert ref["rc"] == 0
    for no_compact in (False, True):
        if no_compact:
            os.environ["AGZ_NO_COMPACT"] = "1"
        try:
            with M.Engine(g, n, V, seed=91, game_id_base=700, nn_mode=M.NN_BF16) as e:
                e.set_network(net)
                st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
                s = e.samples()
                form = e.search_form()[0]
        finally:
            os.environ.pop("AGZ_NO_COMPACT", None)
        assert ("rows by legal rank" in form) == (not no_compact), form      # (the last plies of a generation have few legal actions)
        assert st["valid"]
        assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
        for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
            assert_same_bits(s[key], ref[key], key)


@pytest.mark.parametrize("name,n,V,H,T", [("tictactoe", 256, 16, 128, 6), ("connect4", 48, 16, 128, 2), ("gobang9", 16, 8, 128, 1), ("reversi6", 24, 12, 64, 1)])
def test_bf16_selfplay_generation_matches_oracle_bitwise(name, n, V, H, T):
    """A whole self-play generation in the benchmarked bf16 mode (device ply loop, whole-search kernel) == the oracle's generation
    with the bf16 MFMA model: every sample, move, value and final position."""
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    ref = O.selfplay(og, onet.bf16(), n, V, 1.5, 25, 77, 500)
    assert ref["rc"] == 0
    with M.Engine(g, n, V, seed=77, game_id_base=500, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
        s = e.samples()
    assert st["valid"]
    assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
    for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
        assert_same_bits(s[key], ref[key], key)


@pytest.mark.parametrize("name,L,V,H,T", [
    ("gobang9", 64, 64, 128, 6), ("connect4", 64, 32, 128, 6), ("hex9", 32, 48, 128, 2), ("reversi8", 32, 32, 128, 2),
    # the trunks the reference ships (main*.jl:123-128: ressimplesf(..., 512, 4|6|8)) on BASELINE configs 3-5: k_mlp_big
    ("gobang9", 136, 24, 512, 8), ("hex9", 40, 128, 512, 8), ("reversi8", 136, 20, 512, 8), ("connect4", 48, 16, 512, 4),
    ("gobang9", 40, 12, 512, 6), ("gobang9", 40, 12, 256, 3)])
def test_bf16_teacher_forced_parity(name, L, V, H, T):
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    roots = common.diverse_roots(og, L, seed=5)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots)
    t.reset()
    worst = np.zeros(2)
    orows = np.arange(L) if H <= 128 else np.arange(0, L, 8)      # the scalar C forwards of a 512-wide net are slow: every 8th leaf
    with M.Engine(g, L, V, seed=9, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(common.pos_bytes(roots))
        e.search_begin(1.5, True, 2)
        for k in range(V):
            e.rollout_select(k, last=(k == V - 1))
            t.select(9, 2, k, 1.5)
            assert_same_bits(e.leaf(), t.leaf(), f"leaf @rollout {k}")
            assert_same_bits(e.leaf_batch(), t.encode_leaves(), f"leaf planes @rollout {k}")
            e.rollout_eval()
            pr, v = e.get_eval()
            if k % 4 == 0 or k == V - 1:
                worst = np.maximum(worst, check_network_outputs(e, onet, t.encode_leaves(), orows, f"rollout {k}"))
            lg, _ = e.get_logits()                               # the engine's softmax of its own logits == the oracle's bf16-mode softmax
            sm = lg.copy()
            for row in sm:
                O.lib().agzo_softmax_bf16mode(row.ctypes.data, g.A)
            assert_same_bits(pr, sm, f"softmax @rollout {k}")
            t.expand(pr, True, 9, 2, k)
            t.backup(v, 9, 2, k)
            e.rollout_expand_backup()
        e.search_end()
        assert_same_bits(e.root_visits(), t.root_visits(), "visits")
        assert np.abs(e.root_q() - t.root_q()).max() <= 1e-4
        assert_same_bits(e.root_q(), t.root_q(), "q")
        assert_same_bits(e.node_count(), t.newindex(), "newindex")
    print(f"\n[bf16 {name} {H}x{T}] bit-identical to the bf16 MFMA model; worst logit err vs the fp32 oracle {worst[0]:.2e} (value {worst[1]:.2e})")


@pytest.mark.parametrize("name,L,H,T", [
    ("gobang9", 8400, 512, 8),      # > 8192 leaves: 128 leaves per workgroup, ragged last tile
    ("hex9", 300, 512, 8), ("reversi8", 129, 512, 8), ("gobang9", 777, 512, 4), ("connect4", 200, 512, 6), ("gobang9", 300, 256, 6),
    ("gobang9", 20000, 128, 6), ("connect4", 300, 128, 6), ("tictactoe", 100, 64, 2)])
def test_bf16_network_kernels_match_oracle(name, L, H, T, monkeypatch):
    """Every bf16 network kernel (k_mlp_big 32- and 128-leaf builds, k_mlp_wave, the per-layer k_layer_bf16 fallback) on
    batches of real positions against the oracle forwards (DenseNet.jl:294-304): logits and values of a sample of leaves (first,
    last = ragged tile, strided) bit for bit against the bf16 MFMA model and within the bf16 bound of the fp32 forward."""
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    base = common.diverse_roots(og, min(L, 320), seed=17, max_prefix=min(og.ML - 2, 24))
    roots = [base[i % len(base)] for i in range(L)]
    rows = np.unique(np.concatenate([np.arange(0, min(L, 40)), np.arange(max(0, L - 40), L), np.arange(0, L, max(1, L // 64))]))
    res = {}
    for env in ({}, {"AGZ_NO_FUSED_NN": "1"}):
        monkeypatch.delenv("AGZ_NO_FUSED_NN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with M.Engine(g, L, 4, seed=3, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(common.pos_bytes(roots))
            e.search_begin(1.5, True, 0)
            e.rollout_select(0)
            planes = e.leaf_batch()
            e.rollout_eval()
            w = check_network_outputs(e, onet, planes, rows, f"{name} {H}x{T} {env}")
            res[str(env)] = e.get_logits()
            print(f"\n[{name} L={L} {H}x{T} {env}] bit-identical to the bf16 MFMA model on {len(rows)} leaves; vs fp32: logit {w[0]:.2e} value {w[1]:.2e}")
    a, b = res["{}"], res[str({"AGZ_NO_FUSED_NN": "1"})]
    assert_same_bits(a[0], b[0], "logits: one-launch kernel vs per-layer kernels")
    assert_same_bits(a[1], b[1], "values: one-launch kernel vs per-layer kernels")


@pytest.mark.parametrize("mode", [M.NN_BF16, M.NN_EXACT])
def test_fused_search_equals_stepwise(mode):
    g, og = spec("gobang9")
    net, _ = nets(g, og, 128, 6)
    roots = common.pos_bytes(common.diverse_roots(og, 40, seed=8))
    L, V = 40, 32
    with M.Engine(g, L, V, seed=4, nn_mode=mode) as e:
        e.set_network(net)
        e.set_roots(roots)
        e.search(V, cpuct=1.5, training=True, step=1)
        a = (e.policy(), e.root_visits(), e.root_q(), e.leaf(), e.node_count())
        e.set_roots(roots)
        e.search_begin(1.5, True, 1)
        for k in range(V):
            e.rollout_select(k, last=(k == V - 1))
            e.rollout_eval()
            e.rollout_expand_backup()
        e.search_end()
        b = (e.policy(), e.root_visits(), e.root_q(), e.leaf(), e.node_count())
    for x, y, w in zip(a, b, ("policy", "visits", "q", "leaf", "count")):
        assert_same_bits(x, y, w)


def test_exact_selfplay_generation_matches_golden():
    """BASELINE.json configs[0] family: Gobang N=3, 128x6 net, whole generation on the device."""
    z = np.load(os.path.join(GOLD, "selfplay_tictactoe.npz"))
    g, og = spec("tictactoe")
    net, _ = nets(g, og, int(z["H"]), int(z["T"]), int(z["netseed"]))
    n = int(z["ngames"])
    with M.Engine(g, n, int(z["V"]), seed=int(z["seed"]), game_id_base=int(z["base"]), nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        st = e.selfplay(n, int(z["V"]), cpuct=float(z["cpuct"]), tau_plies=int(z["tau"]))
        s = e.samples()
    assert st["valid"] and st["faults"] == 0
    assert [st["wins"], st["draws"], st["losses"], st["total_plies"]] == z["wdl"].tolist()
    assert st["nsamples"] == len(z["ply"])
    for key in ("game_id", "ply", "move", "player", "state", "fstate"):
        assert_same_bits(s[key], z[key], key)
    assert_same_bits(s["policy"], z["policy"], "policy")
    assert_same_bits(s["value"], z["value"], "value")


@pytest.mark.parametrize("name,n,V,H,T,tau", [
    ("tictactoe", 256, 16, 128, 6, 25),      # BASELINE.json configs[0]
    ("connect4", 48, 16, 32, 2, 25), ("reversi6", 24, 12, 32, 1, 25), ("hex5", 32, 16, 32, 1, 4), ("gobang9", 12, 8, 32, 1, 25)])
def test_exact_selfplay_generation_matches_oracle(name, n, V, H, T, tau):
    g, og = spec(name)
    net, onet = nets(g, og, H, T)
    ref = O.selfplay(og, onet, n, V, 1.5, tau, 77, 500)
    assert ref["rc"] == 0
    with M.Engine(g, n, V, seed=77, game_id_base=500, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        st = e.selfplay(n, V, cpuct=1.5, tau_plies=tau)
        s = e.samples()
    assert st["valid"]
    assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
    for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
        assert_same_bits(s[key], ref[key], key)


def test_mcts_module_api_fills_pool_like_reference():
    g, og = spec("tictactoe")
    net, onet = nets(g, og, 32, 1)
    buf = ag.PoolSample(g, 1000)
    stats, valid = M.mcts(net, 8, 16, buf, cpuct=1.5, seed=3, nn_mode=M.NN_EXACT)
    ref = O.selfplay(og, onet, 16, 8, 1.5, 25, 3, 0)
    assert valid and buf.length_buffer() == ref["n"] == stats["nsamples"]
    assert_same_bits(buf.policy[:ref["n"]], ref["policy"], "pool policy")
    assert np.array_equal(buf.state[:ref["n"]], ref["state"]) and np.array_equal(buf.fstate[:ref["n"]], ref["fstate"])
    assert np.array_equal(buf.value[:ref["n"]], ref["value"])


def test_samples_go_straight_into_the_pool_and_wrap_like_the_reference_ring():
    """PoolSample.push_from_engine (agz_get_samples unpacking into the ring's own arrays) == push_generation(samples()), also when
    the write wraps around the end of the ring (mainGobang.jl:54-68 newindex logic)."""
    g, _ = spec("connect4")
    net = ag.SNetwork2.random(g, 64, 1)
    with M.Engine(g, 200, 8, seed=4, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        st = e.selfplay(200, 8, cpuct=1.5, tau_plies=25)
        n = st["nsamples"]
        for length in (3 * n, n + n // 2):                 # second generation wraps in the smaller ring
            a, b = ag.PoolSample(g, length), ag.PoolSample(g, length)
            for _ in range(2):
                ia = a.push_from_engine(e)
                ib = b.push_generation(e.samples())
                assert np.array_equal(ia, ib)
            assert (a.currentIndex, a.full) == (b.currentIndex, b.full)
            for name in ("state", "policy", "player", "value", "fstate"):
                assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_julia_position_image_roundtrip_through_set_roots():
    g, og = spec("reversi8")
    net, onet = nets(g, og, 32, 1)
    roots = common.diverse_roots(og, 6, seed=2)
    img = np.concatenate([O.pos_image(og, p) for p in roots])
    with M.Engine(g, 6, 8, seed=1, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        e.set_roots(img, fmt=M.POS_JULIA)
        e.search(8, cpuct=1.5, training=True, step=0)
        a = e.policy()
        e.set_roots(common.pos_bytes(roots), fmt=M.POS_COMPACT)
        e.search(8, cpuct=1.5, training=True, step=0)
        assert_same_bits(a, e.policy(), "policy")


# ---- full-size, size-independent properties (BASELINE.json metric shape) -------------------------------
def test_full_size_properties_and_sharding_invariance():
    g, og = spec("gobang9")
    net, _ = nets(g, og, 128, 6)
    L, V = 32768, 64
    with M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(None, L=L)
        e.search(V, cpuct=1.5, training=True, step=0)
        vis, pol, cnt = e.root_visits(), e.policy(), e.node_count()
        q, leaf = e.root_q(), e.leaf()
        assert (vis.sum(1) == V - 1).all()                       # rollout 1 only expands the root
        assert (cnt >= 2).all() and (cnt <= V).all()
        assert np.isfinite(pol).all() and np.allclose(pol.sum(1), 1.0, atol=5e-3)
        p, n, r = e.counters()
        assert r == L * V and n == int((cnt - 1).sum()) and p >= L * (V - 1)
        e.set_roots(None, L=L)                                   # idempotence: same seed, same bits
        e.search(V, cpuct=1.5, training=True, step=0)
        assert_same_bits(e.root_visits(), vis, "visits rerun")
        assert_same_bits(e.policy(), pol, "policy rerun")
    # sharding: games 1000..1063 computed in a 64-slot engine must equal the same game ids inside the big one
    with M.Engine(g, 64, V, seed=1, game_id_base=1000, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(None, L=64)
        e.search(V, cpuct=1.5, training=True, step=0)
        assert_same_bits(e.root_visits(), vis[1000:1064], "shard visits")
        assert_same_bits(e.policy(), pol[1000:1064], "shard policy")
        assert_same_bits(e.root_q(), q[1000:1064], "shard q")
        assert_same_bits(e.leaf(), leaf[1000:1064], "shard leaf")
        assert_same_bits(e.node_count(), cnt[1000:1064], "shard node count")


@pytest.mark.parametrize("L", [20000, 32768])
def test_whole_search_kernel_at_scale_equals_two_kernel_form(L, monkeypatch):
    """The register budgets the big batches actually run (k_search_small<..,4,3> for 16384 < L <= 24576, <..,4,4> above) against
    the two-kernel form (k_rollout_eager + k_mlp_wave per rollout) on the same 2000 game ids: visits, q, policy, leaf, node count."""
    g, _ = spec("gobang9")
    net = ag.SNetwork2.random(g, 128, 6)
    V = 64
    with M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(None, L=L)
        e.search(V, cpuct=1.5, training=True, step=3)
        a = (e.root_visits(), e.policy(), e.root_q(), e.leaf(), e.node_count())
    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    with M.Engine(g, 2000, V, seed=1, game_id_base=L - 2000, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(None, L=2000)
        e.search(V, cpuct=1.5, training=True, step=3)
        b = (e.root_visits(), e.policy(), e.root_q(), e.leaf(), e.node_count())
    for x, y, w in zip(a, b, ("visits", "policy", "q", "leaf", "node count")):
        assert_same_bits(x[L - 2000:], y, w)


def test_errors_are_reported_not_swallowed():
    g, _ = spec("tictactoe")
    with M.Engine(g, 4, 8) as e:
        with pytest.raises(ag.AgzError):
            e.set_roots(None, L=5)                               # more than max_games
        e.set_roots(None, L=4)
        with pytest.raises(ag.AgzError):
            e.search(8)                                          # no network loaded
        e.set_network(ag.SNetwork2.random(g, 32, 1))
        with pytest.raises(ag.AgzError):
            e.search(9)                                          # V > max_visits
        with pytest.raises(ag.AgzError):
            e.rollout_expand_backup()                            # nothing evaluated
        e.set_roots(None, L=0)                                   # empty batch is legal
        e.search(8)
        assert e.policy().shape == (0, 9)


# ---- further shapes of BASELINE.json's configs -----------------------------------------------------------
def test_hex_128_rollouts_exact_parity():
    """configs[3] family: Hex 9x9 with V = 128 (two meta registers / 128-node trees), irregular-action PUCT stress."""
    g, og = spec("hex9")
    net, onet = nets(g, og, 128, 2)
    L, V = 16, 128
    roots = common.diverse_roots(og, L, seed=13, max_prefix=30)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots)
    t.search(onet, V, 1.5, True, 5, 9)
    with M.Engine(g, L, V, seed=5, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        e.set_roots(common.pos_bytes(roots))
        e.search(V, cpuct=1.5, training=True, step=9)
        assert_same_bits(e.root_visits(), t.root_visits(), "visits")
        assert_same_bits(e.policy(), t.policy(), "policy")
        assert_same_bits(e.root_q(), t.root_q(), "q")
        assert_same_bits(e.node_count(), t.newindex(), "newindex")


def test_reversi8_generation_with_passes_exact_parity():
    """configs[4] family: Reversi 8x8 self-play including pass moves (action 64) and game-end by double pass."""
    g, og = spec("reversi8")
    net, onet = nets(g, og, 32, 1)
    ref = O.selfplay(og, onet, 16, 8, 1.5, 25, 9, 0)
    with M.Engine(g, 16, 8, seed=9, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        st = e.selfplay(16, 8, cpuct=1.5, tau_plies=25)
        s = e.samples()
    assert st["valid"] and st["nsamples"] == ref["n"]
    for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
        assert_same_bits(s[key], ref[key], key)


@pytest.mark.parametrize("name,n,V,H,T,tau", [
    ("tictactoe", 64, 16, 32, 1, 15), ("connect4", 24, 12, 32, 2, 15), ("gobang9", 10, 8, 32, 1, 6), ("reversi6", 12, 8, 32, 1, 15)])
@pytest.mark.parametrize("first", [0, 1])
def test_duel_matches_oracle(name, n, V, H, T, tau, first):
    """mcts(actor1, actor2, visits, ngames; cpuct=2f0) (mcts_gpu.jl:581-651): training=false (no root mix), tau = 1 over ALL
    actions for round < 15 then argmax, the actor alternates by ply parity.  EXACT mode: W/D/L and every move of every game
    equal the oracle's restatement, for two different networks and both orders."""
    g, og = spec(name)
    a, oa = nets(g, og, H, T, seed=11)
    b, ob = nets(g, og, H, T, seed=22)
    ref = O.duel(og, oa, ob, n, V, 2.0, tau, 31, 200, first)
    assert ref["rc"] == 0
    with M.Engine(g, n, V, seed=31, game_id_base=200, nn_mode=M.NN_EXACT) as e:
        e.set_network(a, 0)
        e.set_network(b, 1)
        wdl = e.duel(n, V, cpuct=2.0, tau_plies=tau, first=first)
        s = e.samples()                         # the ply loop records (game, ply, move) for duels as well
    assert wdl == ref["wdl"], (wdl, ref["wdl"])
    assert sum(wdl) == n
    moves = np.full_like(ref["moves"], -1)
    moves[s["game_id"].astype(np.int64) - 200, s["ply"]] = s["move"]
    assert np.array_equal(moves, ref["moves"])
    assert np.array_equal(np.bincount(s["game_id"] - 200, minlength=n), ref["nplies"])


def test_duelnetwork_halves_and_fresh_seeds():
    """duelnetwork (mcts_gpu.jl:653-668): second half with the roles swapped and (d2, n2, v2) read back mirrored.  Wrapper
    calls without a seed draw a fresh Philox key per call (the reference's randomness is unseeded): two generations with
    the same network must not replay the same games; an explicit seed reproduces."""
    g, og = spec("tictactoe")
    a, oa = nets(g, og, 32, 1, seed=11)
    b, ob = nets(g, og, 32, 1, seed=22)
    v, n, d = M.duelnetwork(a, b, 16, 64, g, seed=77, nn_mode=M.NN_EXACT)
    r1 = O.duel(og, oa, ob, 32, 16, 2.0, 15, 77, 0, 0)["wdl"]
    r2 = O.duel(og, ob, oa, 32, 16, 2.0, 15, 78, 0, 0)["wdl"]
    assert (v, n, d) == (r1[0] + r2[2], r1[1] + r2[1], r1[2] + r2[0])
    bufs = []
    for seed in (None, None, 5, 5):
        buf = ag.PoolSample(g, 4000)
        st, valid = M.mcts(a, 8, 64, buf, cpuct=1.5, nn_mode=M.NN_EXACT, seed=seed)
        assert valid
        bufs.append((st["nsamples"], buf.policy[:st["nsamples"]].copy()))
    assert bufs[2][0] == bufs[3][0] and np.array_equal(bufs[2][1], bufs[3][1])
    assert bufs[0][0] != bufs[1][0] or not np.array_equal(bufs[0][1], bufs[1][1])


def test_packed_device_records_roundtrip():
    """The multi-GPU exchange path: agz_get_samples_packed into DEVICE memory (k_pack_samples) -> shard.unpack_records must
    equal agz_get_samples and the oracle's samples."""
    import torch
    from alphagpu_amd import shard
    g, og = spec("connect4")
    net, onet = nets(g, og, 32, 2)
    n, V = 40, 12
    ref = O.selfplay(og, onet, n, V, 1.5, 25, 13, 900)
    with M.Engine(g, n, V, seed=13, game_id_base=900, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
        cap = n * g.max_plies
        dev = torch.empty(cap * g.rec_bytes, dtype=torch.uint8, device="cuda")
        cnt = e.samples_packed_into(dev.data_ptr(), cap)
        e.synchronize()
        s = e.samples()
    assert cnt == st["nsamples"] == ref["n"]
    u = shard.unpack_records(dev.cpu().numpy(), cnt, g)
    for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
        assert_same_bits(u[key], s[key], "packed vs agz_get_samples: " + key)
        assert_same_bits(u[key], ref[key], "packed vs oracle: " + key)
    rb = dev.cpu().numpy()[: cnt * g.rec_bytes].reshape(cnt, g.rec_bytes)
    assert not rb[:, 17:20].any() and not rb[:, 20 + 4 * g.A + 2 * g.VS + g.FS:].any()      # padding is zeroed


def test_duel_runs_and_is_consistent():
    """mcts(actor1, actor2, ...) (mcts_gpu.jl:581-651): same net on both sides, every game ends, W+D+L = ngames."""
    g, _ = spec("tictactoe")
    net = ag.SNetwork2.random(g, 32, 1)
    wdl = M.mcts_duel(net, net, 16, 64, g, cpuct=2.0, seed=5)
    assert sum(wdl) == 64 and all(x >= 0 for x in wdl)
    v, n, d = M.duelnetwork(net, net, 16, 64, g, seed=5)
    assert v + n + d == 64


def _bf16_search_bits(g, net, L, V, H):
    with M.Engine(g, L, V, seed=11, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.set_roots(None, L=L)
        e.search(V, cpuct=1.5, training=True, step=2)
        return e.root_visits().copy(), e.policy().copy(), e.root_q().copy()


@pytest.mark.parametrize("H,T", [(128, 6), (64, 3), (128, 1), (512, 2), (256, 3)])
def test_bf16_network_kernels_agree_bitwise(H, T, monkeypatch):
    """The latency-first network kernel (agz_nn_wave.hpp, the default) at other tile counts / prefetch depths and the per-layer
    MFMA kernels (agz_nn.hpp) run the same MFMA sequence: identical bits, also for a ragged last tile and for layer counts that
    need identity padding groups."""
    g, _ = spec("gobang9")
    net = ag.SNetwork2.random(g, H, T)
    L, V = 300, 24
    ref = _bf16_search_bits(g, net, L, V, H)              # default path (H = 128: the whole-search kernel)
    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")             # from here on: one tree launch + one network launch per rollout
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    got = _bf16_search_bits(g, net, L, V, H)
    for a, b, what in zip(got, ref, ("visits", "policy", "q")):
        assert_same_bits(a, b, what + " two-kernel form")
    for env in ({"AGZ_NN_WAVE_LT": "4", "AGZ_NN_WAVE_DEPTH": "4"}, {"AGZ_NN_WAVE_LT": "2"}, {"AGZ_NN_WAVE_LT": "8"}, {"AGZ_NO_FUSED_NN": "1"}):
        for k in ("AGZ_NN_WAVE_LT", "AGZ_NN_WAVE_DEPTH", "AGZ_NO_FUSED_NN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = _bf16_search_bits(g, net, L, V, H)
        for a, b, what in zip(got, ref, ("visits", "policy", "q")):
            assert_same_bits(a, b, what + " " + str(env))


def test_sub_batch_chains_do_not_change_results(monkeypatch):
    """AGZ_CHAINS=k cuts the batch into k sub-batches on parallel streams; every per-game quantity is keyed by game
    id, so the cut must be invisible."""
    g, _ = spec("connect4")
    net = ag.SNetwork2.random(g, 64, 2)
    L, V = 700, 32
    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")             # the two-kernel form (chains apply to it only)
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    ref = _bf16_search_bits(g, net, L, V, 64)
    monkeypatch.setenv("AGZ_CHAINS", "3")
    got = _bf16_search_bits(g, net, L, V, 64)
    for a, b, what in zip(got, ref, ("visits", "policy", "q")):
        assert_same_bits(a, b, what)


def test_wide_trunk_leaf_tiles_agree_bitwise(monkeypatch):
    """The stand-alone wide-trunk network k_mlp_big runs 32, 64 or 128 leaves per workgroup depending on the launch size (forced
    here with AGZ_BIG_MT): each build gives the same bits as the per-layer kernels, ragged last tile."""
    g, _ = spec("connect4")
    net = ag.SNetwork2.random(g, 512, 1)
    L, V = 8400, 3
    monkeypatch.setenv("AGZ_CHAINS", "1")
    monkeypatch.setenv("AGZ_BIG_MAXL", "0")               # two kernels per rollout (no k_search_big)
    monkeypatch.setenv("AGZ_NO_FUSED_NN", "1")
    ref = _bf16_search_bits(g, net, L, V, 512)
    monkeypatch.delenv("AGZ_NO_FUSED_NN")
    for mt in ("2", "4", "8", None):
        if mt is None:
            monkeypatch.delenv("AGZ_BIG_MT")
        else:
            monkeypatch.setenv("AGZ_BIG_MT", mt)
        with M.Engine(g, L, V, seed=5, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=0)
            assert e.search_form()[1].startswith("k_mlp_big<H=512,MT=" + (mt or "4"))
        got = _bf16_search_bits(g, net, L, V, 512)
        for a, b, what in zip(got, ref, ("visits", "policy", "q")):
            assert_same_bits(a, b, what + f" MT={mt}")


@pytest.mark.parametrize("name,L,V", [("gobang9", 300, 24), ("connect4", 77, 36), ("hex9", 40, 64), ("hex9", 44, 128), ("reversi8", 130, 32),
                                      ("reversi6", 90, 20), ("tictactoe", 500, 8)])
def test_whole_search_kernel_agrees_bitwise(name, L, V, monkeypatch):
    """k_search_small (one launch per mcts_single, 16 or 32 games per workgroup, default up to 16384 games) runs the same
    tree step and network bodies as the two stand-alone kernels (trees of up to 128 nodes, a multiple of 4): identical bits,
    ragged last workgroup."""
    g, _ = spec(name)
    net = ag.SNetwork2.random(g, 128, 2)

    def run():
        with M.Engine(g, L, V, seed=11, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=2)
            return e.root_visits().copy(), e.policy().copy(), e.root_q().copy(), e.leaf().copy(), e.node_count().copy()

    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    ref = run()                                     # two kernels per rollout
    monkeypatch.delenv("AGZ_SMALL_MAXL")
    monkeypatch.delenv("AGZ_SMALL4_MAXL")
    # 16 games per workgroup (default: sparse waves at these sizes), dense waves, 4 games per wave, 32 games per workgroup
    # ... and every register budget of the 32-game build (launch bounds for 2 / 3 / 4 workgroups per CU: k_search_small<..,4,2|3|4>)
    # ... and work lists that overflow their LDS part into global memory (16 bytes = 4 entries, and no LDS part at all), odd numbers
    # of games per wave, IEEE '/' instead of the guarded fast divisions
    for env in ({}, {"AGZ_SMALL_GPW": "8"}, {"AGZ_SMALL_GPW": "4"}, {"AGZ_SMALL_MAXL": "0"},
                {"AGZ_SMALL_MAXL": "0", "AGZ_SMALL4_OCC": "0"}, {"AGZ_SMALL_MAXL": "0", "AGZ_SMALL4_OCC": "1"},
                {"AGZ_SMALL_MAXL": "0", "AGZ_SMALL4_OCC": "2"}, {"AGZ_SMALL_GPW": "8", "AGZ_WL_LDS_BYTES": "16"},
                {"AGZ_SMALL_MAXL": "0", "AGZ_WL_LDS_BYTES": "0"}, {"AGZ_SMALL_GPW": "3"}, {"AGZ_SMALL_GPW": "5", "AGZ_SMALL_MAXL": "0"},
                {"AGZ_NO_FASTDIV": "1"}):
        monkeypatch.delenv("AGZ_SMALL_GPW", raising=False)
        monkeypatch.delenv("AGZ_SMALL_MAXL", raising=False)
        monkeypatch.delenv("AGZ_SMALL4_OCC", raising=False)
        monkeypatch.delenv("AGZ_WL_LDS_BYTES", raising=False)
        monkeypatch.delenv("AGZ_NO_FASTDIV", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for a, b, what in zip(got, ref, ("visits", "policy", "q", "leaf", "node_count")):
            assert_same_bits(a, b, what + " " + str(env))


@pytest.mark.parametrize("name,L,V,T", [("gobang9", 300, 24, 2), ("hex9", 70, 128, 1), ("reversi8", 1100, 16, 1), ("connect4", 2100, 12, 1)])
def test_whole_search_kernel_for_wide_trunks_agrees_bitwise(name, L, V, T, monkeypatch):
    """k_search_big (512-wide trunk, one launch per mcts_single, 32 games per 8-wave workgroup; sparse waves at these sizes) runs the
    same tree step and network bodies as k_rollout_eager + k_mlp_big: identical bits, ragged last workgroup, V = 128 trees."""
    g, _ = spec(name)
    net = ag.SNetwork2.random(g, 512, T)

    def run():
        with M.Engine(g, L, V, seed=11, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=2)
            return e.root_visits().copy(), e.policy().copy(), e.root_q().copy(), e.leaf().copy(), e.node_count().copy(), e.search_form()[0]

    monkeypatch.setenv("AGZ_BIG_MAXL", "0")
    ref = run()                                     # two kernels per rollout
    assert ref[5].startswith("k_rollout_eager")
    monkeypatch.delenv("AGZ_BIG_MAXL")
    for env in ({}, {"AGZ_SMALL_GPW": "8"}, {"AGZ_SMALL_GPW": "2"}, {"AGZ_SMALL_GPW": "8", "AGZ_WL_LDS_BYTES": "16"}):
        monkeypatch.delenv("AGZ_SMALL_GPW", raising=False)
        monkeypatch.delenv("AGZ_WL_LDS_BYTES", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        assert got[5].startswith("k_search_big")
        for a, b, what in zip(got[:5], ref[:5], ("visits", "policy", "q", "leaf", "node_count")):
            assert_same_bits(a, b, what + " " + str(env))


def test_duel_with_two_different_128_wide_networks_is_deterministic_and_symmetric():
    """mcts(actor1, actor2, ...) on the whole-search kernel (H = 128): two different weight slots alternate by ply parity.
    Same seeds -> same result; swapping who moves first swaps the roles (W/L mirror when the nets are swapped too)."""
    g, _ = spec("connect4")
    a, b = ag.SNetwork2.random(g, 128, 2, seed=1), ag.SNetwork2.random(g, 128, 2, seed=2)
    w1 = M.mcts_duel(a, b, 16, 96, g, cpuct=2.0, seed=5)
    w2 = M.mcts_duel(a, b, 16, 96, g, cpuct=2.0, seed=5)
    assert list(w1) == list(w2) and sum(w1) == 96


def test_profiling_counters_and_busy_time_are_consistent():
    """agz_get_kernel_times / agz_get_tree_busy_ms / agz_get_counters: busy time (union of launch intervals) never exceeds the
    summed launch time, both are positive, and the descent counters do not depend on the execution form."""
    g, _ = spec("gobang9")
    net = ag.SNetwork2.random(g, 128, 2)
    res = []
    for env in ({}, {"AGZ_SMALL_MAXL": "0", "AGZ_SMALL4_MAXL": "0", "AGZ_CHAINS": "2"}):
        for k in ("AGZ_SMALL_MAXL", "AGZ_SMALL4_MAXL", "AGZ_CHAINS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            with M.Engine(g, 600, 16, seed=3, nn_mode=M.NN_BF16) as e:
                e.set_network(net)
                e.set_profiling(1)
                e.set_roots(None, L=600)
                e.kernel_times(reset=True)
                e.search(16, cpuct=1.5, training=True, step=0)
                tree, nn, launches = e.kernel_times()
                busy = e.tree_busy_ms()
                p, n, ro = e.counters()
                assert launches >= 1 and tree > 0 and 0 < busy <= tree * 1.001
                res.append((p, n, ro))
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert res[0] == res[1] and res[0][2] == 600 * 16


# ---- the game plugins as the DEVICE runs them, against published counts ------------------------------------------------------
# (tests/test_oracle_games.py pins the ORACLE with the same numbers; here Game<FAM,NC>::canPlay / play / isOver of agz_games.hpp —
#  the code every tree and ply kernel is instantiated from — run on the GPU: agz_perft)
def test_device_perft_othello():
    from alphagpu_amd.game import perft
    g = ag.GameSpec("reversi8")
    assert [perft(g, d)[0] for d in range(1, 9)] == [4, 12, 56, 244, 1396, 8200, 55092, 390216]


def test_device_perft_tictactoe_full_game_tree():
    from alphagpu_amd.game import perft
    nodes, term = perft(ag.GameSpec("gobang", 3, 3), 9)
    assert term == [131184, 46080, 77904] and sum(term) == 255168       # X wins / draws / O wins: every complete game
    assert nodes == 127872                                               # games that last all 9 plies: 81792 wins on the last cell + 46080 draws


def test_device_perft_connect4():
    from alphagpu_amd.game import perft
    g = ag.GameSpec("connect4")
    assert [perft(g, d)[0] for d in range(1, 9)] == [7, 49, 343, 2401, 16807, 117649, 823536, 5673234]


@pytest.mark.parametrize("name", ["gobang9", "hex9", "gobang13", "hex11", "reversi6"])
def test_device_perft_equals_the_oracle(name):
    """boards of two and three 64-bit chunks, Hex's padded board, Reversi 6x6: the device counts == the oracle's (which the naive
    array models and the published tables pin)"""
    from alphagpu_amd.game import perft
    kind, n, k = common.GAMES[name]
    g, og = ag.GameSpec(kind, n, k), O.make_game(kind, n, k)
    for d in ((1, 2, 3) if g.A > 40 else (1, 2, 3, 4, 5, 6)):
        nodes, term = O.perft(og, O.pos_init(og), d)
        assert perft(g, d) == (int(nodes), [int(x) for x in term]), (name, d)


# ---- narrow lane-groups: 4 or 2 lanes per tree (16 / 32 trees per wave) ---------------------------------------------------------
@pytest.mark.parametrize("name,L,V,groups", [("gobang9", 300, 24, ("4",)), ("gobang9", 1000, 64, ("4",)), ("connect4", 333, 36, ("4", "2")),
                                             ("connect4", 2100, 64, ("4", "2"))])
def test_whole_search_kernel_with_narrow_lane_groups_agrees_bitwise(name, L, V, groups, monkeypatch):
    """k_search_small<..., G = 4 | 2>: the same tree step with 4 or 2 lanes per tree (24 actions per lane on a 9x9 board; Connect4's 7
    actions in 4 x 4 or 2 x 4 slots, its records laid out for that row width) — same bits as the two-kernel form with 8 lanes per tree, in
    both register budgets, ragged last wave and workgroup, work lists that overflow into global memory."""
    g, _ = spec(name)
    net = ag.SNetwork2.random(g, 128, 2)

    def run():
        with M.Engine(g, L, V, seed=11, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=2)
            return e.root_visits().copy(), e.policy().copy(), e.root_q().copy(), e.leaf().copy(), e.node_count().copy(), e.search_form()[0]

    monkeypatch.setenv("AGZ_NARROW", "-1")
    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    ref = run()                                     # two kernels per rollout, 8 lanes per tree
    assert ref[5].startswith("k_rollout_eager")
    monkeypatch.delenv("AGZ_SMALL_MAXL")
    monkeypatch.delenv("AGZ_SMALL4_MAXL")
    monkeypatch.setenv("AGZ_NARROW_MINL", "0")
    for grp in groups:
        # (32 trees per wave: a workgroup's tables take more than half a CU's LDS — one workgroup per CU, the one-wave-per-SIMD build only)
        for env in ({"AGZ_NARROW_OCC": "0"}, {"AGZ_NARROW_OCC": "1"}, {"AGZ_NARROW_OCC": "0" if grp == "4" else "1", "AGZ_WL_LDS_BYTES": "16"}):
            monkeypatch.delenv("AGZ_WL_LDS_BYTES", raising=False)
            monkeypatch.setenv("AGZ_NARROW", grp)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = run()
            assert f",G={grp}>" in got[5], got[5]
            for a, b, what in zip(got[:5], ref[:5], ("visits", "policy", "q", "leaf", "node_count")):
                assert_same_bits(a, b, f"{what} G={grp} {env}")


@pytest.mark.parametrize("L,V", [(333, 36), (2100, 64)])
def test_sparse_waves_of_the_narrow_search_agree_bitwise(L, V, monkeypatch):
    """k_search_small<F_C4, ..., WV = 4, G = 4, GPW = 8> (round 6): eight Connect4 games per wave of sixteen 4-lane groups, the groups without a game on
    work items, four waves per SIMD — what a full-batch Connect4 search runs; forced at a small size here (AGZ_NARROW_SPARSE=2): same bits as the
    two-kernel form with 8 lanes per tree, ragged last wave and workgroup, a work list that overflows into global memory"""
    g, _ = spec("connect4")
    net = ag.SNetwork2.random(g, 128, 2)

    def run():
        with M.Engine(g, L, V, seed=12, nn_mode=M.NN_BF16) as e:
            e.set_network(net)
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=3)
            return e.root_visits().copy(), e.policy().copy(), e.root_q().copy(), e.leaf().copy(), e.node_count().copy(), e.search_form()[0]

    monkeypatch.setenv("AGZ_NARROW", "-1")
    monkeypatch.setenv("AGZ_SMALL_MAXL", "0")
    monkeypatch.setenv("AGZ_SMALL4_MAXL", "0")
    ref = run()
    assert ref[5].startswith("k_rollout_eager")
    monkeypatch.delenv("AGZ_SMALL_MAXL")
    monkeypatch.delenv("AGZ_SMALL4_MAXL")
    monkeypatch.setenv("AGZ_NARROW", "4")
    monkeypatch.setenv("AGZ_NARROW_MINL", "0")
    monkeypatch.setenv("AGZ_NARROW_SPARSE", "2")
    for wl in (None, "16"):
        if wl:
            monkeypatch.setenv("AGZ_WL_LDS_BYTES", wl)
        got = run()
        assert "WV=4" in got[5] and ",G=4>" in got[5] and "8 games per tree wave" in got[5], got[5]
        for a, b, what in zip(got[:5], ref[:5], ("visits", "policy", "q", "leaf", "node_count")):
            assert_same_bits(a, b, f"{what} sparse 4-lane waves, work list {wl}")


@pytest.mark.parametrize("name,n,V,grp", [("gobang9", 40, 16, "4"), ("connect4", 70, 16, "4"), ("connect4", 70, 16, "2")])
def test_generation_with_narrow_lane_groups_equals_the_oracle(name, n, V, grp, monkeypatch):
    """a whole self-play generation through the narrow builds at every ply (Gobang 9x9: rows by action, then by legal rank 16 and 8 per lane):
    sample for sample the oracle's generation"""
    g, og = spec(name)
    net, onet = ag.SNetwork2.random(g, 128, 2), O.OracleNet(og, 128, 2)
    monkeypatch.setenv("AGZ_NARROW", grp)
    monkeypatch.setenv("AGZ_NARROW_MINL", "0")
    forms = set()
    with M.Engine(g, n, V, seed=6, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
        forms.add(e.search_form()[0])
        s = e.samples()
    assert st["valid"] and all(f",G={grp}>" in f for f in forms), forms
    ref = O.selfplay(og, onet.bf16(), n, V, 1.5, 25, 6, 0)
    assert ref["n"] == len(s["ply"])
    for k in ("game_id", "ply", "move", "player", "state", "fstate", "value", "policy"):
        assert parity.same_bits(s[k], ref[k]), f"{name} G={grp}: {k}"


# ---- more games than slots: finished games' slots are refilled --------------------------------------------------------------
@pytest.mark.parametrize("name,slots,ngames,V,mode", [("tictactoe", 16, 100, 8, "exact"), ("gobang9", 24, 70, 16, "bf16"), ("connect4", 40, 150, 12, "bf16"),
                                                      ("reversi6", 16, 50, 8, "exact"), ("hex5", 8, 40, 16, "bf16")])
def test_selfplay_with_more_games_than_slots_equals_the_lockstep_oracle(name, slots, ngames, V, mode):
    """agz_selfplay(ngames > max_games): a slot whose game has ended takes the next game that has not started yet (k_advance), every game
    keeps its own ply (key of its uniforms, tau rule, sample index) — sample for sample the generation the oracle plays in lock step
    over ngames slots, in PoolSample order (ply-major, then game id)."""
    g, og = spec(name)
    net, onet = nets(g, og, 32, 1) if mode == "exact" else (ag.SNetwork2.random(g, 128, 2), O.OracleNet(og, 128, 2))
    ref = O.selfplay(og, onet if mode == "exact" else onet.bf16(), ngames, V, 1.5, 25, 9, 1000)
    with M.Engine(g, slots, V, seed=9, game_id_base=1000, nn_mode=M.NN_EXACT if mode == "exact" else M.NN_BF16, sample_capacity_games=ngames) as e:
        e.set_network(net)
        st = e.selfplay(ngames, V, cpuct=1.5, tau_plies=25)
        s = e.samples()
        assert st["valid"] and st["nsamples"] == ref["n"] and st["wins"] + st["draws"] + st["losses"] == ngames
        assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
        for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
            assert_same_bits(s[key], ref[key], key)
        # ... and the engine is reusable: a lock-step generation afterwards (fewer games than slots)
        st2 = e.selfplay(slots // 2, V, cpuct=1.5, tau_plies=25)
        s2 = e.samples()
    ref2 = O.selfplay(og, onet if mode == "exact" else onet.bf16(), slots // 2, V, 1.5, 25, 9, 1000)
    assert st2["valid"] and st2["nsamples"] == ref2["n"]
    for key in ("game_id", "ply", "move", "policy", "value"):
        assert_same_bits(s2[key], ref2[key], key + " (second call)")


def test_selfplay_refill_needs_sample_capacity():
    g, _ = spec("tictactoe")
    with M.Engine(g, 8, 8, seed=1, nn_mode=M.NN_BF16) as e:
        e.set_network(ag.SNetwork2.random(g, 128, 2))
        with pytest.raises(Exception):
            e.selfplay(20, 8)


# ---- the two-kernel form on the big boards (rows of 16 / 24 actions per lane) -------------------------------------------------
@pytest.mark.parametrize("name,n,V,H,T,mode,seed", [("gobang13", 16, 64, 256, 3, "exact", 42), ("gobang13", 24, 32, 256, 2, "bf16", 5),
                                                    ("hex11", 16, 48, 256, 2, "exact", 7)])
def test_two_kernel_generation_on_big_boards_equals_the_oracle(name, n, V, H, T, mode, seed):
    """k_rollout_eager<KPL = 24 / 16> + the stand-alone network kernels over a whole generation.  The first case is the one the round-4
    fuzz run (profiles/r04_fuzz_parity.txt, set 2) caught: the 3-waves-per-SIMD build with the register prefetch of 24-action rows lost
    one node-count increment at ply 7 of game 42010 (the build now prefetches through registers up to 16 actions per lane only)."""
    g, og = spec(name)
    net, onet = ag.SNetwork2.random(g, H, T, 0x5EED + seed), O.OracleNet(og, H, T, 0x5EED + seed)
    with M.Engine(g, n, V, seed=seed, game_id_base=1000 * seed, nn_mode=M.NN_EXACT if mode == "exact" else M.NN_BF16) as e:
        e.set_network(net)
        st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
        assert e.search_form()[0].startswith("k_rollout_eager"), e.search_form()
        s = e.samples()
    ref = O.selfplay(og, onet if mode == "exact" else onet.bf16(), n, V, 1.5, 25, seed, 1000 * seed)
    assert st["valid"] and st["nsamples"] == ref["n"]
    for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
        assert_same_bits(s[key], ref[key], f"{name} {mode}: {key}")


# ---- chains of self-play calls: the next call's games start in the slots the running call leaves free -----------------------------
@pytest.mark.parametrize("name,slots,N,V,mode", [("tictactoe", 16, 60, 8, "exact"), ("gobang9", 24, 50, 16, "bf16"), ("connect4", 32, 90, 12, "bf16"),
                                                 ("reversi6", 16, 40, 8, "exact")])
def test_chain_of_selfplay_calls_returns_the_oracles_games_call_by_call(name, slots, N, V, mode):
    """agz_selfplay_chain: three calls of a chain (N, N and N / 2 games; the first two announce the size of the next one, the last one 0).
    Every call returns exactly ITS games — ids running on through the chain — sample for sample as the oracle's lock-step generation over
    all the chain's games has them (PoolSample order within the call), although most of a call's games were started, and some finished,
    while the call before it was still running; the sample store is a ring that wraps in the third call."""
    g, og = spec(name)
    net, onet = nets(g, og, 32, 1) if mode == "exact" else (ag.SNetwork2.random(g, 128, 2), O.OracleNet(og, 128, 2))
    # (the second pattern asks for FEWER games than it announced: the surplus stays in flight through a call that announces nothing, and the
    #  call after it picks it up)
    calls = [(N, N), (N, N // 2), (N // 2, 0)] if name != "connect4" else [(N, N), (N // 4, 0), (N // 2, N // 4), (N // 4, 0)]
    total = sum(n for n, _ in calls)
    ref = O.selfplay(og, onet if mode == "exact" else onet.bf16(), total, V, 1.5, 25, 9, 700)
    with M.Engine(g, slots, V, seed=9, game_id_base=700, nn_mode=M.NN_EXACT if mode == "exact" else M.NN_BF16, sample_capacity_games=2 * N + 7) as e:
        e.set_network(net)
        k0, rollouts = 0, 0
        for i, (n, nxt) in enumerate(calls):
            st = e.selfplay_chain(n, nxt, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
            sel = (ref["game_id"] >= 700 + k0) & (ref["game_id"] < 700 + k0 + n)
            assert st["valid"] and st["nsamples"] == int(sel.sum()) == len(s["ply"]), (i, st["nsamples"], int(sel.sum()))
            for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
                assert_same_bits(s[key], ref[key][sel], f"call {i}: {key}")
            first = sel & (ref["ply"] == 0)                      # the side to move at ply 0 is player +1: value = (1 + res) / 2
            res = np.rint(2.0 * ref["value"][first] - 1.0).astype(int)
            assert (st["wins"], st["draws"], st["losses"]) == (int((res == 1).sum()), int((res == 0).sum()), int((res == -1).sum()))
            assert st["total_plies"] == int(sel.sum()) - n
            if nxt and i == 0:                                   # games of the next call are in flight: their key must not change
                with pytest.raises(Exception):
                    e.set_seed(1234)
            k0 += n
            rollouts += st["rollouts"]
        # ... the work of the chain is the work of its games, no more (nothing is searched twice or dropped)
        assert rollouts >= V * len(ref["ply"])
        # a call of its own afterwards starts over (ids from game_id_base, a fresh batch)
        st = e.selfplay(slots, V, cpuct=1.5, tau_plies=25)
        s = e.samples()
    ref2 = O.selfplay(og, onet if mode == "exact" else onet.bf16(), slots, V, 1.5, 25, 9, 700)
    assert st["valid"] and st["nsamples"] == ref2["n"]
    for key in ("game_id", "ply", "move", "policy", "value"):
        assert_same_bits(s[key], ref2[key], key + " (call of its own after the chain)")


def test_chain_needs_sample_capacity_for_both_calls():
    g, _ = spec("tictactoe")
    with M.Engine(g, 8, 8, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=20) as e:
        e.set_network(ag.SNetwork2.random(g, 128, 2))
        with pytest.raises(Exception):
            e.selfplay_chain(16, 16, 8)
        assert e.selfplay_chain(12, 8, 8)["valid"] and e.selfplay_chain(8, 0, 8)["valid"]
        # the chain has run dry (nothing announced, nothing in flight): a further call of it starts its games itself, ids running on
        st = e.selfplay_chain(6, 0, 8)
        ids = np.unique(e.samples()["game_id"])
        assert st["valid"] and list(ids) == list(range(20, 26)) and st["wins"] + st["draws"] + st["losses"] == 6
