"""The C oracle against the widened second restatement (tests/ref_selfplay.py + tests/ref_games.py + tests/ref_transliteration.py, all
written from the Julia text alone and run here WITHOUT any oracle code under them): snetwork2's forward, whole self-play generations
(samples, PoolSample contents, W / D / L, tot_length) and two-actor games, bit for bit — on the golden self-play fixture, the golden
duel fixtures and further generations of every game.  CPU only."""
import os

import numpy as np
import pytest

import common
import oracle_lib as O
import ref_games as RG
import ref_selfplay as RS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ref_net(onet):
    return RS.snetwork2.from_flat(onet.inp, onet.H, onet.T, onet.A, onet.W0, onet.Wres, onet.Wp, onet.bp, onet.Wv, onet.bv)


def test_philox_known_answers():
    """Random123's kat_vectors for philox4x32-10"""
    assert RS.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert RS.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert RS.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


@pytest.mark.parametrize("name,H,T", [("tictactoe", 128, 6), ("connect4", 32, 2), ("gobang9", 64, 1), ("reversi8", 32, 1)])
def test_snetwork2_forward_equals_the_oracle(name, H, T):
    """DenseNet.jl:294-304 in numpy (k-ordered fma chains, the defined exp) == agzo_forward + agzo_softmax, logits, values and priors"""
    kind, n, k = common.GAMES[name]
    og = O.make_game(kind, n, k)
    onet = O.OracleNet(og, H, T, 11)
    onet.bp[:] = np.linspace(-0.3, 0.2, og.A).astype(np.float32); onet.bv[:] = 0.1   # (Flux initialises biases to 0: make them count)
    onet._sync()
    rng = np.random.default_rng(3)
    planes = (rng.random((9, 2 * og.VS)) < 0.3).astype(np.float32)
    lg, v = onet.logits(planes)
    pr, _ = onet.forward(planes)
    pol, val = ref_net(onet)(planes.T.copy())
    assert np.array_equal(common.bits(pol.T), common.bits(lg)) and np.array_equal(common.bits(val[0]), common.bits(v))
    rpr, rv = ref_net(onet).actor(planes)
    assert np.array_equal(common.bits(rpr), common.bits(pr)) and np.array_equal(common.bits(rv), common.bits(v))


def run_ref_selfplay(name, onet, ngames, V, cpuct, seed, base):
    kind, n, k = common.GAMES[name]
    rg = RG.make(kind, n, k)
    cap = ngames * (rg.maxLengthGame + 70)
    buf = RS.PoolSample(cap, rg.VectorizedState, rg.maxActions, rg.FeatureSize)
    r = RS.mcts(ref_net(onet).actor, V, ngames, buf, rg, cpuct=cpuct, seed=seed, game_id_base=base)
    assert r["valid"] and not buf.full
    nsmp = len(r["order"])
    out = dict(n=nsmp, wins=r["v"], draws=r["n"], losses=r["d"], total_plies=r["tot_length"])
    pool = [buf.pool[idx - 1] for idx, _, _, _ in r["order"]]
    out["state"] = np.array([s.state for s in pool], np.int8); out["policy"] = np.array([s.policy for s in pool], np.float32)
    out["player"] = np.array([s.player for s in pool], np.int8); out["value"] = np.array([s.value for s in pool], np.float32)
    out["fstate"] = np.array([s.fstate for s in pool], np.int8)
    out["game_id"] = np.array([g for _, g, _, _ in r["order"]], np.uint32); out["ply"] = np.array([p for _, _, p, _ in r["order"]], np.int32)
    out["move"] = np.array([m for _, _, _, m in r["order"]], np.int32)
    return out


KEYS = ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value")


def test_selfplay_loop_reproduces_the_golden_generation():
    """tests/golden/selfplay_tictactoe.npz (BASELINE config 1's family: Gobang N = 3, 32 games x 16 rollouts, 128 x 6 network), made by
    the C oracle: the transliterated loop — search, network, games, PoolSample, decode, move choice — gives the same 228 samples"""
    z = np.load(os.path.join(GOLD, "selfplay_tictactoe.npz"))
    og = O.make_game("gobang", 3, 3)
    onet = O.OracleNet(og, int(z["H"]), int(z["T"]), int(z["netseed"]))
    r = run_ref_selfplay("tictactoe", onet, int(z["ngames"]), int(z["V"]), float(z["cpuct"]), int(z["seed"]), int(z["base"]))
    assert r["n"] == len(z["ply"]) and [r["wins"], r["draws"], r["losses"], r["total_plies"]] == list(z["wdl"])
    for k in KEYS:
        a, b = r[k], z[k]
        assert np.array_equal(common.bits(a), common.bits(b)) if a.dtype == np.float32 else np.array_equal(a, b), k


@pytest.mark.parametrize("name,ngames,V,H,T,cpuct", [("connect4", 5, 8, 16, 1, 1.5), ("gobang9", 3, 8, 16, 1, 1.5), ("hex5", 4, 8, 16, 1, 2.0),
                                                     ("reversi6", 3, 8, 16, 1, 1.5), ("reversi8", 2, 6, 16, 1, 1.5)])
def test_selfplay_loop_reproduces_the_oracle_on_every_game(name, ngames, V, H, T, cpuct):
    kind, n, k = common.GAMES[name]
    og = O.make_game(kind, n, k)
    onet = O.OracleNet(og, H, T, 5)
    ref = O.selfplay(og, onet, ngames, V, cpuct, 25, 9, 40)
    r = run_ref_selfplay(name, onet, ngames, V, cpuct, 9, 40)
    assert ref["rc"] == 0 and r["n"] == ref["n"]
    assert (r["wins"], r["draws"], r["losses"], r["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
    for k_ in KEYS:
        a, b = r[k_], ref[k_]
        assert np.array_equal(common.bits(a), common.bits(b)) if a.dtype == np.float32 else np.array_equal(a, b), k_


@pytest.mark.parametrize("name", ["tictactoe", "connect4"])
def test_two_actor_loop_reproduces_the_golden_duels(name):
    """tests/golden/duel_*.npz (made by the C oracle's agzo_duel): mcts(actor1, actor2, visits, ngames) :581-651 transliterated — every
    move of every game and [v, n, d], with either actor moving first"""
    z = np.load(os.path.join(GOLD, f"duel_{name}.npz"))
    kind, n, k = common.GAMES[name]
    og, rg = O.make_game(kind, n, k), RG.make(kind, n, k)
    n1, n2 = O.OracleNet(og, int(z["H"]), int(z["T"]), int(z["netseed"])), O.OracleNet(og, int(z["H"]), int(z["T"]), int(z["netseed"]) + 1)
    a1, a2 = ref_net(n1).actor, ref_net(n2).actor
    for first in (0, 1):
        wdl, moves = RS.mcts_duel(a1 if first == 0 else a2, a2 if first == 0 else a1, int(z["V"]), int(z["ngames"]), rg, cpuct=float(z["cpuct"]),
                                  seed=int(z["seed"]), game_id_base=int(z["base"]), tau_rounds=int(z["tau"]))
        # [v, n, d] counts res == 1 / 0 / -1 (:618-624): the side that moves FIRST is player +1 (duelnetwork swaps the triple itself, :664)
        assert wdl == [int(x) for x in z[f"wdl{first}"]], (first, wdl)
        for g in range(int(z["ngames"])):
            npl = int(z[f"nplies{first}"][g])
            assert moves[int(z["base"]) + g] == list(z[f"moves{first}"][g][:npl]), (first, g)
