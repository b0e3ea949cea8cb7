"""The C oracle against an independent numpy transliteration of mcts_gpu.jl (tests/ref_transliteration.py) on the golden searches:
visits, Q, policy_final, leaves and node counts bit for bit.  (CPU only; the reference itself cannot run here — no Julia.)"""
import glob
import os

import numpy as np
import pytest

import common
import oracle_lib as O
import ref_transliteration as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_ref(og, onet, roots, ids, V, cpuct, training, seed, step):
    t = R.RefTree(og, roots, V, ids)
    t.mcts_single(lambda planes: onet.forward(planes), V, training=training, cpuct=cpuct, seed=seed, step=step)
    L = len(roots)
    return dict(policy=t.policy_final[1:, 1:].T.copy(), visits=t.nvisits[1:, 1, 1:].T.copy(), q=t.q[1:, 1, 1:].T.copy(),
                leaf=(t.leaf[1:] - 1).astype(np.int32), newindex=t.newindex[1:].astype(np.int32), root_planes=t.root_batch,
                L=L)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "search_*.npz"))))
def test_transliteration_reproduces_the_oracle_on_the_golden_searches(path):
    z = np.load(path)
    name = os.path.basename(path)[len("search_"):-4]
    kind, n, k = common.GAMES[name]
    og = O.make_game(kind, n, k)
    onet = O.OracleNet(og, int(z["H"]), int(z["T"]), int(z["netseed"]))
    roots = common.pos_from_bytes(z["roots"])
    r = run_ref(og, onet, roots, z["game_ids"], int(z["V"]), float(z["cpuct"]), bool(z["training"]), int(z["seed"]), int(z["step"]))
    for key in ("visits", "q", "policy", "root_planes"):
        assert np.array_equal(common.bits(r[key]), common.bits(z[key])), key
    assert np.array_equal(r["leaf"], z["leaf"]) and np.array_equal(r["newindex"], z["newindex"])


@pytest.mark.parametrize("name,L,V,training,cpuct", [("gobang9", 6, 40, True, 1.5), ("connect4", 6, 48, False, 2.0), ("reversi6", 4, 40, True, 0.7),
                                                      ("tictactoe", 12, 30, True, 1.5)])
def test_transliteration_reproduces_the_oracle_on_deeper_searches(name, L, V, training, cpuct):
    """longer searches than the fixtures (deeper trees, terminal leaves with the Float64 value path, Newton with many children)"""
    kind, n, k = common.GAMES[name]
    og = O.make_game(kind, n, k)
    onet = O.OracleNet(og, 32, 1, 7)
    roots = common.diverse_roots(og, L, seed=11, max_prefix=(6 if name == "tictactoe" else 14))
    ids = (50 + 11 * np.arange(L)).astype(np.uint32)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots, ids)
    t.search(onet, V, cpuct, training, 3, 17)
    r = run_ref(og, onet, roots, ids, V, cpuct, training, 3, 17)
    assert np.array_equal(common.bits(r["visits"]), common.bits(t.root_visits()))
    assert np.array_equal(common.bits(r["q"]), common.bits(t.root_q()))
    assert np.array_equal(common.bits(r["policy"]), common.bits(t.policy()))
    assert np.array_equal(r["leaf"], t.leaf()) and np.array_equal(r["newindex"], t.newindex())
