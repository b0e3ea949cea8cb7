"""Generates tests/golden/*.npz from the CPU oracle (oracle/agz_oracle.c).

The reference has no golden vectors and cannot run here (Julia absent), so these fixtures pin the ORACLE's
outputs: they guard the oracle against regressions (-m "not gpu") and are what the HIP path must reproduce
bit for bit in EXACT mode (-m gpu).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = [
    # name,       L,  V,  H,  T, cpuct, training
    ("tictactoe", 16, 16, 32, 2, 1.5, 1),
    ("gobang9", 8, 16, 64, 2, 1.5, 1),
    ("connect4", 8, 24, 32, 2, 2.0, 1),
    ("hex5", 8, 16, 32, 1, 1.5, 0),
    ("hex9", 4, 12, 32, 1, 1.5, 1),
    ("reversi8", 8, 16, 32, 2, 1.5, 1),
    ("reversi6", 8, 20, 32, 1, 1.5, 1),
]
SEED, NETSEED = 1, 0x5EED

for name, L, V, H, T, cpuct, training in CASES:
    kind, n, k = common.GAMES[name]
    g = O.make_game(kind, n, k)
    net = O.OracleNet(g, H, T, NETSEED)
    roots = common.diverse_roots(g, L, seed=7)
    ids = (np.arange(L) * 3 + 5).astype(np.uint32)
    t = O.OracleTree(g, L, V)
    t.set_roots(roots, ids)
    pc, vc = t.search(net, V, cpuct, training, SEED, 3, capture=True)
    np.savez_compressed(
        os.path.join(HERE, f"search_{name}.npz"),
        roots=common.pos_bytes(roots), game_ids=ids, L=L, V=V, H=H, T=T, cpuct=np.float32(cpuct), training=training,
        seed=SEED, netseed=NETSEED, step=3,
        policy=t.policy(), visits=t.root_visits(), q=t.root_q(), leaf=t.leaf(), newindex=t.newindex(),
        root_planes=t.root_planes(), prior_last=pc[-1], v_last=vc[-1], counters=np.array(t.counters()))
    print(name, "ok", t.counters())

# one tiny self-play generation: BASELINE.json configs[0] family (Gobang N=3, 128x6 net), 32 games x 16 rollouts
g = O.make_game("gobang", 3, 3)
net = O.OracleNet(g, 128, 6, NETSEED)
s = O.selfplay(g, net, 32, 16, 1.5, 25, SEED, 100)
np.savez_compressed(os.path.join(HERE, "selfplay_tictactoe.npz"),
                    ngames=32, V=16, H=128, T=6, cpuct=np.float32(1.5), tau=25, seed=SEED, netseed=NETSEED, base=100,
                    **{k: v for k, v in s.items() if isinstance(v, np.ndarray)},
                    wdl=np.array([s["wins"], s["draws"], s["losses"], s["total_plies"]]))
print("selfplay ok", s["n"], s["wins"], s["draws"], s["losses"])

# two-actor games (duelnetwork's halves, mcts_gpu.jl:581-651): moves of every game and [v, n, d], both orders
for name, ngames, V, H, T in (("tictactoe", 24, 12, 32, 1), ("connect4", 10, 12, 32, 1)):
    kind, n, k = common.GAMES[name]
    g = O.make_game(kind, n, k)
    n1, n2 = O.OracleNet(g, H, T, NETSEED), O.OracleNet(g, H, T, NETSEED + 1)
    out = {}
    for first in (0, 1):
        d = O.duel(g, n1, n2, ngames, V, 2.0, 15, SEED + 4, 300, first)
        assert d["rc"] == 0
        out[f"wdl{first}"], out[f"moves{first}"], out[f"nplies{first}"] = np.array(d["wdl"]), d["moves"], d["nplies"]
    np.savez_compressed(os.path.join(HERE, f"duel_{name}.npz"), ngames=ngames, V=V, H=H, T=T, cpuct=np.float32(2.0), tau=15, seed=SEED + 4,
                        netseed=NETSEED, base=300, **out)
    print("duel", name, out["wdl0"], out["wdl1"])
