# a two-network duel at a batch that runs k_search_big4 (one 128-game workgroup per CU) against the same duel with AGZ_BIG4=0 (two 64-game workgroups per CU):
# W / D / L must be identical (results depend on game ids only)
import os, sys
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec("gobang", 9, 5)
n0, n1 = ag.SNetwork2.random(g, 512, 1, 11), ag.SNetwork2.random(g, 512, 1, 22)
L, V = 20000, 16
out = []
for b in ("1", "0"):
    os.environ["AGZ_BIG4"] = b
    with M.Engine(g, L, V, seed=7, nn_mode=M.NN_BF16) as e:
        e.set_network(n0); e.set_network(n1, which=1)
        r = []
        for first in (0, 1):
            r.append(tuple(e.duel(L, V, cpuct=1.5, tau_plies=15, first=first)))
        out.append((r, e.search_form()[0].split(" (")[0]))
    print("AGZ_BIG4=" + b, out[-1])
assert out[0][0] == out[1][0], out
print("identical")
