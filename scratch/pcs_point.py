# profile point for PC sampling: a few first-ply searches (32768 games x 64 rollouts, Gobang 9x9, 128x6)
import sys, os
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L, V = int(os.environ.get("LL", "32768")), 64
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
for r in range(int(os.environ.get("REPS", "6"))):
    e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
e.close(); print("done")
