# one whole generation (Gobang 9x9, 32768 games x 64 rollouts, 128x6) for rocprofv3 --pmc: per-variant counters of the ply loop's searches
import sys, os
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, 32768, 64, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
st = e.selfplay(32768, 64, cpuct=1.5, tau_plies=25)
print("plies", st["plies"], "samples", st["nsamples"])
e.close()
