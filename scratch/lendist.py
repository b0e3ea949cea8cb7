import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = ag.GameSpec('gobang', 9, 5)
net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, 64, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
st = e.selfplay(L, 64, cpuct=2.0, tau_plies=25)
print(st)
n = C = None
import ctypes
# games alive per ply from the per-sample ply index
s = e.samples()
cnt = np.bincount(s["ply"])
print("alive per ply:", list(cnt))
e.close()
