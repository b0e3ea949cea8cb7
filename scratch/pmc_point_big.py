# one mcts_single of BASELINE config 3 (Gobang 9x9, 512x8, L = 32768, V = 64): the two-kernel form in three sub-batch chains
import sys, os, json
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L, V = 32768, 64
g = ag.GameSpec('gobang', 9, 5)
net = ag.SNetwork2.random(g, 512, 8)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
for r in range(2):
    e.set_roots(None, L=L)
    e.search(V, cpuct=1.5, training=True, step=0)
    e.synchronize()
print(json.dumps({"form": e.search_form(), "flops_per_leaf": 4444160, "leaves_per_search": L * V}))
e.close()
