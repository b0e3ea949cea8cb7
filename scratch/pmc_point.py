# one mcts_single at the profile point (L = 32768, V = 64, Gobang 9x9, 128x6, single chain): prints the algorithmic bytes
import sys, os, json
sys.path.insert(0, os.getcwd())
os.environ["AGZ_CHAINS"] = "1"
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import bench
L, V = 32768, 64
g = ag.GameSpec('gobang', 9, 5)
net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
for r in range(2):
    e.set_roots(None, L=L)
    e.kernel_times(reset=True)
    e.search(V, cpuct=1.5, training=True, step=0)
    e.synchronize()
p, n, ro = e.counters()
alg = bench.algorithmic_bytes(g, p, n, ro, g.pos_image_bytes)
print(json.dumps({"sum_p": p, "sum_new": n, "rollouts": ro, "launches_per_search": 1, "algorithmic_bytes_per_search_launch": alg}))
e.close()
