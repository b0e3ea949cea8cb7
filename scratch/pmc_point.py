# one mcts_single at the profile point of a BASELINE config (L = 32768, first ply, single chain): prints the algorithmic bytes / flops.
# env CFG = 0 (headline, default) | 2..5 (bench.py CONFIGS)
import sys, os, json
sys.path.insert(0, os.getcwd())
os.environ["AGZ_CHAINS"] = os.environ.get("AGZ_CHAINS", "1")
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import bench
cfg = int(os.environ.get("CFG", "0"))
c = dict(game="gobang", n=9, nvict=5, games=32768, rollouts=64, filters=128, towers=6) if cfg == 0 else bench.CONFIGS[cfg]
L, V = c["games"], c["rollouts"]
g = ag.GameSpec(c["game"], c["n"], c["nvict"])
net = ag.SNetwork2.random(g, c["filters"], c["towers"])
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
e.set_profiling(2)
for r in range(2):
    e.set_roots(None, L=L)
    e.kernel_times(reset=True)
    e.search(V, cpuct=1.5, training=True, step=0)
    e.synchronize()
p, n, ro = e.counters()
alg = bench.algorithmic_bytes(g, p, n, ro, g.pos_image_bytes)
print(json.dumps({"cfg": cfg, "sum_p": p, "sum_new": n, "rollouts": ro, "launches_per_search": 1, "algorithmic_bytes_per_search_launch": alg,
                  "nn_leaves": e.nn_leaves(), "flops_per_leaf": bench.nn_flops_per_leaf(g, c["filters"], c["towers"]), "form": e.search_form()}))
e.close()
