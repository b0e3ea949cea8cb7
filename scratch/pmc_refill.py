# what bench.py times, for rocprofv3 --pmc: ONE agz_selfplay call of GENS x 32768 games on 32768 slots (finished games' slots refilled) of a
# BASELINE config; prints the algorithmic bytes / flops of the call from the device counters.  env CFG = 0 (headline) | 2..5, GENS (default 2)
import sys, os, json
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import bench
cfg, gens = int(os.environ.get("CFG", "0")), int(os.environ.get("GENS", "2"))
c = dict(game="gobang", n=9, nvict=5, games=32768, rollouts=64, filters=128, towers=6) if cfg == 0 else bench.CONFIGS[cfg]
L, V = c["games"], c["rollouts"]
g = ag.GameSpec(c["game"], c["n"], c["nvict"])
net = ag.SNetwork2.random(g, c["filters"], c["towers"])
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=gens * L)
e.set_network(net)
e.set_profiling(1)
e.kernel_times(reset=True)
st = e.selfplay(gens * L, V, cpuct=1.5, tau_plies=25)
tree_ms, nn_ms, launches = e.kernel_times()
p, n, ro = e.counters()
alg = bench.algorithmic_bytes(g, p, n, ro, g.pos_image_bytes)
print(json.dumps({"cfg": cfg, "gens": gens, "sum_p": p, "sum_new": n, "rollouts": ro, "search_launches": launches, "algorithmic_bytes_of_the_call": alg,
                  "flops_per_leaf": bench.nn_flops_per_leaf(g, c["filters"], c["towers"]), "rounds": st["plies"], "samples": st["nsamples"],
                  "search_kernel_ms": tree_ms, "form": e.search_form()}))
e.close()
