import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = ag.GameSpec(os.environ.get('GK', 'gobang'), int(os.environ.get('GN', '9')), int(os.environ.get('GV', '5')))
net = ag.SNetwork2.random(g, int(os.environ.get('NH', '128')), int(os.environ.get('NT', '6')))
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
e.set_profiling(os.environ.get("NOPROF") is None)
for r in range(reps):
    e.set_roots(None, L=L)
    e.kernel_times(reset=True)
    t0 = time.perf_counter()
    e.search(V, cpuct=1.5, training=True, step=0)
    t_enq = time.perf_counter() - t0
    e.synchronize()
    dt = time.perf_counter() - t0
    tree, nn, k = e.kernel_times()
    p, n, ro = e.counters()
    print(f"search {r}: enq {t_enq*1e3:.2f} ms wall {dt*1e3:.2f} ms  tree {tree:.2f} ms ({k} launches, {tree/max(k,1)*1e3:.1f} us avg)  nn {nn:.2f} ms  p/rollout {p/ro:.2f}  rollouts/s {ro/dt/1e6:.1f}M")
e.close()
