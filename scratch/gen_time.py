import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = 32768
g = ag.GameSpec('gobang', 9, 5)
net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, 64, seed=1, nn_mode=M.NN_BF16)
e.set_network(net)
e.set_profiling(int(os.environ.get("PROF", "0")))
for i in range(3):
    t0 = time.perf_counter()
    st = e.selfplay(L, 64, cpuct=1.5, tau_plies=25)
    dt = time.perf_counter() - t0
    print(f"gen {i}: {dt*1e3:.1f} ms  {st['rollouts']/dt/1e6:.1f} M rollouts/s  plies {st['plies']}")
e.close()
