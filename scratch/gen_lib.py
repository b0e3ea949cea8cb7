# whole generations (32768 games x 64 rollouts, Gobang 9x9, 128x6) with an alternative build of the library: scratch/gen_lib.py libagz_X.so [reps]
import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', sys.argv[1])
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, 32768, 64, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
for r in range(reps + 1):
    e.set_seed(1 + r)
    t0 = time.perf_counter(); st = e.selfplay(32768, 64, cpuct=1.5, tau_plies=25); dt = time.perf_counter() - t0
    print(f"{sys.argv[1]} generation {r}: {dt*1e3:.1f} ms  {st['rollouts']/dt/1e6:.1f} M rollouts/s  plies {st['plies']}")
e.close()
