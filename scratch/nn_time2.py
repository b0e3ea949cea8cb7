# stand-alone wide-trunk network launches (512x8, Gobang 9x9): time per launch by leaves and leaf tiles per workgroup
import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
if os.environ.get('LIB'): aglib.LIB_PATH = os.path.join(os.getcwd(), os.environ['LIB'])
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 512, 8)
for L in (32768, 16384, 8192):
    for mt in ("8", "4", "2"):
        os.environ["AGZ_BIG_MT"] = mt
        e = M.Engine(g, L, 8, seed=1, nn_mode=M.NN_BF16); e.set_network(net); e.set_roots(None, L=L)
        e.L.agz_search_begin(e.h, 1.5, 1, 0); e.L.agz_rollout_select(e.h, 0, 0)
        for _ in range(3): e.L.agz_rollout_eval(e.h)
        e.synchronize(); t0 = time.perf_counter(); reps = 20
        for _ in range(reps): e.L.agz_rollout_eval(e.h)
        e.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"{os.environ.get('LIB', 'default'):24s} L={L} MT={mt}: {dt*1e6:7.1f} us per forward (incl. softmax launch), {L*4.44e6/dt/1e15:.3f} PFLOP/s  [{e.search_form()[1][:40]}]", flush=True)
        e.close()
