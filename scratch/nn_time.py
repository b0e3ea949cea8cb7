# stand-alone wide-trunk network launches: time per launch by leaves (stepwise API: select once, then repeated evals)
import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = int(sys.argv[1]); reps = 20
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 512, 8)
e = M.Engine(g, L, 8, seed=1, nn_mode=M.NN_BF16); e.set_network(net); e.set_roots(None, L=L)
e.L.agz_search_begin(e.h, 1.5, 1, 0); e.L.agz_rollout_select(e.h, 0, 0)
for _ in range(3): e.L.agz_rollout_eval(e.h)
e.synchronize(); t0 = time.perf_counter()
for _ in range(reps): e.L.agz_rollout_eval(e.h)
e.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"L={L} MT={os.environ.get('AGZ_BIG_MT','-')} lo={os.environ.get('AGZ_BIG_LO','0')}: {dt*1e6:.1f} us per forward, {L*4.44e6/dt/1e15:.3f} PFLOP/s  [{e.search_form()[1][:28]}]")
e.close()
