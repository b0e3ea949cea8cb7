"""phases of a rollout in k_search_big as wave 0 of every workgroup sees them (-DAGZ_BIGSTAMPS build in scratch/libagz_bigst.so): tree step,
barrier in front of the network pass, network pass, barrier behind it — config 3 (Gobang 9x9, 512x8), batch sizes in SIZES"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_bigst.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V = 64
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 512, 8)
for L in [int(x) for x in os.environ.get("SIZES", "32768,16384").split(",")]:
    e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net); e.set_profiling(1)
    out = (C.c_ulonglong * 32)()
    e.L.agz_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
    e.L.agz_debug_stamps(e.h, out, 1); e.kernel_times(reset=True)
    e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
    e.L.agz_debug_stamps(e.h, out, 1)
    ms, _, n = e.kernel_times()
    wgs = out[20]
    print(f"== L={L}: {ms / max(n, 1):.3f} ms per search [{e.search_form()[0][:70]}]  {wgs} workgroups; cycles per rollout (s_memtime, 100 MHz ticks x?):")
    tot = sum(out[16:20])
    for name, v in zip(("tree step", "barrier before the pass", "network pass", "barrier behind the pass"), out[16:20]):
        print(f"   {name:28s} {v / wgs / V:10.0f}   {100 * v / tot:5.1f} %")
    print(f"   total per rollout {tot / wgs / V:10.0f} -> x {V} = {tot / wgs:12.0f} per search")
    e.close()
