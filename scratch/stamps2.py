"""per-phase cycles of one rollout of a tree wave and of the network body (-DAGZ_STAMPS build in scratch/libagz_dbg.so), at the batch sizes
given in SIZES (sparse waves included: the wave count comes from the search form)"""
import sys, os, re, ctypes as C
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_dbg.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V = 64
g = ag.GameSpec(os.environ.get('GK', 'gobang'), int(os.environ.get('GN', '9')), int(os.environ.get('GV', '5'))); net = ag.SNetwork2.random(g, 128, 6)
names = ['0 prologue', '1 expand (+ sampling of the first visit)', '2 values', '3 item: fetch + row loads (wait)', '4 item: edge backup, q patch, re-sum', '5 item: scatter, lambda, alpha0', '6 item: Newton', '7 item: policy row', '8 item: running sums + sampling + store', '9 fence after items', '10 descent: root word', '11 descent: child word (wait)', '12 descent: step', '13 create + encode', '14 bookkeeping + wait for the other tree waves (first barrier)', '15 network + last barrier']
nn_names = ['weight requests', 'first barrier (wait for the tree waves)', 'B reads + MFMA issue (waits for weights)', 'epilogue (waits for MFMAs)', 'layer barriers', 'head']
for L in [int(x) for x in os.environ.get("SIZES", "256,1024,4096").split(",")]:
    e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
    e.set_profiling(1)
    e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
    out = (C.c_ulonglong * 32)()
    e.L.agz_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    e.L.agz_debug_stamps(e.h, out, 1)
    e.kernel_times(reset=True)
    e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
    e.L.agz_debug_stamps(e.h, out, 1)
    tree_ms, _, launches = e.kernel_times()
    form = e.search_form()[0]
    m = re.search(r"(\d+) games per workgroup, (\d+) per tree wave", form)
    gpwg, gpw = int(m.group(1)), int(m.group(2))
    tw = gpwg // gpw
    wgs = (L + gpwg - 1) // gpwg
    waves = wgs * tw * 65
    nwaves = wgs * (8 if "TW=8" in form else 4) * 64
    print(f"== L={L}  {tree_ms / max(launches,1):.3f} ms per search  [{form}]  tree waves {wgs * tw}")
    tot = sum(out[:16])
    for n, v in zip(names, out[:16]):
        print(f"{n:60s} {v/waves:10.0f} cyc/wave-rollout  {100*v/tot:5.1f}%")
    print(f"total cyc per wave-rollout {tot / waves:.0f}   (x 65 = {tot / waves * 65 / 1e6:.3f} M cycles per search)")
    print('network body, cycles per wave and rollout:')
    for n, v in zip(nn_names, out[16:22]): print(f"   {n:44s} {v/nwaves:9.0f}")
    e.close()
