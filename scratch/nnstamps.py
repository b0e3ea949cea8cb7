import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_dbg.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V, L = 8, 32768
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
out = (C.c_ulonglong * 8)()
e.L.agz_debug_nn_stamps.argtypes = [C.c_void_p, C.c_void_p]
e.L.agz_debug_nn_stamps(e.h, out)
e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
e.L.agz_debug_nn_stamps(e.h, out)
names = ['0 prologue (planes+W0 stage)', '1 prefetch issue + acc zero', '2 MFMA loop', '3 epilogue', '4 barrier A', '5 commit', '6 barrier B', '7']
n = 256 * V
for nm, v in zip(names, out): print(f"{nm:32s} {v/n:9.0f} cyc per WG-launch")
print('total', sum(out) / n)
