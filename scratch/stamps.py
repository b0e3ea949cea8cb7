import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_dbg.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V, L = 64, int(os.environ.get("LL", "32768"))
g = ag.GameSpec(os.environ.get('GK', 'gobang'), int(os.environ.get('GN', '9')), int(os.environ.get('GV', '5'))); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
out = (C.c_ulonglong * 32)()
e.L.agz_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
e.L.agz_debug_stamps(e.h, out, 1)
e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
e.L.agz_debug_stamps(e.h, out, 1)
names_reg = ['0 meta stage', '1 newton: fast branch', '2 expand', '3 newton: slow branch', '4 backup', '5 fence', '6 newton: step+loop', '7 round: row load+philox', '8 round: stats/prior_rem/alpha0', '9 round: child table compaction', '10 round: newton', '11 round: policy', '12 round: child lookup/end', '13 round: sampling', '14 tail: create+planes', '15 writeback']
names = ['0 prologue', '1 expand (+ sampling of the first visit)', '2 values', '3 item: fetch + row loads (wait)', '4 item: edge backup, q patch, re-sum', '5 item: scatter, lambda, alpha0', '6 item: Newton', '7 item: policy row', '8 item: running sums + sampling + store', '9 fence after items', '10 descent: root word', '11 descent: child word (wait)', '12 descent: step', '13 create + encode', '14 bookkeeping + wait for the other tree waves (first barrier)', '15 network + last barrier']
tot = sum(out[:16])
G = 8; waves = (L * G // 64) * 65
for n, v in zip(names, out[:16]):
    print(f"{n:32s} {v/waves:10.0f} cyc/wave  {100*v/tot:5.1f}%")
print('total cyc/wave', tot / waves)

nn_names = ['weight requests', 'first barrier (wait for the tree waves)', 'B reads + MFMA issue (waits for weights)', 'epilogue (waits for MFMAs)', 'layer barriers', 'head']
nnw = (L // (16 if L <= 8192 else 32)) * 4 * 64
print('network body, cycles per wave and rollout:')
for n, v in zip(nn_names, out[16:22]): print(f"   {n:44s} {v/nnw:9.0f}")
