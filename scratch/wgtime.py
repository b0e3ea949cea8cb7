"""Spread of the workgroups' run times inside the full-batch search launches of a refilled call (-DAGZ_WGTIME build, scratch/libagz_wgt.so):
per launch, mean(workgroup time) / (last end - first start) — the bound of what a barrier-free per-workgroup ply loop could win."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_wgt.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V, L = 64, 32768
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=6 * L); e.set_network(net)
st = e.selfplay(5 * L, V, cpuct=1.5, tau_plies=25)
print("plies", st["plies"], "rollouts/s %.1f M" % (st["rollouts"] / st["total_seconds"] / 1e6), e.search_form()[0][:80])
out = np.zeros((256, 512, 2), dtype=np.uint64)
e.L.agz_debug_wgtimes.argtypes = [C.c_void_p, C.c_void_p]
e.L.agz_debug_wgtimes(e.h, out.ctypes.data_as(C.c_void_p))
o = out.astype(np.int64)
rows = []
for s in range(256):
    t0, t1 = o[s, :, 0], o[s, :, 1]
    if t0.min() == 0: continue
    span = t1.max() - t0.min(); d = t1 - t0
    rows.append((s, span / 100.0, d.mean() / 100.0, d.min() / 100.0, d.max() / 100.0, (t0.max() - t0.min()) / 100.0, d.mean() / span))
rows = np.array(rows)
full = rows[(rows[:, 1] < 6000) & (rows[:, 1] > 3000)]
print("launches with stamps:", len(rows), " full-batch-like:", len(full))
print("per launch (us): span %.0f  wg mean %.0f  min %.0f  max %.0f  start skew %.0f   mean/span %.3f (min %.3f max %.3f)" % (
    full[:, 1].mean(), full[:, 2].mean(), full[:, 3].mean(), full[:, 4].mean(), full[:, 5].mean(), full[:, 6].mean(), full[:, 6].min(), full[:, 6].max()))
# per-CU view: the two workgroups of a CU are not known; percentiles of the workgroup times of one launch
s = int(full[len(full) // 2, 0]); d = (o[s, :, 1] - o[s, :, 0]) / 100.0
print("one launch (step %d): percentiles of workgroup time us: " % s, np.percentile(d, [0, 5, 25, 50, 75, 95, 100]).round(0))
e.close()
