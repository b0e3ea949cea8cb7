import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L, V = int(sys.argv[1]), 64
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
def depths_of_search(step):
    e.L.agz_search_begin(e.h, C.c_float(1.5), 1, step)
    prev = np.zeros(L, np.uint32); out = []
    for k in range(V):
        e.L.agz_rollout_select(e.h, k, int(k == V - 1)); e.L.agz_rollout_eval(e.h); e.L.agz_rollout_expand_backup(e.h)
        cur = np.zeros(L, np.uint32); e.L.agz_debug_slot_depths(e.h, cur.ctypes.data_as(C.c_void_p))
        out.append((cur - prev).astype(np.int64)); prev = cur
    e.L.agz_search_end(e.h)
    return np.array(out)          # [V][L] depth (expanded nodes traversed) per rollout
def waves(d, order):              # sum over waves of 8 games of (max depth + 1 round overheadless)
    dd = d[order]; pad = (-len(dd)) % 8
    dd = np.concatenate([dd, np.zeros(pad, dd.dtype)]).reshape(-1, 8)
    return dd.max(1).sum()
# play a few plies first so that roots differ
e.set_roots(None, L=L)
st = e.selfplay(L, 16, cpuct=1.5, tau_plies=25) if len(sys.argv) > 2 else None
e.set_roots(None, L=L)
D = depths_of_search(0)
ident = np.arange(L)
tot_ideal = 0; tot_ident = 0; tot_prev = 0; tot_ema = 0; tot_perfect = 0
ema = np.zeros(L)
for k in range(1, V):
    d = D[k]
    tot_ideal += d.sum() / 8.0                      # lower bound: every lane busy
    tot_ident += waves(d, ident)
    tot_perfect += waves(d, np.argsort(d, kind='stable'))
    tot_prev += waves(d, np.argsort(D[k - 1], kind='stable'))
    tot_ema += waves(d, np.argsort(ema, kind='stable'))
    ema = 0.7 * ema + 0.3 * d
def regroup(d, wg):
    dd = d.copy(); pad = (-len(dd)) % wg
    dd = np.concatenate([dd, np.zeros(pad, dd.dtype)]).reshape(-1, wg)
    tot = 0
    for r in range(1, int(dd.max()) + 1):
        n_r = (dd >= r).sum(1)
        tot += np.ceil(n_r / 8.0).sum()
    return tot
tot_rg32 = sum(regroup(D[k], 32) for k in range(1, V))
tot_rg16 = sum(regroup(D[k], 16) for k in range(1, V))
tot_rg64 = sum(regroup(D[k], 64) for k in range(1, V))
print(f"regrouped within workgroups of 16 / 32 / 64 games: {tot_rg16:.0f} / {tot_rg32:.0f} / {tot_rg64:.0f}")
print(f"L={L}: wave-rounds  ideal {tot_ideal:.0f}  slot order {tot_ident}  perfect sort {tot_perfect}  sort by prev depth {tot_prev}  sort by ema {tot_ema}")
print("mean depth per rollout idx (every 8th):", [round(float(D[k].mean()), 2) for k in range(0, V, 8)], "max", int(D.max()))
