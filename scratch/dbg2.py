import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O
for name in ('connect4', 'gobang9'):
    kind, n, k = common.GAMES[name]
    g, og = ag.GameSpec(kind, n, k), O.make_game(kind, n, k)
    net, onet = ag.SNetwork2.random(g, 128, 6), O.OracleNet(og, 128, 6)
    L, V = 64, 4
    roots = common.diverse_roots(og, L, seed=5, max_prefix=20)
    with M.Engine(g, L, V, seed=9, nn_mode=M.NN_BF16) as e:
        e.set_network(net); e.set_roots(common.pos_bytes(roots))
        e.search_begin(1.5, True, 2)
        e.rollout_select(0, last=False); e.rollout_eval()
        pr, v = e.get_eval()
        planes = e.leaf_batch()
    olg, ov = onet.logits(planes)
    opr, _ = onet.forward(planes)
    print(name, 'logit range', olg.min(), olg.max(), 'v range', ov.min(), ov.max())
    print('  prior err', np.abs(pr - opr).max(), 'v err', np.abs(v - ov).max())
    i = int(np.argmax(np.abs(v - ov)))
    print('  worst v game', i, 'gpu', v[i], 'ora', ov[i], 'planes sum', planes[i].sum())
    j = int(np.argmax(np.abs(pr - opr).max(1)))
    print('  worst p game', j, 'gpu', pr[j][:7], 'ora', opr[j][:7])
