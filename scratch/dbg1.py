import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O
z = np.load('tests/golden/search_hex9.npz')
g, og = ag.GameSpec('hex', 9, 0), O.make_game('hex', 9, 0)
net, onet = ag.SNetwork2.random(g, 32, 1), O.OracleNet(og, 32, 1)
L, V = 4, 12
roots = common.pos_from_bytes(z['roots'])
t = O.OracleTree(og, L, V); t.set_roots(roots, z['game_ids']); t.reset()
with M.Engine(g, L, V, seed=1, nn_mode=M.NN_EXACT) as e:
    e.set_network(net); e.set_roots(z['roots'], game_ids=z['game_ids'])
    e.search_begin(1.5, True, 3)
    for k in range(V):
        e.rollout_select(k, last=True); t.select(1, 3, k, 1.5)
        gp, op = e.policy(), t.root_policy_row()
        d = common.bits(gp) != common.bits(op)
        print('rollout', k, 'leaf eq', np.array_equal(e.leaf(), t.leaf()), 'policy diff', int(d.sum()), 'per game', d.sum(1).tolist(),
              'maxrel', float(np.max(np.abs(gp-op)/(np.abs(op)+1e-30))))
        if d.any():
            i = int(np.argmax(d.sum(1))); ks = np.where(d[i])[0][:4]
            print('  game', i, 'k', ks, 'gpu', gp[i, ks], 'ora', op[i, ks], 'ratio', (gp[i,ks].astype(np.float64)/op[i,ks]))
        e.rollout_eval(); pr, v = e.get_eval()
        opr, ov = onet.forward(t.encode_leaves())
        print('   eval prior eq', np.array_equal(common.bits(pr), common.bits(opr)), 'v eq', np.array_equal(common.bits(v), common.bits(ov)))
        t.expand(opr, True); t.backup(ov)
        e.rollout_expand_backup()
