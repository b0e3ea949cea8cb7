# reproduce the one policy value of scratch/fuzz_generation.py case 15 that differs from the oracle: which network output differs?
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O
name, n, V, H, T, seed = "reversi8", 400, 64, 128, 2, 15
GID, PLY = 15369, 15
kind, nn, k = common.GAMES[name]
g, og = ag.GameSpec(kind, nn, k), O.make_game(kind, nn, k)
net, onet = ag.SNetwork2.random(g, H, T, 0x5EED + seed), O.OracleNet(og, H, T, 0x5EED + seed)
with M.Engine(g, n, V, seed=seed, game_id_base=1000 * seed, nn_mode=M.NN_BF16) as e:
    e.set_network(net)
    st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
    s = e.samples()
sel = np.where(s["game_id"] == GID)[0]
sel = sel[np.argsort(s["ply"][sel])]
moves = [int(s["move"][i]) for i in sel]
pol_gen = s["policy"][sel[PLY]].copy()
p = O.pos_init(og)
for m in moves[:PLY]:
    p = O.play(og, p, m)
root = [p]
ob = onet.bf16()
# oracle search of this root alone
t = O.OracleTree(og, 1, V); t.set_roots(root, np.array([GID], np.uint32)); t.search(ob, V, 1.5, True, seed, PLY)
out = {}
with M.Engine(g, 1, V, seed=seed, nn_mode=M.NN_BF16) as e:
    e.set_network(net)
    e.set_roots(common.pos_bytes(root), game_ids=np.array([GID], np.uint32))
    e.search(V, cpuct=1.5, training=True, step=PLY)
    pol_one = e.policy()[0].copy()
    print("whole search alone == generation sample:", np.array_equal(pol_one.view(np.uint32), pol_gen.view(np.uint32)),
          " == oracle:", np.array_equal(pol_one.view(np.uint32), t.policy()[0].view(np.uint32)),
          " visits equal:", np.array_equal(e.root_visits(), t.root_visits()), " q equal:", np.array_equal(e.root_q().view(np.uint32), t.root_q().view(np.uint32)))
    dq = np.where(e.root_q()[0].view(np.uint32) != t.root_q()[0].view(np.uint32))[0]
    print("q differs at actions", dq, e.root_q()[0][dq], t.root_q()[0][dq])
    # stepwise: compare the network outputs with the oracle's bf16 model at every rollout
    e.set_roots(common.pos_bytes(root), game_ids=np.array([GID], np.uint32))
    e.search_begin(1.5, True, PLY)
    nbad = 0
    for kk in range(V):
        e.rollout_select(kk, last=(kk == V - 1))
        planes = e.leaf_batch().copy()
        e.rollout_eval()
        lg, v = e.get_logits()
        olg, ov = ob.logits_bf16(planes)
        bl = np.where(lg[0].view(np.uint32) != olg[0].view(np.uint32))[0]
        bv = v[0].view(np.uint32) != ov[0].view(np.uint32)
        if len(bl) or bv:
            nbad += 1
            print(f"rollout {kk}: logits differ at {bl}: gpu {lg[0][bl]} oracle {olg[0][bl]}; value gpu {v[0]!r} oracle {ov[0]!r} differ={bool(bv)}  leaf {e.leaf()[0]}")
            out[f"planes_{kk}"] = planes[0]; out[f"lg_gpu_{kk}"] = lg[0]; out[f"v_gpu_{kk}"] = v[0]
        e.rollout_expand_backup()
    e.search_end()
    print("stepwise policy == whole search:", np.array_equal(e.policy()[0].view(np.uint32), pol_one.view(np.uint32)), " network mismatches:", nbad)
os.makedirs(os.path.join(ROOT, "gpurun_out", "repro15"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "repro15", "mismatch.npz"), **out)
