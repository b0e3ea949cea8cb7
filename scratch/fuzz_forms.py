# Full-size cross-checks of the execution forms against each other (GPU only, no oracle): the one-launch searches (k_search_small /
# k_search_big) vs two kernels per rollout, searches and whole generations, bit for bit.
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def run(kind, n, k, L, V, H, T, seed, gen):
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    g = ag.GameSpec(kind, n, k); net = ag.SNetwork2.random(g, H, T, 77 + seed)
    with M.Engine(g, L, V, seed=seed, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        if gen:
            st = e.selfplay(L, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
            out = [s[x] for x in ("game_id", "ply", "move", "policy", "value", "state")]
        else:
            e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=3)
            out = [e.root_visits(), e.policy(), e.root_q(), e.leaf(), e.node_count()]
        form = e.search_form()[0]
    import hashlib
    return form, [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in out]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    a = json.loads(sys.argv[2]); print(json.dumps(run(*a))); sys.exit(0)
cases = [("gobang", 9, 5, 32768, 64, 128, 6, 1, 0), ("gobang", 9, 5, 32768, 64, 128, 6, 2, 0), ("gobang", 9, 5, 24576, 64, 128, 6, 3, 0), ("hex", 9, 0, 32768, 128, 128, 6, 4, 0),
         ("connect4", 0, 0, 32768, 64, 128, 6, 5, 0), ("reversi8", 0, 0, 32768, 64, 128, 6, 6, 0), ("reversi6", 0, 0, 20000, 48, 128, 3, 7, 0),
         ("gobang", 9, 5, 16384, 64, 512, 8, 8, 0), ("hex", 9, 0, 12000, 128, 512, 8, 9, 0), ("reversi8", 0, 0, 16384, 64, 512, 8, 10, 0),
         ("gobang", 9, 5, 32768, 64, 128, 6, 11, 1), ("reversi8", 0, 0, 32768, 64, 128, 6, 12, 1), ("gobang", 9, 5, 8192, 64, 512, 8, 13, 1)]
bad = 0
for c in cases:
    res = []
    for env in ({}, {"AGZ_SMALL_MAXL": "0", "AGZ_SMALL4_MAXL": "0", "AGZ_BIG_MAXL": "0"}):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child", json.dumps(c)], env=e, capture_output=True, text=True)
        if out.returncode: print(out.stderr[-400:])
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
    same = res[0][1] == res[1][1]
    bad += not same
    print(f"{c}: {res[0][0][:40]} | {res[1][0][:36]} -> {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
print("forms:", "all identical" if not bad else f"{bad} differ")
sys.exit(1 if bad else 0)
