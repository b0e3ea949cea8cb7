"""Rows of each wave's own width (AGZ_DYN) against rows by action: one search of 32768 games whose roots all have the same number of stones
(plies 0, 9, 17, 25, 33, 41, 49, 57), and batches of mixed plies in random order and ordered by ply.   python scratch/dyn_time.py"""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
L = int(os.environ.get("L", "32768")); V = 64
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    batches = pickle.load(open("/tmp/dyn_roots.pkl", "rb"))
    g = ag.GameSpec("gobang", 9, 5)
    net = ag.SNetwork2.random(g, 128, 6)
    e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16)
    e.set_network(net); e.set_profiling(1)
    out = []
    lb = int(os.environ.get("AGZ_LEGAL_BOUND", "0"))
    for name, roots in batches:
        if lb and (not name.startswith("ply") or 81 - int(name[3:]) > lb): continue
        e.set_roots(roots); e.search(V, cpuct=1.5, training=True, step=0)
        e.kernel_times(reset=True)
        for _ in range(3):
            e.set_roots(roots); e.search(V, cpuct=1.5, training=True, step=0)
        tree, nn, launches = e.kernel_times()
        out.append(f"{name}: {tree / launches:6.3f}")
    print(f"[{sys.argv[2]:>12s}] " + "  ".join(out) + "   " + e.search_form()[0][:60], flush=True)
    e.close()
else:
    import common, oracle_lib as O
    og = O.make_game("gobang", 9, 5)
    rng = np.random.default_rng(5)
    def at_ply(p, n):
        out = []
        while len(out) < n:
            q = O.pos_init(og); ok = True
            for _ in range(p):
                legal = [a for a in range(og.A) if O.can_play(og, q, a)]
                q2 = O.play(og, q, legal[int(rng.integers(len(legal)))])
                if O.is_over(og, q2)[0]: ok = False; break
                q = q2
            if ok: out.append(bytes(q))
        return out
    pool = {p: at_ply(p, 64) for p in range(0, 58)}
    def tile(plies):
        return np.frombuffer(b"".join(pool[int(p)][i % 64] for i, p in enumerate(plies)), np.uint8).copy()
    batches = [(f"ply{p}", tile(np.full(L, p))) for p in (0, 9, 17, 25, 33, 41, 49, 57)]
    mixed = rng.integers(0, 49, L)
    batches.append(("mixed", tile(mixed)))
    batches.append(("sorted", tile(np.sort(mixed)[::-1])))
    s = np.sort(mixed)[::-1].reshape(-1, 64)            # 64-game workgroups: pair an old one with a young one on a CU (b and 511 - b)
    pickle.dump(batches, open("/tmp/dyn_roots.pkl", "wb"))
    for rep in range(2):
        for name, kv in (("by action", {"AGZ_CLS": "-1"}), ("2 classes", {"AGZ_CLS": "2"}), ("3 classes", {"AGZ_CLS": "3"}), ("static 8", {"AGZ_CLS": "-1", "AGZ_LEGAL_BOUND": "64"}), ("static 4", {"AGZ_CLS": "-1", "AGZ_LEGAL_BOUND": "32"})) + ((("wave width", {"AGZ_CLS": "-1", "AGZ_DYN": "1"}),) if os.environ.get("WITH_DYN") else ()):
            env = dict(os.environ); env.update(kv)
            subprocess.call([sys.executable, __file__, "--one", name], env=env)
