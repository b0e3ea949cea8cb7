# the root of game 42010 at ply 7 (gobang13, exact 256x3, seed 42), searched with V = 1..64 by two library builds: first V at which they differ
import sys, os, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "--run":
    import alphagpu_amd.lib as aglib
    if sys.argv[2] != "default": aglib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    import common, oracle_lib as O
    roots = pickle.load(open("/tmp/roots13.pkl", "rb"))
    g = ag.GameSpec("gobang", 13, 5)
    net = ag.SNetwork2.random(g, 256, 3, 0x5EED + 42)
    out = {}
    with M.Engine(g, 16, 64, seed=42, nn_mode=M.NN_EXACT) as e:
        e.set_network(net)
        for V in range(2, 65):
            e.set_roots(roots["pos"], game_ids=roots["ids"])
            e.search(V, cpuct=1.5, training=True, step=roots["step"])
            out[V] = (e.root_visits().copy(), e.root_q().copy(), e.policy().copy(), e.leaf().copy(), e.node_count().copy())
    pickle.dump(out, open(sys.argv[3], "wb"))
else:
    import common, oracle_lib as O
    og = O.make_game("gobang", 13, 5)
    onet = O.OracleNet(og, 256, 3, 0x5EED + 42)
    # replay the oracle generation and keep the positions of all 16 games at ply 7
    ref = O.selfplay(og, onet, 16, 64, 1.5, 25, 42, 42000)
    pos = []
    for gi in range(16):
        p = O.pos_init(og)
        moves = {int(pl): int(mv) for gid, pl, mv in zip(ref["game_id"], ref["ply"], ref["move"]) if gid == 42000 + gi}
        for pl in range(7): p = O.play(og, p, moves[pl])
        pos.append(p)
    pickle.dump({"pos": common.pos_bytes(pos), "ids": np.arange(42000, 42016, dtype=np.uint32), "step": 7}, open("/tmp/roots13.pkl", "wb"))
    libs = sys.argv[1:]
    for i, lib in enumerate(libs):
        subprocess.check_call([sys.executable, __file__, "--run", lib, f"/tmp/out13_{i}.pkl"])
    b = pickle.load(open("/tmp/out13_0.pkl", "rb"))        # the first library is the reference
    names = ("visits", "q", "policy", "leaf", "node_count")
    for li in range(1, len(libs)):
        a = pickle.load(open(f"/tmp/out13_{li}.pkl", "rb"))
        first = None
        for V in range(2, 65):
            bad = [(n, np.nonzero((x.view(np.uint32) != y.view(np.uint32)).reshape(16, -1).any(1))[0].tolist()) for n, x, y in zip(names, a[V], b[V]) if not np.array_equal(x.view(np.uint32), y.view(np.uint32))]
            if bad:
                first = (V, bad); break
        print("==", libs[li], "vs", libs[0], ":", "identical for V = 2..64" if first is None else f"first difference at V = {first[0]}: {first[1]}")
