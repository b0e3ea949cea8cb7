# A/B of library builds on the GPU: scratch/ab.py [lib.so ...]   ("main" = alphagpu_amd/libagz.so; others are looked up in scratch/)
# per build (own process): first-ply search time at several batch sizes, whole generations, and a hash of the results of fixed
# searches (identical hashes = identical bits: the main build is the oracle-checked one)
import sys, os, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time, hashlib
    import numpy as np
    sys.path.insert(0, os.getcwd())
    import alphagpu_amd.lib as aglib
    name = sys.argv[2]
    if name != "main": aglib.LIB_PATH = os.path.join(os.getcwd(), "scratch", name)
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    H, T = int(os.environ.get("NH", "128")), int(os.environ.get("NT", "6"))
    gk, gn, gv = os.environ.get("GK", "gobang"), int(os.environ.get("GN", "9")), int(os.environ.get("GV", "5"))
    V = int(os.environ.get("VV", "64"))
    g = ag.GameSpec(gk, gn, gv); net = ag.SNetwork2.random(g, H, T)
    out = {"lib": name}
    sizes = [int(x) for x in os.environ.get("SIZES", "32768,24576,16384,4096,512").split(",")]
    e = M.Engine(g, max(sizes), V, seed=1, nn_mode=M.NN_BF16); e.set_network(net); e.set_profiling(1)
    for L in sizes:
        ts = []
        for r in range(4):
            e.set_roots(None, L=L); e.kernel_times(reset=True)
            e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
            tree, nn, k = e.kernel_times(); ts.append(tree + nn)
        h = hashlib.sha1()
        for a in (e.root_visits(), e.root_q(), e.policy(), e.leaf(), e.node_count()): h.update(np.ascontiguousarray(a).tobytes())
        out[f"ply_ms_{L}"] = round(min(ts[1:]), 3); out[f"hash_{L}"] = h.hexdigest()[:12]
        out[f"form_{L}"] = e.search_form()[0][:44]
    e.set_profiling(0)
    gens = []
    for r in range(int(os.environ.get("GENS", "3"))):
        e.set_seed(1 + r)
        t0 = time.perf_counter(); st = e.selfplay(max(sizes), V, cpuct=1.5, tau_plies=25); dt = time.perf_counter() - t0
        gens.append(round(dt * 1e3, 1))
    if gens:
        out["gen_ms"] = gens; out["gen_Mrps"] = round(st["rollouts"] / (min(gens) * 1e-3) / 1e6, 1); out["samples"] = st["nsamples"]
    e.close()
    print(json.dumps(out))
    sys.exit(0)
libs = sys.argv[1:] or ["main"]
for rep in range(int(os.environ.get("REPS", "1"))):
    for l in libs:
        r = subprocess.run([sys.executable, __file__, "--child", l], capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + l + " " + r.stderr[-400:]))
