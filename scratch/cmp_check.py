# rows by legal rank: is the form used, what does it buy (generation time with and without), same samples?
import sys, os, time, hashlib, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import alphagpu_amd.lib as aglib
    if os.environ.get("LIBNAME"): aglib.LIB_PATH = os.path.join(os.getcwd(), "scratch", os.environ["LIBNAME"])
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    gk, gn, gv, V = os.environ.get("GK", "gobang"), int(os.environ.get("GN", "9")), int(os.environ.get("GV", "5")), int(os.environ.get("VV", "64"))
    L = int(os.environ.get("LL", "32768"))
    g = ag.GameSpec(gk, gn, gv); net = ag.SNetwork2.random(g, int(os.environ.get('NH', '128')), int(os.environ.get('NT', '6')))
    e = M.Engine(g, L, V, seed=3, nn_mode=M.NN_BF16); e.set_network(net)
    ts = []
    for r in range(3):
        t0 = time.perf_counter(); st = e.selfplay(L, V, cpuct=1.5, tau_plies=25); ts.append(time.perf_counter() - t0)
    h = "%d/%d/%d plies %d" % (st["wins"], st["draws"], st["losses"], st["plies"])
    print(json.dumps({"lib": os.environ.get("LIBNAME", "main"), "no_compact": os.environ.get("AGZ_NO_COMPACT"), "gen_ms": [round(t * 1e3, 1) for t in ts], "samples": int(st["nsamples"]), "hash": h, "form": e.search_form()[0][:90]}))
    sys.exit(0)
runs = [(None, None), ("1", None), (None, None), ("1", None)]
if os.environ.get("LIBS"): runs = [(None, l if l != "main" else None) for l in os.environ["LIBS"].split(",")] * 2
for nc, lib in runs:
    env = dict(os.environ)
    if nc: env["AGZ_NO_COMPACT"] = nc
    else: env.pop("AGZ_NO_COMPACT", None)
    if lib: env["LIBNAME"] = lib
    else: env.pop("LIBNAME", None)
    print(subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1], flush=True)
