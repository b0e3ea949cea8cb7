"""A/B of library builds on one box: for every libagz build given on the command line (a path, or 'default'), the first-ply search at
32768 / 16384 games and the rate of a refilled call of 4 x 32768 games (what bench.py times), each in a process of its own.
   python scratch/ab_lib.py default scratch/libagz_x.so ...        (GAME=gobang|connect4, H=128|512, T=6|8)"""
import os, sys, subprocess, time
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    lib = sys.argv[2]
    sys.path.insert(0, os.getcwd())
    import alphagpu_amd.lib as aglib
    if lib != "default":
        aglib.LIB_PATH = os.path.join(os.getcwd(), lib)
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    game = os.environ.get("GAME", "gobang")
    g = ag.GameSpec('gobang', 9, 5) if game == "gobang" else ag.GameSpec(game)
    H, T, V = int(os.environ.get("H", "128")), int(os.environ.get("T", "6")), int(os.environ.get("V", "64"))
    net = ag.SNetwork2.random(g, H, T)
    e = M.Engine(g, 32768, V, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=4 * 32768)
    e.set_network(net)
    e.set_profiling(1)
    out = []
    for L in (32768, 16384):
        e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0)
        e.kernel_times(reset=True)
        for _ in range(4):
            e.set_roots(None, L=L); e.search(V, cpuct=1.5, training=True, step=0)
        tree, nn, launches = e.kernel_times()
        out.append(f"L={L}: {tree / max(launches, 1):6.3f} ms")
    e.set_profiling(0)
    e.selfplay(32768, V, cpuct=1.5)
    rates = []
    for i in range(2):
        e.set_seed(3 + i)
        t0 = time.perf_counter(); st = e.selfplay(4 * 32768, V, cpuct=1.5, tau_plies=25); dt = time.perf_counter() - t0
        rates.append(st["rollouts"] / dt / 1e6)
    print(f"{lib:28s} {'  '.join(out)}   refilled 4 x 32768: {rates[0]:6.1f} / {rates[1]:6.1f} M rollouts/s   [{e.search_form()[0][:70]}]", flush=True)
    e.close()
else:
    for rep in range(int(os.environ.get("REPS", "2"))):
        for lib in sys.argv[1:]:
            subprocess.call([sys.executable, __file__, "--one", lib])
