# Larger randomized parity runs than the pytest suite affords: whole self-play generations in the benchmarked bf16 mode against the
# oracle's generation (bf16 MFMA model), bit for bit, for several games / seeds / network sizes.  Test infrastructure (uses oracle/).
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O

cases = [("gobang9", 640, 64, 128, 6, 11), ("gobang9", 320, 64, 128, 6, 12), ("connect4", 1200, 64, 128, 6, 13), ("hex9", 200, 128, 128, 2, 14),
         ("reversi8", 400, 64, 128, 2, 15), ("reversi6", 600, 48, 128, 3, 16), ("gobang9", 96, 64, 512, 8, 17), ("reversi8", 96, 32, 512, 8, 18),
         ("hex9", 64, 128, 512, 4, 19), ("tictactoe", 3000, 16, 128, 6, 20)]
if os.environ.get("FUZZ_SET") == "2":                          # other shapes: V = 128 on Gobang, 256-wide trunks, 13x13, Hex 11x11, wide Connect4
    cases = [("gobang9", 200, 128, 128, 6, 41), ("gobang13", 96, 64, 256, 3, 42), ("hex11", 120, 40, 128, 2, 43), ("connect4", 200, 64, 512, 8, 44),
             ("reversi6", 200, 64, 512, 2, 45), ("gobang9", 160, 64, 256, 4, 46), ("hex5", 1500, 24, 128, 1, 47), ("gobang13", 64, 64, 128, 2, 48)]
if os.environ.get("FUZZ_SET") == "3":                          # thousands of games with cheap searches: every sparse-wave shape (1..8 games per wave,
    cases = [("connect4", 7000, 16, 128, 1, 51), ("tictactoe", 12000, 8, 128, 1, 52), ("reversi6", 3000, 12, 128, 1, 53)]   # 16- and 32-game workgroups) on the way down
off = int(os.environ.get("FUZZ_SEED_OFFSET", "0"))             # other seeds (networks, roots, uniforms): FUZZ_SEED_OFFSET=100 ...
cases = [(a, b, c, d, e, f + off) for a, b, c, d, e, f in cases]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[5]) in sys.argv[1:] or c[0] in sys.argv[1:]]
bad = 0
for name, n, V, H, T, seed in cases:
    kind, nn, k = common.GAMES[name]
    g, og = ag.GameSpec(kind, nn, k), O.make_game(kind, nn, k)
    net, onet = ag.SNetwork2.random(g, H, T, 0x5EED + seed), O.OracleNet(og, H, T, 0x5EED + seed)
    t0 = time.perf_counter()
    exact = os.environ.get("FUZZ_EXACT") == "1"                # the bit-exact fp32 mode instead of the benchmarked bf16 mode
    ref = O.selfplay(og, onet if exact else onet.bf16(), n, V, 1.5, 25, seed, 1000 * seed)
    t1 = time.perf_counter()
    slots = max(8, n // int(os.environ.get("FUZZ_SLOT_DIV", "1")))   # FUZZ_SLOT_DIV=3: a third of the games in flight, finished games' slots refilled
    with M.Engine(g, min(slots, n), V, seed=seed, game_id_base=1000 * seed, nn_mode=M.NN_EXACT if exact else M.NN_BF16, sample_capacity_games=n) as e:
        e.set_network(net)
        form = ""
        if os.environ.get("FUZZ_CHAIN") == "1":                # the n games as a chain of three calls (agz_selfplay_chain), merged back into PoolSample order
            parts, st, sizes = [], None, [n // 2, n // 4, n - n // 2 - n // 4]
            for i, ng in enumerate(sizes):
                sti = e.selfplay_chain(ng, sizes[i + 1] if i + 1 < len(sizes) else 0, V, cpuct=1.5, tau_plies=25)
                parts.append(e.samples())
                st = sti if st is None else {k: (st[k] + sti[k] if k in ("nsamples", "wins", "draws", "losses", "rollouts", "total_plies") else (st[k] and sti[k] if k == "valid" else sti[k])) for k in sti}
            cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
            order = np.lexsort((cat["game_id"], cat["ply"]))
            s = {k: v[order] for k, v in cat.items()}
        else:
            st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
        form = e.search_form()[0].split(" (")[0]
    ok = st["valid"] and ref["rc"] == 0 and st["nsamples"] == ref["n"]
    diff = {}
    if ok:
        for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
            a, b = np.ascontiguousarray(s[key]), np.ascontiguousarray(ref[key])
            same = a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
            if not same:
                diff[key] = int((a.view(np.uint8) != b.view(np.uint8)).sum()) if a.shape == b.shape else -1
                if a.shape == b.shape and a.ndim == 2:
                    for i, j in np.argwhere(a.view(np.uint32 if a.dtype == np.float32 else a.dtype) != b.view(np.uint32 if b.dtype == np.float32 else b.dtype))[:8]:
                        print(f"   {key}[{i},{j}]: gpu {a[i, j]!r} ({a[i, j].view(np.uint32) if a.dtype == np.float32 else ''})  oracle {b[i, j]!r} "
                              f"({b[i, j].view(np.uint32) if b.dtype == np.float32 else ''})  game {s['game_id'][i]} ply {s['ply'][i]} move {s['move'][i]}")
    bad += (not ok) or bool(diff)
    print(f"{name} n={n} V={V} {H}x{T} seed={seed} [{form}]: samples {st['nsamples']} vs {ref['n']}  W/D/L {st['wins']}/{st['draws']}/{st['losses']}  "
          f"{'IDENTICAL' if ok and not diff else 'DIFFERENT ' + str(diff)}  (oracle {t1 - t0:.1f}s)", flush=True)
print("fuzz:", "all identical" if not bad else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)
