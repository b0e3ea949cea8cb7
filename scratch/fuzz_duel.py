# Larger randomized duel parity runs than the pytest suite affords: mcts(actor1, actor2, ...) (mcts_gpu.jl:581-651) in the bf16 mode
# (whole-search kernels) against the oracle's duel with the bf16 MFMA model: W/D/L and every move.  Test infrastructure (uses oracle/).
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O
off = int(os.environ.get("FUZZ_SEED_OFFSET", "0"))
cases = [("gobang9", 300, 32, 128, 6, 15, 0, 31), ("connect4", 600, 32, 128, 6, 15, 1, 32), ("reversi8", 200, 32, 128, 2, 15, 0, 33),
         ("hex9", 100, 64, 128, 2, 15, 1, 34), ("gobang9", 64, 32, 512, 8, 15, 1, 35), ("tictactoe", 2000, 16, 128, 6, 15, 0, 36)]
bad = 0
for name, n, V, H, T, tau, first, seed in cases:
    seed += off
    kind, nn, k = common.GAMES[name]
    g, og = ag.GameSpec(kind, nn, k), O.make_game(kind, nn, k)
    a, oa = ag.SNetwork2.random(g, H, T, 11 + seed), O.OracleNet(og, H, T, 11 + seed)
    b, ob = ag.SNetwork2.random(g, H, T, 22 + seed), O.OracleNet(og, H, T, 22 + seed)
    t0 = time.perf_counter()
    ref = O.duel(og, oa.bf16(), ob.bf16(), n, V, 2.0, tau, seed, 1000 * seed, first)
    t1 = time.perf_counter()
    with M.Engine(g, n, V, seed=seed, game_id_base=1000 * seed, nn_mode=M.NN_BF16) as e:
        e.set_network(a, 0); e.set_network(b, 1)
        wdl = e.duel(n, V, cpuct=2.0, tau_plies=tau, first=first)
        s = e.samples()
    moves = np.full_like(ref["moves"], -1)
    moves[s["game_id"].astype(np.int64) - 1000 * seed, s["ply"]] = s["move"]
    ok = ref["rc"] == 0 and list(wdl) == list(ref["wdl"]) and np.array_equal(moves, ref["moves"])
    bad += not ok
    print(f"{name} n={n} V={V} {H}x{T} first={first} seed={seed}: W/D/L {list(wdl)} vs {list(ref['wdl'])}  moves differ in {int((moves != ref['moves']).any(axis=1).sum())} games  "
          f"{'IDENTICAL' if ok else 'DIFFERENT'}  (oracle {t1 - t0:.1f}s)", flush=True)
print("duel fuzz:", "all identical" if not bad else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)
