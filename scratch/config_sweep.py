# BASELINE.json configs at reduced game counts: whole generations in bf16 mode, sanity of the statistics
import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfgs = [("connect4", 0, 0, 64, 128, 6), ("gobang", 9, 5, 64, 512, 8), ("hex", 9, 0, 128, 512, 8), ("reversi8", 0, 0, 64, 512, 8),
        ("reversi6", 0, 0, 64, 128, 6), ("gobang", 13, 5, 64, 256, 4), ("gobang", 3, 3, 16, 128, 6),
        ("hex", 9, 0, 128, 128, 6), ("gobang", 13, 5, 64, 128, 6), ("reversi8", 0, 0, 64, 128, 6), ("gobang", 9, 5, 64, 128, 6)]
for kind, n, nv, V, H, T in cfgs:
    g = ag.GameSpec(kind, n, nv)
    net = ag.SNetwork2.random(g, H, T)
    with M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        e.selfplay(L, V, cpuct=1.5, tau_plies=25)          # warm-up
        t0 = time.perf_counter()
        st = e.selfplay(L, V, cpuct=1.5, tau_plies=25)
        dt = time.perf_counter() - t0
    print(f"{kind}{n or ''} V={V} {H}x{T}: valid={st['valid']} faults={st['faults']} plies={st['plies']} samples={st['nsamples']} "
          f"W/D/L={st['wins']}/{st['draws']}/{st['losses']} rollouts/s={st['rollouts']/dt/1e6:.1f}M", flush=True)
