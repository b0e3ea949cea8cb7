# whole generations at full size on the whole-search kernel for every game shape (128-wide networks): stability / "faute" check
import sys, os, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfgs = [("connect4", 0, 0, 64, 6), ("gobang", 9, 5, 64, 6), ("hex", 9, 0, 128, 4), ("reversi8", 0, 0, 64, 4), ("reversi6", 0, 0, 48, 2),
        ("gobang", 13, 5, 32, 2), ("hex", 11, 0, 40, 2), ("gobang", 3, 3, 16, 6), ("hex", 5, 0, 24, 1), ("gobang", 7, 4, 50, 3)]
for kind, n, nv, V, T in cfgs:
    g = ag.GameSpec(kind, n, nv)
    net = ag.SNetwork2.random(g, 128, T)
    with M.Engine(g, L, V, seed=7, nn_mode=M.NN_BF16) as e:
        e.set_network(net)
        t0 = time.perf_counter()
        st = e.selfplay(L, V, cpuct=1.5, tau_plies=25)
        dt = time.perf_counter() - t0
        assert st["valid"] and st["faults"] == 0 and st["wins"] + st["draws"] + st["losses"] == L, st
    print(f"{kind}{n or ''} V={V} 128x{T} L={L}: plies={st['plies']} samples={st['nsamples']} W/D/L={st['wins']}/{st['draws']}/{st['losses']} "
          f"{st['rollouts']/dt/1e6:.1f}M rollouts/s ({dt:.2f}s)", flush=True)
print("stress ok")
