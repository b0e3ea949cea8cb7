import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec("gobang", 3, 3); net = ag.SNetwork2.random(g, 128, 1)
L = 32768; cap = int(os.environ.get("CAP", "99304"))
with M.Engine(g, L, 8, seed=5, nn_mode=M.NN_BF16, sample_capacity_games=cap) as e:
    e.set_network(net)
    k0 = 0
    for ng, nxt in [(65536, 32768), (32768, 32768), (32768, 0)]:
        st = e.selfplay_chain(ng, nxt, 8, cpuct=1.5)
        s = e.samples()
        ids = np.unique(s["game_id"])
        miss = np.setdiff1d(np.arange(k0, k0 + ng, dtype=np.uint32), ids)
        print(f"call k0={k0} ng={ng} nxt={nxt}: nsamples {st['nsamples']} W/D/L {st['wins']}/{st['draws']}/{st['losses']} plies {st['plies']} ids {ids.min()}..{ids.max()} ({len(ids)} games), missing {len(miss)}" + (f" [{miss.min()}..{miss.max()}]" if len(miss) else ""), flush=True)
        k0 += ng
