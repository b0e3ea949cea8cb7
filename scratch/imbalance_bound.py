# upper bound of what balancing the tree waves of a workgroup could gain: the four waves of every workgroup search the SAME eight
# games (game ids replicated), so that they arrive at the barriers together; compared with the ordinary search of 32768 distinct games
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L, V = 32768, 64
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net); e.set_profiling(1)
slots = np.arange(L)
for name, ids in (("distinct games", slots.astype(np.uint32)), ("4 waves of a workgroup share 8 games", ((slots // 32) * 8 + slots % 8).astype(np.uint32)),
                  ("all waves: the same 8 games", (slots % 8).astype(np.uint32))):
    for r in range(3):
        e.set_roots(None, L=L, game_ids=ids) if True else None
        e.kernel_times(reset=True)
        e.search(V, cpuct=1.5, training=True, step=0); e.synchronize()
        t, _, k = e.kernel_times()
    p, n, ro = e.counters()
    print(f"{name}: {t:.3f} ms  p/rollout {p/ro:.2f}")
e.close()
