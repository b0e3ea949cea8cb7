import sys, os
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
L = int(sys.argv[1]); H = int(sys.argv[2]); T = int(sys.argv[3]); V = int(sys.argv[4]) if len(sys.argv) > 4 else 64
game = sys.argv[5] if len(sys.argv) > 5 else "gobang"
g = ag.GameSpec(game, 9 if game in ("gobang", "hex") else 0, 5 if game == "gobang" else 0)
net = ag.SNetwork2.random(g, H, T)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16); e.set_network(net)
st = e.selfplay(L, V, cpuct=1.5, tau_plies=25)
print(L, H, T, V, game, os.environ.get("AGZ_CHAINS"), os.environ.get("AGZ_SMALL_MAXL"), "valid", st["valid"], "faults", st["faults"], "plies", st["plies"], e.search_form()[0][:40])
for i in range(2):
    e.set_seed(2 + i)
    if os.environ.get("PROF"): e.set_profiling(int(os.environ["PROF"]))
    st = e.selfplay(L, V, cpuct=1.5, tau_plies=25)
    print(" gen", i + 2, "valid", st["valid"], "faults", st["faults"], "plies", st["plies"])
