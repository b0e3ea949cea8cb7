"""Where the waves of the persistent self-play kernel spend their cycles (-DAGZ_PSTAMPS build, scratch/libagz_ps.so): prints the split
search / ply step / flag barriers of a refilled call on the headline shape."""
import sys, os
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', 'libagz_ps.so')
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
V, L = 64, 32768
g = ag.GameSpec('gobang', 9, 5); net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, L, V, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=6 * L); e.set_network(net)
st = e.selfplay(4 * L, V, cpuct=1.5, tau_plies=25)
print("rollouts/s %.1f M" % (st["rollouts"] / st["total_seconds"] / 1e6), "samples per game %.1f" % (st["nsamples"] / (4.0 * L)), e.search_form()[0][:80])
e.close()
