"""host side of the ply loop: wall time and CPU time of agz_selfplay with the host thread sleeping through the searches (default) and
spinning on the scan's word (AGZ_PLY_SPIN=1)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec('gobang', 9, 5)
net = ag.SNetwork2.random(g, 128, 6)
for label, env in (("sleep", {}), ("spin", {"AGZ_PLY_SPIN": "1"})):
    os.environ.pop("AGZ_PLY_SPIN", None)
    os.environ.update(env)
    e = M.Engine(g, 32768, 64, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=4 * 32768)
    e.set_network(net)
    e.selfplay(32768, 64, cpuct=1.5)
    for ng in (1, 1, 4, 4):
        e.set_seed(7 + ng)
        c0, t0 = time.process_time(), time.perf_counter()
        st = e.selfplay(ng * 32768, 64, cpuct=1.5, tau_plies=25)
        c1, t1 = time.process_time(), time.perf_counter()
        print(f"{label:5s} {ng} x 32768 games: {1e3 * (t1 - t0) / ng:7.1f} ms per generation  {st['rollouts'] / (t1 - t0) / 1e6:6.1f} M rollouts/s   host CPU {100 * (c1 - c0) / (t1 - t0):5.1f} % of a core   rounds {st['plies']}", flush=True)
    e.close()
