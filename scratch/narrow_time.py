"""A/B of the one-launch search forms on one box: first-ply search time at several batch sizes and whole generations, per environment
setting (the switches are read by agz_create, so one process can walk through them).
usage: python scratch/narrow_time.py [game] ; game = gobang | connect4"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M

game = sys.argv[1] if len(sys.argv) > 1 else "gobang"
g = ag.GameSpec('gobang', 9, 5) if game == "gobang" else ag.GameSpec(game)
net = ag.SNetwork2.random(g, 128, 6)
V = 64
ENVS = [("base", {"AGZ_NARROW": "-1"}), ("G=4", {"AGZ_NARROW": "4", "AGZ_NARROW_MINL": "0"})]
if game == "connect4":
    ENVS.append(("G=2", {"AGZ_NARROW": "2", "AGZ_NARROW_MINL": "0"}))
KEYS = ("AGZ_NARROW", "AGZ_NARROW_MINL", "AGZ_NARROW_OCC")
sizes = [int(x) for x in os.environ.get("SIZES", "32768,24576,16384,8192,2048").split(",")]
for label, env in ENVS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    e = M.Engine(g, 32768, V, seed=1, nn_mode=M.NN_BF16)
    e.set_network(net)
    e.set_profiling(1)
    for L in sizes:
        e.set_roots(None, L=L)
        e.search(V, cpuct=1.5, training=True, step=0)
        e.kernel_times(reset=True)
        for _ in range(3):
            e.set_roots(None, L=L)
            e.search(V, cpuct=1.5, training=True, step=0)
        tree, nn, launches = e.kernel_times()
        print(f"{game} {label:5s} L={L:6d}: {tree / max(launches, 1):7.3f} ms per search   [{e.search_form()[0][:90]}]", flush=True)
    e.set_profiling(0)
    for i in range(3):
        e.set_seed(1 + i)
        t0 = time.perf_counter()
        st = e.selfplay(32768, V, cpuct=1.5, tau_plies=25)
        dt = time.perf_counter() - t0
        print(f"{game} {label:5s} generation {i}: {dt * 1e3:7.1f} ms  {st['rollouts'] / dt / 1e6:6.1f} M rollouts/s  plies {st['plies']} samples {st['nsamples']}", flush=True)
    e.close()
