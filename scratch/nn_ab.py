import sys, os, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
def run(L):
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    g = ag.GameSpec('gobang', 9, 5)
    net = ag.SNetwork2.random(g, 128, 6)
    e = M.Engine(g, L, 64, seed=1, nn_mode=M.NN_BF16)
    e.set_network(net); e.set_roots(None, L=L)
    e.search(64, cpuct=1.5, training=True, step=0); e.synchronize()
    out = np.concatenate([e.policy().ravel(), e.root_visits().ravel(), e.root_q().ravel()])
    e.close(); return out
if len(sys.argv) > 2:
    np.save(sys.argv[2], run(int(sys.argv[1])))
else:
    L = sys.argv[1]
    for tag, env in (("wave", {}), ("f3", {"AGZ_NN_WAVE_MAXL": "0"})):
        subprocess.check_call([sys.executable, __file__, L, f"/tmp/nn_{tag}.npy"], env={**os.environ, **env})
    a, b = np.load("/tmp/nn_wave.npy"), np.load("/tmp/nn_f3.npy")
    print("L", L, "identical:", np.array_equal(a, b), "maxdiff", float(np.abs(a - b).max()))
