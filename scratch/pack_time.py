# where a 4-generation call with packed delivery spends its time: the call, the packing of its records, the D2H copy
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec('gobang', 9, 5); G = 32768; gp = 4
net = ag.SNetwork2.random(g, 128, 6)
e = M.Engine(g, G, 64, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=gp * G)
e.set_network(net)
e.selfplay(gp * G, 64, cpuct=1.5)
n = e.num_samples(); rb = g.rec_bytes
cap = int(n * 1.2)
dbuf = torch.empty(cap * rb, dtype=torch.uint8, device="cuda"); hbuf = torch.empty(cap * rb, dtype=torch.uint8).pin_memory()
for i in range(3):
    e.set_seed(5 + i)
    t0 = time.perf_counter(); st = e.selfplay(gp * G, 64, cpuct=1.5); t1 = time.perf_counter()
    n = e.samples_packed_into(dbuf.data_ptr(), cap); t2 = time.perf_counter()
    hbuf[: n * rb].copy_(dbuf[: n * rb]); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"call of {gp} generations {t1 - t0:.3f} s ({st['rollouts'] / (t1 - t0) / 1e6:.1f} M rollouts/s), pack {n} records {1e3 * (t2 - t1):.1f} ms, D2H {1e3 * (t3 - t2):.1f} ms ({n * rb / (t3 - t2) / 1e9:.1f} GB/s)", flush=True)
