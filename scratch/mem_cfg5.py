"""HBM of one rank of BASELINE config 5 (Reversi 8x8, 32768 slots, V = 64, 512x8; bench.py's engine: sample store for 1 + 2 generations in flight) and of
its exchange buffers at 8 ranks (capacity = 32768 games x the longest game): hipMemGetInfo before / after."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
g = ag.GameSpec("reversi8", 0, 0)
free0, tot = torch.cuda.mem_get_info()
e = M.Engine(g, 32768, 64, seed=1, nn_mode=M.NN_BF16, sample_capacity_games=3 * 32768)
e.set_network(ag.SNetwork2.random(g, 512, 8))
free1, _ = torch.cuda.mem_get_info()
cap = 32768 * g.max_plies
blk = 32 + cap * g.rec_bytes
print("engine (32768 slots, V=64, samples of 3 x 32768 games): %.2f GB;  rec_bytes %d, max_plies %d;  exchange buffers at 8 ranks, capacity %d records: 2 x 9 x %.3f GB = %.2f GB;  device total %.0f GB"
      % ((free0 - free1) / 1e9, g.rec_bytes, g.max_plies, cap, blk / 1e9, 18 * blk / 1e9, tot / 1e9))
e.close()
