# single-rank RCCL check of the sample exchange code path (device tensors, asynchronous all_gather_into_tensor)
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M, shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = ag.GameSpec("gobang", 9, 5); net = ag.SNetwork2.random(g, 128, 2)
G, V = 2048, 16
eng = M.Engine(g, G, V, device=0, seed=1, game_id_base=shard.shard_base(0, G)); eng.set_network(net)
rb = g.rec_bytes
bufs = [torch.empty(G * g.max_plies * rb, dtype=torch.uint8, device="cuda") for _ in range(2)]
pend = [None, None]
for step in range(3):
    st = eng.selfplay(G, V, cpuct=1.5, tau_plies=25)
    k = step & 1
    if pend[k] is not None: pend[k].wait()
    n = eng.samples_packed_into(bufs[k].data_ptr(), G * g.max_plies)
    pend[k] = shard.allgather_records_async(bufs[k], n, rb)
for p in pend:
    if p is not None:
        out, counts = p.wait()
        n = int(counts[0].item())
        rec = shard.unpack_records(out[0].cpu().numpy(), n, g)
        assert n == st["nsamples"] and rec["ply"].min() == 0 and (rec["game_id"] < G).all()
torch.cuda.synchronize(); dist.barrier(); dist.destroy_process_group(); eng.close()
print("nccl single-rank exchange ok:", n, "records of", rb, "bytes")
