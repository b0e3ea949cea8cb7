"""Multi-GPU sharding of a self-play generation.

Games are independent, so rank r of W owns game ids [r*G, (r+1)*G) with its own engine, stream and network
replica; nothing is exchanged during the generation.  The ONE exchange step is an all-gather of the packed
sample records at generation end (RCCL over xGMI on GPUs: torch.distributed backend "nccl"; "gloo" on CPU
for tests).  Because every uniform is keyed by the global game id, the gathered samples are identical to a
single-GPU run over all W*G games.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_base(rank, games_per_rank):
    """game_id_base of a rank."""
    return int(rank) * int(games_per_rank)


class PendingGather:
    """An all-gather of sample records in flight.  `wait()` returns (gathered [world, max_n*rec_bytes], counts) once the
    collective has completed for the HOST (the engine writes the source buffer on its own stream, so stream-ordered
    completion in torch's sense is not enough to reuse it)."""

    def __init__(self, work, out, counts, src):
        self.work, self.out, self.counts, self.src = work, out, counts, src     # src kept alive until completion

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.out.is_cuda:
                torch.cuda.current_stream(self.out.device).synchronize()
            self.work = None
            self.src = None
        return self.out, self.counts


def allgather_records_async(local, n_local, rec_bytes, group=None):
    """Start the exchange step: counts (tiny, blocking), then the padded records as an asynchronous collective, so that
    the next generation's kernels overlap the transfer (xGMI is per-link bound: a ring all-gather of 8 x 0.75 GB takes
    ~0.1 s, a fifth of a generation).  local: uint8 tensor with >= n_local*rec_bytes bytes (device or host); it must not
    be modified until wait() returns."""
    world = dist.get_world_size(group)
    dev = local.device
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    max_n = int(counts.max().item())
    need = max_n * rec_bytes
    if local.numel() < need:
        pad = torch.zeros(need, dtype=torch.uint8, device=dev)
        pad[: local.numel()] = local
        local = pad
    src = local[:need].contiguous()
    out = torch.empty(world * need, dtype=torch.uint8, device=dev)
    work = dist.all_gather_into_tensor(out, src, group=group, async_op=True)
    return PendingGather(work, out.view(world, need), counts, src)


def allgather_records(local, n_local, rec_bytes, group=None):
    """local: uint8 tensor holding >= n_local*rec_bytes bytes (device or host).  Returns (gathered uint8 tensor
    [world, max_n*rec_bytes], counts int64 tensor [world]).  Two collectives: counts, then padded records."""
    return allgather_records_async(local, n_local, rec_bytes, group).wait()


def unpack_records(buf, n, game):
    """uint8 array of n packed records (agz.h agz_get_samples_packed layout) -> dict of arrays."""
    rb, A, VS, FS = game.rec_bytes, game.A, game.VS, game.FS
    a = np.ascontiguousarray(buf[: n * rb]).reshape(n, rb)
    hdr = a[:, :20].copy()
    return dict(
        game_id=hdr[:, 0:4].copy().view(np.uint32).reshape(n), ply=hdr[:, 4:8].copy().view(np.int32).reshape(n),
        move=hdr[:, 8:12].copy().view(np.int32).reshape(n), value=hdr[:, 12:16].copy().view(np.float32).reshape(n),
        player=hdr[:, 16].copy().view(np.int8).reshape(n),
        policy=a[:, 20:20 + 4 * A].copy().view(np.float32).reshape(n, A),
        state=a[:, 20 + 4 * A:20 + 4 * A + 2 * VS].copy().view(np.int8).reshape(n, 2 * VS),
        fstate=a[:, 20 + 4 * A + 2 * VS:20 + 4 * A + 2 * VS + FS].copy().view(np.int8).reshape(n, FS))


def merge_poolsample_order(parts):
    """Concatenate per-rank sample dicts and restore the reference's PoolSample push order
    (ply-major, then game id — mcts_gpu.jl:513-516 with order-preserving compaction :550-553)."""
    keys = parts[0].keys()
    cat = {k: np.concatenate([p[k] for p in parts]) for k in keys}
    order = np.lexsort((cat["game_id"], cat["ply"]))
    return {k: v[order] for k, v in cat.items()}
