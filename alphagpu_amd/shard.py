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


def allgather_records(local, n_local, rec_bytes, group=None):
    """local: uint8 tensor holding >= n_local*rec_bytes bytes (device or host).  Returns (gathered uint8 tensor
    [world, max_n*rec_bytes], counts int64 tensor [world]).  Two collectives: counts, then padded records."""
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
    out = torch.empty(world * need, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, local[:need].contiguous(), group=group)
    return out.view(world, need), counts


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
