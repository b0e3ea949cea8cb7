"""N>1 path on CPU: world_size-2 gloo all-gather of packed sample records + PoolSample-order merge.
The records are produced by the oracle's self-play for two game-id shards and must merge into exactly
the samples of one un-sharded run (results are keyed by game id, never by rank or slot)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import alphagpu_amd as ag
from alphagpu_amd import shard
import oracle_lib as O


def _pack(s, game):
    n, rb, A, VS, FS = s["n"], game.rec_bytes, game.A, game.VS, game.FS
    a = np.zeros((n, rb), np.uint8)
    a[:, 0:4] = s["game_id"].view(np.uint8).reshape(n, 4)
    a[:, 4:8] = s["ply"].view(np.uint8).reshape(n, 4)
    a[:, 8:12] = s["move"].view(np.uint8).reshape(n, 4)
    a[:, 12:16] = s["value"].view(np.uint8).reshape(n, 4)
    a[:, 16] = s["player"].view(np.uint8)
    a[:, 20:20 + 4 * A] = s["policy"].view(np.uint8).reshape(n, 4 * A)
    a[:, 20 + 4 * A:20 + 4 * A + 2 * VS] = s["state"].view(np.uint8)
    a[:, 20 + 4 * A + 2 * VS:20 + 4 * A + 2 * VS + FS] = s["fstate"].view(np.uint8)
    return a.reshape(-1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    game = ag.GameSpec("gobang", 3, 3)
    og = O.make_game("gobang", 3, 3)
    net = O.OracleNet(og, 16, 1)
    G = 6
    s = O.selfplay(og, net, G, 8, 1.5, 25, 11, shard.shard_base(rank, G))
    local = torch.from_numpy(_pack(s, game))
    out, counts = shard.allgather_records(local, s["n"], game.rec_bytes)
    parts = [shard.unpack_records(out[r].numpy(), int(counts[r]), game) for r in range(world)]
    merged = shard.merge_poolsample_order(parts)
    if rank == 0:
        q.put({k: v for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_equals_unsharded_run():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    og = O.make_game("gobang", 3, 3)
    ref = O.selfplay(og, O.OracleNet(og, 16, 1), 12, 8, 1.5, 25, 11, 0)
    assert len(merged["ply"]) == ref["n"]
    for k in ("game_id", "ply", "move", "player", "state", "fstate", "value", "policy"):
        assert np.array_equal(merged[k], ref[k]), k
