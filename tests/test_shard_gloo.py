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
    # the pipelined form (what bench.py runs): one asynchronous collective per generation, the count in the buffer's header, no
    # read-back when it is issued.  Four "generations", each waited for after the next one was issued (double buffering): the first two
    # are sent with the largest count of their own exchange (gathered first: nothing to predict from yet), the next two with the count agreed from the gathered counts of the first; the last outgrows that
    # prediction on rank 1 only and is completed by the second (blocking) step inside wait() -- every one must deliver exactly the records each rank packed.
    rb = game.rec_bytes
    ex = shard.RecordExchange(40 * game.max_plies, rb, slack=0.0)
    sent, pipelined_ok = [], True
    pend = []
    for gen, ng in enumerate((6, 6, 6, 6 if rank == 0 else 40)):
        sg = O.selfplay(og, net, ng, 8, 1.5, 25, 11 + gen, shard.shard_base(rank, 40))
        buf = ex.new_buffer("cpu")
        pk = torch.from_numpy(_pack(sg, game))
        buf[shard.HEADER: shard.HEADER + pk.numel()] = pk
        pend.append((ex.start(buf, sg["n"]), pk, sg["n"]))
        sent.append(pend[-1][0].sent)
        if gen == 0:                       # double buffering: generation k is waited for after generation k + 1 was issued ...
            continue
        pg, pk0, n0 = pend[gen - 1]
        parts_g, counts_g = pg.wait()
        pipelined_ok &= int(counts_g[rank]) == n0 and bool((parts_g[rank] == pk0).all())
    pg, pk0, n0 = pend[-1]
    parts_g, counts_g = pg.wait()
    pipelined_ok &= int(counts_g[rank]) == n0 and bool((parts_g[rank] == pk0).all()) and int(counts_g.max()) == int(counts_g[1])
    if rank == 0:
        q.put(dict(merged=dict(merged), sent=sent, tails=ex.tails, ok=pipelined_ok, cap=ex.cap))
    else:
        q.put(dict(ok=pipelined_ok))
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
    res = [q.get(timeout=120), q.get(timeout=120)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r["ok"] for r in res)
    r0 = [r for r in res if "merged" in r][0]
    merged = r0["merged"]
    # generations 0 and 1 are issued before any collective has been waited for: the ranks gather their counts first (the blocking step of
    # a run's first exchanges) and send the largest, not the capacity; 2 and 3 travel with the count agreed from the gathered counts
    # without any read-back; only generation 3 needs the second step
    cap, sent = r0["cap"], r0["sent"]
    assert 0 < sent[0] < cap and 0 < sent[1] < cap and sent[2] < cap and sent[3] == sent[2] and r0["tails"] == 1, (sent, cap, r0["tails"])
    og = O.make_game("gobang", 3, 3)
    ref = O.selfplay(og, O.OracleNet(og, 16, 1), 12, 8, 1.5, 25, 11, 0)
    assert len(merged["ply"]) == ref["n"]
    for k in ("game_id", "ply", "move", "player", "state", "fstate", "value", "policy"):
        assert np.array_equal(merged[k], ref[k]), k


def _worker8(rank, world, port, q):
    """eight ranks, blocking exchange only: one shard of Connect4-free TicTacToe games per rank -> the un-sharded run"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    game = ag.GameSpec("gobang", 3, 3)
    og = O.make_game("gobang", 3, 3)
    net = O.OracleNet(og, 16, 1)
    G = 3
    s = O.selfplay(og, net, G, 8, 1.5, 25, 21, shard.shard_base(rank, G))
    out, counts = shard.allgather_records(torch.from_numpy(_pack(s, game)), s["n"], game.rec_bytes)
    if rank == 0:
        merged = shard.merge_poolsample_order([shard.unpack_records(out[r].numpy(), int(counts[r]), game) for r in range(world)])
        q.put(dict(merged=dict(merged), counts=[int(c) for c in counts]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_allgather_equals_unsharded_run():
    """BASELINE config 5's shape of the exchange (8 shards, one all-gather at generation end) on CPU: gloo, world size 8."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    og = O.make_game("gobang", 3, 3)
    ref = O.selfplay(og, O.OracleNet(og, 16, 1), 24, 8, 1.5, 25, 21, 0)
    merged = res["merged"]
    assert sum(res["counts"]) == ref["n"] == len(merged["ply"]) and len(res["counts"]) == 8
    for k in ("game_id", "ply", "move", "player", "state", "fstate", "value", "policy"):
        assert np.array_equal(merged[k], ref[k]), k
