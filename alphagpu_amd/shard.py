"""Multi-GPU sharding of a self-play generation.

Games are independent, so rank r of W owns game ids [r*G, (r+1)*G) with its own engine, stream and network
replica; nothing is exchanged during the generation.  The ONE exchange step is an all-gather of the packed
sample records at generation end (RCCL over xGMI on GPUs: torch.distributed backend "nccl"; "gloo" on CPU
for tests).  Because every uniform is keyed by the global game id, the gathered samples are identical to a
single-GPU run over all W*G games.

The exchange itself lives behind the C ABI (include/agz.h agz_comm_*: libagz binds RCCL and owns the gather buffers, allocated once);
`CommExchange` below is a thin caller of it — what a Julia host does through julia/AlphaGPUAMD.jl `mcts_sharded`.  The torch.distributed
forms remain for the CPU tests (gloo) and as the reference the C-ABI path is compared with.

Forms of the exchange:
  * `CommExchange`       — through the C ABI (RCCL inside libagz): blocking `allgather()` or pipelined `start()` / `wait()`;
  * `allgather_records`  — blocking: counts, then the records padded to the largest count (tests, one-off callers);
  * `RecordExchange`     — pipelined: ONE asynchronous collective per generation that carries the rank's record count in a
    16-byte header in front of its records, issued WITHOUT any host synchronisation or blocking read (the host never waits
    for another rank between two generations); the counts are read when the caller waits for the collective.
"""
import numpy as np
import torch
import torch.distributed as dist

HEADER = 16          # bytes in front of a rank's records in a RecordExchange buffer: int64 record count, int64 reserved


def shard_base(rank, games_per_rank):
    """game_id_base of a rank."""
    return int(rank) * int(games_per_rank)


def allgather_records(local, n_local, rec_bytes, group=None):
    """Blocking exchange.  local: uint8 tensor holding >= n_local*rec_bytes bytes (device or host).  Returns (parts, counts):
    parts[r] = uint8 tensor with the counts[r]*rec_bytes bytes of rank r's records.  Two collectives: counts, then the
    records padded to the largest count."""
    world = dist.get_world_size(group)
    dev = local.device
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.cpu()
    need = int(counts.max()) * rec_bytes
    if local.numel() < need:
        pad = torch.zeros(need, dtype=torch.uint8, device=dev)
        pad[: local.numel()] = local
        local = pad
    out = torch.empty(world * need, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, local[:need].contiguous(), group=group)
    out = out.view(world, need)
    return [out[r, : int(counts[r]) * rec_bytes] for r in range(world)], counts


class PendingGather:
    """A RecordExchange collective in flight.  `wait()` returns (parts, counts) once the collective has completed for the
    HOST (the engine writes the source buffer on its own stream, so stream-ordered completion in torch's sense is not
    enough to reuse it)."""

    def __init__(self, ex, work, out, src, sent, units=1):
        self.ex, self.work, self.out, self.src, self.sent, self.units = ex, work, out, src, sent, units
        self.result = None

    def wait(self):
        if self.result is not None:
            return self.result
        ex, rb = self.ex, self.ex.rb
        self.work.wait()
        if self.out.is_cuda:
            torch.cuda.current_stream(self.out.device).synchronize()
        world = self.out.shape[0]
        counts = self.out[:, :8].contiguous().view(torch.int64).reshape(world).cpu()
        max_n = int(counts.max())
        if max_n > ex.cap:
            raise RuntimeError(f"a rank produced {max_n} records, more than the exchange capacity {ex.cap}")
        tail = None
        if max_n > self.sent:
            # some rank produced more records than every rank agreed to send in the first collective (the prediction from the
            # previous generations was too small): every rank sees the same counts, so all of them take this branch together
            # and gather the rest, blocking.  Rare by construction (slack over the largest count seen so far).
            tn = (max_n - self.sent) * rb
            tail = torch.empty(world, tn, dtype=torch.uint8, device=self.out.device)
            dist.all_gather_into_tensor(tail.view(-1), self.src[HEADER + self.sent * rb: HEADER + max_n * rb].contiguous(), group=ex.group)
            if tail.is_cuda:
                torch.cuda.current_stream(tail.device).synchronize()
            ex.tails += 1
        parts = []
        for r in range(world):
            n = int(counts[r])
            if n <= self.sent:
                parts.append(self.out[r, HEADER: HEADER + n * rb])
            else:
                parts.append(torch.cat([self.out[r, HEADER: HEADER + self.sent * rb], tail[r, : (n - self.sent) * rb]]))
        ex._observe(max_n, self.units)
        self.work = self.src = None
        self.result = (parts, counts)
        return self.result


class RecordExchange:
    """The exchange step of a generation as one asynchronous all-gather, issued without reading anything back.

    Every rank sends `HEADER + sent * rec_bytes` bytes where `sent` is a record count ALL ranks agree on — for the first exchange
    of a run the largest count of that exchange (one blocking gather of the counts), from then on without talking to each other:
    the largest count any rank has produced
    in a generation whose collective has been waited for, plus a slack, rounded up — derived only from gathered counts, which
    are identical on every rank, and updated inside wait(), which every rank calls in the same order.  A rank that produces
    more than `sent` records is completed by a second (blocking) collective inside wait().  The record count of the rank
    travels in the header, so the host reads no count before the collective is issued."""

    def __init__(self, capacity_records, rec_bytes, group=None, slack=1.0 / 32):
        self.cap, self.rb, self.group, self.slack = int(capacity_records), int(rec_bytes), group, float(slack)
        self.seen_max = None
        self.tails = 0               # collectives that needed the second step (diagnostics)

    def buffer_bytes(self):
        return HEADER + self.cap * self.rb

    def new_buffer(self, device):
        """A send buffer: the engine writes its packed records at data_ptr() + HEADER (agz_get_samples_packed)."""
        return torch.zeros(self.buffer_bytes(), dtype=torch.uint8, device=device)

    def agreed_count(self, units=1):
        if self.seen_max is None:
            return self.cap
        n = int(self.seen_max * units * (1.0 + self.slack)) + 64
        return min(self.cap, (n + 255) & ~255)

    def _observe(self, max_n, units=1):
        r = max_n / float(units)
        self.seen_max = r if self.seen_max is None else max(self.seen_max, r)

    def start(self, buf, n_local, units=1):
        """buf: a new_buffer() tensor whose records region holds n_local records; it must not be modified until wait().
        units: what the record count scales with (games of the generation; the same on every rank): the agreed size is predicted
        from the largest count PER UNIT seen so far, so that calls of different sizes share one prediction.
        A rank whose records exceed the capacity still takes part in the collective (its count travels in the header): wait() raises on
        EVERY rank — they all see the same counts — instead of one rank bailing out here and the others blocking in the all-gather."""
        world = dist.get_world_size(self.group)
        buf[:8].view(torch.int64).fill_(int(n_local))          # (a fill with a scalar argument: nothing is read back, no host buffer)
        if self.seen_max is None:
            # the FIRST exchange of a run has no count to predict from: the ranks gather their counts (the one blocking read of the run)
            # and send exactly the largest — not the capacity, which for calls of several generations is W x 12 GB of gathered records
            cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=buf.device)
            counts = torch.zeros(world, dtype=torch.int64, device=buf.device)
            dist.all_gather_into_tensor(counts, cnt, group=self.group)
            sent = min(self.cap, (int(counts.max()) + 255) & ~255)
        else:
            sent = self.agreed_count(units)
        nbytes = HEADER + sent * self.rb
        out = torch.empty(world, nbytes, dtype=torch.uint8, device=buf.device)
        work = dist.all_gather_into_tensor(out.view(-1), buf[:nbytes], group=self.group, async_op=True)
        return PendingGather(self, work, out, buf, sent, units)


class CommExchange:
    """The exchange step through the C ABI: agz_comm_create / agz_allgather_samples* (RCCL bound by libagz, buffers allocated once:
    2 x (1 + world) x (16 + capacity x rec_bytes) bytes per rank).  The 128-byte RCCL id is made on rank 0 and shipped to the other ranks
    by `broadcast` (default: torch.distributed's object broadcast over the initialised process group — any backend — when world > 1).

    allgather()            -> (parts, counts): blocking, counts then records
    start(units) / wait()  -> pipelined: one collective per call, every rank sends the record count all ranks agree on (predicted from
                              the counts of the exchanges waited for so far, like RecordExchange); parts[r] are host uint8 arrays."""

    def __init__(self, engine, rank, world, capacity_records, broadcast=None, slack=1.0 / 32):
        import ctypes as C
        self.C, self.e, self.L = C, engine, engine.L
        self.rank, self.world, self.cap, self.rb = int(rank), int(world), int(capacity_records), int(engine.game.rec_bytes)
        self.slack, self.seen_max, self.units, self.tails = float(slack), None, [], 0
        self.last_statuses = np.zeros(self.world, np.int32)
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            rc = self.L.agz_comm_unique_id(C.byref(uid))
            if rc:
                raise RuntimeError("agz_comm_unique_id: " + (self.L.agz_comm_last_error(None) or b"").decode())
        blob = bytes(uid)
        if self.world > 1:
            if broadcast is None:
                def broadcast(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            blob = broadcast(blob)
        self.h = C.c_void_p()
        # (RCCL prints a version banner on the C-level stdout when a communicator is made: it goes to stderr — a caller's stdout may be a
        #  protocol, e.g. bench.py's ONE JSON line)
        import os
        import sys
        sys.stdout.flush()
        libc, saved = C.CDLL(None), os.dup(1)
        os.dup2(2, 1)
        try:
            rc = self.L.agz_comm_create(engine.h, self.rank, self.world, blob, self.cap, C.byref(self.h))
        finally:
            libc.fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        if rc:
            raise RuntimeError("agz_comm_create: " + (self.L.agz_comm_last_error(None) or b"").decode())

    def close(self):
        if self.h:
            self.L.agz_comm_destroy(self.h)
            self.h = None

    def _chk(self, rc, what):
        if rc:
            raise RuntimeError(what + ": " + (self.L.agz_comm_last_error(self.h) or b"").decode())

    def _parts(self, counts):
        parts = []
        for r in range(self.world):
            a = np.empty(int(counts[r]) * self.rb, np.uint8)
            self._chk(self.L.agz_comm_fetch_records(self.h, r, a.ctypes.data_as(self.C.c_void_p), 0, int(counts[r])), "agz_comm_fetch_records")
            parts.append(a)
        return parts

    def allgather(self, status=0, fetch=True):
        """Blocking exchange (counts, then records).  status: the return code of this rank's self-play call — a rank whose call failed
        STILL calls this (with its code; it sends no records): see statuses()."""
        counts, st = (self.C.c_int64 * self.world)(), (self.C.c_int32 * self.world)()
        self._chk(self.L.agz_allgather_samples_status(self.e.h, self.h, int(status), counts, st), "agz_allgather_samples_status")
        self.last_statuses = np.array(st[:], np.int32)
        counts = np.array(counts[:], np.int64)
        return (self._parts(counts) if fetch else None), counts

    def statuses(self):
        """the status words every rank sent with the exchange last waited for (0 = its self-play call succeeded): identical on all ranks,
        so that they fail, or go on, together"""
        return self.last_statuses

    def fetch_last(self, counts):
        """the records of the exchange last waited for, rank by rank, in host memory"""
        return self._parts(counts)

    def agreed_count(self, units=1):
        if self.seen_max is None:
            return self.cap
        n = int(self.seen_max * units * (1.0 + self.slack)) + 64
        return min(self.cap, (n + 255) & ~255)

    def start(self, units=1, send_records=None, status=0, fetch=True):
        """Issue the collective for the engine's last self-play call (status: its return code; a failed call still enters the collective,
        without records).  The first exchange of a run (nothing to predict from) is the blocking form; its result is kept for wait()
        (fetch=False: its records stay on the device like those of the pipelined form)."""
        if self.seen_max is None and send_records is None:
            parts, counts = self.allgather(status, fetch=fetch)
            self.units.append((units, (parts, counts, self.last_statuses)))
            return
        self._chk(self.L.agz_comm_post_status(self.h, int(status)), "agz_comm_post_status")
        self._chk(self.L.agz_allgather_samples_start(self.e.h, self.h, int(send_records if send_records is not None else self.agreed_count(units))), "agz_allgather_samples_start")
        self.units.append((units, None))

    def wait(self, fetch=True):
        """-> (parts, counts) of the OLDEST collective in flight.  fetch=False: parts is None — the gathered records stay in the exchange's
        device buffer (agz_comm_records_device / fetch_last()) until the next-but-one start(); a host loop that trains on the GPU, or a
        benchmark, does not pay a device-to-host copy of every rank's records per call.  statuses() then holds every rank's status word."""
        units, done = self.units[0]
        if done is None:
            counts, mx = (self.C.c_int64 * self.world)(), self.C.c_int64(0)
            rc = self.L.agz_allgather_samples_wait(self.h, counts, self.C.byref(mx))
            self.units.pop(0)                    # (the C side retires the slot whatever the outcome: the two queues stay in step)
            self._chk(rc, "agz_allgather_samples_wait")
            st = (self.C.c_int32 * self.world)()
            self._chk(self.L.agz_comm_get_statuses(self.h, st), "agz_comm_get_statuses")
            self.last_statuses = np.array(st[:], np.int32)
            counts = np.array(counts[:], np.int64)
            done = (self._parts(counts) if fetch else None, counts)
        else:
            self.units.pop(0)
            self.last_statuses = done[2]
            done = (done[0], done[1])           # (the blocking form fetched at start() — or not, if start() was told fetch=False)
        if not self.last_statuses.any():
            r = float(done[1].max()) / float(units)
            self.seen_max = r if self.seen_max is None else max(self.seen_max, r)
        return done


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
