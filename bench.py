#!/usr/bin/env python3
"""bench.py — self-play rollouts/s, the BASELINE.json metric, for every BASELINE config through one harness.

A step = ONE whole self-play generation: `--games` games per GPU x `--rollouts` rollouts per move, random-init snetwork2
`--filters` x `--towers`, bf16 MFMA network + fp32 strict-IEEE tree arithmetic, all games played to the end on the device
(mcts(), mcts_gpu.jl:477-579).  Inputs (start positions, weights) are resident in HBM before the timed region.
value = rollouts of the games all ranks RETURNED in the timed region (samples x V; every game played to its end) / max-over-ranks time
(`value_executed`: the rollouts executed inside the region, whichever game they belong to).  With N > 1 ranks each rank plays its own shard
of game ids and every call ends with the RCCL all-gather of its packed sample records.

Scheduling of the K timed steps (what the "scheduling" field of the line spells out): by default the warm-up and timed calls are ONE
chain of agz_selfplay_chain calls on `--games` slots — a slot whose game has ended takes the next game that has not started, of the
running call or of the next one — so every search runs on a full batch; `--no-chain` plays each call on its own (it ends on the
batch of its last games running out), `--lockstep` plays K separate generations the way the reference does (the batch shrinks as
games end).  Every game's samples are the same in all three (keyed by game id and the game's own ply; parity-tested).

The default is the configuration the metric is quoted on (Gobang 9x9 Nvict=5, 32768 x 64, 128x6).  `--config k` selects
BASELINE.json configs[k-1] (2: Connect4 128x6, 3: Gobang 9x9 512x8, 4: Hex 9x9 V=128 512x8, 5: Reversi 8x8 512x8 — the
per-GPU shard of the 8-GPU run), or give --game/--n/--nvict/--filters/--towers/--rollouts/--games directly.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (same guide)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; one VALU instruction of a 64-wide wavefront occupies a SIMD for 4 cycles

CONFIGS = {                    # BASELINE.json configs[k-1]
    1: dict(game="gobang", n=3, nvict=3, games=256, rollouts=16, filters=128, towers=6),
    2: dict(game="connect4", n=0, nvict=0, games=32768, rollouts=64, filters=128, towers=6),
    3: dict(game="gobang", n=9, nvict=5, games=32768, rollouts=64, filters=512, towers=8),
    4: dict(game="hex", n=9, nvict=0, games=32768, rollouts=128, filters=512, towers=8),
    5: dict(game="reversi8", n=0, nvict=0, games=32768, rollouts=64, filters=512, towers=8),
}


def algorithmic_bytes(game, sum_p, sum_new, rollouts, S):
    """B = p(18A+16) + new*2S + (48+4VS) + (4A+4) + 8A + S per (game, rollout)."""
    A, VS = game.A, game.VS
    return sum_p * (18 * A + 16) + sum_new * 2 * S + rollouts * ((48 + 4 * VS) + (4 * A + 4) + 8 * A + S)


def nn_flops_per_leaf(game, H, T):
    """2 (in H + T H^2 + (A+1) H)."""
    return 2.0 * (2 * game.VS * H + T * H * H + (game.A + 1) * H)


def game_label(args):
    if args.game == "gobang":
        return f"gobang{args.n}x{args.n}_nvict{args.nvict}", f"Gobang {args.n}x{args.n}"
    if args.game == "hex":
        return f"hex{args.n}x{args.n}", f"Hex {args.n}x{args.n}"
    return args.game, {"connect4": "Connect4", "reversi8": "Reversi 8x8", "reversi6": "Reversi 6x6"}[args.game]


def cpu_baseline(args):
    """fast_mcts.jl restatement (oracle/agz_oracle.c agzo_fmcts_selfplay, kind 'port') on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    og = O.make_game(args.game, args.n, args.nvict)
    net = O.OracleNet(og, args.filters, args.towers)
    cores = os.cpu_count() or 1
    ngames, plies = 4 * cores, 1
    t0 = time.time()
    r = O.fmcts_selfplay(og, net, ngames, args.rollouts, args.cpuct, 25, 1, cores, plies)     # calibration
    dt = max(time.time() - t0, 1e-3)
    plies = max(1, min(og.ML, int(args.cpu_seconds / dt)))
    t0 = time.time()
    r = O.fmcts_selfplay(og, net, ngames, args.rollouts, args.cpuct, 25, 1, cores, plies)
    dt = time.time() - t0
    return {"value": r / dt, "unit": "rollouts/s", "cores": cores, "kind": "port",
            "sample": f"{ngames} games x first {plies} plies x {args.rollouts} readouts, fast_mcts.jl semantics, "
                      f"fp32 {args.filters}x{args.towers} net, OpenMP over games ({r} rollouts in {dt:.1f}s)"}


def launch_ranks(n):
    """`python bench.py --gpus N` as typed: start the N rank processes as CHILDREN (torch.distributed.run, one rank per GPU) before
    anything in this process touches the GPU, relay rank 0's JSON line (the children inherit stdout) and exit with their status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed generations (the default is what the round driver passes; 20 generations of the headline config take 3 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, help="BASELINE.json configs[k-1] (1..5); 0 = the headline metric config")
    ap.add_argument("--game", default="gobang", choices=["gobang", "connect4", "hex", "reversi8", "reversi6"])
    ap.add_argument("--games", type=int, default=32768, help="games per GPU (--samples, mainGobang.jl:96)")
    ap.add_argument("--rollouts", type=int, default=64, help="--rollout, mainGobang.jl:100")
    ap.add_argument("--cpuct", type=float, default=1.5, help="--cpuct, mainGobang.jl:113")
    ap.add_argument("--n", type=int, default=9)
    ap.add_argument("--nvict", type=int, default=5)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--towers", type=int, default=6)
    ap.add_argument("--mode", choices=["bf16", "exact"], default="bf16")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-delivery", action="store_true", help="skip the extra (untimed) generation that measures sample delivery to the host")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only for single-GPU smoke tests of the N>1 path")
    ap.add_argument("--lockstep", action="store_true", help="K separate lock-step generations of --games games (the reference's call pattern: the batch "
                    "shrinks as games end) instead of ONE call that plays K x --games games on --games slots, finished games' slots refilled")
    ap.add_argument("--gens-per-call", type=int, default=0, help="generations' worth of games one agz_selfplay call plays on the engine's slots (bounds the "
                    "sample store: ~1 GB per generation of Gobang 9x9); 0 = up to 32 on one GPU, up to 8 per rank with several (the exchange buffers "
                    "scale with it)")
    ap.add_argument("--no-chain", action="store_true", help="every agz_selfplay call on its own (ends on a batch that runs out) instead of a chain of "
                    "calls in which a call starts the next call's games in the slots it leaves free (agz_selfplay_chain)")
    ap.add_argument("--exchange", action="store_true", help="run the exchange step (agz_comm_*: RCCL all-gather of the call's records through the C ABI) even "
                    "with ONE rank, inside the timed region as with N > 1 (tests: the bench's multi-GPU path end to end on one GPU)")
    ap.add_argument("--exchange-impl", choices=["torch", "abi"], default="torch", help="the exchange step of N > 1 ranks over RCCL: torch.distributed's "
                    "all_gather_into_tensor (shard.RecordExchange: the form every multi-rank test of this repository has run — gloo with 2 and 8 ranks, RCCL "
                    "with one) or the C ABI (agz_comm_*: libagz binds RCCL itself; shard.CommExchange — has only ever run with ONE rank: no box with two "
                    "GPUs was available; tests/test_gpu_scale_parity.py compares the two forms as soon as two devices are visible)")
    ap.add_argument("--exchange-plies", type=int, default=0, help="capacity of the exchange buffers in records per game (0 = the longest game the rules "
                    "allow, game.max_plies: a call can never outgrow it)")
    ap.add_argument("--dump-records", default="", help="rank 0 writes the gathered samples of the LAST timed generation (PoolSample order) to this .npz")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.config:
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for k, v in CONFIGS[args.config].items():
            if "--" + k not in given:             # (an option typed next to --config wins: `--config 5 --games 256` is config 5's shape on a smaller batch)
                setattr(args, k, v)
    if args.game in ("connect4", "reversi8", "reversi6"):
        args.n = args.nvict = 0
    if args.game == "hex":
        args.nvict = 0

    import torch
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    from alphagpu_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` or under torch.distributed.run with --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libagz has no CPU path)")
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs ({ndev} visible); `--backend gloo` runs the N > 1 path with all ranks on one GPU (smoke test)")
    dev = local_rank % max(ndev, 1)             # (several ranks on one GPU only with --backend gloo)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    game = ag.GameSpec(args.game, args.n, args.nvict)
    net = ag.SNetwork2.random(game, args.filters, args.towers)
    G, V = args.games, args.rollouts
    # scheduling of the K timed generations: by default ONE agz_selfplay call plays K x G games on the engine's G slots — a slot whose game
    # has ended takes the next game that has not started yet, so every mcts_single of the timed region runs on a full batch of G games
    # (the configuration the metric names) instead of the shrinking batches of a generation's tail.  Every game's samples are exactly those
    # of lock-step generations (results are keyed by game id and the game's own ply; tests/test_gpu_scale_parity.py checks slices of such a
    # run against the oracle).  --lockstep: K separate calls of G games; the line carries that number too (value_lockstep_generations).
    world_ = int(os.environ.get("WORLD_SIZE", "1"))
    # several GPUs: the exchange buffers (allocated once, agz_comm_create) scale with a call's records x the ranks — 2 slots x (1 + world) x
    # capacity x rec_bytes per rank; calls are sized so that the gathered records stay under ~16 GB per rank: 8 ranks -> 1 generation per call
    # (0.76 GB of Gobang 9x9 records per rank and generation), 4 -> 2, 2 -> 4
    gpc = args.gens_per_call if args.gens_per_call > 0 else (32 if world_ == 1 else max(1, 8 // world_))
    gens_cap = 1 if args.lockstep else max(1, min(gpc, max(args.steps, args.warmup, 1)))

    def calls(k):                           # K generations as calls of at most gens_cap generations each
        return [gens_cap] * (k // gens_cap) + ([k % gens_cap] if k % gens_cap else [])

    # The calls of a run form a CHAIN (agz_selfplay_chain, include/agz.h): game ids run on from call to call under one seed, and while a
    # call's last games run out the slots that come free start games of the NEXT call (up to FILL generations' worth), which stay in flight
    # when the call returns — no call of the run ends on a batch that runs out, the untimed last one aside.  The timed region therefore
    # completes its K generations' worth of games on full batches from its first search to its last; it inherits the games the warm-up left in
    # flight and leaves as many in flight itself; `value` counts the rollouts of the games it RETURNS, `value_executed` those executed inside it.
    chain = not args.lockstep and not args.no_chain
    FILL = 2
    gens_run = max(1, args.warmup + args.steps)
    eng = M.Engine(game, G, V, device=dev, seed=1, game_id_base=shard.shard_base(rank, (gens_run if chain else gens_cap) * G),
                   nn_mode=M.NN_BF16 if args.mode == "bf16" else M.NN_EXACT, sample_capacity_games=(gens_cap + (FILL if chain else 0)) * G)
    eng.set_network(net)
    rb = game.rec_bytes
    # the exchange of generation k overlaps generation k+1: two sample buffers, at most two collectives in flight; a collective is
    # issued without any read-back (the rank's record count travels in the buffer's header: shard.RecordExchange)
    # The exchange step.  RCCL (backend "nccl"): through the C ABI — shard.CommExchange is a thin caller of agz_comm_create /
    # agz_allgather_samples_start / _wait; libagz binds RCCL itself and owns the gather buffers, allocated ONCE here.  Capacity: a call's
    # games x the longest game the rules allow (--exchange-plies bounds it: a call that outgrows a smaller capacity fails on every rank with
    # a message, it does not hang).  "gloo" (smoke tests with all ranks on one GPU): the torch.distributed form, staged through host memory.
    use_abi = (world > 1 and args.backend == "nccl" and args.exchange_impl == "abi") or (world == 1 and args.exchange)
    cap_records = gens_cap * G * max(1, min(game.max_plies, args.exchange_plies) if args.exchange_plies > 0 else game.max_plies)
    cex = shard.CommExchange(eng, rank, world, cap_records) if use_abi else None
    ex = shard.RecordExchange(gens_cap * G * game.max_plies, rb) if world > 1 and not use_abi else None
    sample_bufs = [ex.new_buffer("cuda") for _ in range(2)] if ex else None
    host_bufs = [ex.new_buffer("cpu").pin_memory() for _ in range(2)] if ex and args.backend != "nccl" else None
    inflight = [None, None]
    last_gather = [None]
    nstep = [0]
    fault = [0]

    def step(ngen=1, nxt=None):
        if chain and nxt is not None:           # a call of the run's chain: `nxt` generations' worth of the next call's games may start early
            # (FILL generations' worth of LATER games may start early whatever the size of the next call: with calls of one generation —
            #  8 ranks — the next call alone is not enough to keep the slots busy while a call's longest games finish: 529 vs 591 M rollouts/s)
            st = eng.selfplay_chain(ngen * G, FILL * G, V, cpuct=args.cpuct, tau_plies=25)
        else:
            eng.set_seed(1 + nstep[0])          # a fresh Philox key per call, as the reference's unseeded draws
            st = eng.selfplay(ngen * G, V, cpuct=args.cpuct, tau_plies=25)
        # an illegal sampled move ("faute", mcts_gpu.jl:526-529) voids the run — but a rank must NOT leave before the exchange: the others
        # would wait inside the all-gather for ever.  The rank takes part (its status word travels with the records, agz_comm_post_status;
        # the torch form gathers whatever it has) and every rank stops behind the timed region, together (the all-reduce of `fault` below).
        if not st["valid"]:
            fault[0] = 1
        k = nstep[0] & 1
        nstep[0] += 1
        if cex is not None:                 # the one exchange step, through the C ABI: at most two collectives in flight
            if len(cex.units) == 2:
                last_gather[0] = cex.wait(fetch=False)   # (the gathered records stay on the device: a trainer on the GPU reads them there)
                fault[0] |= int(cex.statuses().any())
            cex.start(units=ngen, status=0 if st["valid"] else -5, fetch=False)
        elif world > 1:                     # ... the torch.distributed form (gloo smoke path)
            if inflight[k] is not None:
                inflight[k].wait()          # the collective that read sample_bufs[k] two generations ago
            n = eng.samples_packed_into(sample_bufs[k].data_ptr() + shard.HEADER, gens_cap * G * game.max_plies)
            if args.backend == "nccl":
                inflight[k] = ex.start(sample_bufs[k], n, units=ngen)
            else:                               # gloo smoke path: stage through host memory
                lo, hi = shard.HEADER, shard.HEADER + n * rb
                host_bufs[k][lo:hi].copy_(sample_bufs[k][lo:hi])
                inflight[k] = ex.start(host_bufs[k], n, units=ngen)
            last_gather[0] = inflight[k]
        return st

    def fence():
        eng.synchronize()
        while cex is not None and cex.units:
            last_gather[0] = cex.wait(fetch=False)
            fault[0] |= int(cex.statuses().any())
        for k in range(2):
            if inflight[k] is not None:
                inflight[k].wait()
                inflight[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # which kernels does this configuration run?  One launch per ply (whole mcts_single in k_search_small): events around every
    # launch.  Two kernels per rollout (wide trunks, V > 64): events around ~10^4 launches cost ~10 % of a generation, so only
    # every 4th search is instrumented (profiling bit 2); the fractions are taken over the instrumented searches.
    # the kernel the roofline object names is the one the FIRST ply (all G games alive) dispatches to -- the plies of the tail
    # run smaller-batch variants of the same kernel (agz_get_search_form reports the last search): one untimed probe search
    eng.set_profiling(1)
    eng.set_roots(None, L=G)
    eng.search(V, cpuct=args.cpuct, training=True, step=0)
    form_tree, form_nn = eng.search_form()
    whole = form_tree.startswith("k_search_small") or form_tree.startswith("k_search_big")   # one launch per ply at every size
    plan = [(ng, False) for ng in calls(max(args.warmup, 0))] + [(ng, True) for ng in calls(args.steps)]   # (generations, timed) per call of the run
    nxt_of = [plan[i + 1][0] if i + 1 < len(plan) else FILL for i in range(len(plan))]                     # (the last call announces games that nobody will ask for)
    eng.set_seed(1)
    for i, (ng, timed) in enumerate(plan):
        if not timed:
            step(ng, nxt_of[i])
    eng.set_profiling(1 if whole else 7)
    eng.kernel_times(reset=True)
    fence()
    t0 = time.perf_counter()
    rollouts = 0
    search_s = 0.0
    plies = 0
    nsamples = 0
    for i, (ng, timed) in enumerate(plan):
        if not timed:
            continue
        st = step(ng, nxt_of[i])
        rollouts += st["rollouts"]
        search_s += st["search_seconds"]
        plies += st["plies"]
        nsamples += st["nsamples"]
    fence()
    dt = time.perf_counter() - t0
    tree_ms, nn_ms, launches = eng.kernel_times()
    busy_ms = eng.tree_busy_ms()      # union of the launch intervals: sub-batch chains run launches side by side
    # the persistent form (k_selfplay_small: ONE launch per agz_selfplay call, every workgroup loops over the plies of its own games): the
    # unit the roofline object is quoted per — "a launch" in SURVEY 8(d)'s sense, one mcts_single of G games x V rollouts — is then a
    # PLY-EQUIVALENT of the persistent launch: kernel time x (G x V) / rollouts executed
    age_searches, age_ranked, age_moved = eng.age_stats()
    form_run = eng.search_form()[0]
    persistent = form_run.startswith("k_selfplay_")
    kernel_launches = launches
    if persistent:
        form_tree, form_nn = eng.search_form()
        whole = True
    sum_p, sum_new, r_cnt = eng.counters()
    nn_leaves = eng.nn_leaves()

    if chain:
        eng.set_roots(None, L=0)            # the chain ends here: the games the timed region left in flight are dropped
    if args.dump_records:                   # (tests: the samples of the last timed generation as the host sees them)
        import numpy as np
        if cex is not None:
            counts = last_gather[0][1]
            parts = cex.fetch_last(counts)
            if rank == 0:
                np.savez(args.dump_records, **shard.merge_poolsample_order([shard.unpack_records(parts[r], int(counts[r]), game) for r in range(world)]))
        elif world > 1:
            parts, counts = last_gather[0].wait()
            if rank == 0:
                merged = shard.merge_poolsample_order([shard.unpack_records(parts[r].cpu().numpy(), int(counts[r]), game) for r in range(world)])
                np.savez(args.dump_records, **merged)
        else:
            np.savez(args.dump_records, **eng.samples())

    # host delivery (SURVEY §8d "end-to-end"): the reference's generation ends with the samples in the host PoolSample
    # (mcts_gpu.jl:515, mainGobang.jl:54-80).  Outside the timed region: (a) one generation + packed records D2H into pinned host
    # memory + push into the PoolSample arrays, serially; (b) the PIPELINED host loop a trainer would run: the records of
    # generation k are packed into device buffer k & 1, copied to pinned host memory on a second stream and unpacked into the
    # PoolSample ring by a host thread (agz_unpack_records) while generation k + 1 runs on the engine's stream.
    host = None
    if world == 1 and not args.no_host_delivery:
        import threading
        import queue
        eng.set_profiling(0)
        buf = ag.PoolSample(game, 2_000_000)               # mainGobang.jl:130
        eng.samples_packed_host(); eng.samples_into(None, None, None, None, None)   # (staging buffers of the delivery paths are allocated once per run, not per generation)
        eng.synchronize()
        h0 = time.perf_counter()
        st = step()
        h1 = time.perf_counter()
        recs = eng.samples_packed_host()
        h2 = time.perf_counter()
        # ... and into the PoolSample arrays: agz_get_samples (D2H into the engine's pinned staging buffer + unpack on the host
        # cores) followed by the ring-buffer push
        buf.push_from_engine(eng)
        h3 = time.perf_counter()
        host = {"generation_s": h1 - h0, "packed_records_to_pinned_host_s": h2 - h1, "samples_into_PoolSample_s": h3 - h2,
                "bytes": int(recs.size), "samples": int(recs.shape[0]),
                "lockstep_generation_rollouts_per_s": st["rollouts"] / (h1 - h0),
                "rollouts_per_s_with_host_delivery": st["rollouts"] / (h2 - h0),
                "rollouts_per_s_with_delivery_into_PoolSample_serial": st["rollouts"] / (h1 - h0 + h3 - h2)}
        # (b) pipelined: calls of `gp` generations' worth of games (refilled slots, as in the timed region); the records of call k travel
        # and are unpacked while call k + 1 runs
        # (gp generations per call: every call ends with the plies in which its last games run out, ~60 ms for the headline config — four
        #  generations per call spread them thinner than two did; the buffers are sized by one untimed call, not by the longest game possible)
        # (a chain of calls of two generations each, as a trainer's loop would run them; without chaining four generations per call, so that
        #  the ~60 ms in which a call's last games run out are spread thinner)
        gp = min(gens_cap, 2 if chain else 4)
        step(gp, gp)
        cap = min(gp * G * game.max_plies, int(eng.num_samples() * 1.2) + 4096)
        dbuf = [torch.empty(cap * rb, dtype=torch.uint8, device="cuda") for _ in range(2)]
        hbuf = [torch.empty(cap * rb, dtype=torch.uint8).pin_memory() for _ in range(2)]
        side = torch.cuda.Stream()
        free = [threading.Semaphore(1), threading.Semaphore(1)]
        jobs = queue.Queue()
        err = []
        d_copy, d_unpack = [], []

        def deliver():
            try:
                while True:
                    job = jobs.get()
                    if job is None:
                        return
                    k, n = job
                    c0 = time.perf_counter()
                    with torch.cuda.stream(side):
                        hbuf[k][: n * rb].copy_(dbuf[k][: n * rb], non_blocking=True)
                    side.synchronize()
                    c1 = time.perf_counter()
                    buf.push_packed(hbuf[k][: n * rb].numpy(), n)
                    d_copy.append(c1 - c0); d_unpack.append(time.perf_counter() - c1)
                    free[k].release()
            except Exception as e:          # noqa: BLE001 — reported by the main thread
                err.append(e)
                free[0].release(); free[1].release()

        th = threading.Thread(target=deliver, daemon=True)
        th.start()
        ncalls = 16 // gp if gp in (1, 2, 4) else 4   # (the delivery of the last call is not hidden: amortised over the calls)
        p_rollouts = 0
        p0 = time.perf_counter()
        for i in range(ncalls):
            k = i & 1
            free[k].acquire()               # the delivery of call i - 2 has left buffer k
            st = step(gp, gp)
            n = eng.samples_packed_into(dbuf[k].data_ptr(), cap)
            jobs.put((k, n))
            p_rollouts += st["rollouts"]
        jobs.put(None)
        th.join()
        p1 = time.perf_counter()
        if err:
            raise err[0]
        if chain:
            eng.set_roots(None, L=0)
        host["pipelined_generations"] = ncalls * gp
        host["pipelined_calls"] = ncalls
        host["pipelined_copy_s_per_call"] = d_copy
        host["pipelined_unpack_s_per_call"] = d_unpack
        host["pipelined_wall_s"] = p1 - p0
        host["rollouts_per_s_with_delivery_into_PoolSample"] = p_rollouts / (p1 - p0)
        host["poolsample_length_after"] = buf.length_buffer()
        del dbuf, hbuf

    cdev = "cuda" if args.backend == "nccl" else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    tot = torch.tensor([float(rollouts), float(nsamples)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max, total_rollouts, nsamples_all = float(tmax.item()), float(tot[0].item()), float(tot[1].item())
    flt = torch.tensor([float(fault[0])], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(flt, op=dist.ReduceOp.MAX)
    if flt.item() > 0:                          # every rank leaves here, behind the last collective of the run
        if cex is not None:
            cex.close()
        eng.close()
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit("illegal move sampled ('faute') on some rank: the run is void")

    if rank == 0:
        S = game.pos_image_bytes
        alg = algorithmic_bytes(game, sum_p, sum_new, r_cnt, S)
        gl, gname = game_label(args)
        # HBM bytes per launch: NOT measured in this run (counters need rocprofv3 passes of their own) — the ratio traffic /
        # algorithmic bytes of the committed PMC passes (profiles/pmc_traffic.json, one key per configuration) times this run's
        # algorithmic bytes.  A whole-generation ratio (separate FETCH_SIZE / WRITE_SIZE passes over every launch of a generation,
        # summed over the kernel variants of its plies) is used where one was measured, the first-ply ratio otherwise; the
        # object names its source.  A BASELINE configuration without a committed pass is an error, not a silent null.
        if persistent:
            launches = max(r_cnt / float(G * V), 1e-9)                  # ply-equivalents (G games x V rollouts) of the persistent launches
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = f"{gl}_{V}_{args.filters}x{args.towers}"
        traffic, traffic_source = None, "none: no committed PMC pass for this shape (profiles/pmc_traffic.json has no key %s)" % key
        valu_obj = None         # VALU-issue roofline of the same kernel: SQ_INSTS_VALU (PMC pass) x 4 cycles / (SIMDs x clock x time)
        baseline_shape = any(all(getattr(args, k) == v for k, v in c.items()) for kk, c in CONFIGS.items() if kk != 1) or \
            (args.game, args.n, args.nvict, G, V, args.filters, args.towers) == ("gobang", 9, 5, 32768, 64, 128, 6)
        if key not in pm and baseline_shape and args.mode == "bf16":
            raise SystemExit(f"profiles/pmc_traffic.json has no entry {key}: a BASELINE configuration must carry measured HBM traffic")
        if key in pm and args.mode == "bf16":
            e = pm[key]
            if "generation" in e:
                ratio = e["generation"]["traffic_over_algorithmic"]
                traffic_source = f"profiles/pmc_traffic.json[{key}].generation ({e['generation']['source']}): whole-generation ratio"
            else:
                ratio = e["traffic_over_algorithmic"]
                traffic_source = f"profiles/pmc_traffic.json[{key}]: first-ply ratio (32768 games) applied to every ply of the generation"
            traffic = ratio * alg / max(launches, 1)
            vi = e.get("generation", {}).get("valu_insts_per_rollout", e.get("valu_insts_per_rollout"))
            if vi is not None and busy_ms > 0:
                insts = vi * r_cnt                                      # wave-instructions of the instrumented searches
                valu_obj = {"bound": "valu", "achieved": insts / (busy_ms * 1e-3) / 1e9, "peak": SIMDS * CLOCK_HZ / 4 / 1e9,
                            "unit": "G wave-instructions/s", "frac": insts * 4 / (SIMDS * CLOCK_HZ * busy_ms * 1e-3),
                            "note": "SQ_INSTS_VALU per (game, rollout) of the committed PMC pass (" + traffic_source + ") x this run's "
                                    "rollouts; 4 cycles per wave-instruction on 1024 SIMDs at 2.4 GHz (the counters show ~2.0 GHz under "
                                    "this load: the real issue utilisation is ~1.2 x higher)"}
        hbm_achieved = alg / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0   # aggregate over the launches in flight together
        if whole or nn_leaves == 0:
            # the network forward runs inside the search kernel: its time is not separable, the fraction is taken against the
            # whole launch (a lower bound of the MFMA pipe's rate while the network phase runs)
            flops = nn_flops_per_leaf(game, args.filters, args.towers) * r_cnt
            nn_t_ms, nn_note = busy_ms, "network inside the whole-search kernel: flops / whole-launch time (lower bound)"
        else:
            # plies of the two-kernel form only (big batches): leaves and HIP-event time of the stand-alone network launches; the
            # small-batch plies of the same generation run the network inside k_search_big / k_search_small
            flops = nn_flops_per_leaf(game, args.filters, args.towers) * nn_leaves
            nn_t_ms, nn_note = nn_ms, "stand-alone network launches of the instrumented searches (leaves and HIP-event time of those launches only)"
        mfma_achieved = flops / (nn_t_ms * 1e-3) / 1e12 if nn_t_ms > 0 else 0.0
        tree_obj = {"kernel": form_tree, "bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": alg / max(launches, 1), "avg_launch_ms": tree_ms / max(launches, 1),
                    "launches": launches, "launch_concurrency": tree_ms / busy_ms if busy_ms > 0 else None,
                    "mean_depth_p": sum_p / max(r_cnt, 1),
                    "kernel_launches": kernel_launches, "kernel_ms_total": tree_ms,
                    "note": ("PERSISTENT kernel: one launch per agz_selfplay call, every workgroup loops over the plies of its own games (search, move "
                             "choice, play / isOver, sample capture, refill inside the kernel).  `launches`, `avg_launch_ms` and "
                             "`algorithmic_bytes_per_launch` are per PLY-EQUIVALENT (G games x V rollouts = one mcts_single of the batch): "
                             "kernel_ms_total x G x V / rollouts; rocprofv3 shows `kernel_launches` launches of k_selfplay_small whose TOTAL "
                             "duration is kernel_ms_total" if persistent else
                             "one launch = one ply of the generation (all games alive, V rollouts)" if whole else
                             "one launch = one rollout of all games alive; busy time = union of the launch intervals of the sub-batch chains")
                            + "; algorithmic bytes are the tree path's (SURVEY 8d)"}
        nn_traffic = None       # HBM bytes per stand-alone network launch (PMC passes of the config's first ply x this run's leaves per launch)
        if not whole and nn_leaves > 0 and "nn_hbm_bytes_per_leaf" in pm.get(key, {}):
            nn_traffic = pm[key]["nn_hbm_bytes_per_leaf"] * nn_leaves / max(launches * V / (V + 1.0), 1.0)
        nn_obj = {"kernel": form_nn, "bound": "mfma", "achieved": mfma_achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": mfma_achieved / MFMA_PEAK_TFLOPS, "traffic": nn_traffic,
                  "flops_per_leaf": nn_flops_per_leaf(game, args.filters, args.towers), "leaves": (r_cnt if (whole or nn_leaves == 0) else nn_leaves),
                  "time_ms": nn_t_ms, "note": nn_note}
        # the dominant kernel of the configuration: the network when its launches take longer than the tree kernel's
        nn_dominant = (not whole) and nn_ms > busy_ms
        if whole and args.filters >= 512:
            # one launch per search with a 512-wide trunk (k_search_big): the network pass inside it is what the launch waits for (its
            # matrix work is 30 x the 128-wide trunk's) — the launch is priced against the MFMA peak, whole-launch time as the denominator
            nn_dominant = True
            nn_obj["kernel"] = form_tree + " [" + form_nn + "]"
            nn_obj["traffic"] = traffic; nn_obj["traffic_source"] = traffic_source
            nn_obj["avg_launch_ms"] = tree_ms / max(launches, 1); nn_obj["launches"] = launches
            nn_obj["kernel_launches"] = kernel_launches; nn_obj["kernel_ms_total"] = tree_ms
            if persistent:
                nn_obj["note"] += "; PERSISTENT kernel (one launch per agz_selfplay call): launches / avg_launch_ms are per ply-equivalent (G games x V rollouts)"
        out = {
            "metric": f"self-play rollouts/sec at {G} games x {V} rollouts, {gname}",
            # value = the rollouts of the games RETURNED in the timed region (their samples x V: every one of them played to its end, K x G games
            # for K steps) / time.  The rollouts EXECUTED inside the region (`value_executed`) also count work on games the region leaves in flight
            # for the next call and miss the work done earlier on the games it inherited; the two meet as the run gets longer.
            "value": nsamples_all * V / dt_max, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{gl}_{G}games_per_gpu_x{V}rollouts_snetwork2_{args.filters}x{args.towers}_full_generation",
                       "baseline_config": args.config if args.config else "headline (metric)",
                       "games_per_gpu": G, "rollouts_per_move": V, "cpuct": args.cpuct, "tau_plies": 25,
                       "tree_arithmetic": "f32 strict IEEE", "network": "bf16 MFMA, fp32 accumulate" if args.mode == "bf16" else "f32 exact",
                       "parallelism": (f"game-shard x{world}, RCCL all-gather of samples at the end of every call"
                                       + (" (through the C ABI: agz_comm_*)" if use_abi else " (torch.distributed)")) if (world > 1 or use_abi) else "single GPU"},
            "roofline": nn_obj if nn_dominant else tree_obj,
            "roofline_other": tree_obj if nn_dominant else nn_obj,
            "roofline_valu": valu_obj,
            "scheduling": ("lock-step: K separate generations of G games, the batch shrinks as games end (the reference's call pattern)" if args.lockstep else
                           (f"a chain of agz_selfplay_chain calls of {gens_cap} x G games on G slots: a slot whose game has ended takes the next game that has "
                            f"not started — of this call or, once those have all started, of the next one (up to {FILL} x G of them, left in flight when the "
                            "call returns): every search of the timed region runs on a full batch of G games; it completes its K x G games, inherits the games "
                            "the warm-up left in flight and leaves as many in flight; value = rollouts of the K x G games it RETURNS / time; every game's samples are those of a "
                            "lock-step run over the run's games (keyed by game id and the game's own ply)" if chain else
                            f"agz_selfplay calls of {gens_cap} x G games on G slots, each on its own: a slot whose game has ended takes the next game that has "
                            "not started (per-game samples identical to lock-step generations); a call ends on the batch of its last games running out")),
            "value_calls_on_their_own": None if chain else rollouts / dt,
            "value_executed": total_rollouts / dt_max,
            "value_by_returned_games": nsamples_all * V / dt_max,
            "value_lockstep_generations": (rollouts / dt if args.lockstep else (host or {}).get("lockstep_generation_rollouts_per_s")),
            "value_with_host_delivery": (host or {}).get("rollouts_per_s_with_delivery_into_PoolSample"),
            "rank0": {"search_only_rollouts_per_s": rollouts / search_s if search_s > 0 else None,
                      "search_kernel_ms": tree_ms, "network_kernel_ms": nn_ms, "search_ms": search_s * 1e3,
                      "instrumented_rollouts": r_cnt, "plies": plies, "samples": nsamples, "wall_s": dt,
                      "age_classes": ({"searches_of_a_game": age_searches, "with_rows_by_legal_rank": age_ranked, "fraction": age_ranked / max(age_searches, 1),
                                       "games_migrated": age_moved} if persistent else None),
                      "host_delivery": host},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if cex is not None:
        cex.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
