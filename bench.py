#!/usr/bin/env python3
"""bench.py — self-play rollouts/s, the BASELINE.json metric.

A step = ONE whole self-play generation of the configuration the metric is quoted on: Gobang 9x9 (Nvict=5),
32768 games per GPU x 64 rollouts per move, random-init 128x6 snetwork2, bf16 MFMA network + fp32 tree
arithmetic, all games played to the end on the device (mcts(), mcts_gpu.jl:477-579).  Inputs (start positions,
weights) are resident in HBM before the timed region.  value = rollouts of all ranks / max-over-ranks time.
With N > 1 ranks each rank plays its own shard of game ids and the step ends with the RCCL all-gather of the
packed sample records.

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


def algorithmic_bytes(game, sum_p, sum_new, rollouts, S):
    """B = p(18A+16) + new*2S + (48+4VS) + (4A+4) + 8A + S per (game, rollout)."""
    A, VS = game.A, game.VS
    return sum_p * (18 * A + 16) + sum_new * 2 * S + rollouts * ((48 + 4 * VS) + (4 * A + 4) + 8 * A + S)


def cpu_baseline(args):
    """fast_mcts.jl restatement (oracle/agz_oracle.c agzo_fmcts_selfplay, kind 'port') on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    og = O.make_game("gobang", args.n, args.nvict)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
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
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only for single-GPU smoke tests of the N>1 path")
    args = ap.parse_args()

    import torch
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    from alphagpu_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libagz has no CPU path)")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)             # (several ranks on one GPU only with --backend gloo)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    game = ag.GameSpec("gobang", args.n, args.nvict)
    net = ag.SNetwork2.random(game, args.filters, args.towers)
    G, V = args.games, args.rollouts
    eng = M.Engine(game, G, V, device=dev, seed=1, game_id_base=shard.shard_base(rank, G),
                   nn_mode=M.NN_BF16 if args.mode == "bf16" else M.NN_EXACT)
    eng.set_network(net)
    eng.set_profiling(1)          # HIP events around every launch of the search kernel (one launch per ply: mcts_single in one kernel)
    rb = game.rec_bytes
    # the exchange of generation k overlaps generation k+1: two sample buffers, at most two collectives in flight
    sample_bufs = [torch.empty(G * game.max_plies * rb, dtype=torch.uint8, device="cuda") for _ in range(2)] if world > 1 else None
    inflight = [None, None]
    nstep = [0]

    def step():
        st = eng.selfplay(G, V, cpuct=args.cpuct, tau_plies=25)
        if not st["valid"]:
            raise SystemExit("illegal move sampled ('faute')")
        if world > 1:                       # the one exchange step: all-gather of the generated samples
            k = nstep[0] & 1
            nstep[0] += 1
            if inflight[k] is not None:
                inflight[k].wait()          # the collective that read sample_bufs[k] two generations ago
            n = eng.samples_packed_into(sample_bufs[k].data_ptr(), G * game.max_plies)
            if args.backend == "nccl":
                inflight[k] = shard.allgather_records_async(sample_bufs[k], n, rb)
            else:                               # gloo smoke path: stage through host memory
                inflight[k] = shard.allgather_records_async(sample_bufs[k][: n * rb].cpu(), n, rb)
        return st

    def fence():
        eng.synchronize()
        for k in range(2):
            if inflight[k] is not None:
                inflight[k].wait()
                inflight[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.kernel_times(reset=True)
    fence()
    t0 = time.perf_counter()
    rollouts = 0
    search_s = 0.0
    plies = 0
    nsamples = 0
    for _ in range(args.steps):
        st = step()
        rollouts += st["rollouts"]
        search_s += st["search_seconds"]
        plies += st["plies"]
        nsamples += st["nsamples"]
    fence()
    dt = time.perf_counter() - t0
    tree_ms, nn_ms, launches = eng.kernel_times()
    busy_ms = eng.tree_busy_ms()      # union of the launch intervals: sub-batch chains run launches side by side
    sum_p, sum_new, r_cnt = eng.counters()

    cdev = "cuda" if args.backend == "nccl" else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    tot = torch.tensor([float(rollouts)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt_max, total_rollouts = float(tmax.item()), float(tot.item())

    if rank == 0:
        S = game.pos_image_bytes
        alg = algorithmic_bytes(game, sum_p, sum_new, r_cnt, S)
        traffic = None          # HBM bytes per launch from the committed PMC passes, scaled by this run's algorithmic bytes
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if args.n == 9 and args.rollouts == 64:
                traffic = pm["traffic_over_algorithmic"] * alg / max(launches, 1)
        except Exception:
            traffic = None
        achieved = alg / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0   # aggregate over the launches in flight together
        out = {
            "metric": "self-play rollouts/sec at 32768 games x 64 rollouts, Gobang 9x9",
            "value": total_rollouts / dt_max, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"gobang{args.n}x{args.n}_nvict{args.nvict}_{G}games_per_gpu_x{V}rollouts_"
                                   f"snetwork2_{args.filters}x{args.towers}_full_generation",
                       "games_per_gpu": G, "rollouts_per_move": V, "cpuct": args.cpuct, "tau_plies": 25,
                       "tree_arithmetic": "f32 strict IEEE", "network": "bf16 MFMA, fp32 accumulate" if args.mode == "bf16" else "f32 exact",
                       "parallelism": f"game-shard x{world}, RCCL all-gather of samples at generation end" if world > 1 else "single GPU"},
            "roofline": {"kernel": "k_search_small (a whole mcts_single per launch: 65 x {expand+backup+select+encode, network forward}; "
                                   "8 lanes per game tree, node rows in registers, 16/32 games per workgroup)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg / max(launches, 1), "avg_launch_ms": tree_ms / max(launches, 1),
                         "launches": launches, "launch_concurrency": tree_ms / busy_ms if busy_ms > 0 else None,
                         "mean_depth_p": sum_p / max(r_cnt, 1),
                         "note": "one launch = one ply of the generation (all games alive, 64 rollouts); algorithmic bytes are the tree path's "
                                 "(SURVEY 8d) - the network weights stream from L2 and the kernel is bound by instruction issue / latency, not HBM"},
            "rank0": {"search_only_rollouts_per_s": rollouts / search_s if search_s > 0 else None,
                      "search_kernel_ms": tree_ms, "search_ms": search_s * 1e3,
                      "plies": plies, "samples": nsamples,
                      "wall_s": dt},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
