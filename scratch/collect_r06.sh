#!/bin/bash
# Runs on the GPU box (gpurun): round 6's measurement set -> gpurun_out/$1/ (copied into profiles/ by scratch/install_r06.py)
#   bench lines: headline, BASELINE configs 2-5, lock-step generations (headline, config 3), exact mode (fp32 network: the reference's own precision),
#   the exchange step with one rank (C ABI), A/Bs of the round's switches (AGZ_NXL, AGZ_NARROW_SPARSE, AGZ_RESERVE_CUS);
#   rocprofv3 --kernel-trace --stats of the bench command of the headline AND of configs 2-5 (+ the bench line printed under the profiler);
#   PMC passes (FETCH_SIZE, WRITE_SIZE, instruction mix, L2) over a refilled call of every config; unit utilisation of configs 0, 2, 3
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-host-delivery"
if [ -z "$PMC_ONLY" ]; then
python bench.py --steps 20 --warmup 5 > $out/bench_headline.json 2> $out/bench_headline.err
for c in 2 3 4 5; do timeout 900 python bench.py --config $c --steps 20 --warmup 5 > $out/bench_config$c.json 2> $out/bench_config$c.err; done
python bench.py --steps 4 --warmup 1 --lockstep $B > $out/bench_headline_lockstep.json 2> $out/bench_headline_lockstep.err
AGZ_PERSIST=1 python bench.py --steps 4 --warmup 1 --lockstep $B > $out/bench_headline_lockstep_persistent.json 2> $out/bench_headline_lockstep_persistent.err
timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --lockstep $B > $out/bench_config3_lockstep.json 2> $out/bench_config3_lockstep.err
timeout 900 python bench.py --steps 6 --warmup 2 --mode exact $B > $out/bench_headline_exact.json 2> $out/bench_headline_exact.err
python bench.py --steps 6 --warmup 2 --exchange --gens-per-call 2 $B > $out/bench_headline_exchange_1rank.json 2> $out/bench_headline_exchange_1rank.err
for n in 4 8; do AGZ_RESERVE_CUS=$n python bench.py --steps 6 --warmup 2 --exchange --gens-per-call 2 $B > $out/bench_headline_exchange_1rank_reserve$n.json 2> $out/bench_headline_exchange_1rank_reserve$n.err; done
AGZ_NXL=0 python bench.py --steps 20 --warmup 5 $B > $out/bench_headline_nxl0.json 2> $out/bench_headline_nxl0.err
AGZ_NARROW_SPARSE=0 python bench.py --config 2 --steps 20 --warmup 5 $B > $out/bench_config2_dense_waves.json 2> $out/bench_config2_dense_waves.err
# rocprofv3 kernel stats of the bench command itself (the program directly behind --): headline and every config
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats_headline -o x --output-format csv -- python3 bench.py --steps 20 --warmup 5 $B > $out/bench_under_rocprof.json 2> $out/stats_headline.log
for c in 2 3 4 5; do timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_config$c -o x --output-format csv -- python3 bench.py --config $c --steps 20 --warmup 5 $B > $out/bench_config${c}_under_rocprof.json 2> $out/stats_config$c.log; done
fi
if [ -z "$BENCH_ONLY" ]; then
rm -f $out/pmc_refill_summary.txt
declare -A GENS_OF=([0]=8 [2]=8 [3]=4 [4]=3 [5]=4)
for cfg in 0 2 3 4 5; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $c | cut -c1-12 | tr " " _)
    CFG=$cfg GENS=${GENS_OF[$cfg]} timeout 600 rocprofv3 --pmc $c --kernel-trace -d $out/p${cfg}_$n -o x --output-format csv -- python3 scratch/pmc_refill.py > $out/p${cfg}_$n.log 2>&1
    echo "# cfg $cfg: rocprofv3 --pmc $c --kernel-trace -- python3 scratch/pmc_refill.py (CFG=$cfg GENS=${GENS_OF[$cfg]})   sums over the self-play launches of the call" >> $out/pmc_refill_summary.txt
    python3 - $out/p${cfg}_$n >> $out/pmc_refill_summary.txt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void agz::", "")
    if "k_search" not in k and "k_selfplay" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg): print("sum", k, "launches", max(cnt[k].values()), {c: round(v) for c, v in agg[k].items()})
t = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void agz::", "")
    if "k_search" in k or "k_selfplay" in k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(dur): print("dur", k, len(dur[k]), "avg us", round(sum(dur[k]) / len(dur[k]) / 1e3, 1))
PY
    grep -h algorithmic_bytes_of_the_call $out/p${cfg}_$n.log | tail -1 >> $out/pmc_refill_summary.txt
    rm -rf $out/p${cfg}_$n
  done
done
CFGS="0 2 3" bash scratch/pmc_util_persist.sh $1 > $out/pmc_util.log 2>&1
if [ -f scratch/libagz_dbg.so ]; then { echo "# python scratch/stamps.py  (-DAGZ_STAMPS build of the final library: cycles per wave and rollout by phase, first ply of the headline shape, 32768 games, k_search_small)"; python scratch/stamps.py; } > $out/phase_cycles_stamps.txt 2>&1; fi
fi
for d in $out/stats_*; do find $d -name "*kernel_trace.csv" -delete; done
ls $out
