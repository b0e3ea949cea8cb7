#!/bin/bash
# Runs on the GPU box (gpurun): round 5's measurement set -> gpurun_out/$1/ (copied into profiles/ by scratch/install_r05.py)
#   bench lines: headline (default = persistent kernel with age classes), the same without age classes, with one launch per ply (AGZ_PERSIST=0),
#   lock-step generations (the reference's call pattern), BASELINE configs 2-5; rocprofv3 kernel stats of the headline bench command;
#   PMC passes (FETCH_SIZE, WRITE_SIZE, instruction mix; one pass each) over a refilled call of every config; the workgroup-spread measurement
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ -z "$PMC_ONLY" ]; then
python bench.py --steps 20 --warmup 5 > $out/bench_headline.json 2> $out/bench_headline.err
AGZ_AGE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-delivery > $out/bench_headline_noage.json 2> $out/bench_headline_noage.err
AGZ_PERSIST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-delivery > $out/bench_headline_plyloop.json 2> $out/bench_headline_plyloop.err
python bench.py --steps 4 --warmup 1 --lockstep --no-host-delivery --no-cpu-baseline > $out/bench_headline_lockstep.json 2> $out/bench_headline_lockstep.err
python bench.py --steps 6 --warmup 2 --exchange --gens-per-call 2 --no-cpu-baseline --no-host-delivery > $out/bench_headline_exchange_1rank.json 2> $out/bench_headline_exchange_1rank.err
for c in 2 3 4 5; do timeout 900 python bench.py --config $c --steps 20 --warmup 5 > $out/bench_config$c.json 2> $out/bench_config$c.err; done
# (the wide trunks with two 64-game workgroups per CU — k_selfplay_big<WG=2>, the form before k_selfplay_big4 — on the same box; the headline with 64-game workgroups of eight waves)
for c in 3 4 5; do AGZ_BIG4=0 timeout 900 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-host-delivery > $out/bench_config${c}_two_workgroups.json 2> $out/bench_config${c}_two_workgroups.err; done
AGZ_PERSIST_TW=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-delivery > $out/bench_headline_tw8.json 2> $out/bench_headline_tw8.err
for c in 3; do timeout 600 python bench.py --config $c --steps 3 --warmup 1 --lockstep --no-cpu-baseline --no-host-delivery > $out/bench_config${c}_lockstep.json 2> $out/bench_config${c}_lockstep.err; done
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o x --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-delivery > $out/bench_under_rocprof.json 2> $out/stats.log
fi
rm -f $out/pmc_refill_summary.txt
# (a call of several generations: the plies in which a call's last games run out — partly empty workgroups — are a small part of it, as in
#  the chained calls bench.py times)
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
if [ -z "$PMC_ONLY" ] && [ -f scratch/libagz_wgt.so ]; then
# diagnostic builds (scratch/libagz_wgt.so: -DAGZ_WGTIME, scratch/libagz_ps.so: -DAGZ_PSTAMPS; made by `make OUT=... BUILD=... EXTRA=...`)
{ echo "# AGZ_PERSIST=0 python scratch/wgtime.py  (-DAGZ_WGTIME build): when do the 512 workgroups of a full-batch k_search_small launch start and end?"; AGZ_PERSIST=0 python scratch/wgtime.py; } > $out/workgroup_spread.txt 2>&1
{ echo "# python scratch/pstamps.py  (-DAGZ_PSTAMPS build): where the waves of the persistent kernel spend their cycles"; python scratch/pstamps.py; } > $out/persistent_phase_shares.txt 2>&1
fi
find $out/stats -name "*kernel_trace.csv" -delete
ls $out
