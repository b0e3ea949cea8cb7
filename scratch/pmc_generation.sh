#!/bin/bash
# counters per search-kernel variant over one generation -> gpurun_out/$1/gen_summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -c1-12 | tr " " _)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/g_$n -o x --output-format csv -- python3 scratch/pmc_generation.py > $out/g_$n.log 2>&1
  echo "# rocprofv3 --pmc $c --kernel-trace -- python3 scratch/pmc_generation.py   (per launch: average over the launches of the variant)" >> $out/gen_summary.txt
  python3 - $out/g_$n >> $out/gen_summary.txt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void agz::", "")
    if "k_search_small" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg): print(k, "launches", max(cnt[k].values()), {c: round(v / cnt[k][c]) for c, v in agg[k].items()})
t = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void agz::", "")
    if "k_search_small" in k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(dur): print("dur", k, len(dur[k]), "avg us", round(sum(dur[k]) / len(dur[k]) / 1e3, 1))
PY
done
cat $out/gen_summary.txt
