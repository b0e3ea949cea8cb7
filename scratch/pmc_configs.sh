#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of one first-ply search of BASELINE configs 2..5 -> gpurun_out/$1/cfg_summary.txt
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in 2 3 4 5; do
  export CFG=$cfg
  [ $cfg -ge 3 ] && unset AGZ_CHAINS   # (the wide-trunk configs in their default dispatch: sub-batch chains)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace -d $out/c${cfg}_$c -o x --output-format csv -- python3 scratch/pmc_point.py > $out/c${cfg}_$c.log 2>&1
    echo "# cfg $cfg $c" >> $out/cfg_summary.txt
    python scratch/pmc_summary2.py $out/c${cfg}_$c 2>&1 | grep -v "k_advance\|k_scan\|k_compact\|k_fold\|^dur" >> $out/cfg_summary.txt
    python - $out/c${cfg}_$c >> $out/cfg_summary.txt <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("<")[0].split("(")[0]; tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in tot: print("sum", k, int(tot[k]), "launches", n[k])
PY
    grep -h sum_p $out/c${cfg}_$c.log | tail -1 >> $out/cfg_summary.txt
  done
done
cat $out/cfg_summary.txt
