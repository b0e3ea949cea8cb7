#!/bin/bash
# stamps of the big4 kernel + A/B against the two-workgroup form on one box: scratch/ab_big4.sh [config] [steps]
CFG=${1:-3}; STEPS=${2:-6}
mkdir -p gpurun_out/r05c
if [ -f scratch/libagz_b4s.so ]; then AGZ_LIB_PATH=$PWD/scratch/libagz_b4s.so AGZ_BIG4=1 timeout 300 python bench.py --config $CFG --steps 4 --warmup 2 --no-host-delivery --no-cpu-baseline 2>&1 | grep -v "^{" | tail -2; fi
for b in ${AB:-0 1 0 1}; do AGZ_BIG4=$b timeout 300 python bench.py --config $CFG --steps $STEPS --warmup 2 --no-host-delivery --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r05c/big4_cfg${CFG}_$b.json; python -c "
import json; d=json.load(open('gpurun_out/r05c/big4_cfg${CFG}_$b.json')); r=d['roofline']; print('BIG4=$b', round(d['value']/1e6,1), round(d.get('value_executed',0)/1e6,1), round(r['frac'],4), round(r['avg_launch_ms'],3), r['kernel'][:40])"; done
