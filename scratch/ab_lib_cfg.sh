#!/bin/bash
# A/B of library builds on a BASELINE config: CFG=3 STEPS=6 scratch/ab_lib_cfg.sh default scratch/libagz_x.so ...
CFG=${CFG:-3}; STEPS=${STEPS:-6}
for lib in "$@"; do
  if [ "$lib" = default ]; then L=""; else L="$PWD/$lib"; fi
  AGZ_LIB_PATH=$L timeout 300 python bench.py --config $CFG --steps $STEPS --warmup 2 --no-host-delivery --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s value %.1f M  executed %.1f M  frac %.4f  ply-equiv %.3f ms' % ('$lib', d['value']/1e6, d['value_executed']/1e6, r['frac'], r['avg_launch_ms']))"
done
