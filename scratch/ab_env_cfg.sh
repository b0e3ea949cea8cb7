#!/bin/bash
# A/B of environment settings on a BASELINE config: CFG=3 STEPS=6 scratch/ab_env_cfg.sh "AGZ_X=1" "AGZ_X=0" ...   (CFG=0: the headline, 20 steps)
CFG=${CFG:-3}; STEPS=${STEPS:-6}
for setting in "$@"; do
  if [ "$CFG" = 0 ]; then A="--steps 20 --warmup 5"; else A="--config $CFG --steps $STEPS --warmup 2"; fi
  env $setting timeout 400 python bench.py $A --no-host-delivery --no-cpu-baseline $BENCH_ARGS 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg $CFG %-28s value %.1f M  executed %.1f M  %s frac %.4f  ply-equiv %.3f ms  %s' % ('$setting', d['value']/1e6, d['value_executed']/1e6, r['bound'], r['frac'], r['avg_launch_ms'], r['kernel'][:48]))"
done
