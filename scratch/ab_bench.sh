#!/bin/bash
# A/B of environment settings on the headline bench: scratch/ab_bench.sh "AGZ_X=1" "AGZ_X=2 AGZ_Y=3" ...   (one bench run per setting)
for setting in "$@"; do
  env $setting timeout 300 python bench.py --steps 20 --warmup 5 --no-host-delivery --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-44s value %.1f M  executed %.1f M  frac %.4f  ply-equiv %.3f ms  %s' % ('$setting', d['value']/1e6, d['value_executed']/1e6, r['frac'], r['avg_launch_ms'], r['kernel'][:40]), (d['rank0'].get('age_classes') or {}).get('fraction'), (d['rank0'].get('age_classes') or {}).get('games_migrated'))
"
done
