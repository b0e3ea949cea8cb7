#!/bin/bash
# A/B of environment settings on lock-step generations of the headline config: scratch/ab_env_lockstep.sh "AGZ_X=1" "AGZ_Y=2" ...
for setting in "$@"; do
  env $setting timeout 300 python bench.py --steps 4 --warmup 1 --lockstep --no-host-delivery --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s lock-step %.1f M rollouts/s  frac %.4f  avg launch %.3f ms  %s' % ('$setting', d['value']/1e6, r['frac'], r['avg_launch_ms'], r['kernel'][:48]))"
done
