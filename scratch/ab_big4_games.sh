#!/bin/bash
# k_selfplay_big4 against two 64-game workgroups per CU on the 512x8 trunk of the other built-in games: scratch/ab_big4_games.sh
for gm in "connect4 7" "reversi6 6" "hex 7"; do
  set -- $gm
  for b in 0 1; do
    AGZ_BIG4=$b timeout 300 python bench.py --game $1 --n $2 --filters 512 --towers 8 --steps 6 --warmup 2 --no-host-delivery --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s BIG4=$b  value %.1f M  executed %.1f M  %s frac %.4f  ply-equiv %.3f ms  %s' % ('$1 $2', d['value']/1e6, d['value_executed']/1e6, r['bound'], r['frac'], r['avg_launch_ms'], r['kernel'][:42]))"
  done
done
