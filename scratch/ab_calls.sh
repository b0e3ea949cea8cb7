#!/bin/bash
# how much do small calls cost?  headline bench with K generations per call, with and without the exchange step (one rank through the C ABI)
for args in "--gens-per-call 1" "--gens-per-call 2" "--gens-per-call 4" "--gens-per-call 1 --exchange" "--gens-per-call 2 --exchange" "--gens-per-call 4 --exchange"; do
  timeout 300 python bench.py --steps 12 --warmup 4 --no-host-delivery --no-cpu-baseline $args 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-32s value %.1f M  executed %.1f M  ms/step %.1f  frac %.4f' % ('$args', d['value']/1e6, d['value_executed']/1e6, d['ms_per_step'], d['roofline']['frac']))"
done
