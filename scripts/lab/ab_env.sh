#!/bin/bash
# same-box A/B of one environment switch: ab_env.sh VAR  (VAR=0 vs VAR=1, twice each, f32_split bench without probes)
V=$1
for i in 1 2; do
  for val in 0 1; do
    env $V=$val python bench.py --dtype f32_split --steps 12 --warmup 4 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val', d['value'], d['ms_per_step'])"
  done
done
