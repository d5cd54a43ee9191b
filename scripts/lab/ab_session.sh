#!/bin/bash
# same-box A/B: session-start tree vs current tree, twice each
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for t in old new; do
    if [ $t = old ]; then d=$R/scripts/lab/_build/ab_old; else d=$R; fi
    (cd $d && python bench.py --dtype f32_split --steps 12 --warmup 4 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])")
  done
done
