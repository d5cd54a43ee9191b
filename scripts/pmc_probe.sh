#!/bin/bash
# rocprofv3 PMC pass over `bench.py --probe-only` (GPU box).  usage: pmc_probe.sh <dtype> "<counters>" <outdir-under-gpurun_out>
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/$3" -o p -- python3 "$ROOT/bench.py" --probe-only --dtype $1 > /dev/null 2>&1
python3 - "$ROOT/gpurun_out/$3" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("attn_", "gemm_", "cons_")):
        d[n.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k)
    for c, xs in sorted(v.items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
