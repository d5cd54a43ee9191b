#!/bin/bash
# per-kernel durations of `bench.py --probe-only` (GPU box).  usage: trace_probe.sh <dtype> <outdir-under-gpurun_out> [ENV=VAL ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
dt=$1; out=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/$out" -o p -- python3 "$ROOT/bench.py" --probe-only --dtype $dt > /dev/null 2>&1
python3 - "$ROOT/gpurun_out/$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("attn_", "gemm_", "cons_")):
        print("%-60s calls %4s avg %9.1f us" % (n.split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
