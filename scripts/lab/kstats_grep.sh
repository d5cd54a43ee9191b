#!/bin/bash
# GPU lab: rocprofv3 kernel stats of a short f32_split bench, rows matching a pattern.   usage: kstats_grep.sh "<python-regex>" [dtype]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/_ks"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/_ks" -o p -- python3 "$ROOT/bench.py" --dtype ${2:-f32_split} --steps 6 --warmup 3 --no-cpu-baseline --no-infer > /dev/null 2>&1
F=$(find "$ROOT/gpurun_out/_ks" -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$1" <<PY
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
tot = 0.0
for r in rows:
    if pat.search(r["Name"]):
        tot += float(r["TotalDurationNs"]) / 9e6
        print("%-88s %5s x %8.1f us = %7.3f ms/step" % (r["Name"][:88], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 9e6))
print("matched total %.3f ms/step; all kernels %.3f ms/step" % (tot, sum(float(r["TotalDurationNs"]) for r in rows) / 9e6))
PY
rm -rf "$ROOT/gpurun_out/_ks"
