#!/bin/bash
# rocprofv3 passes over scripts/lab/attn_gen.py (GPU box): kernel-trace stats, then one SQ counter pass.
# usage: pmc_attn.sh <outdir under gpurun_out> [B] [T]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/scripts/lab/attn_gen.py" ${2:-32} ${3:-785} > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc" -o p -- python3 "$ROOT/scripts/lab/attn_gen.py" ${2:-32} ${3:-785} > "$OUT/pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/t_kernel_stats.csv", recursive=True)
if f:
    print("== kernel stats (avg ns)")
    for r in csv.DictReader(open(f[0])):
        if "attn" in r["Name"] or "x3" in r["Name"]:
            print("   %-60s calls %5s avg %10.0f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
f = glob.glob(out + "/pmc/**/p_counter_collection.csv", recursive=True)
if f:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "attn" in r["Kernel_Name"] or "x3" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== counters (avg per dispatch)")
    for k, v in d.items():
        print(k)
        for c, xs in sorted(v.items()):
            print("   %-28s %16.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
