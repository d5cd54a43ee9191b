#!/bin/bash
# rocprofv3 SQ counter passes over one split-product GEMM shape (GPU box).  usage: pmc_split_gemm.sh <outdir under gpurun_out> mode N K [math]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$ROOT/scripts/lab/one_split_gemm.py" $2 $3 $4 ${5:-1} > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and "epilogue" not in r["Kernel_Name"]:
            d[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc1/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(k, " avg %.1f us" % (sum(dur[k]) / max(1, len(dur[k])) / 1e3))
    for c, xs in sorted(v.items()):
        print("   %-32s %16.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
