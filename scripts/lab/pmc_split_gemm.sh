#!/bin/bash
# rocprofv3 SQ counter passes over one split-product GEMM shape (GPU box).  usage: pmc_split_gemm.sh <outdir under gpurun_out> mode N K [math]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/$1"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_REQ_sum TCC_TAG_STALL_sum SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
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
