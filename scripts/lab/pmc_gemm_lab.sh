#!/bin/bash
# rocprofv3 PMC pass over the lab binary (GPU box).  usage: pmc_gemm_lab.sh "<-D flags>" "<counters>" outdir
cd "$(dirname "$0")"
rm -f /tmp/gemm_lab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $1 gemm_f32_lab.hip -o /tmp/gemm_lab 2>&1 | grep -E "error"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $3 -o p -- /tmp/gemm_lab 25120 3072 768 > /dev/null 2>&1
python3 - "$3" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "gemm_f32_kernel" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k)
    for c, xs in sorted(v.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
