#!/bin/bash
# usage: run_attn_trace.sh name [lib]
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export ACR_LAB_LIB=$ROOT/$2
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/at_$1 -o t -- python3 $ROOT/scripts/lab/attn_x3_trace.py > $ROOT/gpurun_out/at_$1.log 2>&1
python3 $ROOT/scripts/lab/kstats.py $(find $ROOT/gpurun_out/at_$1 -name "*kernel_trace.csv") attn_ x3_ > $ROOT/gpurun_out/at_$1.txt 2>&1
rm -rf $ROOT/gpurun_out/at_$1
