#!/bin/bash
# usage (on the GPU box): scripts/lab/run_gemm_lab.sh "<-D flags variant 1>" "<-D flags variant 2>" ...
cd "$(dirname "$0")"
for v in "$@"; do
  echo "=== variant: [$v]"
  rm -f /tmp/gemm_lab; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $v gemm_f32_lab.hip -o /tmp/gemm_lab 2>&1 | grep -E "error" 
  /tmp/gemm_lab 25120 3072 768
  /tmp/gemm_lab 25120 768 768
  /tmp/gemm_lab 4096 4096 4096
done
