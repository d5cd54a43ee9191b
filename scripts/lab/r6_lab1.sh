#!/bin/bash
# round-6 lab session 1 (GPU box): GroupNorm with one workgroup per CU (LDS pad), image-product MFMA order / count timing builds,
# accuracy of the emulated three-MFMA (fp16 x 2) product
cd "$(dirname "$0")/../.."
B=scripts/lab/_build
for v in "" gnpad50000 gnpad90000; do
  echo "== gn_time ${v:-product}"
  if [ -n "$v" ]; then ACR_LAB_LIB=$PWD/$B/libacr_$v.so python scripts/lab/gn_time.py 2>/dev/null; else python scripts/lab/gn_time.py 2>/dev/null; fi
done
for v in "" gareuse gmfma4 gmfma3 ""; do
  echo "== gemm_x3_time ${v:-product}"
  if [ -n "$v" ]; then ACR_LAB_LIB=$PWD/$B/libacr_$v.so python scripts/lab/gemm_x3_time.py 2>/dev/null; else python scripts/lab/gemm_x3_time.py 2>/dev/null; fi
done
echo "== split_fp16x2_accuracy"
python scripts/lab/split_fp16x2_accuracy.py 2>&1 | grep -v amdgpu.ids
