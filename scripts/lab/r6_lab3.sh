#!/bin/bash
# round-6 lab session 3: register-resident GroupNorm + ReLU mask bytes
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py -q -k "groupnorm or stem_kernels_full_size" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -k "train_hybrid_64 or train_hybrid_448 or reproducible or full_batch" 2>&1 | tail -3
echo "== gn_time NEW"; python scripts/lab/gn_time.py 2>/dev/null
echo "== same-box A/B: round-5 tree (old) vs current (new)"
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} bash scripts/lab/ab_session.sh
