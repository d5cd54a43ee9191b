#!/bin/bash
# round-6 lab session 3: register-resident GroupNorm for every group of the step (forward) / groups of <= 13 slots (backward)
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py -q -k "groupnorm or stem_kernels_full_size" 2>&1 | tail -3
echo "== gn_time OLD kernels (control build)"; ACR_LAB_LIB=$PWD/scripts/lab/_build/libacr_gnpad50000.so python scripts/lab/gn_time.py 2>/dev/null
echo "== gn_time NEW"; python scripts/lab/gn_time.py 2>/dev/null
echo "== same-box A/B: round-5 tree (old) vs current (new)"
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} scripts/lab/ab_session.sh
