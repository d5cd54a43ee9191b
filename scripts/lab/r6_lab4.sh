#!/bin/bash
# round-6 lab session 4: attention backward with the dK/dV and dQ workgroups of a (sample, head) dispatched side by side
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -2
for i in 1 2; do
echo "== attn_gen OLD order"; ACR_LAB_LIB=$PWD/scripts/lab/_build/libacr_attnold.so python scripts/lab/attn_gen.py 32 785 2>/dev/null | grep "split x3 \|split x3$" | grep -v notail
echo "== attn_gen NEW order"; python scripts/lab/attn_gen.py 32 785 2>/dev/null | grep "split x3" | grep -v notail
done
echo "== T = 2305 B = 16"; ACR_LAB_LIB=$PWD/scripts/lab/_build/libacr_attnold.so python scripts/lab/attn_gen.py 16 2305 2>/dev/null | grep "split x3" | grep -v notail; python scripts/lab/attn_gen.py 16 2305 2>/dev/null | grep "split x3" | grep -v notail
echo "== same-box A/B: round-5 tree (old) vs current (new)"
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} bash scripts/lab/ab_session.sh
