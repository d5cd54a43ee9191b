#!/bin/bash
# round-6 lab session 5: head-mean stream walking the samples in the order of recency of the forward that wrote their scores
cd "$(dirname "$0")/../.."
for i in 1 2; do
echo "== attn_gen product"; python scripts/lab/attn_gen.py 32 785 2>/dev/null | grep "split x3" | grep -v notail
echo "== attn_gen pmean recency order"; ACR_LAB_LIB=$PWD/scripts/lab/_build/libacr_pmrecent.so python scripts/lab/attn_gen.py 32 785 2>/dev/null | grep "split x3" | grep -v notail
done
