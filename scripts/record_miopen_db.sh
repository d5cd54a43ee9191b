#!/bin/bash
# Record MIOpen's user find-db for the stem convolution shapes of the bench (both precisions, training + CAM generation) on an
# MI355X:  scripts/record_miopen_db.sh  -> gpurun_out/miopen_db/*.{udb,ufdb}.txt, to be merged into acr_wsss_amd/miopen_db/
# (scripts/merge_miopen_db.py).  A fresh, empty user db makes every convolution run MIOpen's full Find once.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export MIOPEN_USER_DB_PATH="$ROOT/gpurun_out/miopen_db"
rm -rf "$MIOPEN_USER_DB_PATH"; mkdir -p "$MIOPEN_USER_DB_PATH"
cd "$ROOT"
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/miopen_record.json 2> gpurun_out/miopen_record.err
echo "rc $? ; $(ls -la $MIOPEN_USER_DB_PATH | tail -3)"
