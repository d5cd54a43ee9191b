#!/bin/bash
# Round profile set (GPU box): rocprofv3 kernel traces of the bench in both precisions + PMC traffic passes over the probe.
#   usage: scripts/profile_round.sh <tag> ["<dtypes>"]    -> gpurun_out/<tag>_{f32,bf16}/, gpurun_out/<tag>_pmc{F,W}_{f32,bf16}/
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1
cd /tmp && export TMPDIR=/tmp
for dt in ${2:-f32_split f32 bf16}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${tag}_${dt}" -o p -- python3 "$ROOT/bench.py" --dtype $dt --steps 6 --warmup 3 --no-cpu-baseline --no-infer > "$ROOT/gpurun_out/${tag}_${dt}.json" 2> "$ROOT/gpurun_out/${tag}_${dt}.err"
  echo "trace $dt done: $(cut -c1-120 "$ROOT/gpurun_out/${tag}_${dt}.json")"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${tag}_pmcF_${dt}" -o p -- python3 "$ROOT/bench.py" --probe-only --dtype $dt > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${tag}_pmcW_${dt}" -o p -- python3 "$ROOT/bench.py" --probe-only --dtype $dt > /dev/null 2>&1
  echo "pmc $dt done"
done
# the judged summaries: built on the box (the traces are too large to travel), copied where gpurun merges them back from
cd "$ROOT" && python scripts/make_in_step.py gpurun_out "$tag" && mkdir -p "gpurun_out/profiles_$tag" && cp profiles/${tag}_* "gpurun_out/profiles_$tag/" && for dt in ${2:-f32_split f32 bf16}; do rm -rf "gpurun_out/${tag}_${dt}" "gpurun_out/${tag}_pmcF_${dt}" "gpurun_out/${tag}_pmcW_${dt}"; done
