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
# Round 6: the clock x matrix-pipe-occupancy table and the CAM-generation trace come from THIS run too -- one script,
# one tree, one commit, so they cannot drift from the step breakdown again.  (Counters in a pass of their own: --pmc + --kernel-trace only.)
case " ${2:-f32_split f32 bf16} " in *" f32_split "*)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${tag}_busy" -o p -- python3 "$ROOT/bench.py" --dtype f32_split --steps 3 --warmup 2 --no-cpu-baseline --no-infer --no-roofline > /dev/null 2>&1
  (echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --dtype f32_split --steps 3 --warmup 2 --no-roofline   [$tag tree, scripts/profile_round.sh]"; python3 "$ROOT/scripts/lab/mfma_busy_from_pmc.py" "$ROOT/gpurun_out/${tag}_busy") > "$ROOT/profiles/${tag}_clock_mfma_busy_f32_split.txt" 2>&1
  echo "busy pass done"
  rocprofv3 --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${tag}_infer" -o p -- python3 "$ROOT/scripts/lab/infer_list_busy.py" 5 > "$ROOT/gpurun_out/${tag}_infer.log" 2>&1
  (echo "# CAM generation, f32_split, scales {0.5,1,1.5,2}, acr_wsss_amd.infer_cam.infer_cam_list with its defaults (batches of 8, pass graphs on): rocprofv3 --kernel-trace"; echo "# -- python3 scripts/lab/infer_list_busy.py 5; the last 3 batches = 24 images, figures PER IMAGE   [$tag tree, scripts/profile_round.sh]"; echo "# (the four scales of a batch run on streams of their own: kernel durations overlap, so GPU busy = the SUM of durations exceeds the span)"; grep "img/s" "$ROOT/gpurun_out/${tag}_infer.log" | sed 's/^/# under the tracer: /'; python3 "$ROOT/scripts/lab/infer_trace_summary.py" "$(ls "$ROOT"/gpurun_out/${tag}_infer/*/*kernel_trace.csv "$ROOT"/gpurun_out/${tag}_infer/*kernel_trace.csv 2>/dev/null | head -1)" 3 40 4 8) > "$ROOT/profiles/${tag}_infer_trace_summary_f32_split.txt" 2>&1
  echo "infer trace done"
  rm -rf "$ROOT/gpurun_out/${tag}_busy" "$ROOT/gpurun_out/${tag}_infer"
;; esac
# the judged summaries: built on the box (the traces are too large to travel), copied where gpurun merges them back from
cd "$ROOT" && python scripts/make_in_step.py gpurun_out "$tag" && mkdir -p "gpurun_out/profiles_$tag" && cp profiles/${tag}_* "gpurun_out/profiles_$tag/" && for dt in ${2:-f32_split f32 bf16}; do rm -rf "gpurun_out/${tag}_${dt}" "gpurun_out/${tag}_pmcF_${dt}" "gpurun_out/${tag}_pmcW_${dt}"; done
