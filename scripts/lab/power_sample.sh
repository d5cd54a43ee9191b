#!/bin/bash
# GPU box: sample rocm-smi power / clocks every 0.2 s while the bench's f32_split (then f32, bf16) steps run.  usage: power_sample.sh <mode> <steps>
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mode=${1:-f32_split}; steps=${2:-40}
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > "$ROOT/gpurun_out/power_$mode.jsonl" &
SP=$!
python3 "$ROOT/bench.py" --dtype $mode --steps $steps --warmup 3 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | cut -c1-160
kill $SP
python3 - "$ROOT/gpurun_out/power_$mode.jsonl" <<'PY'
import json, sys
pw, sclk = [], []
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except Exception:
        continue
    for card, v in d.items():
        for k, x in v.items():
            if "ower" in k and "W" in k:
                try: pw.append(float(x))
                except Exception: pass
            if "sclk" in k and "clock speed" in k:
                try: sclk.append(float(str(x).strip("()Mhz ")))
                except Exception: pass
        break
pw_run = sorted(pw)[len(pw) // 3:]          # the upper two thirds: samples taken while the steps ran
print("power samples %d: max %.0f W, mean of the upper two thirds %.0f W; sclk samples %d: %s" % (len(pw), max(pw or [0]), sum(pw_run) / max(1, len(pw_run)), len(sclk), sorted(set(int(s) for s in sclk))[-6:]))
PY
