#!/bin/bash
cd "$(dirname "$0")/../.."
for args in "64 4 step f32_split det" "96 2 loop f32_split det" "448 2 loop f32_split" "448 2 loop f32_split det"; do
  python scripts/lab/step_repeat.py $args 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using" | grep "^det\|^step"
done
python scripts/lab/split_fp16x2_accuracy.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|lt = " | tail -5
python -m pytest tests/test_dp_gpu.py -q -x -k hybrid -s 2>&1 | grep "hybrid DP\|passed\|failed"
