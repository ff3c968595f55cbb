#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_persistent.py -q -k "norm_fold or persistent or ablation or ties" > gpurun_out/r4e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4e_pytest.log
for b in 5 8; do
  for r in 4 8; do
    MV_NORM_FOLD_MAX_ROWS=$r timeout -k 10 300 python bench.py --batch $b --steps 64 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r4e_b${b}_r${r}.log 2> gpurun_out/r4e_b${b}_r${r}.err; rc=$?
    python - <<PY
import json
l=[x for x in open("gpurun_out/r4e_b${b}_r${r}.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("B=$b norm-fold rows<=$r rc=$rc ms_per_step", d.get("ms_per_step"), "tok/s", d.get("value"))
PY
  done
done
