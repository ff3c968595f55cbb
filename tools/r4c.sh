#!/bin/bash
# round 4, third GPU call: the persistent small-batch decode kernel — parity first (bounded by timeout), then timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_persistent.py -q > gpurun_out/r4c_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -25 gpurun_out/r4c_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for b in 1 8; do
  for p in 0 1; do
    MV_PERSISTENT=$p timeout -k 10 300 python bench.py --batch $b --steps 64 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r4c_b${b}_p${p}.log 2> gpurun_out/r4c_b${b}_p${p}.err; echo "B=$b persistent=$p rc=$?"
    python - <<PY
import json
l=[x for x in open("gpurun_out/r4c_b${b}_p${p}.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("   ms_per_step", d.get("ms_per_step"), "tok/s", d.get("value"))
PY
  done
done
