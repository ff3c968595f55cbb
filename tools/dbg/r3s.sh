#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py -m gpu -x -q -k "decode or split or e2e or generate" > gpurun_out/r3s_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3s_pytest.log
tail -5 gpurun_out/r3s_pytest.log
grep -q "rc=0" gpurun_out/r3s_pytest.log || exit 1
for b in 1 8 32; do
  timeout -k 10 300 python bench.py --batch $b --steps 64 --warmup 8 --no-secondary --no-cpu-baseline > gpurun_out/r3s_b$b.log 2>&1
  tail -1 gpurun_out/r3s_b$b.log | cut -c1-130
done
