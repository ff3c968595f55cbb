#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_ops.py -m gpu -x -q -k "prefill" > gpurun_out/r3q_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3q_pytest.log
tail -4 gpurun_out/r3q_pytest.log
grep -q "rc=0" gpurun_out/r3q_pytest.log || exit 1
timeout -k 10 400 python tools/bench_kernels.py --prefill-only --variants 0,4 > gpurun_out/r3q3_bench.log 2>&1
grep "variant" gpurun_out/r3q3_bench.log | tail -24
