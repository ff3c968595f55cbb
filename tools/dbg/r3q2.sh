#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "prefill_mfma_random" > gpurun_out/r3q_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3q_pytest.log
tail -5 gpurun_out/r3q_pytest.log
grep -q "rc=0" gpurun_out/r3q_pytest.log || exit 1
# 5 = kernel; 21 = no K/V refresh; 773 = no exponentials, no maxima; 789 = neither, nor refresh
timeout -k 10 400 python tools/bench_kernels.py --prefill-only --variants 0,5,21,773,789 > gpurun_out/r3q2_bench.log 2>&1
grep "4x4096\|16x1024 H=28" gpurun_out/r3q2_bench.log | tail -20
