#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "prefill_mfma_random" > gpurun_out/r3q_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3q_pytest.log
tail -25 gpurun_out/r3q_pytest.log
grep -q "rc=0" gpurun_out/r3q_pytest.log || exit 1
timeout -k 10 300 python tools/bench_kernels.py --prefill-only --variants 0,5 > gpurun_out/r3q_bench.log 2>&1
tail -30 gpurun_out/r3q_bench.log
