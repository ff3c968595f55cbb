#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py -x -q -k "prefill or logits or chunked or mixed" > gpurun_out/r3k_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3k_pytest.log
python tools/bench_kernels.py --prefill-only --variants 0,32,4 > gpurun_out/r3k_prefill.log 2>&1; echo "rc=$?"; cat gpurun_out/r3k_prefill.log
