#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sampl or nucleus or topc or mass or draw or top_" > gpurun_out/r3o_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3o_pytest.log
tail -15 gpurun_out/r3o_pytest.log
