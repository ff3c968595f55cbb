#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rccl.py tests/test_tp.py -m gpu -x -q -k "nucleus or world1 or llm_spawned_tp2" > gpurun_out/r3n_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3n_pytest.log
tail -30 gpurun_out/r3n_pytest.log
