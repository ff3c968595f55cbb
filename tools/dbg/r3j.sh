#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_ops.py -x -q -k "mass or topc or sampl" > gpurun_out/r3j_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/r3j_ops.log
python -m pytest tests/test_tp.py -x -q -m gpu -k "spawned" > gpurun_out/r3j_tp.log 2>&1; echo "tp rc=$?"; tail -15 gpurun_out/r3j_tp.log
