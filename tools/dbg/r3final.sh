#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3final_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3final_pytest.log
tail -6 gpurun_out/r3final_pytest.log
grep -q "rc=0" gpurun_out/r3final_pytest.log || exit 1
