#!/bin/bash
set -o pipefail
python tools/kv_stride_probe.py --cases 4864,4864@40,4864@78,16384,8192@50,4864@20 > gpurun_out/r3c_kvstride.log 2>&1; echo "kv rc=$?"; cat gpurun_out/r3c_kvstride.log
python -m pytest tests -m gpu -x -q > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3c_pytest.log
