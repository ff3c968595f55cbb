#!/bin/bash
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3r_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3r_pytest.log
