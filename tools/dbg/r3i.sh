#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_ops.py -x -q -k "mid_m or split_k" > gpurun_out/r3i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3i_pytest.log
python tools/tune_mid.py --models qwen3-1.7b,qwen2-7b,lm-heads --ms 64,128,192,256 --merge mini-vllm_amd/tuned/mid_m_gfx950.json --out gpurun_out/mid_m_gfx950_r3.json > gpurun_out/r3i_tune.log 2>&1; echo "tune rc=$?"; tail -60 gpurun_out/r3i_tune.log
