#!/bin/bash
set -o pipefail
python tools/dbg/mid_ablate.py > gpurun_out/r3f_ablate.log 2>&1; echo "rc=$?"; cat gpurun_out/r3f_ablate.log
python tools/bench_mid.py --models qwen3-1.7b --only qkv,gate_up --cfgs 11,23,21,14,22,20 > gpurun_out/r3f_mid.log 2>&1; echo "rc=$?"; cat gpurun_out/r3f_mid.log
