#!/bin/bash
python tools/bench_kernels.py --prefill-only --variants 0,64,128,192 > gpurun_out/r3l_prefill.log 2>&1; echo "rc=$?"; grep "4x4096\|16x1024 H=28" gpurun_out/r3l_prefill.log | tail -24
