#!/bin/bash
# the N = 2 form of the driver's bench command on ONE GPU (ranks share the card; gloo stands in for RCCL): the launcher
# path, the per-rank all-reduce probe and the cross-rank token digest
mkdir -p gpurun_out
MINIVLLM_TP_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 16 --warmup 4 --batch 64 --no-secondary --no-cpu-baseline > gpurun_out/r3tp2_bench.log 2> gpurun_out/r3tp2_bench.err
echo "rc=$?"
tail -1 gpurun_out/r3tp2_bench.log | cut -c1-900
