#!/bin/bash
# round 4, second GPU call: barrier / hand-off price probe, the N=1 bench line, the N=2 rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 tools/dbg/barrier_probe.bin > gpurun_out/r4b_barrier.log 2>&1; echo "barrier probe rc=$?"
cat gpurun_out/r4b_barrier.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4b_bench.log 2> gpurun_out/r4b_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r4b_bench.log; grep "\[bench" gpurun_out/r4b_bench.err | tail -40; tail -3 gpurun_out/r4b_bench.err
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4b_bench_tp2.log 2> gpurun_out/r4b_bench_tp2.err; echo "bench tp2 rc=$?"
tail -c 1500 gpurun_out/r4b_bench_tp2.log; grep "\[bench" gpurun_out/r4b_bench_tp2.err | tail -30; tail -5 gpurun_out/r4b_bench_tp2.err
