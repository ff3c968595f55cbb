#!/bin/bash
# round 3, first GPU call: GPU tests, the new bench line (N=1 with secondary records), bench --gpus 2 started bare, KV stride probe
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench.log 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r3a_bench.log
python tools/kv_stride_probe.py > gpurun_out/r3a_kvstride.log 2>&1; echo "kvstride rc=$?"
cat gpurun_out/r3a_kvstride.log
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3a_bench_tp2.log 2> gpurun_out/r3a_bench_tp2.err; echo "bench tp2 rc=$?"
tail -c 2500 gpurun_out/r3a_bench_tp2.log; tail -5 gpurun_out/r3a_bench_tp2.err
