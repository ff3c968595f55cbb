#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r3q_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3q_pytest.log
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3q_bench_tp2.log 2> gpurun_out/r3q_bench_tp2.err; echo "bench tp2 rc=$?"; tail -c 1500 gpurun_out/r3q_bench_tp2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
