#!/bin/bash
# round 4, first GPU call: GPU tests (new: spawned-TP add_request/step relay, block sizes 128/256, paged-prefix fuzz over
# block sizes), the new N=1 bench line (rooflines on every record), the N=2 rehearsal (two ranks time-slicing the card)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4a_pytest.log
tail -5 gpurun_out/r4a_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.log 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r4a_bench.log; tail -30 gpurun_out/r4a_bench.err
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4a_bench_tp2.log 2> gpurun_out/r4a_bench_tp2.err; echo "bench tp2 rc=$?"
tail -c 1500 gpurun_out/r4a_bench_tp2.log; grep "\[bench" gpurun_out/r4a_bench_tp2.err | tail -30; tail -5 gpurun_out/r4a_bench_tp2.err
