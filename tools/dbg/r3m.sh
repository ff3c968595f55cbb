#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r3m_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3m_pytest.log
python tools/bench_kernels.py --prefill-only --variants 0,4,3 > gpurun_out/r3m_prefill.log 2>&1; echo "rc=$?"; tail -24 gpurun_out/r3m_prefill.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3m_bench.log 2> gpurun_out/r3m_bench.err; echo "bench rc=$?"; head -c 600 gpurun_out/r3m_bench.log
