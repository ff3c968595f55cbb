#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3final_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r3final_pytest.log
tail -4 gpurun_out/r3final_pytest.log
grep -q "rc=0" gpurun_out/r3final_pytest.log || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final_smoke.log 2>&1; tail -1 gpurun_out/r3final_smoke.log
python bench.py > gpurun_out/r3final_bench.log 2> gpurun_out/r3final_bench.err; tail -1 gpurun_out/r3final_bench.log | cut -c1-400
