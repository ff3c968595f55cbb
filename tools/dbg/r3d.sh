#!/bin/bash
set -o pipefail
B="python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary"
run() { echo "== $1"; env $1 $B 2> gpurun_out/r3d_err.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'tok/s', d['value'], 'attn us', d['roofline']['us_per_launch'])
"; }
MV_ROPE_FOLD=0 MV_KV_PREFETCH=1 MV_KV_PREFETCH_MIN_ROWS=1 python -m pytest tests/test_gpu_e2e.py -x -q -k "graph or speculat" > gpurun_out/r3d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3d_pytest.log
run MV_KV_PREFETCH=0
run MV_KV_PREFETCH=2
run MV_KV_PREFETCH=1
run MV_KV_PREFETCH=3
run "MV_KV_PREFETCH=2 MV_KV_PREFETCH_FLAGS=1"
run MV_KV_PREFETCH=4
run MV_KV_PREFETCH=0
run MV_KV_PREFETCH=6
run "MV_KV_PREFETCH=0 MV_W_PREFETCH=7"
run "MV_KV_PREFETCH=2 MV_W_PREFETCH=7"
run "MV_KV_PREFETCH=0 MV_W_PREFETCH=1"
run MV_KV_PREFETCH=0
