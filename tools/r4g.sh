#!/bin/bash
# round 4: profile artefacts (rocprofv3 traces, PMC passes, decode sweeps) + the barrier probe with mode 3
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 tools/dbg/barrier_probe.bin > gpurun_out/r4g_barrier.log 2>&1; echo "barrier probe rc=$?"; cat gpurun_out/r4g_barrier.log | grep "mode"
bash tools/profiles_round.sh r04 2>&1 | tail -30
