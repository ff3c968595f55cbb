#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/dbg/persist_dbg.py > gpurun_out/r4d_dbg.log 2>&1; echo "dbg rc=$?"
grep -v "amdgpu.ids\|tokenizer" gpurun_out/r4d_dbg.log | tail -12 | cut -c1-600
