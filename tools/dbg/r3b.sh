#!/bin/bash
set -o pipefail
python tools/dbg/chunk_dbg.py > gpurun_out/r3b_chunk.log 2>&1; echo "chunk rc=$?"; tail -8 gpurun_out/r3b_chunk.log
python tools/kv_stride_probe.py --cases 4864,4864+42278584320,5120,6144,8192,12288,16384,4608 > gpurun_out/r3b_kvstride.log 2>&1; echo "kv rc=$?"; cat gpurun_out/r3b_kvstride.log
