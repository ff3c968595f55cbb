#!/bin/bash
python tools/dbg/mid_ablate.py > gpurun_out/r3g_ablate.log 2>&1; echo "rc=$?"; cat gpurun_out/r3g_ablate.log
