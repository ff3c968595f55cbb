#!/bin/bash
timeout -k 10 120 tools/dbg/wreg_probe.bin > gpurun_out/r3n_wreg.log 2>&1; echo "rc=$?"; cat gpurun_out/r3n_wreg.log
