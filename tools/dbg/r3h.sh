#!/bin/bash
python tools/dbg/mid_clock.py > gpurun_out/r3h_clock.log 2>&1; echo "rc=$?"; cat gpurun_out/r3h_clock.log
