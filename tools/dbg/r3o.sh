#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_e2e.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/bench_kernels.py --prefill-only --variants 0,4,3 > $O/prefill_kernel.log 2>&1; grep variant $O/prefill_kernel.log | tail -24
csvof() { ls $1/*/run_$2.csv $1/run_$2.csv 2>/dev/null | head -1; }
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o run -- python3 $3 > $O/pmc_$1.log 2>&1; cp $(csvof $O/pmc_$1 counter_collection) $O/pmc_$1_counter_collection.csv; rm -rf $O/pmc_$1; }
pmc prefill_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/prefill_probe.py
pmc prefill_waves "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" tools/prefill_probe.py
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python tools/pmc_summary.py $O/pmc_prefill_mfma_counter_collection.csv $c | grep prefill; done
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES; do python tools/pmc_summary.py $O/pmc_prefill_waves_counter_collection.csv $c | grep prefill; done
python tools/bench_prefill.py --model qwen2-7b --cases 16x1024,4x4096 2>/dev/null | grep "^|"
python tools/bench_prefill.py --model qwen3-1.7b --cases 16x1024,4x4096 2>/dev/null | grep "^|"
