# round 2: prefill artefacts (whole-step tokens/s at the config-4 shapes; PMC of prefill_mfma_kernel in two passes)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f
mkdir -p $O
python tools/bench_prefill.py --model qwen2-7b > $O/prefill_7b.log 2>&1
python tools/bench_prefill.py --model qwen3-1.7b > $O/prefill_17b.log 2>&1
echo tokens done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o run -- python3 tools/prefill_probe.py > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/pmc2 -o run -- python3 tools/prefill_probe.py > $O/pmc2.log 2>&1
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python tools/pmc_summary.py $(ls $O/pmc1/*/run_counter_collection.csv $O/pmc1/run_counter_collection.csv 2>/dev/null | head -1) $c | grep prefill; done > $O/pmc_summary.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES; do python tools/pmc_summary.py $(ls $O/pmc2/*/run_counter_collection.csv $O/pmc2/run_counter_collection.csv 2>/dev/null | head -1) $c | grep prefill; done >> $O/pmc_summary.txt
cp $(ls $O/pmc1/*/run_counter_collection.csv $O/pmc1/run_counter_collection.csv 2>/dev/null | head -1) $O/pmc_prefill_mfma_counter_collection.csv
cp $(ls $O/pmc2/*/run_counter_collection.csv $O/pmc2/run_counter_collection.csv 2>/dev/null | head -1) $O/pmc_prefill_waves_counter_collection.csv
rm -rf $O/pmc1 $O/pmc2
echo done
