set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $(ls $O/pmc_fetch/*/run_counter_collection.csv $O/pmc_fetch/run_counter_collection.csv 2>/dev/null | head -1) FETCH_SIZE > $O/pmc_fetch_summary.txt
python tools/pmc_summary.py $(ls $O/pmc_write/*/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv 2>/dev/null | head -1) WRITE_SIZE > $O/pmc_write_summary.txt
cp $(ls $O/pmc_fetch/*/run_counter_collection.csv $O/pmc_fetch/run_counter_collection.csv 2>/dev/null | head -1) $O/pmc_fetch_counter_collection.csv
cp $(ls $O/pmc_write/*/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv 2>/dev/null | head -1) $O/pmc_write_counter_collection.csv
rm -rf $O/pmc_fetch $O/pmc_write
echo pmc done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof17 -o run -- python3 bench.py --no-cpu-baseline --steps 24 > $O/bench17_prof.log 2>&1
T=$(ls $O/prof17/*/run_kernel_trace.csv $O/prof17/run_kernel_trace.csv 2>/dev/null | head -1)
python tools/prof_summary.py $T --steps 16 > $O/decode_qwen3-1.7b_B256_ctx1024.txt
cp $(dirname $T)/run_kernel_stats.csv $O/kernel_stats_full_run_qwen3-1.7b.csv
rm -rf $O/prof17
echo trace17 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof7 -o run -- python3 bench.py --no-cpu-baseline --steps 24 --model qwen2-7b > $O/bench7_prof.log 2>&1
T=$(ls $O/prof7/*/run_kernel_trace.csv $O/prof7/run_kernel_trace.csv 2>/dev/null | head -1)
python tools/prof_summary.py $T --steps 16 > $O/decode_qwen2-7b_B256_ctx1024.txt
cp $(dirname $T)/run_kernel_stats.csv $O/kernel_stats_full_run_qwen2-7b.csv
rm -rf $O/prof7
echo trace7 done
python bench.py > $O/bench17.log 2>&1
python bench.py --no-cpu-baseline --model qwen2-7b > $O/bench7.log 2>&1
python bench.py --no-cpu-baseline --auto-kv > $O/bench17_autokv.log 2>&1
echo bench done
