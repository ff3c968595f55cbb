# round 2, last refresh: full GPU test-suite, headline traces + bench lines on the final code
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -2 $O/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof17 -o run -- python3 bench.py --no-cpu-baseline --steps 24 > $O/bench17_prof.log 2>&1
T=$(ls $O/prof17/*/run_kernel_trace.csv $O/prof17/run_kernel_trace.csv 2>/dev/null | head -1)
python tools/prof_summary.py $T --steps 16 > $O/decode_qwen3-1.7b_B256_ctx1024.txt
cp $(dirname $T)/run_kernel_stats.csv $O/kernel_stats_full_run_qwen3-1.7b.csv
rm -rf $O/prof17
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof7 -o run -- python3 bench.py --no-cpu-baseline --steps 24 --model qwen2-7b > $O/bench7_prof.log 2>&1
T=$(ls $O/prof7/*/run_kernel_trace.csv $O/prof7/run_kernel_trace.csv 2>/dev/null | head -1)
python tools/prof_summary.py $T --steps 16 > $O/decode_qwen2-7b_B256_ctx1024.txt
cp $(dirname $T)/run_kernel_stats.csv $O/kernel_stats_full_run_qwen2-7b.csv
rm -rf $O/prof7
python bench.py > $O/bench17.log 2>&1
python bench.py --no-cpu-baseline --model qwen2-7b > $O/bench7.log 2>&1
python bench.py --no-cpu-baseline --auto-kv > $O/bench17_autokv.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
echo done
