# round 2, after the RoPE fold: full GPU suite, sweeps, B=1 trace, bench lines
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -2 $O/tests.log
python tools/sweep_decode.py > $O/sweep17.log 2>&1
python tools/sweep_decode.py --model qwen2-7b > $O/sweep7.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -o run -- python3 bench.py --no-cpu-baseline --batch 1 --steps 24 > $O/b1.log 2>&1
python tools/prof_summary.py $(ls $O/prof_b1/*/run_kernel_trace.csv $O/prof_b1/run_kernel_trace.csv 2>/dev/null | head -1) --steps 16 > $O/decode_qwen3-1.7b_B1_ctx1024.txt
rm -rf $O/prof_b1
python bench.py > $O/bench17.log 2>&1
python bench.py --no-cpu-baseline --model qwen2-7b > $O/bench7.log 2>&1
python bench.py --no-cpu-baseline --auto-kv > $O/bench17_autokv.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -1 $O/smoke.log
echo done
