# round 2, final refresh of the sweep tables and the B=1 trace (run from the repo root on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c
mkdir -p $O
python tools/sweep_decode.py > $O/sweep17.log 2>&1
echo sweep17 done
python tools/sweep_decode.py --model qwen2-7b > $O/sweep7.log 2>&1
echo sweep7 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -o run -- python3 bench.py --no-cpu-baseline --batch 1 --steps 24 > $O/b1.log 2>&1
python tools/prof_summary.py $(ls $O/prof_b1/*/run_kernel_trace.csv $O/prof_b1/run_kernel_trace.csv 2>/dev/null | head -1) --steps 16 > $O/decode_qwen3-1.7b_B1_ctx1024.txt
rm -rf $O/prof_b1
python tools/bench_linear.py --packed --no-sweep-packed --ms 1,4,8,16 > $O/small_m_table.log 2>&1
echo done
