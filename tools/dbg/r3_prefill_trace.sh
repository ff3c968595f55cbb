#!/bin/bash
# per-kernel times of one prefill shape (rocprofv3 kernel trace of tools/bench_prefill.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3pt
mkdir -p $O
for m in qwen3-1.7b qwen2-7b; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o run -- python3 tools/bench_prefill.py --model $m --cases 16x1024 --reps 3 > $O/bench_$m.log 2>&1
  f=$(ls $O/prof_$m/*/run_kernel_stats.csv $O/prof_$m/run_kernel_stats.csv 2>/dev/null | head -1)
  cp $f $O/kernel_stats_prefill_$m.csv
  rm -rf $O/prof_$m
  echo "$m done"
done
