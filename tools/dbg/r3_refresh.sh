#!/bin/bash
# end-of-round refresh of the decode artefacts that later commits changed (split-KV combine kernel, sampler histograms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r
mkdir -p $O
csvof() { ls $1/*/run_$2.csv $1/run_$2.csv 2>/dev/null | head -1; }
trace() { # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python3 bench.py --no-cpu-baseline --no-secondary --steps 24 "$@" > $O/bench_$name.log 2>&1
  python tools/prof_summary.py $(csvof $O/prof_$name kernel_trace) --steps 16 > $O/decode_$name.txt
  cp $(dirname $(csvof $O/prof_$name kernel_trace))/run_kernel_stats.csv $O/kernel_stats_full_run_$name.csv
  rm -rf $O/prof_$name
  echo "trace $name done"
}
trace qwen3-1.7b_B256_ctx1024
trace qwen3-1.7b_B1_ctx1024 --batch 1
python tools/sweep_decode.py > $O/sweep17.log 2>&1; echo sweep17 done
python tools/sweep_decode.py --model qwen2-7b > $O/sweep7.log 2>&1; echo sweep7 done
python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo bench done
echo all done
