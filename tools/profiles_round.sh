#!/bin/bash
# The profile artefacts of a round, in one GPU call: decode traces (1.7B, 7B, B=1: rocprofv3 --kernel-trace --stats),
# PMC passes in runs of their own (decode attention FETCH/WRITE_SIZE, mid-M projections, prefill kernel MFMA-busy / wave
# time), decode sweeps, the bench lines.  Usage on the GPU box:  bash tools/profiles_round.sh r04   -> gpurun_out/r04p/...;
# the summaries to be judged are then copied into profiles/ (tracked) by hand with the round prefix.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}p
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
trace qwen2-7b_B256_ctx1024 --model qwen2-7b
trace qwen3-1.7b_B1_ctx1024 --batch 1
pmc() { # name, counters (quoted), probe script
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/pmc_$1 -o run -- python3 $3 > $O/pmc_$1.log 2>&1
  cp $(csvof $O/pmc_$1 counter_collection) $O/pmc_$1_counter_collection.csv
  rm -rf $O/pmc_$1
  echo "pmc $1 done"
}
pmc mid_m_fetch_size FETCH_SIZE tools/pmc_probe_mid.py
pmc mid_m_write_size WRITE_SIZE tools/pmc_probe_mid.py
pmc decode_fetch_size FETCH_SIZE tools/pmc_probe.py
pmc decode_write_size WRITE_SIZE tools/pmc_probe.py
pmc prefill_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/prefill_probe.py
pmc prefill_waves "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" tools/prefill_probe.py
for c in FETCH_SIZE; do python tools/pmc_summary.py $O/pmc_mid_m_fetch_size_counter_collection.csv $c; done > $O/pmc_mid_m_summary.txt
python tools/pmc_summary.py $O/pmc_mid_m_write_size_counter_collection.csv WRITE_SIZE >> $O/pmc_mid_m_summary.txt
python tools/pmc_summary.py $O/pmc_decode_fetch_size_counter_collection.csv FETCH_SIZE > $O/pmc_decode_summary.txt
python tools/pmc_summary.py $O/pmc_decode_write_size_counter_collection.csv WRITE_SIZE >> $O/pmc_decode_summary.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python tools/pmc_summary.py $O/pmc_prefill_mfma_counter_collection.csv $c | grep prefill; done > $O/pmc_prefill_summary.txt
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES; do python tools/pmc_summary.py $O/pmc_prefill_waves_counter_collection.csv $c | grep prefill; done >> $O/pmc_prefill_summary.txt
python tools/sweep_decode.py > $O/sweep17.log 2>&1; echo sweep17 done
python tools/sweep_decode.py --model qwen2-7b > $O/sweep7.log 2>&1; echo sweep7 done
python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2> $O/bench_default.err; echo bench done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --auto-kv > $O/bench_autokv.log 2>&1
python tools/bench_kernels.py --prefill-only --variants 0,4,3 > $O/prefill_kernel.log 2>&1
echo all done
