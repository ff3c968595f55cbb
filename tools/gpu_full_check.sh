#!/bin/bash
# One GPU call that checks a round: the full GPU suite, the driver's bench command, the default user path (--auto-kv), the N = 2 plan and
# the N = 8 record flow rehearsed with two ranks on the one card.  Logs: gpurun_out/r4f_*
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4f_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4f_pytest.log
tail -5 gpurun_out/r4f_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4f_bench.log 2> gpurun_out/r4f_bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --auto-kv > gpurun_out/r4f_bench_autokv.log 2> gpurun_out/r4f_bench_autokv.err; echo "bench auto-kv rc=$?"
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4f_bench_tp2.log 2> gpurun_out/r4f_bench_tp2.err; echo "bench tp2 rc=$?"
grep "\[bench" gpurun_out/r4f_bench_tp2.err | tail -4
MINIVLLM_TP_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --rehearse-n8 > gpurun_out/r4f_bench_n8flow.log 2> gpurun_out/r4f_bench_n8flow.err; echo "bench n8-flow rehearsal rc=$?"
grep "\[bench" gpurun_out/r4f_bench_n8flow.err | tail -8; tail -3 gpurun_out/r4f_bench_n8flow.err
python - <<'PY'
import json
for f in ("r4f_bench", "r4f_bench_autokv", "r4f_bench_tp2", "r4f_bench_n8flow"):
    l = [x for x in open(f"gpurun_out/{f}.log") if x.startswith("{")]
    if not l:
        print(f, "NO LINE"); continue
    d = json.loads(l[-1])
    print(f, d["value"], d["ms_per_step"], [ (s.get("model"), s.get("status") or s.get("value")) for s in (d.get("secondary") or [])])
PY
