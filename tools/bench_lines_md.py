#!/usr/bin/env python3
"""profiles/rNN_bench_lines.md from the logs of tools/gpu_full_check.sh (gpurun_out/r4f_*.log): the driver's command in full,
a summary table of its records, the auto-kv line, the N = 2 rehearsal and the N = 8 flow rehearsal (abridged)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
passed = sys.argv[2] if len(sys.argv) > 2 else "?"


def line(f):
    rows = [x for x in open(f"gpurun_out/{f}.log") if x.startswith("{")]
    return rows[-1].strip()


def rf(r):
    x = r.get("roofline")
    return "–" if not x else f"{x['kernel']}: {x['achieved']} {x['unit']} = {x['frac']} of {x['peak']}"


def short(r):
    r = dict(r)
    for k in ("roofline", "projection_roofline", "step_roofline"):
        if r.get(k):
            r[k] = {kk: r[k][kk] for kk in ("kernel", "achieved", "frac", "heads_per_rank") if kk in r[k]}
    c = r.get("config")
    if c and c.get("tensor_parallel"):
        tp = dict(c["tensor_parallel"])
        tp["allreduce_probe_per_rank"] = "(per-rank HIP-event times: see the log)"
        r["config"] = dict(c, tensor_parallel=tp)
    return r


d = json.loads(line("r4f_bench"))
out = [f"# bench.py lines, round {tag[1:]} (one MI355X; `tools/gpu_full_check.sh`, logs `gpurun_out/r4f_*`)\n",
       f"All four commands ran in ONE GPU call behind the full `pytest -m gpu` suite ({passed} passed).  The driver's own line of round 3 was "
       f"7.083 ms / 36 143 tok/s (`BENCH_r03.json`); this box read {d['ms_per_step']} ms for the same command.\n",
       "## `python bench.py --steps 20 --warmup 5` (the driver's command; N = 1: headline + `projection_roofline` + `step_roofline` + "
       "`secondary` + both `cpu_baseline` samples)\n", "```\n" + line("r4f_bench") + "\n```\n"]
rows = ["| record | value | ms | roofline (dominant kernel) | other |", "|---|---|---|---|---|"]
rows.append(f"| headline: {d['config']['workload']} | {d['value']} tok/s | {d['ms_per_step']} / step | {rf(d)} | projections "
            f"{d['projection_roofline']['frac']} of HBM peak, {d['projection_roofline']['second_bound']['frac']} of the per-CU ingest bound; "
            f"whole step {d['step_roofline']['frac']} of 8 TB/s |")
for s in d["secondary"]:
    if s["metric"].startswith("decode"):
        pr, sr = s.get("projection_roofline"), s.get("step_roofline")
        extra = (f"projections {pr['frac']} of HBM peak, {pr['second_bound']['frac']} of ingest bound; step {sr['frac']}" if pr
                 else "(sampled rows: the radix-select path of `sample_kernel`)")
        rows.append(f"| {s['config']['workload']} | {s['value']} tok/s | {s['ms_per_step']} / step | {rf(s)} | {extra} |")
    else:
        x = s["roofline"]
        rows.append(f"| prefill {s['model']} {s['prompts']} x {s['prompt_len']} | {s['value']} tok/s | {s['ms_per_batch']} / batch | {rf(s)} | "
                    f"flash kernel = {x['share_of_step']} of the step; MFMA-busy {x['mfma_busy']} (PMC profile) |")
out.append("Summary of that line:\n\n" + "\n".join(rows) + "\n")
a = json.loads(line("r4f_bench_autokv"))
out.append("## `--auto-kv --no-secondary --no-cpu-baseline` (KV pool sized from free HBM = the default user path)\n")
out.append(f"{a['value']} tok/s, {a['ms_per_step']} ms per step, kv_blocks {a['config']['kv_blocks']}, attention {a['roofline']['achieved']} GB/s "
           f"({a['roofline']['frac']}).\n")
t = json.loads(line("r4f_bench_tp2"))
out.append("## `MINIVLLM_TP_BACKEND=gloo python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline`, started BARE (rehearsal of the N = 2 plan)\n")
out.append("Two ranks time-slicing the ONE GPU (gloo + the peer-to-peer all-reduce kernel): a functional run of `record_plan(2)` — Qwen3-1.7B "
           "TP=2, then Qwen2-7B TP=2 decode + prefill 16 x 1024 / 4 x 4096 — **its timings are not a scaling point** (they swing several-fold "
           "from run to run with the hardware scheduler's time slices).\n")
out.append("```\n" + json.dumps({k: (v if k != "secondary" else [short(s) for s in v]) for k, v in short(t).items()
                                  if k != "cpu_baseline"}) + "\n```\n")
n = json.loads(line("r4f_bench_n8flow"))
out.append("## `... --gpus 2 --rehearse-n8` (the N = 8 record FLOW on stand-ins: rejection record, then a third engine at Qwen2-72B's layer shape)\n")
out.append("```\n" + json.dumps([short(s) for s in n["secondary"]]) + "\n```\n")
open(f"profiles/{tag}_bench_lines.md", "w").write("\n".join(out))
print("wrote", f"profiles/{tag}_bench_lines.md")
