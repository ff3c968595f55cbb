#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel time inside the timed decode
steps only (graph capture / warm-up / the post-run roofline probe are cut away).  A decode step
ends with exactly one sample_kernel dispatch, so the last K of them delimit the last K steps.

    python tools/prof_summary.py gpurun_out/prof/<pid>_kernel_trace.csv --steps 8 > profiles/rNN_decode.txt
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return f"hipBLASLt GEMM {m.group(1) if m else ''}"
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--skip-last", type=int, default=4,
                    help="decode steps at the end of the run that are NOT part of the timed loop (bench.py's eager "
                         "roofline probe: one warm-up + three probe steps)")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "sample_kernel" in r[2]]
    if a.skip_last:
        ends = ends[:-a.skip_last]
    if len(ends) < a.steps + 1:
        raise SystemExit("not enough sample_kernel dispatches in the trace")
    lo, hi = ends[-a.steps - 1] + 1, ends[-1] + 1
    window = rows[lo:hi]
    wall = (window[-1][1] - window[0][0]) / a.steps
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in window:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    busy = sum(v[1] for v in agg.values()) / a.steps
    print(f"# {a.trace}: last {a.steps} decode steps; GPU span per step {wall/1e3:.1f} us, "
          f"kernel-busy per step {busy/1e3:.1f} us, {len(window)//a.steps} dispatches per step")
    print(f"{'kernel':92s} {'calls/step':>10s} {'avg us':>9s} {'us/step':>9s} {'% busy':>7s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:92s} {c/a.steps:10.1f} {t/c/1e3:9.2f} {t/a.steps/1e3:9.1f} {100*t/a.steps/busy:7.1f}")


if __name__ == "__main__":
    main()
