"""Library GEMM rates on the prefill shapes (tokens x K) @ (N x K)^T, bf16: default backend vs the other BLAS library vs TunableOp."""
import os, sys, time, json
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
SHAPES = {"qwen2-7b": [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944)],
          "qwen3-1.7b": [(4096, 2048), (2048, 2048), (12288, 2048), (2048, 6144)]}
T = 16384


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def run(tag):
    out = {}
    for model, shapes in SHAPES.items():
        tot = 0.0
        for n, k in shapes:
            x = torch.randn(T, k, device=dev).bfloat16()
            w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
            t = timed(lambda: F.linear(x, w))
            out[f"{model} {n}x{k}"] = round(2 * T * n * k / t / 1e12, 1)
            tot += t
        out[f"{model} sum_ms"] = round(tot * 1e3, 3)
    print(tag, json.dumps(out), flush=True)


print("preferred:", torch.backends.cuda.preferred_blas_library())
run("default")
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        run(lib)
    except Exception as e:
        print(lib, "failed", e)
torch.backends.cuda.preferred_blas_library("default")
try:
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(200)
    tn.set_filename("gpurun_out/tunableop_probe.csv")
    run("tunableop(first,tuning)")
    run("tunableop(tuned)")
    tn.write_file()
except Exception as e:
    print("tunable failed", e)
