import sys, os, time, tempfile, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
args = argparse.Namespace(model="qwen3-1.7b", batch=256, ctx=1024, steps=8, warmup=2, gpus=1, eager=False, auto_kv=False, synthetic_kv=False)
with tempfile.TemporaryDirectory() as d:
    eng = bench.build_engine(args, d)
    from minivllm_amd import SamplingParams
    g = torch.Generator().manual_seed(0)
    sp = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    for _ in range(256):
        eng.add_request(torch.randint(0, 151936, (1024,), generator=g).tolist(), sp)
    ts = []
    from minivllm_amd import hip_ops
    import torch.nn.functional as F
    acc = {}
    def wrap(mod, name):
        real = getattr(mod, name)
        def f(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = real(*a, **k)
            torch.cuda.synchronize(); acc.setdefault(name, []).append(time.perf_counter() - t)
            return r
        setattr(mod, name, f)
    if os.environ.get("PF_TRACE"):
        for n in ("linear_small_m", "linear_mid_m", "prefill_varlen", "rope_qk_store", "rmsnorm", "add_rmsnorm", "silu_and_mul", "sample", "pack_weight"):
            wrap(hip_ops, n)
        wrap(F, "linear"); wrap(F, "embedding")
    while eng.scheduler.waiting:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    for k, v in acc.items():
        print(k, "first", round(v[0] * 1e3, 2), "ms; max", round(max(v) * 1e3, 2), "; calls", len(v), "; total", round(sum(v) * 1e3, 1))
    print("prefill step ms:", [round(t * 1e3, 1) for t in ts])
    print("tokens/s overall", round(256 * 1024 / sum(ts)), " steady (steps 3..):", round((len(ts) - 2) * 16384 / sum(ts[2:])))
    eng.exit()
