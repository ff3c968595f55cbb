"""Per-rank device time of the gather-free nucleus sequence (collectives excluded: one GPU) against the full-row sampler."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from minivllm_amd import hip_ops as H

dev = torch.device("cuda", 0)
B, V = 256, 151936
g = torch.Generator().manual_seed(0)
full = (torch.randn(B, V, generator=g) * 2.5).bfloat16().to(dev)
temps = torch.full((B,), 0.8, device=dev)
top_ps = torch.full((B,), 0.9, device=dev)
min_ps = torch.zeros(B, device=dev)
ks = torch.full((B,), -1, dtype=torch.int64, device=dev)
rng = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


out = {"full_row_sampler_us": timed(lambda: H.sample(full, temps, ks, top_ps, min_ps, seed=1, offset=1))}
for tp in (2, 4, 8):
    shard = full[:, : V // tp].contiguous()
    row_max = full.float().amax(dim=1)

    def seq():
        rm = shard.amax(dim=1).float()
        h0 = H.shard_key_hist(shard, row_max, temps, None)
        st, pf = H.nucleus_pick(h0, 0, top_ps, min_ps, temps, row_max, shard.dtype)
        h1 = H.shard_key_hist(shard, row_max, temps, pf)
        fl = H.nucleus_pick(h1, 1, top_ps, min_ps, temps, row_max, shard.dtype, st, pf)
        m = H.mask_below(shard, fl)
        loc, stats = H.sample(m, temps, None, None, None, seed=1, offset=0, offset_dev=rng, return_stats=True)
        buf = H.mass_pack(stats, loc, 0, tp, V // tp, B, shard.dtype)
        return H.mass_merge(buf, B, temps, 1, 0, rng)

    pf0 = torch.zeros(B, dtype=torch.int32, device=dev)
    out[f"tp{tp}"] = {
        "sequence_us": timed(seq),
        "hist_level0_us": timed(lambda: H.shard_key_hist(shard, row_max, temps, None)),
        "hist_level1_us": timed(lambda: H.shard_key_hist(shard, row_max, temps, pf0)),
        "mask_us": timed(lambda: H.mask_below(shard, row_max - 3)),
        "local_sample_us": timed(lambda: H.sample(shard, temps, None, None, None, seed=1, offset=1, return_stats=True)),
        "shard_MB": shard.numel() * 2 / 1e6,
    }
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/nuc_time.json", "w"), indent=1)
