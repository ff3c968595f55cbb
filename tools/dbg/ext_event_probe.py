"""Do HIP events recorded INSIDE a captured graph (torch.cuda.Event(external=True) -> event-record nodes) time a kernel?"""
import torch
x = torch.randn(64 << 20, device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
e = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(4)]
with torch.cuda.stream(s):
    y.copy_(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        e[0].record(s)
        y.copy_(x)
        e[1].record(s)
        y.mul_(2.0)
        e[2].record(s)
        y.copy_(x)
        e[3].record(s)
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", rep, [round(e[i].elapsed_time(e[i + 1]) * 1e3, 1) for i in range(3)], "us (copy 256 MiB, mul, copy)")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); y.copy_(x); b.record(); torch.cuda.synchronize()
print("eager copy", round(a.elapsed_time(b) * 1e3, 1), "us")
