import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minivllm_amd import hip_ops as H
torch.manual_seed(0)
dt = torch.float16
m, n, k, splits = 256, 3584, 2048, 4
x = torch.randn(m, k).to(dt).cuda(); w = (torch.randn(n, k) * k ** -0.5).to(dt).cuda()
res = torch.randn(m, n).to(dt).cuda(); nw = (1 + 0.1 * torch.randn(n)).to(dt).cuda()
wp = H.pack_weight(w)
part = H.linear_mid_m_partial(x, wp, n, splits, cfg=1)
s32 = ((part[0] + part[1]) + part[2]) + part[3]
x16 = s32.to(dt)
y_ref, r_ref = H.add_rmsnorm(x16, res.clone(), nw, 1e-6)
y, r = H.add_rmsnorm_partial(part, res.clone(), nw, 1e-6)
print("r equal", torch.equal(r, r_ref), "y equal", torch.equal(y, y_ref), "y mismatches", (y != y_ref).sum().item(),
      "rows with mismatch", (y != y_ref).any(1).sum().item())
sub = (x16.float().abs() < 6.2e-5) & (x16 != 0)
print("subnormal x16 elements", sub.sum().item(), "rows containing one", sub.any(1).sum().item(),
      "rows mismatching that contain one", ((y != y_ref).any(1) & sub.any(1)).sum().item())
tiny = (s32.abs() < 6.2e-5)
print("|sum| < 6.1e-5:", tiny.sum().item())
# which step differs?
y1, r1 = H.add_rmsnorm_partial(s32.unsqueeze(0).contiguous(), res.clone(), nw, 1e-6)
print("pre-summed single split == ref:", torch.equal(y1, y_ref), torch.equal(r1, r_ref), (y1 != y_ref).sum().item())
print("pre-summed single split == 4-split kernel:", torch.equal(y1, y), (y1 != y).sum().item())
rows = (y != y_ref).any(1).nonzero().flatten()[:3].tolist()
for rr in rows:
    cols = (y[rr] != y_ref[rr]).nonzero().flatten().tolist()
    print("row", rr, "cols", cols, [(y[rr, c].item(), y_ref[rr, c].item()) for c in cols])
    # fp64 recomputation of the row
    v = x16[rr].double() + res[rr].double()
    v32 = (x16[rr].float() + res[rr].float())
    rs = 1.0 / torch.sqrt((v32.double() ** 2).mean() + 1e-6)
    print("   exact y:", [(v32[c].double() * rs * nw[c].double()).item() for c in cols])
# bf16 for comparison
dt = torch.bfloat16
