import os, sys
sys.path.insert(0, ".")
import torch
import torch.nn.functional as F
from iif_amd import ops
dev = "cuda:0"
for (n, hw, c) in [(2, 8, 64), (3, 16, 64), (16, 56, 64)]:
    g = torch.Generator().manual_seed(c + hw)
    x = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(dev)
    wt = (torch.randn(c, 9 * c, generator=g) / (9 * c) ** 0.5).bfloat16().to(dev)
    m = n * hw * hw
    out = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
    nt = ops.conv_forward_bnstats(x, wt, 3, 3, 1, 1, out, partial)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float().view(c, 3, 3, c).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = (out.float() - ref).abs()
    print((n, hw, c), "rows", nt, "max err %.3e (ref max %.2f) nan %d" % (err.max().item(), ref.abs().max().item(), torch.isnan(out).sum().item()))
    bad = (err > 0.05).nonzero()
    print("  bad count", bad.shape[0], bad[:6].tolist())
    ps = partial[:nt * 2 * c].view(nt, 2, c).double().sum(0)
    flat = out.double().view(m, c)
    print("  sums err %.2e  sq err %.2e" % (((ps[0] - flat.sum(0)).abs().max() / flat.abs().sum(0).max()).item(), ((ps[1] - (flat * flat).sum(0)).abs().max() / (flat * flat).sum(0).max()).item()))
