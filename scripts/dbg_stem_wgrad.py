"""fp32 4x4/1 pad-2 (space-to-depth stem) weight gradient against an fp64 evaluation, two image sizes."""
import torch
import torch.nn.functional as F
from iif_amd import ops

dev = "cuda:0"
for (n, hs, dt) in ((8, 32, torch.float32), (8, 112, torch.float32), (8, 112, torch.bfloat16), (64, 112, torch.float32)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, hs, hs, 16, generator=g)
    dy = torch.randn(n, hs, hs, 64, generator=g)
    dy = dy - dy.mean((0, 1, 2), keepdim=True)
    if dt == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    dw = ops.conv_wgrad(x.to(dev).to(dt), dy.to(dev).to(dt), 4, 4, 1, 2, workspace=ws)      # [64, 4*4*16] KRSC
    torch.cuda.synchronize()
    xp = F.pad(x.permute(0, 3, 1, 2).double(), (2, 1, 2, 1))
    w = torch.zeros(64, 16, 4, 4, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(xp, w)
    assert out.shape[-1] == hs
    (gw,) = torch.autograd.grad(out, w, dy.permute(0, 3, 1, 2).double())
    ref = gw.permute(0, 2, 3, 1).reshape(64, 256)
    d = dw.double().cpu()
    e = (d - ref).norm() / ref.norm()
    per_tap = ((d - ref).view(64, 4, 4, 16).pow(2).sum((0, 3)).sqrt() / ref.view(64, 4, 4, 16).pow(2).sum((0, 3)).sqrt())
    print(n, hs, dt, "rel L2 %.3e" % e.item())
    print(per_tap)
