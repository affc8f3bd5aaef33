"""Column sums of the BN3-algebra data gradient: GPU result against a float64 evaluation with the GPU's own stacked weights / bias."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iif_amd import ops
DEV = "cuda:0"
n, hw, c, C = 2, 14, 64, 256
m = n * hw * hw
g = torch.Generator().manual_seed(c + C + hw)
dt = torch.bfloat16
a2 = torch.relu(torch.randn(m, c, generator=g)).to(dt)
W = (torch.randn(C, c, generator=g) / c ** 0.5).to(dt)
y = a2.float() @ W.float().t()
mu, var = y.mean(0), y.var(0, unbiased=False)
invstd = torch.rsqrt(var + 1e-5)
gamma = torch.rand(C, generator=g) + 0.5
gt = (torch.randn(m, C, generator=g) * (torch.rand(m, C, generator=g) > 0.4)).to(dt)
d = lambda t: t.to(DEV)
a2d, gtd = d(a2).view(n, hw, hw, c), d(gt).view(n, hw, hw, C)
ldw = c
Wd = d(W).contiguous()
ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
P = ops.conv_wgrad(a2d, gtd, 1, 1, 1, 0, ldw=ldw, workspace=ws)
sums = torch.empty(2, c, device=DEV)
ops.bn_stats_sums(a2d.view(m, c), sums, ops.bn_workspace(m, c, DEV))
npart = 70
part = torch.zeros(npart, 2, C)
for r in range(npart):
    part[r, 0] = gt[r::npart].float().sum(0)
part[:, 1] = 1e30
part = d(part)
stats = torch.zeros(4, C, device=DEV); stats[0] = d(mu); stats[1] = d(invstd)
coef = torch.empty(3, C, device=DEV)
dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
wt = torch.zeros(c, C + c, dtype=dt, device=DEV)
bias = torch.empty(c, device=DEV)
tickets = torch.zeros(64, dtype=torch.int32, device=DEV)
for csum2 in (None, sums[0].contiguous()):
    ops.bn3_algebra_prep(P, Wd, c, part, npart, stats, d(gamma), m, coef, dgam, dbet, wt, bias, ops.bn3_algebra_prep_scratch(C, c, DEV), tickets, colsum2=csum2)
    da = torch.full((n, hw, hw, c), float("nan"), dtype=dt, device=DEV)
    ops.conv_dgrad2_bnbwd(gtd, a2d, wt, bias, da)
    wtc, bc = wt.float().cpu().double(), bias.cpu().double()
    em = gt.double() @ wtc[:, :C].t() + a2.double() @ wtc[:, C:].t() + bc
    got = da.float().cpu().view(m, c).double()
    print("compensation", csum2 is not None, ": colsum max  GPU %.4f   fp64 with the GPU's weights %.4f   that, rounded to bf16 %.4f   max |GPU - rounded| %.4g" % (
        got.sum(0).abs().max(), em.sum(0).abs().max(), em.to(dt).double().sum(0).abs().max(), (got - em.to(dt).double()).abs().max()))
    print("   s1 check: sum g~ max abs", gt.double().sum(0).abs().max().item(), " csum a2 err", (sums[0].cpu().double() - a2.double().sum(0)).abs().max().item())
