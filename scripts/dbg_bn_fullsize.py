"""Accuracy of the fp32 BN forward statistics / backward at the full-size row count, on a replicated batch."""
import torch
from iif_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
C, hw, B, rep = 2048, 49, 8, 32
xs = (torch.randn(B * hw, C) * (0.5 + torch.rand(C)) + torch.randn(C)).float()
gi = torch.randn(B, 1, C) * 1e-3
gs = gi.expand(B, hw, C).reshape(B * hw, C).contiguous().float()
gamma = (0.25 * torch.ones(C)).float()
beta = torch.zeros(C)


def ref(x, g):
    x, g = x.double(), g.double()
    n = x.shape[0]
    mean = x.mean(0); var = x.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    xh = (x - mean) * inv
    dbeta = g.sum(0); dgamma = (g * xh).sum(0)
    dx = gamma.double() * inv * (g - dbeta / n - xh * dgamma / n)
    return mean, inv, dgamma, dbeta, dx


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


for name, reps in (("small", 1), ("full", rep)):
    x = xs.repeat(reps, 1).to(dev); g = (gs.repeat(reps, 1) / reps).to(dev)
    m = x.shape[0]
    stats = torch.zeros(4, C, device=dev); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    ws = ops.bn_workspace(m, C, dev)
    ops.bn_forward_stats(x, gamma.to(dev), beta.to(dev), rm, rv, stats, ws)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev); dx = torch.empty_like(x)
    ops.bn_backward(g, None, x, stats, gamma.to(dev), dgamma, dbeta, dx, ws)
    torch.cuda.synchronize()
    mean, inv, rdg, rdb, rdx = ref(x.cpu(), g.cpu())
    print(name, "stats rows:", [rel(stats[i], r) for i, r in ((0, mean),)], "dgamma %.2e dbeta %.2e dx %.2e" % (rel(dgamma, rdg), rel(dbeta, rdb), rel(dx, rdx)))
    print("   stats[1] vs invstd %.2e  vs var %.2e" % (rel(stats[1], inv), rel(stats[1], x.double().var(0, unbiased=False).cpu())))
