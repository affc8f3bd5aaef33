import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["IIF_BN3_ALGEBRA_PURE_MIN_ELEMS"] = "0"
os.environ["IIF_TWOPASS"] = os.environ.get("TP", "1")
os.environ["IIF_SIDE_STREAMS"] = "1"
import torch
import test_resnet_gpu as T
from iif_amd import _lib
from iif_amd.custom import IIFLoss
arch, C, B, hw = "resnet50", 1000, 32, 64
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
net, sd = T._build(arch, C, torch.bfloat16)
net.load_state_dict(T.damp_residual_branches(sd, arch))
x, y = T._data(B, hw, counts, seed=21)
crit = IIFLoss(T.DS(counts), variant="raw")
net.train()
xd, yd = x.to("cuda:0"), y.to("cuda:0")


def regw(on):
    if on:
        os.environ.pop("IIF_CONV_NO_REGW", None)
    else:
        os.environ["IIF_CONV_NO_REGW"] = "1"
    _lib.lib().iif_conv_reload_env()


def grads():
    loss, _ = net.loss_and_backward(xd, yd, crit)
    torch.cuda.synchronize()
    return loss.item(), net._grad_arena.clone()


regw(True)
l_alg, alg = grads()
plan = net._saved
regw(False)
l_alg0, alg0 = grads()
keep, plan.alg3_units, keep2, plan.twopass_units = plan.alg3_units, set(), plan.twopass_units, set()
l_std0, std0 = grads()
regw(True)
l_std, std = grads()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
print("losses", l_alg, l_alg0, l_std0, l_std)
print("alg regw vs alg tile %.3e   std regw vs std tile %.3e   alg vs std (tile) %.3e   alg vs std (regw) %.3e" % (
    rel(alg, alg0), rel(std, std0), rel(alg0, std0), rel(alg, std)))

# ---- which unit's forward differs?
regw(False)
grads()
ref = [(u.x.clone(), u.stats.clone()) for u in plan.units]
regw(True)
grads()
for i, u in enumerate(plan.units):
    dx = (u.x.float() - ref[i][0].float()).abs().max().item()
    ds = (u.stats - ref[i][1]).abs().max().item()
    if dx > 0 or ds > 1e-6:
        print("unit %d  conv %d->%d k%d s%d  M=%d  |dx| %.3e  |dstats| %.3e  nan %s" % (
            i, u.conv.cin, u.conv.cout, u.conv.k, u.conv.stride, u.n * u.ho * u.wo, dx, ds, torch.isnan(u.stats).any().item()))

print("---- stats of the 128->512 / 512->128 units against float64 of their own stored x")
for tag, on in (("tile", False), ("regw", True)):
    regw(on)
    grads()
    for i, u in enumerate(plan.units):
        if u.n * u.ho * u.wo == 2048 and u.conv.k == 1 and u.conv.stride == 1:
            xx = u.x.double().view(-1, u.conv.cout)
            mean, var = xx.mean(0), xx.var(0, unbiased=False)
            invstd = 1.0 / torch.sqrt(var + 1e-5)
            C = u.conv.cout
            st = u.stats.view(-1)
            e0 = ((st[:C].double() - mean).abs().max() / mean.abs().max()).item()
            e1 = ((st[C:2 * C].double() - invstd).abs().max() / invstd.abs().max()).item()
            print("%s unit %d %d->%d  mean err %.2e  invstd err %.2e" % (tag, i, u.conv.cin, u.conv.cout, e0, e1))
