"""Algebraic BN3 backward inside the net vs the standard route: per-tensor gradient distance, by stage threshold.
    IIF_BN3_ALGEBRA_MIN_ELEMS=0 IIF_BN3_ALGEBRA_MAXC=<64|128|256> python scripts/dbg_bn3_alg.py"""
import os, sys, torch
sys.path.insert(0, '.')
from tests.test_resnet_gpu import _build, _data, damp_residual_branches, DS, DEV
from iif_amd.custom import IIFLoss
arch, C, B, hw = "resnet50", 1000, 32, int(os.environ.get("HW", "64"))
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
net, sd = _build(arch, C, torch.bfloat16)
net.load_state_dict(damp_residual_branches(sd, arch))
x, y = _data(B, hw, counts, seed=21)
crit = IIFLoss(DS(counts), variant="raw")
net.train()
xd, yd = x.to(DEV), y.to(DEV)
net.loss_and_backward(xd, yd, crit)
plan = net._saved
print("algebra units:", len(plan.alg3_units))
fused = net._grad_arena.clone()
keep = plan.alg3_units
plan.alg3_units = set()
net.loss_and_backward(xd, yd, crit)
plain = net._grad_arena.clone()
plan.alg3_units = keep
net.loss_and_backward(xd, yd, crit)
again = net._grad_arena.clone()
print("repeat identical:", torch.equal(fused, again))
errs = []
names = {id(m): n for n, m in net.named_modules()}
for (m_, attr, rows, pitch) in net._param_specs():
    off = net._offsets[(id(m_), attr)][0]
    a_, b_ = fused[off:off + rows * pitch], plain[off:off + rows * pitch]
    errs.append(((a_ - b_).norm().item() / max(b_.norm().item(), 1e-12), names.get(id(m_), "?") + "." + attr))
print("whole", (fused - plain).norm().item() / plain.norm().item())
for e, n in errs:
    if "layer" in n and (".0.conv1" in n or "bn3" in n or "conv3" in n or e > 0.05):
        print("  %-40s %.4f" % (n, e))
