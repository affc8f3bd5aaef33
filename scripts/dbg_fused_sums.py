import sys, torch
sys.path.insert(0, '.')
from tests.test_resnet_gpu import _build, _data, damp_residual_branches, DS, DEV
from iif_amd.custom import IIFLoss
arch, C, B, hw = "resnet50", 1000, 32, 64
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
for seed in (21, 22, 23):
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=seed)
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    xd, yd = x.to(DEV), y.to(DEV)
    net.loss_and_backward(xd, yd, crit)
    plan = net._saved
    fused = net._grad_arena.clone()
    plan.fuse_bwd = False
    net.loss_and_backward(xd, yd, crit)
    plain = net._grad_arena.clone()
    errs = []
    names = {id(m): n for n, m in net.named_modules()}
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        a_, b_ = fused[off:off + rows * pitch], plain[off:off + rows * pitch]
        errs.append(((a_ - b_).norm().item() / max(b_.norm().item(), 1e-12), names.get(id(m_), "?") + "." + attr))
    errs.sort(reverse=True)
    print(seed, "whole", (fused - plain).norm().item() / plain.norm().item(), errs[:4])
