"""bf16-mode gradients against the fp32-mode gradients of the same HIP engine on a conditioned initialisation (last BN gain of
every bottleneck x0.1): per-tensor relative L2 distance.  Measured (MI355X, round 2): median 24 %, max 32-36 % for ResNet50 /
ResNeXt50 on 16-32 images while the loss agrees to 2e-5 — the distance is bf16 storage of activations and gradients through
~50 BN backward passes (the CPU oracle with bf16 storage shows the same level), so a tight bf16 gradient tolerance does not
exist on this recipe; tests/test_resnet_gpu.py therefore compares bf16 gradients with the oracle's own bf16 noise floor."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from oracle import resnet_oracle as R
from tests.test_resnet_gpu import _data, DS, damp_residual_branches
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
DEV = "cuda:0"
for arch, C, B, hw in (("resnet50", 1000, 16, 64), ("resnext50_32x4d", 365, 16, 64), ("resnet50", 1000, 32, 112)):
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    sd = damp_residual_branches(R.init_imagenet(arch, C, seed=3), arch, 0.1)
    x, y = _data(B, hw, counts, seed=13)
    crit = IIFLoss(DS(counts))
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        net = getattr(resnet_pytorch, arch)(num_classes=C, use_norm="None", pretrained="None", compute_dtype=dt)
        net.load_state_dict(sd); net.train()
        loss, _ = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
        grads[dt] = {k: p.grad.detach().double().cpu().clone() if p.grad is not None else None for k, p in net.named_parameters()}
        # grads live in the arena views
        grads[dt] = {k: v.double().cpu().clone() for k, v in zip([k for k, _ in net.named_parameters()], net._grad_views)}
        print(arch, dt, "loss", loss.item())
    errs = []
    for k in grads[torch.float32]:
        a, b = grads[torch.bfloat16][k], grads[torch.float32][k]
        errs.append(((a - b).norm() / b.norm().clamp_min(1e-30)).item())
    errs = np.array(errs)
    keys = list(grads[torch.float32].keys())
    print(arch, B, hw, "bf16 vs fp32 per-tensor rel L2: median %.3e  90%% %.3e  max %.3e (%s)" % (np.median(errs), np.quantile(errs, 0.9), errs.max(), keys[int(errs.argmax())]))
