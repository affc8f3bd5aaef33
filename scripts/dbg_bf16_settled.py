"""Round 6 (review item 7): bf16-mode gradients against fp32-mode gradients of the same HIP engine at a SETTLED point - 224 x 224,
batch 64, damped init, after N SGD steps taken in fp32 mode (BN statistics and weights off the initialisation) - per-tensor
relative L2.   python scripts/dbg_bf16_settled.py [steps] [batch] [lr]"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from oracle import resnet_oracle as R
from tests.test_resnet_gpu import _data, DS, damp_residual_branches
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.002
arch, C, hw = "resnet50", 1000, 224
counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
sd = damp_residual_branches(R.init_imagenet(arch, C, seed=3), arch)
crit = IIFLoss(DS(counts))
net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.float32)
net.load_state_dict(sd); net.train()
for it in range(steps):
    x, y = _data(B, hw, counts, seed=100 + it)
    loss, _ = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
    net.sgd_step(lr, 0.9, 1e-4)
    if it % 5 == 0 or it == steps - 1:
        print("step", it, "loss", loss.item())
settled = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
x, y = _data(B, hw, counts, seed=999)
xd, yd = x.to(DEV), y.to(DEV)
grads = {}
for dt in (torch.float32, torch.bfloat16):
    n2 = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", compute_dtype=dt)
    n2.load_state_dict(settled); n2.train()
    loss, _ = n2.loss_and_backward(xd, yd, crit)
    grads[dt] = [v.double().cpu().clone() for v in n2._grad_views]
    names = [k for k, _ in n2.named_parameters()]
    print(dt, "loss", loss.item())
    del n2
errs = np.array([((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(grads[torch.bfloat16], grads[torch.float32])])
whole = (torch.cat([a.flatten() for a in grads[torch.bfloat16]]) - torch.cat([b.flatten() for b in grads[torch.float32]])).norm() / \
    torch.cat([b.flatten() for b in grads[torch.float32]]).norm()
print("after %d steps, B=%d: whole gradient %.3e; per tensor median %.3e  90%% %.3e  max %.3e (%s)" % (
    steps, B, whole.item(), np.median(errs), np.quantile(errs, 0.9), errs.max(), names[int(errs.argmax())]))
order = np.argsort(-errs)[:8]
for i in order:
    print("   %-40s %.3e" % (names[i], errs[i]))
gb, gf = torch.cat([a.flatten() for a in grads[torch.bfloat16]]), torch.cat([b.flatten() for b in grads[torch.float32]])
print("projection <g_bf16, g_fp32> / <g_fp32, g_fp32>: whole %.4f   cosine %.4f" % ((gb @ gf / (gf @ gf)).item(), (gb @ gf / (gb.norm() * gf.norm())).item()))
proj = np.array([((a.flatten() @ b.flatten()) / (b.flatten() @ b.flatten()).clamp_min(1e-300)).item() for a, b in zip(grads[torch.bfloat16], grads[torch.float32])])
kinds = {"conv": [i for i, n in enumerate(names) if "conv" in n or "downsample.0" in n or n.startswith("fc")],
         "bn.weight": [i for i, n in enumerate(names) if ("bn" in n or "downsample.1" in n) and n.endswith("weight")],
         "bn.bias": [i for i, n in enumerate(names) if ("bn" in n or "downsample.1" in n) and n.endswith("bias")]}
for k, idx in kinds.items():
    p = proj[idx]
    print("   %-10s projection per tensor: min %.3f  median %.3f  max %.3f   relL2 median %.3f" % (k, p.min(), np.median(p), p.max(), np.median(errs[idx])))
