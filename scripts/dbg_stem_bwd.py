"""Stem backward (maxpool -> relu -> bn1 -> conv1) of the engine against an fp64 torch evaluation, given the
engine's own gradient w.r.t. the pooled output."""
import sys
import torch
import torch.nn.functional as F
from oracle import resnet_oracle as R
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss

dev = "cuda:0"
C, B, hw = 1000, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 224
counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]


class DS:
    def get_cls_num_list(self):
        return counts


g = torch.Generator().manual_seed(21)
x = torch.randn(B, 3, hw, hw, generator=g)
prior = torch.tensor(counts, dtype=torch.float64)
y = torch.multinomial(prior / prior.sum(), B, replacement=True, generator=g)
crit = IIFLoss(DS(), variant="raw")
sd = R.init_imagenet("resnet50", C, seed=3)
net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.float32)
net.load_state_dict(sd)
net.train()
loss = crit(net(x.to(dev)), y.to(dev))
loss.backward()
torch.cuda.synchronize()
plan = net._saved
for k, v in plan._grad_pool.items():
    print("pool", k, tuple(v.shape))
gp = plan._grad_pool[("gin", (B, hw // 4, hw // 4, 64), 0)].detach().double().cpu().permute(0, 3, 1, 2)
w = sd["conv1.weight"].double().requires_grad_(True)
o = F.conv2d(x.double(), w, stride=2, padding=3)
o.retain_grad()
bn = F.batch_norm(o, None, None, sd["bn1.weight"].double(), sd["bn1.bias"].double(), True, 0.1, 1e-5)
p = F.max_pool2d(F.relu(bn), 3, 2, 1)
gw, go = torch.autograd.grad(p, [w, o], gp)
mine = dict(net.named_parameters())["conv1.weight"].grad.double().cpu()
print("conv1.weight grad vs fp64 stem backward: %.3e" % ((mine - gw).norm() / gw.norm()).item())
pe = (mine - gw).pow(2).sum((0, 1)).sqrt() / gw.pow(2).sum((0, 1)).sqrt()
print("per tap:\n", pe)
for k, v in plan._grad_pool.items():
    if v.numel() == go.numel():
        d = v.detach().double().cpu().view(B, hw // 2, hw // 2, 64).permute(0, 3, 1, 2)
        print("candidate dx buffer", k, "vs fp64: %.3e" % ((d - go).norm() / go.norm()).item())
        err = (d - go).abs()
        print("  max abs err %.3e at %s ; ref max %.3e" % (err.max().item(), str(torch.nonzero(err == err.max())[0].tolist()), go.abs().max().item()))
        e_rows = (d - go).pow(2).sum((0, 1, 3)).sqrt() / go.pow(2).sum((0, 1, 3)).sqrt()
        print("  per output row:", e_rows[:6], e_rows[-6:])

# are the sparse differences maxpool near-ties?  (two candidates of one window within fp32 rounding)
d = plan._grad_pool[("dy0",)].detach().double().cpu().view(B, hw // 2, hw // 2, 64).permute(0, 3, 1, 2)
bad = torch.nonzero((d - go).abs() > 1e-4 * go.abs().max())
print("elements off by more than 1e-4 of the max:", bad.shape[0], "of", d.numel())
act = F.relu(bn).detach()
for (n_, c_, h_, w_) in bad.tolist()[:8]:
    best = None
    for ph in range(max(0, (h_ - 1) // 2), min(hw // 4, (h_ + 1) // 2 + 1)):
        for pw in range(max(0, (w_ - 1) // 2), min(hw // 4, (w_ + 1) // 2 + 1)):
            win = act[n_, c_, max(0, 2 * ph - 1):2 * ph + 2, max(0, 2 * pw - 1):2 * pw + 2].flatten()
            top = torch.topk(win, 2).values
            gap = ((top[0] - top[1]) / top[0].abs().clamp_min(1e-30)).item()
            best = gap if best is None else min(best, gap)
    print("  (n,c,h,w)=%s  mine %.4e ref %.4e  smallest relative top-2 gap among its windows: %.2e" % ((n_, c_, h_, w_), d[n_, c_, h_, w_].item(), go[n_, c_, h_, w_].item(), best))
