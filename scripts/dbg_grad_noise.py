import sys; sys.path.insert(0,'/root/repo')
import torch
from oracle import iif_oracle as O, resnet_oracle as R
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
arch, C, B, hw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
cifar = arch in R.CIFAR_ARCHS
sd = (R.init_cifar if cifar else R.init_imagenet)(arch, C, seed=3)
net = (getattr(resnet_cifar, arch)(num_classes=C, compute_dtype=torch.float32) if cifar else getattr(resnet_pytorch, arch)(num_classes=C, compute_dtype=torch.float32))
import os
damp = float(os.environ.get("DAMP", "1"))
for k in sd:
    if (k.endswith("bn3.weight") or (k.endswith("bn2.weight") and "resnet50" not in arch)):
        sd[k] = sd[k] * damp
net.load_state_dict(sd)
g = torch.Generator().manual_seed(5)
x = torch.randn(B,3,hw,hw,generator=g)
prior = torch.tensor(counts, dtype=torch.float64)
y = torch.multinomial(prior/prior.sum(), B, replacement=True, generator=g)
table = O.iif_tables(counts)["raw"]
l32, lg32, g32 = R.loss_and_grads({k:v.clone() for k,v in sd.items()}, x, y, table, arch)
sd64 = {k:(v.double() if v.is_floating_point() else v.clone()) for k,v in sd.items()}
l64, lg64, g64 = R.loss_and_grads(sd64, x.double(), y, table.double(), arch)
class DS:
    def get_cls_num_list(self): return counts
net.train()
crit = IIFLoss(DS())
logits = net(x.cuda()); loss = crit(logits, y.cuda()); loss.backward()
def e(a,b): return ((a.double().cpu()-b.double()).norm()/b.double().norm().clamp_min(1e-30)).item()
print("logits gpu-vs-64 %.2e cpu32-vs-64 %.2e" % (e(logits, lg64), e(lg32, lg64)))
print("loss gpu %.8f cpu32 %.8f f64 %.8f" % (loss.item(), l32.item(), l64.item()))
for k,p in net.named_parameters():
    if 'conv' in k or 'linear' in k or 'fc' in k or 'downsample.0' in k:
        print("%-32s gpu-vs-64 %.2e  cpu32-vs-64 %.2e  gpu-vs-cpu32 %.2e" % (k, e(p.grad, g64[k]), e(g32[k], g64[k]), e(p.grad, g32[k])))
