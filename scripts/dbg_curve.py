import sys; sys.path.insert(0,'/root/repo')
import torch
from oracle import iif_oracle as O, resnet_oracle as R
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
arch, C, B, hw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = sys.argv[5] if len(sys.argv) > 5 else "fused"
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
cifar = arch in R.CIFAR_ARCHS
sd = (R.init_cifar if cifar else R.init_imagenet)(arch, C, seed=3)
net = (getattr(resnet_cifar, arch)(num_classes=C, compute_dtype=torch.float32) if cifar else getattr(resnet_pytorch, arch)(num_classes=C, compute_dtype=torch.float32))
net.load_state_dict(sd)
g = torch.Generator().manual_seed(9)
x = torch.randn(B,3,hw,hw,generator=g)
prior = torch.tensor(counts, dtype=torch.float64)
y = torch.multinomial(prior/prior.sum(), B, replacement=True, generator=g)
table = O.iif_tables(counts)["raw"]
class DS:
    def get_cls_num_list(self): return counts
crit = IIFLoss(DS())
ref = {k:v.clone() for k,v in sd.items()}
bufs = {}
net.train()
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
for it in range(3):
    lr = 0.1*O.warmup_factor(it,1000)
    rl,_ = R.train_step(ref, bufs, x, y, table, arch, lr)
    if mode == "fused":
        l,_ = net.loss_and_backward(x.cuda(), y.cuda(), crit)
        gsnap = net.grad_arena.clone()
        net.sgd_step(lr, 0.9, 1e-4)
    else:
        for grp in opt.param_groups: grp["lr"] = lr
        l = crit(net(x.cuda()), y.cuda())
        opt.zero_grad(); l.backward(); opt.step()
    print("it %d ref %.6f mine %.6f" % (it, rl.item(), l.item()))
    worst = []
    for k,v in net.state_dict().items():
        if v.is_floating_point():
            e = ((v.cpu().double()-ref[k].double()).norm()/ref[k].double().norm().clamp_min(1e-30)).item()
            worst.append((e,k))
    worst.sort(reverse=True)
    print("   worst state diffs:", worst[:5])
