"""bf16-mode gradients of the 8-image step and of its 32x replicated 256-image step against the fp32 CPU oracle."""
import sys
import torch
sys.path.insert(0, "tests")
from oracle import iif_oracle as O
from oracle import resnet_oracle as R
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss

dev = "cuda:0"
C, B, rep, hw = 1000, 8, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 224
counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]


class DS:
    def get_cls_num_list(self):
        return counts


g = torch.Generator().manual_seed(21)
x = torch.randn(B, 3, hw, hw, generator=g)
prior = torch.tensor(counts, dtype=torch.float64)
y = torch.multinomial(prior / prior.sum(), B, replacement=True, generator=g)
crit = IIFLoss(DS(), variant="raw")
table = O.iif_tables(counts)["raw"]


def fresh():
    sd = R.init_imagenet("resnet50", C, seed=3)
    for k in sd:
        if k.endswith("bn3.weight") and k.startswith("layer"):
            sd[k] = sd[k] * 0.25
    return sd


_, _, ref = R.loss_and_grads(fresh(), x, y, table, "resnet50")
_, _, refq = R.loss_and_grads(fresh(), x, y, table, "resnet50", q=R.bf16_storage)
l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()   # noqa: E731
e = sorted(l2(refq[k], ref[k]) for k in ref)
print("oracle bf16-storage vs oracle fp32: median %.3e worst %.3e" % (e[len(e) // 2], e[-1]))
got = {}
for name, reps in (("small", 1), ("full", rep)):
    net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.bfloat16)
    net.load_state_dict(fresh())
    net.train()
    loss = crit(net(x.repeat(reps, 1, 1, 1).to(dev)), y.repeat(reps).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    got[name] = {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()}
    for rn, rr in (("fp32 oracle", ref), ("bf16-storage oracle", refq)):
        e = sorted((l2(got[name][k], rr[k]), k) for k in rr)
        print("%-5s vs %-20s median %.3e worst %s" % (name, rn, e[len(e) // 2][0], e[-2:]))
    del net
e = sorted((l2(got["full"][k], got["small"][k]), k) for k in ref)
print("full vs small: median %.3e worst %s" % (e[len(e) // 2][0], e[-2:]))
for k in list(ref)[::-1][:24]:
    print("   %-28s small %.2e full %.2e   full-vs-small %.2e" % (k, l2(got["small"][k], ref[k]), l2(got["full"][k], ref[k]), l2(got["full"][k], got["small"][k])))
