"""Where does the fp32 full-size (replicated batch) backward drift from the 8-image backward?"""
import sys
import torch
sys.path.insert(0, "tests")
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
pools = {}
for name, reps in (("small", 1), ("full", rep)):
    sd = R.init_imagenet("resnet50", C, seed=3)
    for k in sd:
        if k.endswith("bn3.weight") and k.startswith("layer"):
            sd[k] = sd[k] * 0.25
    net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.float32)
    net.load_state_dict(sd)
    net.train()
    logits = net(x.repeat(reps, 1, 1, 1).to(dev))
    loss = crit(logits, y.repeat(reps).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    plan = net._saved
    pools[name] = {k: v.detach().double().cpu() * reps for k, v in plan._grad_pool.items()}
    pools[name]["dlogits"] = plan.dlogits.detach().double().cpu() * reps
    del net
for k, vs in pools["small"].items():
    kf = next((q for q in pools["full"] if q[0] == k[0] and (len(q) < 3 or q[2:] == k[2:]) and (len(q) < 2 or len(q[1]) == len(k[1]) and q[1][1:] == k[1][1:])), None) if isinstance(k, tuple) else k
    if kf is None:
        print("no match", k); continue
    vf = pools["full"][kf]
    r = vf.view(rep, *vs.shape)
    same = all(torch.equal(r[i], r[0]) for i in range(1, rep))
    err = ((r[0] - vs).norm() / max(vs.norm().item(), 1e-30)).item()
    print("%-60s replicas identical: %-5s  full[0] vs small: %.2e" % (str(k)[:60], same, err))
