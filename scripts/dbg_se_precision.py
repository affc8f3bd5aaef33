"""Is the 1e-4-level gradient deviation of the SE networks from the reference fixture a ReLU-tie effect or arithmetic?
HIP fp32 step vs the CPU oracle with the GPU's ReLU decisions replayed: per-tensor relative L2 error, worst first."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import iif_oracle as O, resnet_oracle as R
from tests.test_resnet_gpu import REF_COUNTS, DS, gpu_relu_masks
from iif_amd import resnet_cifar
from iif_amd.custom import IIFLoss
DEV = "cuda:0"
for arch in ("resnet32", "se_resnet32"):
    C, B, hw = 100, 8, 32
    sd = R.init_cifar(arch, C, seed=7)
    x = torch.randn(B, 3, hw, hw, generator=torch.Generator().manual_seed(99))
    g = np.load("tests/golden/%s.npz" % ("g7_nets" if arch == "resnet32" else "g10_se"))
    y = torch.from_numpy(g[arch + "_y"])
    counts = REF_COUNTS[C]()
    net = getattr(resnet_cifar, arch)(num_classes=C, use_norm="None", compute_dtype=torch.float32)
    net.load_state_dict(sd); net.train()
    crit = IIFLoss(DS(counts), variant="raw")
    logits = net(x.to(DEV)); loss = crit(logits, y.to(DEV)); loss.backward()
    masks = R.ReluMasks(gpu_relu_masks(net))
    table = O.iif_tables(counts)["raw"]
    rl, rlog, rg = R.loss_and_grads({k: v.clone() for k, v in sd.items()}, x, y, table, arch, relu_masks=masks)
    rl0, _, rg0 = R.loss_and_grads({k: v.clone() for k, v in sd.items()}, x, y, table, arch)
    print(arch, "mask disagreements with the oracle's own decisions:", masks.disagree, "of", masks.total, "worst |pre-activation|", masks.worst)
    errs = []
    for k, p in net.named_parameters():
        a, b, b0 = p.grad.double().cpu(), rg[k].double(), rg0[k].double()
        errs.append(((a - b).norm().item() / max(b.norm().item(), 1e-12), (a - b0).norm().item() / max(b0.norm().item(), 1e-12), k))
    errs.sort(reverse=True)
    print("   per-tensor rel L2 error vs oracle WITH the GPU's masks (then vs the oracle's own masks):")
    for e, e0, k in errs[:5]:
        print("      %-34s %.2e   %.2e" % (k, e, e0))
