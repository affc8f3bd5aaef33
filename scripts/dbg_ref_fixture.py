"""HIP fp32 steps on the G7/G10 reference fixtures: per-step loss deviation and per-tensor gradient-norm deviation
next to the reference's own fp32-vs-fp64 distance (diagnostic for tests/test_resnet_gpu.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import iif_oracle as O, resnet_oracle as R
from tests.test_resnet_gpu import REF_NET_CASES, REF_COUNTS, DS
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
DEV = "cuda:0"
only = sys.argv[1:]
for fixture, prefix, arch, C, B, hw, damp in REF_NET_CASES:
    if only and prefix not in only:
        continue
    g = np.load(os.path.join("tests/golden", fixture + ".npz"))
    cifar = arch in R.CIFAR_ARCHS
    sd = R.init_cifar(arch, C, seed=7) if cifar else R.init_imagenet(arch, C, seed=7)
    if damp is not None:
        sd = {k: (v * damp if (k.startswith('layer') and k.endswith('bn3.weight')) else v) for k, v in sd.items()}
    x = torch.randn(B, 3, hw, hw, generator=torch.Generator().manual_seed(99))
    y = torch.from_numpy(g[prefix + "_y"])
    net = (getattr(resnet_cifar, arch)(num_classes=C, use_norm="None", compute_dtype=torch.float32) if cifar else
           getattr(resnet_pytorch, arch)(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.float32))
    net.load_state_dict(sd); net.train()
    crit = IIFLoss(DS(REF_COUNTS[C]()), variant="raw")
    l32, l64 = g[prefix + "_losses"], g[prefix + "_losses_f64"]
    for it in range(len(l32)):
        net.zero_grad()
        logits = net(x.to(DEV)); loss = crit(logits, y.to(DEV)); loss.backward()
        if it == 0:
            lg = torch.from_numpy(g[prefix + "_logits0"]); lg64 = torch.from_numpy(g[prefix + "_logits0_f64"])
            print(prefix, "logits: hip-ref32 %.2e  ref32-ref64 %.2e" % ((logits.cpu() - lg).abs().max() / lg.abs().max(), (lg - lg64).abs().max() / lg64.abs().max()))
            keys = g[prefix + "_gradnorm_keys"].tolist(); gn32, gn64 = g[prefix + "_gradnorm0"], g[prefix + "_gradnorm0_f64"]
            grads = dict(net.named_parameters())
            dev = np.array([abs(grads[k].grad.double().norm().item() - a) / max(a, 1e-6 * gn32.max()) for k, a in zip(keys, gn32)])
            noise = np.abs(gn32 - gn64) / np.maximum(gn64, 1e-6 * gn64.max())
            print("   gradnorm: hip-ref32 rms %.2e max %.2e | ref noise rms %.2e max %.2e" % (np.sqrt((dev ** 2).mean()), dev.max(), np.sqrt((noise ** 2).mean()), noise.max()))
            for i in np.argsort(-dev)[:6]:
                print("      %-32s hip dev %.2e  ref noise %.2e  norm %.3e" % (keys[i], dev[i], noise[i], gn32[i]))
        print("   step %d loss hip %.6f ref32 %.6f ref64 %.6f | hip-ref32 %.2e  ref32-ref64 %.2e" % (it, loss.item(), l32[it], l64[it], abs(loss.item() - l32[it]) / l32[it], abs(l32[it] - l64[it]) / l64[it]))
        net.sgd_step(float(g[prefix + "_lr0"]) * O.warmup_factor(it, 1000), 0.9, 1e-4)
    fl = [(k, v) for k, v in net.state_dict().items() if v.is_floating_point()]
    chk = np.array([float(v.double().sum()) for _, v in fl]); l1 = np.array([float(v.double().abs().sum()) for _, v in fl])
    c32, c64 = g[prefix + "_final_checksum"], g[prefix + "_final_checksum_f64"]
    d = np.abs(chk - c32) / np.maximum(l1, 1e-3); n = np.abs(c32 - c64) / np.maximum(l1, 1e-3)
    for i in np.argsort(-d)[:4]:
        print("   final %-34s |hip-ref32|/L1 %.2e  ref noise/L1 %.2e  L1 %.3e" % (fl[i][0], d[i], n[i], l1[i]))
