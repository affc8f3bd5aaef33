"""Per-tensor distance between the step with the algebra routes and the standard step (names printed): debugging aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIF_SIDE_STREAMS", "1")
os.environ["IIF_BN3_ALGEBRA_PURE_MIN_ELEMS"] = sys.argv[1] if len(sys.argv) > 1 else "1e30"
import torch
from tests.test_resnet_gpu import _build, _data, DS, damp_residual_branches, DEV
from iif_amd.custom import IIFLoss
arch, C, B, hw = "resnet50", 1000, 32, 64
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
net, sd = _build(arch, C, torch.bfloat16)
net.load_state_dict(damp_residual_branches(sd, arch))
x, y = _data(B, hw, counts, seed=21)
crit = IIFLoss(DS(counts), variant="raw")
net.train()
xd, yd = x.to(DEV), y.to(DEV)
net.loss_and_backward(xd, yd, crit)
plan = net._saved
alg = net._grad_arena.clone()
keep_ds, plan.ds_alg = plan.ds_alg, {}
net.loss_and_backward(xd, yd, crit)
nods = net._grad_arena.clone()
keep, plan.alg3_units = plan.alg3_units, set()
net.loss_and_backward(xd, yd, crit)
std = net._grad_arena.clone()
names = {id(m): n for n, m in net.named_modules()}
for (m_, attr, rows, pitch) in net._param_specs():
    off = net._offsets[(id(m_), attr)][0]
    a_, n_, b_ = alg[off:off + rows * pitch], nods[off:off + rows * pitch], std[off:off + rows * pitch]
    e = (a_ - b_).norm().item() / max(b_.norm().item(), 1e-12)
    e2 = (n_ - b_).norm().item() / max(b_.norm().item(), 1e-12)
    if e > 3e-2 or "layer1.0.downsample" in names[id(m_)]:
        print("%-40s %-8s alg-vs-std %.3e   (bn3 algebra only)-vs-std %.3e   |std| %.3e" % (names[id(m_)], attr, e, e2, b_.norm().item()))
