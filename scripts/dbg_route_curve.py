"""Six-step bf16 loss curves of the default routes and of the standard backward (the body of
tests/test_resnet_gpu.py::test_bf16_loss_curve_of_the_default_routes_against_the_standard_backward), printed per mode; run
under IIF_AMD_LIB=<another build> to bisect a drift."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIF_SIDE_STREAMS", "1")
os.environ["IIF_BN3_ALGEBRA_PURE_MIN_ELEMS"] = "5e7"
import torch
from tests.test_resnet_gpu import _build, _data, DS, damp_residual_branches, DEV
from iif_amd.custom import IIFLoss
arch, C, B, hw = "resnet50", 1000, 64, 224
counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
x, y = _data(B, hw, counts, seed=77)
xd, yd = x.to(DEV), y.to(DEV)
crit = IIFLoss(DS(counts), variant="raw")
modes = sys.argv[1:] or ["default", "standard"]
for mode in modes:
    for k in ("IIF_NO_BN3_ALGEBRA", "IIF_NO_BWD_FUSE"):
        os.environ.pop(k, None)
    if mode == "standard":
        os.environ["IIF_NO_BN3_ALGEBRA"] = "1"; os.environ["IIF_NO_BWD_FUSE"] = "1"
    if mode == "noalg":
        os.environ["IIF_NO_BN3_ALGEBRA"] = "1"
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    net.train()
    losses = []
    for it in range(6):
        loss, _ = net.loss_and_backward(xd, yd, crit)
        net.sgd_step(0.002, 0.9, 1e-4)
        losses.append(float(loss.item()))
    print(os.environ.get("IIF_AMD_LIB", "current"), mode, " ".join("%.5f" % l for l in losses), flush=True)
    del net
