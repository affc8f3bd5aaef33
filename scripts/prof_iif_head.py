"""The IIF head kernels alone at the mmdet head shapes ([1024, 1204], the reference's bs-4 [2048, 1204]), the
classification shape [256, 1000] and a bandwidth-visible size [65536, 1000] / bf16: run under
`rocprofv3 --kernel-trace --stats` (and the --pmc passes) to get the rocprof-reported duration and HBM bytes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import custom
from iif_amd.mmdet_iif_loss import IIFLoss as DetIIF
dev = "cuda:0"
csvp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/lvis_files/idf_1204.csv")
det = DetIIF(num_classes=1203, path=csvp, variant="raw")
for (B, C, dt) in ((1024, 1204, torch.float32), (2048, 1204, torch.float32), (256, 1000, torch.float32), (65536, 1000, torch.float32), (65536, 1000, torch.bfloat16)):
    x = torch.randn(B, C, device=dev).to(dt)
    y = torch.randint(0, C, (B,), device=dev)
    tab = det.iif_weights if C == 1204 else torch.rand(1, C, device=dev) * 5 + 0.5
    w = torch.ones(B, device=dev)
    for _ in range(20):
        custom._launch_ce(x, tab, y, None, 1.0, w if C == 1204 else None, None, -100, 1.0 / B, True)      # loss + gradient, one launch
        if C == 1204:
            det.get_activation(x)                                                                        # iif_softmax
torch.cuda.synchronize()
print("done")
