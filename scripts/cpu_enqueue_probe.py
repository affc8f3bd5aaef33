"""How long does the host need to enqueue one training step (no waiting on the GPU)?"""
import sys, time, os
sys.path.insert(0, '.')
import torch
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B
dev = torch.device('cuda', 0)
model = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
C, bs = (int(sys.argv[3]) if len(sys.argv) > 3 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 256)
counts = B.lt_counts(C, 1280)
net = getattr(resnet_pytorch, model)(num_classes=C, use_norm="None", pretrained="None", device=dev, compute_dtype=torch.bfloat16)
net.train()
crit = IIFLoss(B._Counts(counts), variant="raw", reduction="mean", device=dev)
x = torch.randn(bs, 3, 224, 224).to(dev); y = torch.randint(0, C, (bs,)).to(dev)
def step():
    loss, _ = net.loss_and_backward(x, y, crit)
    net.sgd_step(1e-4, 0.9, 1e-4)
for _ in range(5): step()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.2f ms, until done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
