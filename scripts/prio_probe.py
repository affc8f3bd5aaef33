"""Does running the compute stream at high priority (side streams normal) shorten the step?  (probe)"""
import sys, time
sys.path.insert(0, '.')
import torch
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B
dev = torch.device('cuda', 0)
C, bs = 1000, 256
counts = B.lt_counts(C, 1280)
net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", device=dev, compute_dtype=torch.bfloat16)
net.train()
crit = IIFLoss(B._Counts(counts), variant="raw", reduction="mean", device=dev)
x = torch.randn(bs, 3, 224, 224).to(dev); y = torch.randint(0, C, (bs,)).to(dev)
def step():
    loss, _ = net.loss_and_backward(x, y, crit)
    net.sgd_step(1e-4, 0.9, 1e-4)
def timeit(tag):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print("%s: %.3f ms/step" % (tag, (time.perf_counter() - t0) / 30 * 1e3))
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
timeit("default stream")
hi = torch.cuda.Stream(priority=-1)
hi.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(hi):
    timeit("high-priority compute stream")
timeit("default stream again")
lo = torch.cuda.Stream(priority=0)
with torch.cuda.stream(lo):
    timeit("plain side stream as compute stream")
