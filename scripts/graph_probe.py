"""Does capturing the whole training step in a hipGraph pay on this stack?  (probe, not part of the product)"""
import sys, time
sys.path.insert(0, '.')
import torch
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B

#   python scripts/graph_probe.py [model] [batch] [image] [classes]
dev = torch.device('cuda', 0)
model = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
img = int(sys.argv[3]) if len(sys.argv) > 3 else 224
C = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
counts = B.lt_counts(C, 1280)
torch.manual_seed(0)
if hasattr(resnet_pytorch, model):
    net = getattr(resnet_pytorch, model)(num_classes=C, use_norm="None", pretrained="None", device=dev, compute_dtype=torch.bfloat16)
else:
    net = getattr(resnet_cifar, model)(num_classes=C, use_norm="None", device=dev, compute_dtype=torch.bfloat16)
net.train()
crit = IIFLoss(B._Counts(counts), variant="raw", reduction="mean", device=dev)
g = torch.Generator().manual_seed(1)
x = torch.randn(bs, 3, img, img, generator=g).to(dev)
y = torch.randint(0, C, (bs,), generator=g).to(dev)

def step():
    loss, _ = net.loss_and_backward(x, y, crit)
    net.sgd_step(1e-4, 0.9, 1e-4)
    return loss

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("eager: %.3f ms/step" % ((time.perf_counter() - t0) * 50))

graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(graph):
    loss = step()
torch.cuda.synchronize()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
print("graph: %.3f ms/step  loss %.4f" % ((time.perf_counter() - t0) * 50, loss.item()))
