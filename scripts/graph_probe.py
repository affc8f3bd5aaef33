"""Does capturing the whole training step in a hipGraph pay on this stack?  (probe, not part of the product)"""
import sys, time
sys.path.insert(0, '.')
import torch
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B

#   python scripts/graph_probe.py [model] [batch] [image] [classes] [--reducer]
#   --reducer: the step with the bucketed gradient reduction driven through a ONE-rank RCCL process group (reducer.force): the
#   reducer's stream is forked off and joined back inside the capture, as it would be with 8 ranks
with_reducer = "--reducer" in sys.argv
argv = [a for a in sys.argv if a != "--reducer"]
dev = torch.device('cuda', 0)
model = argv[1] if len(argv) > 1 else "resnet50"
bs = int(argv[2]) if len(argv) > 2 else 256
img = int(argv[3]) if len(argv) > 3 else 224
C = int(argv[4]) if len(argv) > 4 else 1000
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

reducer = None
if with_reducer:
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    reducer = net.make_reducer()
    reducer.force = True
    print("reducer: %s" % (reducer.describe(),))


def step():
    loss, _ = net.loss_and_backward(x, y, crit, reducer=reducer)
    net.sgd_step(1e-4, 0.9, 1e-4, grad_scale=1.0 if reducer is None else reducer.grad_scale)
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
print("graph: %.3f ms/step  loss %.4f%s" % ((time.perf_counter() - t0) * 50, loss.item(),
                                              "  (collectives captured: %d per step)" % len(reducer.buckets) if reducer is not None else ""))
