"""cProfile of the host side of the training step (which Python frames the per-launch time goes to):
   python scripts/host_profile.py [model] [batch] [image] [classes]"""
import cProfile
import pstats
import sys
sys.path.insert(0, ".")
import torch
from iif_amd import resnet_cifar, resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B

dev = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "resnet32"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
img = int(sys.argv[3]) if len(sys.argv) > 3 else 32
C = int(sys.argv[4]) if len(sys.argv) > 4 else 100
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
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
