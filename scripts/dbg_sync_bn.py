import os, sys, torch, torch.distributed as dist
sys.path.insert(0, "/root/repo")
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
from iif_amd.ddp import broadcast_parameters
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
torch.manual_seed(5)
dt = torch.float32
net = resnet_pytorch.resnet50(num_classes=10, use_norm="None", pretrained="None", device=dev, compute_dtype=dt)
net.train()
if os.environ.get('DAMP'):
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('bn3.weight'): p.mul_(0.1)
g = torch.Generator().manual_seed(91)
x = torch.randn(16, 3, 64, 64, generator=g); y = torch.randint(0, 10, (16,), generator=g)
class D:
    def get_cls_num_list(self): return [500, 300, 200, 120, 80, 50, 30, 20, 10, 5]
crit = IIFLoss(D(), device=dev)
# reference: whole batch, no sync
loss, lg = net.loss_and_backward(x.to(dev), y.to(dev), crit)
ref_stats = [u.stats.clone() for u in net._saved.units]
ref_logits = lg.clone()
ref_grad = net.grad_arena.clone()
def last_unit_check(tag):
    pl = net._saved
    last = pl.blocks[-1]["units"][-1]
    y = last.y.view(-1, last.conv.cout); gbuf = pl._gbuf(("g", pl.final.shape), pl.final.shape).view(-1, last.conv.cout)
    want = ((y > 0).float() * gbuf.float()).double().sum(0)
    got = last.bn._dbeta.double()
    print(tag, "rank", rank, "dbeta vs torch on own y,g: max abs %.3e of max %.3e" % ((want - got).abs().max().item(), want.abs().max().item()), flush=True)
    return (y > 0).clone(), gbuf.clone()
ref_mask, ref_g = last_unit_check("nonsync")
broadcast_parameters(net)
net.enable_sync_bn()
per = 16 // world
xs, ys = x[rank*per:(rank+1)*per].to(dev), y[rank*per:(rank+1)*per].to(dev)
loss, lg = net.loss_and_backward(xs, ys, crit)
plan = net._saved
sm, sg = last_unit_check("sync")
print("rank", rank, "mask flips vs whole-batch run:", (sm != ref_mask[rank*per*4:(rank+1)*per*4]).sum().item(), "of", sm.numel(),
      "g diff", (sg - ref_g[rank*per*4:(rank+1)*per*4] * world).abs().max().item(), flush=True)
gsum = net.grad_arena.clone(); dist.all_reduce(gsum); gsum /= world
if rank == 0:
    for i, (u, r) in enumerate(zip(plan.units, ref_stats)):
        e = (u.stats[:2] - r[:2]).abs().max().item() / r[:2].abs().max().item()
        print(i, u.conv.cout, u.conv.k, "stats rel err %.2e" % e)
        if e > 1e-3:
            break
    print("logits err", (lg - ref_logits[:per]).abs().max().item())
if rank == 0:
    net.grad_arena.copy_(ref_grad); refs = [g.clone() for g in net._grad_views]
    net.grad_arena.copy_(gsum)
    for (name, _), g, r in zip(net.named_parameters(), net._grad_views, refs):
        e = (g - r).abs().max().item() / (r.abs().max().item() + 1e-20)
        print("%-40s |g| %.3e rel err %.2e" % (name, r.abs().max().item(), e))
dist.barrier(); dist.destroy_process_group()
