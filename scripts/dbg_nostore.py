"""Round 6 diagnosis: the never-stored conv3 forward against (a) the standard route, (b) the algebra route with the stored output,
each against an fp32-compute run of the same network (the closest thing to the truth on the box): is the nostore route further
from the truth than the others are from each other?   python scripts/dbg_nostore.py [hw] [B]"""
import os
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["IIF_BN3_ALGEBRA_PURE_MIN_ELEMS"] = "0"
os.environ["IIF_SIDE_STREAMS"] = "1"
import torch
import test_resnet_gpu as T
from iif_amd.custom import IIFLoss
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
arch, C = "resnet50", 1000
counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
x, y = T._data(B, hw, counts, seed=23)
crit = IIFLoss(T.DS(counts), variant="raw")
xd, yd = x.to("cuda:0"), y.to("cuda:0")


def run(dt, env):
    for k in ("IIF_NO_NOSTORE", "IIF_NO_BN3_ALGEBRA"):
        os.environ.pop(k, None)
    os.environ.update(env)
    net, sd = T._build(arch, C, dt)
    net.load_state_dict(T.damp_residual_branches(sd, arch))
    net.train()
    loss, _ = net.loss_and_backward(xd, yd, crit)
    torch.cuda.synchronize()
    plan = net._saved
    print(env, dt, "loss %.6f alg %d nostore %d" % (loss.item(), len(plan.alg3_units), len(getattr(plan, "nostore_units", ()))))
    return net, net._grad_arena.clone()


n32, g32 = run(torch.float32, {})
nns, gns = run(torch.bfloat16, {})
nal, gal = run(torch.bfloat16, {"IIF_NO_NOSTORE": "1"})
nst, gst = run(torch.bfloat16, {"IIF_NO_BN3_ALGEBRA": "1"})
rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
print("whole gradient vs fp32:  nostore %.3e   algebra(stored) %.3e   standard %.3e" % (rel(gns, g32), rel(gal, g32), rel(gst, g32)))
print("nostore vs standard %.3e   algebra vs standard %.3e   nostore vs algebra %.3e" % (rel(gns, gst), rel(gal, gst), rel(gns, gal)))
names = {id(p): n for n, p in nns.named_parameters()}
worst = []
for (m_, attr, rows, pitch) in nns._param_specs():
    off = nns._offsets[(id(m_), attr)][0]
    sl = slice(off, off + rows * pitch)
    nm = names.get(id(getattr(m_, attr)), "?")
    worst.append((rel(gns[sl], g32[sl]), rel(gal[sl], g32[sl]), rel(gst[sl], g32[sl]), nm))
worst.sort(reverse=True)
for w in worst[:25]:
    print("%-40s nostore %.3e  algebra %.3e  standard %.3e" % (w[3], w[0], w[1], w[2]))
