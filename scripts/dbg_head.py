import sys; sys.path.insert(0, '.')
import torch
from iif_amd.custom import IIFLoss
class DS:
    def __init__(s, c): s.c = c
    def get_cls_num_list(s): return s.c
C, B = 100, 16
counts = [500 - 4 * i for i in range(C)]
g = torch.Generator().manual_seed(0)
pred = torch.randn(B, C, generator=g).cuda()
tgt = torch.randint(0, C, (B,), generator=g).cuda()
crit = IIFLoss(DS(counts), variant="raw", reduction="none") if False else IIFLoss(DS(counts), variant="raw", reduction="sum")
p = pred.clone().requires_grad_(True)
loss = crit(p, tgt)
tab = crit.iif["raw"].cuda().view(-1)
z = pred * tab
ref_rows = torch.logsumexp(z, 1) - z[torch.arange(B), tgt]
print("loss", loss.item(), "ref", ref_rows.sum().item())
# per-row via single-row calls
for i in range(B):
    li = crit(pred[i:i+1].clone(), tgt[i:i+1])
    print(i, int(tgt[i]), "%.5f %.5f" % (li.item(), ref_rows[i].item()), "lse-part diff %.5f" % (li.item() - ref_rows[i].item()), " z_t %.4f" % z[i, tgt[i]].item())
