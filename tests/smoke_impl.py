"""Body of __graft_entry__.smoke(): one small invocation of the hot path on
cuda:0, checked against the CPU oracle."""
import torch

from oracle import iif_oracle as O


def run():
    from iif_amd.custom import IIFLoss
    dev = torch.device("cuda", 0)

    class DS:
        def get_cls_num_list(self):
            return O.img_num_per_cls(100, 50000, "exp", 0.01)
    counts = DS().get_cls_num_list()
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(128, 100, generator=g)
    tgt = torch.randint(0, 100, (128,), generator=g)
    crit = IIFLoss(DS(), variant="raw")
    p = pred.to(dev).requires_grad_(True)
    loss = crit(p, tgt.to(dev))
    loss.backward()
    ref_l, ref_d, _ = O.iif_ce_closed_form(pred, tgt, O.iif_tables(counts)["raw"])
    el = abs(loss.item() - ref_l.item()) / abs(ref_l.item())
    eg = (p.grad.cpu().double() - ref_d).abs().max().item() / ref_d.abs().max().item()
    assert el < 1e-4 and eg < 1e-4, (el, eg)
    print("smoke: fused IIF CE loss %.6f (oracle %.6f) rel err loss %.2e grad %.2e" % (loss.item(), ref_l.item(), el, eg))
