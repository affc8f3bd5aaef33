"""FASA pieces on the IIF path against the CPU restatement in oracle.mmdet_iif."""
import pytest
import torch

from oracle import mmdet_iif as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _csv(tmp_path, c, seed=0):
    g = torch.Generator().manual_seed(seed)
    table = torch.rand(c + 1, generator=g) * 4 + 0.5
    table[-1] = 1.0
    path = tmp_path / "idf.csv"
    with open(path, "w") as f:
        f.write("idx,raw\n0,1.0\n")
        for i in range(c):
            f.write("%d,%.9g\n" % (i + 1, table[i].item()))
    return str(path), table


def test_fasa_iif_loss_cums(tmp_path):
    from iif_amd.mmdet_fasa import FasaIIFLoss
    c, n = 40, 300
    path, table = _csv(tmp_path, c)
    crit = FasaIIFLoss(num_classes=c, path=path, variant="raw", loss_weight=1.5, use_cums=True)
    assert crit.reduction == "none" and crit.reduction_old == "mean"
    g = torch.Generator().manual_seed(3)
    ref_cl, ref_cn = torch.zeros(c + 1), torch.zeros(c + 1)
    for step in range(3):
        x = torch.randn(n, c + 1, generator=g) * 2
        y = torch.randint(0, c + 1, (n,), generator=g)
        w = (torch.rand(n, generator=g) > 0.2).float()
        xd = x.to(DEV).requires_grad_(True)
        loss = crit(xd, y.to(DEV), w.to(DEV), avg_factor=float(max(w.sum().item(), 1)))
        loss.backward()
        xr = x.clone().requires_grad_(True)
        rows = 1.5 * M.iif_cross_entropy(xr, y, table.unsqueeze(0), weight=w, reduction="none", avg_factor=None)
        ref = M.fasa_accumulate(rows.detach(), y, ref_cl, ref_cn)
        rows.mean().backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
        assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 1e-5 * xr.grad.abs().max().item()
    assert torch.equal(crit.cum_labels.cpu(), ref_cn)                      # counts are integers: exact
    assert (crit.cum_losses.cpu() - ref_cl).abs().max().item() <= 1e-4 * ref_cl.abs().max().item()
    crit.close_cums()
    assert crit.reduction == "mean" and crit.cum_labels.abs().sum().item() == 0
    out = crit(torch.randn(5, c + 1, device=DEV), torch.randint(0, c + 1, (5,), device=DEV))
    assert out.dim() == 0


def test_feature_bank_update_and_generate():
    from iif_amd.mmdet_fasa import FasaFeatureBank
    c, d = 50, 192
    counts = [max(int(2000 * (0.9 ** i)), 1) for i in range(c)]
    bank = FasaFeatureBank(c, d, counts, dict(decay_ratio=0.1, instance_prob_scale=200.0), device=DEV)
    g = torch.Generator().manual_seed(8)
    fm, fv, fu = torch.zeros(c, d), torch.zeros(c, d), torch.zeros(c)
    for step in range(3):
        n = 97
        emb = torch.randn(n, d, generator=g) * 1.5 + 0.3
        lab = torch.randint(0, 30, (n,), generator=g)             # classes 30.. never seen
        bank.fa_update(emb.to(DEV), lab.to(DEV))
        M.fasa_update(emb, lab, fm, fv, fu, 0.1)
    assert torch.equal(bank.feature_used.cpu(), fu)
    assert (bank.feature_mean.cpu() - fm).abs().max().item() <= 1e-5
    assert (bank.feature_std.cpu() - fv).abs().max().item() <= 1e-4 * fv.abs().max().item()
    rand = torch.rand(c, generator=g)
    normal = torch.randn(c, d, generator=g)
    e, l = bank.fa_generate(rand.to(DEV), normal.to(DEV))
    re, rl = M.fasa_generate(rand, bank.prob_list.cpu(), fu, fm, fv, normal)
    assert len(rl) > 0 and l.cpu().tolist() == rl.tolist()
    assert (e.cpu() - re).abs().max().item() <= 1e-4
    # nothing selected -> the reference's empty lists
    e2, l2 = bank.fa_generate(torch.ones(c, device=DEV) * 2, normal.to(DEV))
    assert e2 == [] and l2 == []


def test_dynamic_sampling_runs():
    from iif_amd.mmdet_fasa import FasaFeatureBank
    c, d = 12, 16

    class L:
        cum_labels = torch.ones(c + 1, device=DEV) * 10
        cum_losses = torch.arange(c + 1, device=DEV).float()
    bank = FasaFeatureBank(c, d, [100] * c, device=DEV)
    bank.feature_mean.data.copy_(torch.randn(c, d, generator=torch.Generator().manual_seed(1)))
    p0 = bank.prob_list.data.clone()
    bank.dynamic_sampling(L)                                          # first call: t0 := t1, no change
    assert torch.equal(bank.prob_list.data, p0) and len(bank.group_cluster_list) >= 1
    L.cum_losses = L.cum_losses * 2                                    # losses rose -> probabilities go down
    bank.dynamic_sampling(L)
    assert (bank.prob_list.data <= p0 + 1e-9).all() and (bank.prob_list.data < p0).any()
