"""Mask-side class-channel selection against the CPU restatement in oracle.mmdet_iif:
gather bit-exact, BCE loss / gradient within 1e-5 relative (fp32)."""
import pytest
import torch

from oracle import mmdet_iif as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,c,hw", [(7, 11, 2), (64, 1203, 28), (1, 3, 28), (33, 80, 14)])
def test_gather_is_bit_exact(n, c, hw):
    from iif_amd.mmdet_mask_loss import gather_class_masks
    g = torch.Generator().manual_seed(n)
    pred = torch.rand(n, c, hw, hw, generator=g)
    labels = torch.randint(0, c, (n,), generator=g)
    out = gather_class_masks(pred.to(DEV), labels.to(DEV))
    assert torch.equal(out.cpu(), M.gather_class_masks(pred, labels))


@pytest.mark.parametrize("n,c,hw,scale", [(3, 11, 2, 1000.0), (64, 1203, 28, 3.0), (33, 80, 14, 1.0)])
def test_mask_cross_entropy_loss_and_grad(n, c, hw, scale):
    from iif_amd.mmdet_mask_loss import mask_cross_entropy
    g = torch.Generator().manual_seed(c)
    pred = torch.randn(n, c, hw, hw, generator=g) * scale          # the reference's doctest uses *1000
    target = torch.rand(n, hw, hw, generator=g)
    labels = torch.randint(0, c, (n,), generator=g)
    pd = pred.to(DEV).requires_grad_(True)
    loss = mask_cross_entropy(pd, target.to(DEV), labels.to(DEV))
    assert loss.shape == (1,)
    (loss * 2.5).sum().backward()
    pr = pred.clone().requires_grad_(True)
    ref = M.mask_cross_entropy(pr, target, labels)
    (ref * 2.5).sum().backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert (pd.grad.cpu() - pr.grad).abs().max().item() <= 1e-5 * pr.grad.abs().max().item()
    # only the selected channels carry gradient
    sel = torch.zeros(n, c, dtype=torch.bool)
    sel[torch.arange(n), labels] = True
    assert pd.grad.cpu()[~sel].abs().max().item() == 0


def test_mask_loss_argument_contract():
    from iif_amd.mmdet_mask_loss import mask_cross_entropy
    p = torch.zeros(2, 3, 4, 4, device=DEV)
    t = torch.zeros(2, 4, 4, device=DEV)
    lb = torch.zeros(2, dtype=torch.long, device=DEV)
    with pytest.raises(AssertionError):
        mask_cross_entropy(p, t, lb, reduction="sum")
    with pytest.raises(AssertionError):
        mask_cross_entropy(p, t, lb, ignore_index=255)
    assert abs(mask_cross_entropy(p, t, lb).item() - 0.6931472) < 1e-6
