"""Native mmdet normed predictors (NormedLinear / IIFNormedLinear / NormedConv2d 1x1) against the CPU
restatement of instance_segmentation/mmdet/models/utils/normed_predictor.py (oracle.mmdet_iif).
fp32, tolerance 1e-4 relative on outputs and every gradient."""
import pytest
import torch

from oracle import mmdet_iif as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("n,d,c,temp,power,iif", [(37, 64, 11, 20, 1.0, False), (256, 1024, 1204, 8, 1.0, True),
                                                   (16, 128, 81, 20, 2.0, False), (5, 96, 30, 8, 1.0, True)])
def test_normed_linear_forward_backward(tmp_path, n, d, c, temp, power, iif):
    from iif_amd.mmdet_normed_predictor import IIFNormedLinear, NormedLinear
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, d, generator=g)
    gy = torch.randn(n, c, generator=g)
    if iif:
        table = torch.rand(c, generator=g) * 5 + 0.5
        table[-1] = 1.0
        path = tmp_path / "idf.csv"
        with open(path, "w") as f:
            f.write("idx,base2_obj\n0,1.0\n")                 # placeholder first row (iif_loss.py:47-50 drops it)
            for i in range(c - 1):
                f.write("%d,%.9g\n" % (i + 1, table[i].item()))
        m = IIFNormedLinear(d, c, tempearture=temp, power=power, path=str(path), variant="base2_obj").to(DEV)
        rows = m.iif_weights.reshape(-1).cpu()
        assert torch.allclose(rows, table, rtol=1e-6, atol=0)
    else:
        m = NormedLinear(d, c, tempearture=temp, power=power).to(DEV)
        rows = None
    with torch.no_grad():
        m.weight.copy_(torch.randn(c, d, generator=g) * 0.05)
        m.bias.copy_(torch.randn(c, generator=g) * 0.1)
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)
    out.backward(gy.to(DEV))
    xr = x.clone().requires_grad_(True)
    w = m.weight.detach().cpu().clone().requires_grad_(True)
    b = m.bias.detach().cpu().clone().requires_grad_(True)
    ref = M.normed_linear(xr, w, b, temp, power, 1e-6, rows)
    ref.backward(gy)
    assert out.shape == ref.shape
    assert rel(out, ref) <= 1e-4
    assert rel(xd.grad, xr.grad) <= 1e-4
    assert rel(m.weight.grad, w.grad) <= 1e-4
    assert rel(m.bias.grad, b.grad) <= 1e-4


def test_normed_linear_empty_batch_and_init():
    from iif_amd.mmdet_normed_predictor import NormedLinear
    m = NormedLinear(64, 10).to(DEV)
    assert abs(m.weight.std().item() - 0.01) < 0.003 and m.bias.abs().max().item() == 0      # :29-32
    out = m(torch.zeros(0, 64, device=DEV, requires_grad=True))
    assert out.shape == (0, 10)
    out.sum().backward()
    assert m.weight.grad.abs().max().item() == 0


def test_normed_conv2d_1x1():
    from iif_amd.mmdet_normed_predictor import NormedConv2d
    g = torch.Generator().manual_seed(4)
    m = NormedConv2d(256, 80, 1, tempearture=20).to(DEV)
    x = torch.randn(3, 256, 14, 14, generator=g)
    gy = torch.randn(3, 80, 14, 14, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)
    out.backward(gy.to(DEV))
    xr = x.clone().requires_grad_(True)
    w = m.weight.detach().cpu().clone().requires_grad_(True)
    b = m.bias.detach().cpu().clone().requires_grad_(True)
    ref = M.normed_conv2d_1x1(xr, w, b, 20, 1.0, 1e-6)
    ref.backward(gy)
    assert rel(out, ref) <= 1e-4 and rel(xd.grad, xr.grad) <= 1e-4
    assert rel(m.weight.grad, w.grad) <= 1e-4 and rel(m.bias.grad, b.grad) <= 1e-4
    # k x k kernels are native since round 3 (test_g12_normed_conv2d pins them); what the MFMA path cannot take still raises
    NormedConv2d(8, 8, 3, padding=1)
    for bad in (dict(groups=2), dict(dilation=2), dict(stride=3)):
        with pytest.raises(NotImplementedError):
            NormedConv2d(8, 8, 3, **bad)
    with pytest.raises(NotImplementedError):
        NormedConv2d(6, 8, 3)                      # input channels not in fours
