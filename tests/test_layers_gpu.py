"""GPU parity of the BN / pooling / glue kernels against torch CPU ops."""
import pytest
import torch
import torch.nn.functional as F

from oracle import iif_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def tol_for(dt):
    return 2e-5 if dt == torch.float32 else 2.0 ** -7


@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (3, 16, 9, 7), (2, 256, 5, 5), (2, 2048, 2, 2), (8, 8, 3, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["relu", "plain", "res", "res2"])
def test_batchnorm_forward_backward(shape, dt, mode):
    from iif_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).to(dt).float()
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.2
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    res = torch.randn(shape, generator=g).to(dt).float()
    gamma2, beta2 = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    gy = torch.randn(shape, generator=g).to(dt).float()
    # --- torch CPU reference
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    resr = res.clone().requires_grad_(True)
    g2r = gamma2.clone().requires_grad_(True); b2r = beta2.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if mode == "res":
        y = y + resr
    if mode == "res2":
        y = y + F.batch_norm(resr, None, None, g2r, b2r, True, 0.1, 1e-5)
    if mode != "plain":
        y = F.relu(y)
    y.backward(gy)
    # --- native
    m = n * h * w
    xd = nhwc(x).to(dt).to(DEV).view(m, c)
    ws = ops.bn_workspace(m, c, DEV)
    stats = torch.empty(4, c, device=DEV)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    ops.bn_forward_stats(xd, gd, bd, rmd, rvd, stats, ws)
    yd = torch.empty_like(xd)
    resd = nhwc(res).to(dt).to(DEV).view(m, c)
    stats2 = None
    if mode == "res2":
        stats2 = torch.empty(4, c, device=DEV)
        ops.bn_forward_stats(resd, gamma2.to(DEV), beta2.to(DEV), None, None, stats2, ws)
    vec = 8 if dt == torch.bfloat16 else 4
    bits = torch.zeros(m * c // vec, dtype=torch.uint8, device=DEV)
    ops.bn_apply(xd, stats, yd, relu=(mode != "plain"), residual=resd if mode in ("res", "res2") else None,
                 residual_stats=stats2, relu_bits=bits)
    tol = tol_for(dt)
    got_y = nchw(yd.view(n, h, w, c).float().cpu())
    assert (got_y - y.detach()).abs().max().item() <= tol * max(1.0, y.detach().abs().max().item())
    assert (rmd.cpu() - rm_ref).abs().max().item() <= 1e-5
    assert (rvd.cpu() - rv_ref).abs().max().item() <= 1e-5 * max(1.0, rv_ref.abs().max().item())
    # backward
    gyd = nhwc(gy).to(dt).to(DEV).view(m, c)
    dgam, dbet = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    dxd = torch.empty_like(xd)
    gm = torch.empty_like(xd) if mode in ("res", "res2") else None
    ops.bn_backward(gyd, yd if mode != "plain" else None, xd, stats, gd, dgam, dbet, dxd, ws, gmasked=gm)
    if mode != "plain":
        # the 1-bit mask path is the same computation with 1/16 of the mask bytes
        packed = (yd.view(-1, vec) > 0).to(torch.int32) * (2 ** torch.arange(vec, device=DEV, dtype=torch.int32))
        assert torch.equal(bits.to(torch.int32), packed.sum(1))
        dg2, db2 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        dx2 = torch.empty_like(xd)
        gm2 = torch.empty_like(xd) if gm is not None else None
        ops.bn_backward(gyd, None, xd, stats, gd, dg2, db2, dx2, ws, gmasked=gm2, relu_bits=bits)
        assert torch.equal(dx2, dxd) and torch.equal(dg2, dgam) and torch.equal(db2, dbet)
        if gm is not None:
            assert torch.equal(gm2, gm)
    gtol = 1e-4 if dt == torch.float32 else 3e-2        # bf16: mask flips on rounded outputs near 0
    ref_dx = xr.grad
    assert (nchw(dxd.view(n, h, w, c).float().cpu()) - ref_dx).abs().max().item() <= gtol * max(1e-3, ref_dx.abs().max().item())
    assert (dgam.cpu() - gr.grad).abs().max().item() <= gtol * max(1.0, gr.grad.abs().max().item())
    assert (dbet.cpu() - br.grad).abs().max().item() <= gtol * max(1.0, br.grad.abs().max().item())
    if mode == "res":
        assert (nchw(gm.view(n, h, w, c).float().cpu()) - resr.grad).abs().max().item() <= gtol * resr.grad.abs().max().item()
    if mode == "res2":
        d2g, d2b = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        dres = torch.empty_like(xd)
        ops.bn_backward(gm, None, resd, stats2, gamma2.to(DEV), d2g, d2b, dres, ws)
        assert (nchw(dres.view(n, h, w, c).float().cpu()) - resr.grad).abs().max().item() <= gtol * max(1e-3, resr.grad.abs().max().item())
        assert (d2g.cpu() - g2r.grad).abs().max().item() <= gtol * max(1.0, g2r.grad.abs().max().item())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_maxpool_and_avgpool(dt):
    from iif_amd import ops
    g = torch.Generator().manual_seed(1)
    for shape in ((2, 64, 12, 12), (3, 8, 7, 9), (1, 16, 112, 112)):
        x = F.relu(torch.randn(shape, generator=g)).to(dt).float().requires_grad_(True)   # many ties at 0
        y = F.max_pool2d(x, 3, 2, 1)
        gy = torch.randn(y.shape, generator=g).to(dt).float()
        y.backward(gy)
        yd, idx = ops.maxpool_forward(nhwc(x.detach()).to(dt).to(DEV), 3, 2, 1)
        assert torch.equal(nchw(yd.float().cpu()), y.detach())
        dx = ops.maxpool_backward(nhwc(gy).to(dt).to(DEV), idx, tuple(nhwc(x.detach()).shape), 3, 2, 1)
        # ties at 0 may route gradient to a different zero element than torch's NCHW scan,
        # ReLU backward zeroes those; compare where x > 0 and total mass
        ref = x.grad
        got = nchw(dx.float().cpu())
        pos = x.detach() > 0
        assert (got[pos] - ref[pos]).abs().max().item() <= tol_for(dt) * max(1.0, ref.abs().max().item())
        assert abs(got.sum().item() - ref.sum().item()) <= 1e-2 * max(1.0, ref.abs().sum().item() ** 0.5)
    x = torch.randn(5, 128, 7, 7, generator=g).to(dt).float().requires_grad_(True)
    y = F.adaptive_avg_pool2d(x, 1).flatten(1)
    gy = torch.randn(y.shape, generator=g).to(dt).float()
    y.backward(gy)
    yd = ops.avgpool_forward(nhwc(x.detach()).to(dt).to(DEV))
    assert (yd.float().cpu() - y.detach()).abs().max().item() <= tol_for(dt) * y.detach().abs().max().item()
    dx = ops.avgpool_backward(gy.to(dt).to(DEV), 49).view(5, 7, 7, 128)
    assert (nchw(dx.float().cpu()) - x.grad).abs().max().item() <= tol_for(dt) * x.grad.abs().max().item()


def test_maxpool_first_max_tie_rule():
    """Exact ties: the gradient goes to the first maximum in (kh, kw) scan order, as torch."""
    from iif_amd import ops
    x = torch.ones(1, 8, 6, 6)
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    y.backward(torch.ones_like(y))
    yd, idx = ops.maxpool_forward(nhwc(x).to(DEV), 3, 2, 1)
    dx = ops.maxpool_backward(torch.ones_like(yd), idx, (1, 6, 6, 8), 3, 2, 1)
    assert torch.equal(nchw(dx.cpu()), xr.grad)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_im2col_stem_equals_conv(dt):
    from iif_amd import ops
    g = torch.Generator().manual_seed(2)
    for (n, hw, cout, r, stride, pad, kp) in ((2, 32, 64, 7, 2, 3, 160), (3, 16, 16, 3, 1, 1, 32)):
        img = torch.randn(n, 3, hw, hw, generator=g)
        w = (torch.randn(cout, 3, r, r, generator=g) * 0.1).to(dt).float()
        ref = F.conv2d(img.to(dt).float(), w, None, stride, pad)
        patches = ops.im2col_nchw(img.to(DEV), r, r, stride, pad, kp, dt)
        w2 = torch.zeros(cout, kp)
        w2[:, :r * r * 3] = w.permute(0, 2, 3, 1).reshape(cout, -1)
        y = ops.conv_forward(patches, w2.to(dt).to(DEV), 1, 1, 1, 0)
        got = nchw(y.float().cpu())
        assert (got - ref).abs().max().item() <= tol_for(dt) * ref.abs().max().item()


def test_weight_transpose_cast_colsum_shortcut():
    from iif_amd import ops
    g = torch.Generator().manual_seed(4)
    w = torch.randn(24, 3, 3, 16, generator=g)                 # KRSC
    w2 = torch.zeros(24, 160); w2[:, :144] = w.reshape(24, -1)
    for dt in (torch.float32, torch.bfloat16):
        wt = torch.full((16, 224), 9.0, dtype=dt, device=DEV)
        ops.weight_transpose(w2.to(DEV), 24, 16, 9, wt)
        ref = w.permute(3, 1, 2, 0).reshape(16, -1).to(dt)     # [c][r][s][k]
        assert torch.equal(wt[:, :216].cpu(), ref) and (wt[:, 216:] == 0).all()
    src = torch.randn(1000, generator=g)
    dst = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    assert torch.equal(ops.cast(src.to(DEV), dst).cpu(), src.to(torch.bfloat16))
    a = torch.randn(37, 104, generator=g)
    out = torch.empty(100, device=DEV)
    ops.colsum_f32(a.to(DEV), 37, 100, 104, out)
    assert (out.cpu() - a[:, :100].sum(0)).abs().max().item() <= 1e-5
    x = torch.randn(2, 16, 8, 8, generator=g).requires_grad_(True)
    y = F.pad(x[:, :, ::2, ::2], (0, 0, 0, 0, 8, 8))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    yd = ops.shortcut_a_forward(nhwc(x.detach()).to(DEV), 32)
    assert torch.equal(nchw(yd.cpu()), y.detach())
    dx = torch.zeros(2, 8, 8, 16, device=DEV)
    ops.shortcut_a_backward_acc(nhwc(gy).to(DEV), dx)
    assert torch.equal(nchw(dx.cpu()), x.grad)


def test_fused_sgd_matches_torch_optim():
    from iif_amd import ops
    g = torch.Generator().manual_seed(5)
    for nesterov in (False, True):
        p = torch.randn(10007, generator=g)
        pd, bd = p.clone().to(DEV), torch.zeros(10007, device=DEV)
        q = p.clone().requires_grad_(True)
        opt = torch.optim.SGD([q], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=nesterov)
        ref_p, ref_b = [p.clone()], [None]
        for it in range(4):
            gr = torch.randn(10007, generator=g)
            lr = 0.1 * O.warmup_factor(it, 1000)
            for grp in opt.param_groups:
                grp["lr"] = lr
            q.grad = gr.clone()
            opt.step()
            O.sgd_step(ref_p, [gr], ref_b, lr, 0.9, 1e-4, nesterov)
            ops.sgd_step(pd, gr.to(DEV), bd, lr, 0.9, 1e-4, nesterov)
        assert (pd.cpu() - q.detach()).abs().max().item() <= 1e-6
        assert (pd.cpu() - ref_p[0]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("cpad", [16, 32])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_stem_space_to_depth_equals_7x7_stride2_conv(dt, cpad):
    """The s2d formulation of the ImageNet stem (4x4/1 conv on the 2x2 space-to-depth image) reproduces
    conv2d(7x7, stride 2, pad 3) forward and its weight gradient."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(8)
    n, hw, cout = 3, 32, 64
    img = torch.randn(n, 3, hw, hw, generator=g)
    w = (torch.randn(cout, 3, 7, 7, generator=g) * 0.1).to(dt).float()
    ref = F.conv2d(img.to(dt).float(), w, None, 2, 3)
    master = torch.zeros(cout, 160)
    master[:, :147] = w.permute(0, 2, 3, 1).reshape(cout, 147)
    x2 = torch.empty(n, hw // 2, hw // 2, cpad, dtype=dt, device=DEV)
    ops.space_to_depth_nchw(img.to(DEV), cpad, x2)
    wp = torch.empty(cout, 16 * cpad, dtype=dt, device=DEV)
    ops.stem_s2d_pack(master.to(DEV), cout, 3, 7, cpad, wp)
    y = ops.conv_forward(x2, wp, 4, 4, 1, 2, out_hw=(hw // 2, hw // 2))
    got = nchw(y.float().cpu())
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= tol_for(dt) * ref.abs().max().item()
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    refdw = torch.nn.grad.conv2d_weight(img.to(dt).float(), w.shape, dy, 2, 3).permute(0, 2, 3, 1).reshape(cout, 147)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    dwp = ops.conv_wgrad(x2, nhwc(dy).to(dt).to(DEV), 4, 4, 1, 2, ldw=16 * cpad, workspace=ws)
    dwm = torch.zeros(cout, 160, device=DEV)
    ops.stem_s2d_unpack_grad(dwp, cout, 3, 7, cpad, dwm)
    wtol = 2e-5 if dt == torch.float32 else 1e-4
    assert (dwm[:, :147].cpu() - refdw).abs().max().item() <= wtol * refdw.abs().max().item()
    assert (dwm[:, 147:] == 0).all()


@pytest.mark.parametrize("dt,n,hw,c,res", [(torch.float32, 3, 49, 64, 1), (torch.bfloat16, 2, 196, 256, 2), (torch.float32, 4, 64, 16, 0)])
def test_se_kernels_against_torch(dt, n, hw, c, res):
    """iif_se_squeeze / iif_se_apply / iif_se_backward_sums / iif_se_backward_form against their
    definitions (SE_Block.forward, resnet_pytorch.py:313-317, around the block's last BN)."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(21)
    h = int(hw ** 0.5)
    x = torch.randn(n, h, h, c, generator=g).to(dt)
    r = torch.randn(n, h, h, c, generator=g).to(dt)
    stats = torch.randn(4, c, generator=g)
    stats2 = torch.randn(4, c, generator=g)
    e = torch.rand(n, c, generator=g)
    xd, rd = x.to(DEV), r.to(DEV)
    sums = torch.empty(n, c, device=DEV)
    ops.se_squeeze(xd, sums)
    ref_sums = x.float().sum(dim=(1, 2))
    assert (sums.cpu() - ref_sums).abs().max().item() <= 1e-5 * max(1.0, ref_sums.abs().max().item())
    y = torch.empty_like(xd)
    bits = torch.empty(n * hw * c // (8 if dt == torch.bfloat16 else 4), dtype=torch.uint8, device=DEV)
    ops.se_apply(xd, stats.to(DEV), e.to(DEV), y, bits, residual=rd if res else None,
                 residual_stats=stats2.to(DEV) if res == 2 else None)
    t = (x.float() * stats[2] + stats[3]) * e[:, None, None, :]
    if res == 1:
        t = t + r.float()
    if res == 2:
        t = t + r.float() * stats2[2] + stats2[3]
    ref_y = t.clamp_min(0)
    assert (y.float().cpu() - ref_y).abs().max().item() <= tol_for(dt) * max(1.0, ref_y.abs().max().item())
    gy = torch.randn(n, h, h, c, generator=g).to(dt)
    gd = gy.to(DEV).clone()
    s1 = torch.empty(n, c, device=DEV)
    s2 = torch.empty(n, c, device=DEV)
    ops.se_backward_sums(gd, bits, xd, s1, s2)
    mask = (y.float().cpu() > 0)
    gm = gy.float() * mask
    assert torch.equal(gd.float().cpu(), gm.to(dt).float())
    assert (s1.cpu() - gm.sum(dim=(1, 2))).abs().max().item() <= 1e-4 * hw ** 0.5
    assert (s2.cpu() - (gm * x.float()).sum(dim=(1, 2))).abs().max().item() <= 2e-4 * hw ** 0.5
    off = torch.randn(n, c, generator=g)
    out = torch.empty_like(gd)
    ops.se_backward_form(gd, e.to(DEV), off.to(DEV), out)
    ref_o = gm.to(dt).float() * e[:, None, None, :] + off[:, None, None, :]
    assert (out.float().cpu() - ref_o).abs().max().item() <= tol_for(dt) * max(1.0, ref_o.abs().max().item())


@pytest.mark.parametrize("n,c,hid,hw", [(5, 16, 4, 64), (8, 64, 16, 9), (13, 256, 16, 196), (6, 2048, 128, 49), (256, 512, 32, 784)])
def test_se_excitation_native_against_torch(n, c, hid, hw):
    """The squeeze-and-excitation MLP (SE_Block.forward, resnet_pytorch.py:306-317: two bias-free linears, ReLU, sigmoid) as
    native launches: forward from the squeeze sums and the BN affine, backward from the per-sample sums of
    iif_se_backward_sums, both weight gradients — against torch autograd in float64 (the kernels are fp32)."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(n * 7 + c)
    sums = torch.randn(n, c, generator=g) * hw ** 0.5
    stats = torch.zeros(4, c)
    stats[2] = torch.rand(c, generator=g) + 0.5
    stats[3] = torch.randn(c, generator=g) * 0.3
    ld1, ld2 = (c + 15) // 16 * 16, (hid + 15) // 16 * 16            # parameter rows are padded to 16 columns in the arena
    w1 = torch.zeros(hid, ld1); w1[:, :c] = torch.randn(hid, c, generator=g) / c ** 0.5
    w2 = torch.zeros(c, ld2); w2[:, :hid] = torch.randn(c, hid, generator=g) / hid ** 0.5
    s1 = torch.randn(n, c, generator=g)
    s2 = torch.randn(n, c, generator=g) * 2.0
    d = lambda t: t.to(DEV)                                          # noqa: E731
    Z = lambda *sh: torch.full(sh, float("nan"), device=DEV)           # noqa: E731
    w1d, w2d = d(w1), d(w2)
    w2t = ops.transpose_f32(w2d[:, :hid], Z(hid, c))
    q, h, e = Z(n, c), Z(n, hid), Z(n, c)
    ops.se_excite_forward(d(sums), d(stats), hw, w1d, w2t, q, h, e)
    # reference in float64
    W1 = w1[:, :c].double().requires_grad_(True)
    W2 = w2[:, :hid].double().requires_grad_(True)
    qr = (stats[2].double() * sums.double() / hw + stats[3].double()).requires_grad_(True)
    hr = torch.relu(qr @ W1.t())
    er = torch.sigmoid(hr @ W2.t())
    assert (q.cpu().double() - qr.detach()).abs().max().item() <= 1e-5 * qr.abs().max().item()
    assert (h.cpu().double() - hr.detach()).abs().max().item() <= 2e-5 * max(1.0, hr.abs().max().item())
    assert (e.cpu().double() - er.detach()).abs().max().item() <= 2e-6
    de = stats[2].double() * s2.double() + stats[3].double() * s1.double()
    er.backward(de)
    dz2, dz1, off = Z(n, c), Z(n, hid), Z(n, c)
    dw1 = torch.zeros(hid, ld1, device=DEV)
    dw2 = torch.zeros(c, ld2, device=DEV)
    ops.se_excite_backward(d(s1), d(s2), d(stats), hw, w1d, w2t, e, h, q, dz2, dz1, off, dw1, dw2)
    rel = lambda a, b: (a.cpu().double() - b).norm().item() / max(b.norm().item(), 1e-30)      # noqa: E731
    assert rel(off, qr.grad / hw) <= 2e-5
    assert rel(dw1[:, :c], W1.grad) <= 2e-5 and rel(dw2[:, :hid], W2.grad) <= 2e-5
    assert not dw1[:, c:].any() and not dw2[:, hid:].any()           # the arena's pad columns stay zero


def test_weight_transpose_batched_matches_per_tensor():
    """One-launch tiled transpose of several [cout][rs*cin] tensors out of a flat fp32 arena == iif_weight_transpose
    of each (pad columns zero), for fp32 and bf16 outputs, ragged channel counts included."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(2)
    shapes = [(64, 64, 9, 576), (1000, 2048, 1, 2048), (40, 24, 9, 224), (256, 64, 1, 64), (8, 8, 9, 80)]
    arena = torch.randn(sum(co * ldw for co, _, _, ldw in shapes) + 64, generator=g).to(DEV)
    for dt in (torch.float32, torch.bfloat16):
        entries, off_in, off_out, views = [], 16, 0, []
        for (co, ci, rs, ldw) in shapes:
            ldwt = (rs * co + 15) // 16 * 16
            entries.append((off_in, off_out, co, ci, rs, ldw, ldwt))
            views.append((off_in, off_out, co, ci, rs, ldw, ldwt))
            off_in += co * ldw
            off_out += (ci * ldwt + 63) // 64 * 64
        table, blocks = ops.wt_table(entries, DEV)
        out = torch.full((off_out,), 7.0, dtype=dt, device=DEV)
        ops.weight_transpose_batched(arena, table, len(entries), blocks, out)
        for (oi, oo, co, ci, rs, ldw, ldwt) in views:
            ref = torch.empty(ci, ldwt, dtype=dt, device=DEV)
            ops.weight_transpose(arena[oi:oi + co * ldw].view(co, ldw), co, ci, rs, ref)
            assert torch.equal(out[oo:oo + ci * ldwt].view(ci, ldwt), ref), (co, ci, rs)


@pytest.mark.parametrize("nt,c", [(700, 64), (6272, 256), (1568, 512), (3000, 40), (513, 2048)])
def test_bn_reduce_and_finalize_in_one_launch_is_bit_identical(nt, c):
    """iif_bn_finalize_stats_fused / iif_bn_backward_partials_fused (slice sums published with agent-scope atomics,
    last block of every 32-channel group finalises) against the two-launch path: same fixed-order sums, bit for bit,
    and the ticket words are left at zero for the next call."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(nt + c)
    m = nt * 128
    partial = torch.randn(nt, 2, c, generator=g).abs_().mul_(100.0)
    partial[:, 1] += partial[:, 0] ** 2 / 128            # keep the variance positive
    partial = partial.to(DEV)
    gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
    tickets = torch.zeros(64, dtype=torch.int32, device=DEV)
    out = []
    for tk in (None, tickets, tickets):
        stats = torch.full((4, c), float("nan"), device=DEV)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        ops.bn_finalize_stats(partial.view(-1), nt, m, c, gamma, beta, rm, rv, stats, scratch=torch.empty(128 * c, device=DEV), tickets=tk)
        out.append((stats, rm, rv))
    assert tickets.abs().sum().item() == 0
    for a_, b_ in zip(out[0], out[1]):
        assert torch.equal(a_, b_)
    for a_, b_ in zip(out[1], out[2]):
        assert torch.equal(a_, b_)
    # backward flavour through bn_backward_partials (bf16 tensors, c % 8 == 0)
    if c % 8 == 0:
        mm = 256
        gy = torch.randn(mm, c, generator=g).to(torch.bfloat16).to(DEV)
        x = torch.randn(mm, c, generator=g).to(torch.bfloat16).to(DEV)
        bits = torch.randint(0, 255, (mm * c // 8,), dtype=torch.uint8, generator=g).to(DEV)
        st = torch.rand(4, c, device=DEV) + 0.5
        res = []
        for tk in (None, tickets):
            dg, db = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
            dx = torch.empty_like(x)
            ws = ops.bn_workspace(mm, c, DEV)
            ops.bn_backward_partials(gy, bits, x, st, gamma, partial.view(-1), nt, dg, db, dx, ws, tickets=tk)
            res.append((dg, db, dx))
        assert tickets.abs().sum().item() == 0
        for a_, b_ in zip(res[0], res[1]):
            assert torch.equal(a_, b_)


def test_ticketed_single_launch_reductions_under_memory_load():
    """The fence-free ticket protocol (bn.hip bn_reduce_finalize_kernel, iif_head.hip finish_with_ticket: partials published
    with agent-scope atomic exchanges, s_waitcnt, relaxed ticket; the last block reads them with agent-scope atomic loads)
    is only as good as the hardware property it rests on, and the bit-identity tests above run on an idle GPU.  Here the
    single-launch forms run 60 times each while a second stream keeps every XCD's L2 and the HBM busy with large copies
    and read-modify-write passes (uneven load, dirty lines in flight): every result must equal the two-launch path bit for
    bit and the ticket words must be back at zero."""
    from iif_amd import custom, ops
    g = torch.Generator().manual_seed(5)
    nt, c = 3136, 256                                      # the 28x28 / 56x56 regime: > 512 partial rows -> two-stage reduce
    m = nt * 128
    partial = torch.randn(nt, 2, c, generator=g).abs_().mul_(100.0)
    partial[:, 1] += partial[:, 0] ** 2 / 128
    partial = partial.to(DEV)
    gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV)
    tickets = torch.zeros(64, dtype=torch.int32, device=DEV)
    scratch = torch.empty(128 * c, device=DEV)

    def finalize(tk):
        stats = torch.full((4, c), float("nan"), device=DEV)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        ops.bn_finalize_stats(partial.view(-1), nt, m, c, gamma, beta, rm, rv, stats, scratch=scratch, tickets=tk)
        return stats
    ref_stats = finalize(None)
    B, C = 4096, 1000
    x = torch.randn(B, C, generator=g).to(DEV)
    y = torch.randint(0, C, (B,), generator=g).to(DEV)
    tab = (torch.rand(1, C, generator=g) * 4 + 0.5).to(DEV)
    loss_ref, _, _ = custom._launch_ce(x, tab, y, None, 1.0, None, None, -100, 1.0 / B, True)
    loss_ref = loss_ref.clone()
    torch.cuda.synchronize()
    noise = torch.cuda.Stream()
    big_a = torch.empty(96 << 20, dtype=torch.float32, device=DEV)          # 384 MB each: beyond the 256 MiB Infinity Cache
    big_b = torch.empty(96 << 20, dtype=torch.float32, device=DEV)
    with torch.cuda.stream(noise):
        for _ in range(40):
            big_b.copy_(big_a); big_a.add_(1.0); big_a[: 1 << 20].mul_(0.5)
    for it in range(60):
        st = finalize(tickets)
        loss, _, _ = custom._launch_ce(x, tab, y, None, 1.0, None, None, -100, 1.0 / B, True)
        assert torch.equal(st, ref_stats), it
        assert torch.equal(loss, loss_ref), it
    torch.cuda.synchronize()
    assert tickets.abs().sum().item() == 0


@pytest.mark.parametrize("shape", [(4, 64, 14, 14), (2, 2048, 2, 2), (6, 24, 5, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("masked", ["bits", "ymask", "plain", "gmasked"])
def test_sync_bn_pieces_of_two_halves_equal_the_whole_batch(shape, dt, masked):
    """SyncBatchNorm split entry points (classification/train.py:190-191): two "ranks" hold the halves of one batch; their
    per-rank sums added (what the all-reduce does) and finalised with the global count give the whole-batch statistics, and the
    backward pieces (local sums -> dgamma / dbeta, total sums -> dx) add up to the whole-batch bn_backward."""
    from iif_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(3 * c + h)
    m, mh = n * h * w, (n // 2) * h * w
    xd = (torch.randn(m, c, generator=g) * 1.3 + 0.2).to(dt).to(DEV)
    gyd = torch.randn(m, c, generator=g).to(dt).to(DEV)
    gd, bd = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.randn(c, generator=g) * 0.2).to(DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    ws = ops.bn_workspace(m, c, DEV)
    stats = torch.empty(4, c, device=DEV)
    ops.bn_forward_stats(xd, gd, bd, rm, rv, stats, ws)
    # forward pieces
    halves = [slice(0, mh), slice(mh, m)]
    sums = [ops.bn_stats_sums(xd[s], torch.empty(2, c, device=DEV), ws).clone() for s in halves]
    total = sums[0] + sums[1]
    rm2, rv2 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    stats2 = torch.empty(4, c, device=DEV)
    ops.bn_finalize_stats(total, 1, m, c, gd, bd, rm2, rv2, stats2, 1e-5, 0.1)
    assert (stats2[:2] - stats[:2]).abs().max().item() <= 2e-5 * stats[:2].abs().max().item()
    assert (rm2 - rm).abs().max().item() <= 1e-5 and (rv2 - rv).abs().max().item() <= 1e-5 * rv.abs().max().item()
    # backward pieces against the whole-batch kernel, on the SAME statistics
    vec = 8 if dt == torch.bfloat16 else 4
    yd = torch.empty_like(xd)
    bits = torch.zeros(m * c // vec, dtype=torch.uint8, device=DEV)
    ops.bn_apply(xd, stats, yd, relu=True, relu_bits=bits)
    ymask = yd if masked in ("ymask", "gmasked") else None
    rbits = bits if masked == "bits" else None
    dg, db, dx = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty_like(xd)
    gm = torch.empty_like(xd) if masked == "gmasked" else None
    ops.bn_backward(gyd, ymask, xd, stats, gd, dg, db, dx, ws, gmasked=gm, relu_bits=rbits)
    local = []
    for s in halves:
        hb = None if rbits is None else rbits[s.start * c // vec:s.stop * c // vec]
        local.append(ops.bn_backward_sums(gyd[s], None if ymask is None else ymask[s], xd[s], stats, torch.empty(2, c, device=DEV), ws,
                                          relu_bits=hb).clone())
    tot = local[0] + local[1]
    dgs, dbs = [], []
    dx2 = torch.empty_like(xd)
    gm2 = torch.empty_like(xd) if gm is not None else None
    coef = torch.empty(3, c, device=DEV)
    for s, lc in zip(halves, local):
        hb = None if rbits is None else rbits[s.start * c // vec:s.stop * c // vec]
        dgh, dbh = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        ops.bn_backward_apply_sums(gyd[s], None if ymask is None else ymask[s], xd[s], stats, gd, lc, tot, float(m), dgh, dbh, dx2[s], coef,
                                   gmasked=None if gm2 is None else gm2[s], relu_bits=hb)
        dgs.append(dgh); dbs.append(dbh)
    tol = 2e-5 if dt == torch.float32 else 2.0 ** -7
    assert ((dgs[0] + dgs[1]) - dg).abs().max().item() <= 2e-5 * max(1.0, dg.abs().max().item())
    assert ((dbs[0] + dbs[1]) - db).abs().max().item() <= 2e-5 * max(1.0, db.abs().max().item())
    assert (dx2.float() - dx.float()).abs().max().item() <= tol * max(1e-3, dx.float().abs().max().item())
    if gm is not None:
        assert torch.equal(gm2, gm)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 112, 112, 64), (3, 17, 23, 64), (1, 9, 9, 32), (2, 8, 8, 128)], ids=lambda s_: "x".join(map(str, s_)))
def test_fused_stem_pool_equals_bn_apply_then_maxpool(dt, shape):
    """iif_maxpool_bn_forward (3/2/1: the row-per-block kernel with nine candidates in flight) = iif_bn_apply (ReLU) followed
    by iif_maxpool_forward: pooled values and arg-max codes bit-identical, including the borders and the first-max tie rule
    (ReLU produces runs of equal zeros)."""
    from iif_amd import _lib, ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(h * 10 + w + c)
    x = (torch.randn(n, h, w, c, generator=g) - 0.3).to(dt).to(DEV)
    m = n * h * w
    stats = torch.zeros(4, c, device=DEV)
    stats[2] = (torch.rand(c, generator=g) * 2 - 0.5).to(DEV)          # some negative gains too
    stats[3] = (torch.randn(c, generator=g) * 0.3).to(DEV)
    act = torch.empty_like(x)
    ops.bn_apply(x.view(m, c), stats, act.view(m, c), relu=True)
    y_ref, idx_ref = ops.maxpool_forward(act, 3, 2, 1)
    y = torch.empty_like(y_ref)
    idx = torch.full_like(idx_ref, 255)
    px = torch.full_like(y_ref, 7.0)
    _lib.check(_lib.lib().iif_maxpool_bn_forward(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(stats), n, h, w, c, 3, 2, 1, _lib.ptr(y),
                                                 _lib.ptr(idx), _lib.ptr(px), _lib.stream_ptr()), "iif_maxpool_bn_forward")
    assert torch.equal(y, y_ref) and torch.equal(idx, idx_ref)
    # pool_x = the raw input at the arg max (code = kh * 3 + kw of the 3 x 3 window at (2 ho - 1, 2 wo - 1))
    ho, wo = y_ref.shape[1], y_ref.shape[2]
    code = idx_ref.long().cpu()
    hh = (2 * torch.arange(ho).view(1, ho, 1, 1) - 1 + code // 3)
    ww = (2 * torch.arange(wo).view(1, 1, wo, 1) - 1 + code % 3)
    nn_ = torch.arange(n).view(n, 1, 1, 1).expand_as(code)
    cc = torch.arange(c).view(1, 1, 1, c).expand_as(code)
    assert torch.equal(px.cpu(), x.cpu()[nn_, hh, ww, cc])
    y2 = torch.empty_like(y_ref)
    _lib.check(_lib.lib().iif_maxpool_bn_forward(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(stats), n, h, w, c, 3, 2, 1, _lib.ptr(y2),
                                                 _lib.ptr(idx), None, _lib.stream_ptr()), "iif_maxpool_bn_forward (no pool_x)")
    assert torch.equal(y2, y_ref)


@pytest.mark.parametrize("gains", ["ordinary", "small_gain_large_shift"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stem_bn_backward_sums_from_the_pooled_tensors(dt, gains):
    """iif_bn_backward_relu_recompute_pooled: the BN-backward column sums of the stem (bn1 -> relu -> 3x3/2 max pool,
    resnet_pytorch.py:284-287) taken from the pooled gradient and the raw stem output at the arg max (pool_x) — dgamma, dbeta
    and dx against the standard route (reduction pass over the scattered gradient and the stem output) on the same tensors.
    ``small_gain_large_shift``: channels with |beta| up to 2000 |gamma| (a pretrained bn1 has such channels) — recovering xhat
    from the pooled ACTIVATION, as round 4 did, turns their sum g xhat into rounding noise (error 2^-9 |xhat + beta / gamma|);
    from pool_x the sums are the standard pass's terms in another order."""
    from iif_amd import _lib, ops
    n, h, w, c = 3, 18, 22, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, c, generator=g).to(dt).to(DEV)
    m = n * h * w
    gamma = (torch.rand(c, generator=g) + 0.5)
    beta = (torch.randn(c, generator=g) * 0.2)
    if gains == "small_gain_large_shift":
        gamma[::2] = 10.0 ** (-3 * torch.rand(c // 2, generator=g) - 0.5) * torch.sign(torch.randn(c // 2, generator=g))
        beta[::2] = 0.3 + torch.rand(c // 2, generator=g)
    gamma, beta = gamma.to(DEV), beta.to(DEV)
    stats = torch.zeros(4, c, device=DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    ops.bn_forward_stats(x.view(m, c), gamma, beta, rm, rv, stats, ops.bn_workspace(m, c, DEV))
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    pooled = torch.empty(n, ho, wo, c, dtype=dt, device=DEV)
    pool_x = torch.empty(n, ho, wo, c, dtype=dt, device=DEV)
    idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=DEV)
    L = _lib.lib()
    _lib.check(L.iif_maxpool_bn_forward(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(stats), n, h, w, c, 3, 2, 1, _lib.ptr(pooled),
                                        _lib.ptr(idx), _lib.ptr(pool_x), _lib.stream_ptr()), "pool")
    gp = torch.randn(n, ho, wo, c, generator=g).to(dt).to(DEV)
    out = {}
    for mode in ("standard", "pooled", "fused"):
        dy0 = torch.empty(n, h, w, c, dtype=dt, device=DEV)
        if mode == "fused":
            # round 5: the max-pool backward gathered inside the normalisation pass (iif_bn_backward_pool_fused); no scattered tensor
            dy0.fill_(float("nan"))
            dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
            ws = ops.bn_workspace(m, c, DEV)
            _lib.check(L.iif_bn_backward_pool_fused(_lib.ptr(gp), _lib.ptr(idx), _lib.ptr(pool_x), _lib.ptr(x), _lib.dtype_code(x), n, h, w, c,
                                                    ho, wo, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(dy0),
                                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "fused")
            out[mode] = (dgam.cpu(), dbet.cpu(), dy0.float().cpu())
            continue
        _lib.check(L.iif_maxpool_backward(_lib.ptr(gp), _lib.ptr(idx), _lib.dtype_code(gp), n, h, w, c, 3, 2, 1, _lib.ptr(dy0),
                                          _lib.stream_ptr()), "pool bwd")
        dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        ws = ops.bn_workspace(m, c, DEV)
        if mode == "standard":
            _lib.check(L.iif_bn_backward_relu_recompute(_lib.ptr(dy0), _lib.ptr(x), _lib.dtype_code(x), m, c, _lib.ptr(stats), _lib.ptr(gamma),
                                                        _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(dy0), _lib.ptr(ws), ws.numel(),
                                                        _lib.stream_ptr()), "std")
        else:
            _lib.check(L.iif_bn_backward_relu_recompute_pooled(_lib.ptr(dy0), _lib.ptr(x), _lib.dtype_code(x), m, c, _lib.ptr(stats),
                                                               _lib.ptr(gamma), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(dy0), _lib.ptr(ws),
                                                               ws.numel(), _lib.ptr(gp), _lib.ptr(pool_x), n * ho * wo, _lib.stream_ptr()),
                       "pooled")
        out[mode] = (dgam.cpu(), dbet.cpu(), dy0.float().cpu())
    # bf16: the scattered gradient of a pixel that is the arg max of several windows was rounded to bf16 once more than the
    # pooled gradients themselves; everything else is summation order
    for a, b in zip(out["fused"], out["pooled"]):            # dgamma, dbeta, dx: bit-identical to the two calls
        assert torch.equal(a, b)
    tol = 2e-5 if dt == torch.float32 else 4e-3
    ref = out["standard"]
    got = out["pooled"]
    assert (got[1] - ref[1]).abs().max().item() <= (3.2e-5 if dt == torch.float32 else 2e-3) * max(1.0, ref[1].abs().max().item())
    # dgamma per channel, so that the small-gain channels are held to the bound themselves (a norm over all channels would hide
    # them).  What separates the two routes is rounding noise of the terms g * xhat of ONE channel: the storage rounding of the
    # scattered gradient (eps of the dtype, independent per element -> eps * l2 of the terms) and the order of the fp32 additions
    # (-> 1e-6 * l1).  Round 4's xhat-from-the-pooled-activation had eps * |xhat + beta / gamma| per term instead: ~1000 x this bound
    # on the channels of the second case.
    dy_s = torch.empty(n, h, w, c, dtype=dt, device=DEV)
    _lib.check(L.iif_maxpool_backward(_lib.ptr(gp), _lib.ptr(idx), _lib.dtype_code(gp), n, h, w, c, 3, 2, 1, _lib.ptr(dy_s),
                                      _lib.stream_ptr()), "pool bwd")
    xf = x.float().view(m, c)
    terms = (dy_s.float().view(m, c) * (torch.addcmul(stats[3], stats[2], xf) > 0) * ((xf - stats[0]) * stats[1])).cpu().double()
    eps = 2.0 ** -24 if dt == torch.float32 else 2.0 ** -9
    bound = 6 * eps * terms.pow(2).sum(0).sqrt() + 4e-6 * terms.abs().sum(0)
    assert ((got[0] - ref[0]).abs().double() <= bound).all(), ((got[0] - ref[0]).abs().double() / bound).max()
    assert (got[0] - ref[0]).norm().item() <= tol * ref[0].norm().item()
    assert (got[2] - ref[2]).norm().item() <= tol * ref[2].norm().item()
