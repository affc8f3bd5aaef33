"""GPU parity of the implicit-GEMM convolution kernels against torch CPU
conv2d (the third-party kernel the reference dispatches to).
fp32 mode: exact-fp32 MFMA, tolerance 1e-5 relative to sum|a*b| scale.
bf16 mode: inputs rounded to bf16 on both sides, fp32 accumulate; the only
difference left is summation order and the final bf16 rounding (2^-8)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def krsc(w, pe):
    """OIHW -> [Cout, R*S*Cin padded to a multiple of pe]"""
    co, ci, r, s = w.shape
    flat = w.permute(0, 2, 3, 1).reshape(co, r * s * ci)
    k = flat.shape[1]
    kp = (k + pe - 1) // pe * pe
    out = torch.zeros(co, kp, dtype=w.dtype)
    out[:, :k] = flat
    return out


CASES = [
    # N, Cin, H, W, Cout, R, stride, pad
    (2, 64, 14, 14, 64, 3, 1, 1),
    (2, 64, 15, 13, 128, 3, 2, 1),
    (3, 128, 9, 9, 256, 1, 1, 0),
    (2, 256, 10, 10, 512, 1, 2, 0),
    (2, 16, 12, 12, 16, 3, 1, 1),       # Cin=16: two taps per bf16 K step, ragged K tail
    (2, 16, 12, 12, 32, 3, 2, 1),
    (1, 32, 7, 7, 1000, 1, 1, 0),       # Cout not a multiple of the tile
    (5, 8, 6, 6, 24, 3, 1, 1),
    (2, 512, 7, 7, 512, 3, 1, 1),       # long K
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_forward_and_dgrad(case, dt):
    from iif_amd import ops
    n, cin, h, w, cout, r, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g).to(dt).float()
    wt = (torch.randn(cout, cin, r, r, generator=g) / (cin * r * r) ** 0.5).to(dt).float()
    pe = 4 if dt == torch.float32 else 8
    ref = F.conv2d(x, wt, None, stride, pad)
    y = ops.conv_forward(nhwc(x).to(dt).to(DEV), krsc(wt, pe).to(dt).to(DEV), r, r, stride, pad)
    got = y.float().cpu().permute(0, 3, 1, 2)
    scale = ref.abs().max().item()
    tol = 2e-5 if dt == torch.float32 else 2.0 ** -7
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= tol * scale, (case, dt)
    # data gradient: dX = conv_transpose(dY, W)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    refdx = torch.nn.grad.conv2d_input(x.shape, wt, dy, stride, pad)
    wtt = wt.permute(1, 0, 2, 3).contiguous()           # [Cin, Cout, R, S] -> rows over (r,s,cout)
    dx = ops.conv_dgrad(nhwc(dy).to(dt).to(DEV), krsc(wtt, pe).to(dt).to(DEV), r, r, stride, pad, (h, w))
    gotdx = dx.float().cpu().permute(0, 3, 1, 2)
    assert (gotdx - refdx).abs().max().item() <= tol * refdx.abs().max().item(), (case, dt)
    # epilogue residual (gradient accumulation at a residual join), aliasing dst
    acc = nhwc(torch.randn(x.shape, generator=g).to(dt).float()).to(dt).to(DEV)
    exp = gotdx + acc.float().cpu().permute(0, 3, 1, 2)
    ops.conv_dgrad(nhwc(dy).to(dt).to(DEV), krsc(wtt, pe).to(dt).to(DEV), r, r, stride, pad, (h, w), out=acc, res=acc)
    assert (acc.float().cpu().permute(0, 3, 1, 2) - exp).abs().max().item() <= tol * exp.abs().max().item()


def test_fc_layer_bias_fp32_out():
    from iif_amd import ops
    g = torch.Generator().manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        for cout in (1000, 365, 100):
            x = torch.randn(37, 2048, generator=g).to(dt).float()
            w = (torch.randn(cout, 2048, generator=g) / 45.0).to(dt).float()
            b = torch.randn(cout, generator=g)
            ref = F.linear(x, w, b)
            y = ops.conv_forward(x.view(37, 1, 1, 2048).to(dt).to(DEV), w.to(dt).to(DEV), 1, 1, 1, 0,
                                 out_dtype=torch.float32, bias=b.to(DEV))
            assert y.dtype == torch.float32
            got = y.cpu().view(37, cout)
            tol = 2e-5 if dt == torch.float32 else 1e-4
            assert (got - ref).abs().max().item() <= tol * ref.abs().max().item(), (dt, cout)


def test_conv_rejects_bad_shapes():
    from iif_amd import ops, _lib
    x = torch.zeros(1, 4, 4, 6, device=DEV)           # Cin=6 not a multiple of 4
    w = torch.zeros(8, 56, device=DEV)
    with pytest.raises(_lib.IIFNativeError):
        ops.conv_forward(x, w, 3, 3, 1, 1)
    x = torch.zeros(1, 4, 4, 8, device=DEV)
    w = torch.zeros(8, 72, device=DEV)
    with pytest.raises(_lib.IIFNativeError):
        ops.conv_forward(x, w, 3, 3, 3, 1)            # stride 3 unsupported


# 3x3/s1/p1 with channels in 64s run the nine-taps-per-block kernel in bf16 (conv3x3_wgrad_halo_kernel): both ring
# leads (W+3 <= 32 | 64), non-square images, W not a multiple of anything, several channel tiles, one image
HALO_WGRAD = [(2, 64, 14, 14, 64, 3, 1, 1), (3, 128, 7, 7, 64, 3, 1, 1), (2, 64, 56, 56, 64, 3, 1, 1), (1, 128, 28, 28, 128, 3, 1, 1),
              (4, 64, 9, 13, 128, 3, 1, 1), (2, 64, 37, 61, 64, 3, 1, 1), (1, 256, 5, 30, 64, 3, 1, 1),
              # padded images shorter than the ring lead (the cursor starts more than one image before pixel 0)
              (4, 64, 2, 2, 64, 3, 1, 1), (3, 128, 1, 1, 64, 3, 1, 1), (2, 64, 4, 3, 128, 3, 1, 1), (40, 64, 1, 2, 64, 3, 1, 1)]
# >= 256 output channels with long kernel rows (the 256-channel tile, 8 waves)
SQUARE_WGRAD = [(2, 512, 14, 14, 256, 1, 1, 0), (2, 256, 15, 13, 256, 3, 2, 1), (3, 768, 7, 7, 512, 1, 1, 0)]
# 1x1 with <= 64 output and 256 k input channels: the 64 x 256 tile (wgrad1x1_64x256_kernel: half of its waves issue a dump piece)
WIDE_X_WGRAD = [(2, 256, 28, 28, 64, 1, 1, 0), (3, 512, 9, 11, 64, 1, 1, 0), (1, 256, 7, 5, 32, 1, 1, 0), (4, 256, 56, 56, 64, 1, 1, 0)]


@pytest.mark.parametrize("case", CASES + [(4, 64, 28, 28, 64, 3, 1, 1), (2, 64, 33, 31, 256, 1, 1, 0)] + HALO_WGRAD + SQUARE_WGRAD + WIDE_X_WGRAD)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_wgrad(case, dt):
    from iif_amd import ops
    n, cin, h, w, cout, r, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    x = torch.randn(n, cin, h, w, generator=g).to(dt).float()
    ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
    dy = torch.randn(n, cout, ho, wo, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(x, (cout, cin, r, r), dy, stride, pad)     # OIHW
    ref = ref.permute(0, 2, 3, 1).reshape(cout, r * r * cin)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    tol = 2e-5 if dt == torch.float32 else 1e-4      # fp32 accumulate of exact bf16 products
    for splits in (0, 1, 3):
        dw = ops.conv_wgrad(nhwc(x).to(dt).to(DEV), nhwc(dy).to(dt).to(DEV), r, r, stride, pad, workspace=ws,
                            splits=splits)
        assert (dw.cpu() - ref).abs().max().item() <= tol * ref.abs().max().item(), (case, dt, splits)
    # padded row pitch: pad columns stay untouched
    ldw = r * r * cin + 16
    out = torch.full((cout, ldw), 7.0, device=DEV)
    ops.conv_wgrad(nhwc(x).to(dt).to(DEV), nhwc(dy).to(dt).to(DEV), r, r, stride, pad, ldw=ldw, out=out, workspace=ws)
    assert (out[:, r * r * cin:] == 7.0).all()
    assert (out[:, :r * r * cin].cpu() - ref).abs().max().item() <= tol * ref.abs().max().item()


def test_wgrad_is_deterministic():
    from iif_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 28, 28, 64, generator=g).to(torch.bfloat16).to(DEV)
    dy = torch.randn(8, 28, 28, 64, generator=g).to(torch.bfloat16).to(DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    a = ops.conv_wgrad(x, dy, 3, 3, 1, 1, workspace=ws)
    b = ops.conv_wgrad(x, dy, 3, 3, 1, 1, workspace=ws)
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 64, 14, 14, 64, 3, 1, 1), (3, 128, 9, 9, 256, 1, 1, 0), (2, 64, 15, 13, 128, 3, 2, 1),
                                  (1, 32, 7, 7, 1000, 1, 1, 0), (5, 8, 6, 6, 24, 3, 1, 1)])
def test_conv_fused_bn_statistics(case):
    """The epilogue's per-tile (sum, sumsq) + iif_bn_finalize_stats == the standalone statistics pass
    over the stored bf16 output; the staged (coalesced) epilogue stores the same values as the oracle conv."""
    from iif_amd import ops
    n, cin, h, w, cout, r, stride, pad = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).to(dt)
    wt = (torch.randn(cout, cin, r, r, generator=g) / (cin * r * r) ** 0.5).to(dt)
    xd, wd = nhwc(x.float()).to(dt).to(DEV), krsc(wt.float(), 8).to(dt).to(DEV)
    ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
    m = n * ho * wo
    out = torch.empty(n, ho, wo, cout, dtype=dt, device=DEV)
    partial = torch.full((((m + 127) // 128) * 2 * cout,), float("nan"), device=DEV)
    nt = ops.conv_forward_bnstats(xd, wd, r, r, stride, pad, out, partial)
    assert nt in ((m + 127) // 128, (m + 255) // 256)        # one partial per pixel tile (128 or 256 rows)
    ref = F.conv2d(x.float(), wt.float(), None, stride, pad)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    plain = ops.conv_forward(xd, wd, r, r, stride, pad)
    assert torch.equal(plain, out)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV)
    s_fused, s_ref = torch.empty(4, cout, device=DEV), torch.empty(4, cout, device=DEV)
    rm1, rv1 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    rm2, rv2 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    ops.bn_finalize_stats(partial, nt, m, cout, gamma, beta, rm1, rv1, s_fused)
    # two-stage reduction path (many partial rows) gives the same statistics
    big = partial[:nt * 2 * cout].view(nt, 2 * cout).repeat(700 // nt + 1, 1).contiguous()
    s_two, s_one = torch.empty(4, cout, device=DEV), torch.empty(4, cout, device=DEV)
    ops.bn_finalize_stats(big, big.shape[0], m * (700 // nt + 1), cout, gamma, beta, None, None, s_two,
                          scratch=torch.empty(128 * cout, device=DEV))
    ops.bn_finalize_stats(big, big.shape[0], m * (700 // nt + 1), cout, gamma, beta, None, None, s_one)
    assert (s_two - s_one).abs().max().item() <= 1e-5 * max(1.0, s_one.abs().max().item())
    ops.bn_forward_stats(out.view(m, cout), gamma, beta, rm2, rv2, s_ref, ops.bn_workspace(m, cout, DEV))
    for a_, b_ in ((s_fused, s_ref), (rm1, rm2), (rv1, rv2)):
        assert (a_ - b_).abs().max().item() <= 1e-5 * max(1.0, b_.abs().max().item())


@pytest.mark.parametrize("case", [(2, 128, 14, 14, 32, 1), (2, 256, 15, 13, 32, 2), (1, 512, 7, 7, 32, 1), (2, 64, 9, 9, 8, 1),
                                  (2, 512, 8, 8, 8, 1)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_grouped_conv3x3(case, dt):
    """ResNeXt-style grouped 3x3 (width -> width, `groups` groups): forward, data gradient and weight
    gradient through the chunk-64 block-diagonal MFMA path against torch's grouped conv2d."""
    from iif_amd import ops
    n, width, h, w, groups, stride = case
    cg = width // groups
    ch = max(64, cg)
    g = torch.Generator().manual_seed(width + groups)
    x = torch.randn(n, width, h, w, generator=g).to(dt).float()
    wt = (torch.randn(width, cg, 3, 3, generator=g) / (9 * cg) ** 0.5).to(dt).float()
    ref = F.conv2d(x, wt, None, stride, 1, 1, groups)
    ldm = (9 * cg + 15) // 16 * 16
    master = torch.zeros(width, ldm)
    master[:, :9 * cg] = wt.permute(0, 2, 3, 1).reshape(width, 9 * cg)
    ldp = 9 * ch
    wp = torch.empty(width, ldp, dtype=dt, device=DEV)
    wpt = torch.empty(width, ldp, dtype=dt, device=DEV)
    ops.group_pack(master.to(DEV), width, cg, ch, 9, wp)
    ops.group_pack(master.to(DEV), width, cg, ch, 9, wpt, transposed=True)
    G = width // ch
    tol = 2e-5 if dt == torch.float32 else 2.0 ** -7
    xd = nhwc(x).to(dt).to(DEV)
    y = ops.conv_forward(xd, wp, 3, 3, stride, 1, groups=G)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
    if dt == torch.bfloat16:      # fused statistics on the grouped path
        m = y.shape[0] * y.shape[1] * y.shape[2]
        partial = torch.empty(((m + 127) // 128) * 2 * width, device=DEV)
        y2 = torch.empty_like(y)
        nt = ops.conv_forward_bnstats(xd, wp, 3, 3, stride, 1, y2, partial, groups=G)
        assert torch.equal(y2, y)
        s = partial[:nt * 2 * width].view(nt, 2, width).sum(0)
        yf = y.float().view(m, width)
        assert (s[0] - yf.sum(0)).abs().max().item() <= 1e-3 * max(1.0, yf.sum(0).abs().max().item())
        assert (s[1] - (yf * yf).sum(0)).abs().max().item() <= 1e-3 * (yf * yf).sum(0).abs().max().item()
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    dyd = nhwc(dy).to(dt).to(DEV)
    refdx = torch.nn.grad.conv2d_input(x.shape, wt, dy, stride, 1, 1, groups)
    dx = ops.conv_dgrad(dyd, wpt, 3, 3, stride, 1, (h, w), groups=G)
    assert (dx.float().cpu().permute(0, 3, 1, 2) - refdx).abs().max().item() <= tol * refdx.abs().max().item()
    refdw = torch.nn.grad.conv2d_weight(x, wt.shape, dy, stride, 1, 1, groups).permute(0, 2, 3, 1).reshape(width, 9 * cg)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    dwp = ops.conv_wgrad(xd, dyd, 3, 3, stride, 1, ldw=ldp, workspace=ws, groups=G)
    dwm = torch.zeros(width, ldm, device=DEV)
    ops.group_unpack_grad(dwp, width, cg, ch, 9, dwm)
    wtol = 2e-5 if dt == torch.float32 else 1e-4
    assert (dwm[:, :9 * cg].cpu() - refdw).abs().max().item() <= wtol * refdw.abs().max().item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dgrad_with_relu_masked_residual(dt):
    """iif_conv_igemm_masked_res: dx = dgrad(dy, w) + res * [bit] with the 1-bit ReLU decisions bn_apply wrote."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(31)
    n, h, cin, cout = 3, 14, 64, 32
    dy = torch.randn(n, h, h, cout, generator=g).to(dt)
    w = (torch.randn(cout, cin, generator=g) * 0.1).to(dt)
    res = torch.randn(n, h, h, cin, generator=g).to(dt)
    pre = torch.randn(n * h * h, cin, generator=g).to(dt)
    # bits through the product path: bn_apply with identity statistics
    stats = torch.zeros(4, cin)
    stats[2] = 1.0
    y = torch.empty(n * h * h, cin, dtype=dt, device=DEV)
    vec = 8 if dt == torch.bfloat16 else 4
    bits = torch.empty(n * h * h * cin // vec, dtype=torch.uint8, device=DEV)
    ops.bn_apply(pre.to(DEV), stats.to(DEV), y, relu=True, relu_bits=bits)
    wt = torch.zeros(cin, 32, dtype=dt, device=DEV)
    ops.weight_transpose(w.float().to(DEV), cout, cin, 1, wt)
    out = ops.conv_dgrad(dy.to(DEV), wt, 1, 1, 1, 0, (h, h), res=res.to(DEV), res_bits=bits)
    ref = dy.float().reshape(-1, cout) @ w.float() + res.float().reshape(-1, cin) * (pre.float() > 0)
    tol = 2e-5 if dt == torch.float32 else 2.0 ** -7
    assert (out.float().cpu().reshape(-1, cin) - ref).abs().max().item() <= tol * ref.abs().max().item()


HALO_CASES = [(4, 128, 28, 28, 128), (3, 256, 14, 14, 256), (7, 512, 7, 7, 512), (2, 128, 9, 11, 256), (1, 64, 5, 30, 136),
              (40, 32, 3, 3, 128)]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_kernel(case, monkeypatch):
    """The LDS halo-window 3x3 kernel (forced onto small grids): forward with fused BN partial sums, data gradient
    with residual, against torch conv2d; and bit-identical to the tap-by-tap kernel's stored output."""
    import subprocess, sys, os, json
    # the dispatch reads its environment once per process: run the comparison in a child with the halo path forced
    code = r"""
import sys, json, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from iif_amd import ops
n, cin, h, w, cout = %r
g = torch.Generator().manual_seed(cin + cout + h)
dt = torch.bfloat16
x = torch.randn(n, cin, h, w, generator=g).to(dt)
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dt)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def krsc(w_):
    co, ci, r, s = w_.shape
    return w_.permute(0, 2, 3, 1).reshape(co, r * s * ci).contiguous()
dev = 'cuda:0'
xd, wd = nhwc(x).to(dev), krsc(wt).to(dev)
ref = F.conv2d(x.float(), wt.float(), None, 1, 1)
m = n * h * w
out = torch.empty(n, h, w, cout, dtype=dt, device=dev)
partial = torch.full((((m + 127) // 128) * 2 * cout,), float('nan'), device=dev)
nt = ops.conv_forward_bnstats(xd, wd, 3, 3, 1, 1, out, partial)
got = out.float().cpu().permute(0, 3, 1, 2)
e_fwd = (got - ref).abs().max().item() / ref.abs().max().item()
ps = partial[:nt * 2 * cout].view(nt, 2, cout).sum(0).cpu()
e_sum = (ps[0] - out.float().cpu().reshape(-1, cout).sum(0)).abs().max().item() / max(1.0, ps[0].abs().max().item())
e_sq = (ps[1] - out.float().cpu().reshape(-1, cout).square().sum(0)).abs().max().item() / max(1.0, ps[1].abs().max().item())
dy = torch.randn(n, cout, h, w, generator=g).to(dt)
res = torch.randn(n, cin, h, w, generator=g).to(dt)
refdx = torch.nn.grad.conv2d_input(x.shape, wt.float(), dy.float(), 1, 1) + res.float()
wtt = krsc(wt.permute(1, 0, 2, 3).contiguous())
dx = ops.conv_dgrad(nhwc(dy).to(dev), wtt.to(dev), 3, 3, 1, 1, (h, w), res=nhwc(res).to(dev))
e_dx = (dx.float().cpu().permute(0, 3, 1, 2) - refdx).abs().max().item() / refdx.abs().max().item()
print(json.dumps({"nt": nt, "m": m, "e_fwd": e_fwd, "e_sum": e_sum, "e_sq": e_sq, "e_dx": e_dx,
                  "digest": float(out.float().double().sum().item())}))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), case)
    outs = {}
    for mode, env in (("halo", {"IIF_CONV_HALO_FORCE": "1"}), ("taps", {"IIF_CONV_NO_HALO": "1"})):
        e = dict(os.environ, **env)
        e.pop("IIF_CONV_NO_HALO" if mode == "halo" else "IIF_CONV_HALO_FORCE", None)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    hres, tres = outs["halo"], outs["taps"]
    n, cin, h, w, cout = case
    span = (256 + w - 1) // w + 1 + 2 * (256 // (h * w) + 1)
    if cin % 32 == 0 and cout >= 128 and cout % 8 == 0 and (span + 2) * (w + 2) <= 512:
        assert hres["nt"] in ((hres["m"] + 255) // 256, (hres["m"] + 127) // 128)   # halo kernel: 256- or 128-pixel tiles
    else:
        assert hres["nt"] == tres["nt"]                             # window too large / shape not covered: tap kernel
    assert hres["e_fwd"] <= 2.0 ** -7 and hres["e_dx"] <= 2.0 ** -7
    assert hres["e_sum"] <= 1e-5 and hres["e_sq"] <= 1e-5
    assert tres["e_fwd"] <= 2.0 ** -7
    # same K order per output element is not guaranteed (chunk-major vs tap-major): compare digests loosely
    assert abs(hres["digest"] - tres["digest"]) <= 1e-3 * max(1.0, abs(tres["digest"]))


# (the fragment kernel exists for 64-channel output tiles only since round 4: Cout and Cin = 64 mod 128, both directions)
V2_CASES = [(2, 64, 56, 56, 64), (1, 64, 5, 30, 192), (5, 64, 14, 14, 320), (6, 192, 9, 11, 64), (40, 64, 5, 5, 64), (3, 192, 17, 13, 192)]


def _pack_frag(w2d, rows, taps, k):
    """bf16 rows [rows][taps * k] -> fragment layout through the C entry point."""
    from iif_amd import ops
    tab, blocks = ops.pack_table([(0, 0, rows, taps, k, w2d.shape[1])], DEV)
    out = torch.empty(rows * taps * k, dtype=w2d.dtype, device=DEV)
    return ops.pack_fragments(w2d, tab, 1, blocks, out)


@pytest.mark.parametrize("case", V2_CASES, ids=["%dx%dx%d_%d_to_%d" % (c[0], c[2], c[3], c[1], c[4]) for c in V2_CASES])
def test_conv3x3_fragment_kernel(case, conv_env):
    """Generation-2 3x3 kernel (weights as MFMA fragments straight from L2 into registers, halo window in LDS, one barrier
    per 32-channel chunk, 128 x 64 wave tiles): the fragment layout itself, forward with fused BN partial sums, data
    gradient with residual, data gradient with the upstream BN-backward sums — against torch conv2d, and the stored
    tensors BIT-IDENTICAL to the round-1 halo kernel where that one applies (same accumulation order)."""
    import torch.nn.functional as F
    from iif_amd import ops
    n, cin, h, w, cout = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g).to(dt)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dt)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()                                    # noqa: E731
    krsc = lambda w_: w_.permute(0, 2, 3, 1).reshape(w_.shape[0], -1).contiguous()          # noqa: E731
    xd, wd = nhwc(x).to(DEV), krsc(wt).to(DEV)
    conv_env(IIF_CONV_V2_FORCE="1", IIF_CONV_NO_REGW="1")   # small grids are refused otherwise (v2_geometry_ok); this test: fragment vs window kernels
    assert ops.conv3x3_frag_ok(n, h, w, cin, cout, dt) and ops.conv3x3_frag_ok(n, h, w, cout, cin, dt)
    wf = _pack_frag(wd, cout, 9, cin)
    # the layout: fragment (row tile, tap, chunk), lane (row & 15, 8 channels of (lane >> 4))
    wv = wd.cpu().view(cout // 16, 16, 9, cin // 32, 4, 8)
    expect = wv.permute(0, 2, 3, 4, 1, 5).contiguous().view(-1)                             # [nt][tap][kc][fc][fr][8]
    assert torch.equal(wf.cpu(), expect)
    ref = F.conv2d(x.float(), wt.float(), None, 1, 1)
    m = n * h * w
    res = {}
    dy = torch.randn(n, cout, h, w, generator=g).to(dt)
    rs = torch.randn(n, cin, h, w, generator=g).to(dt)
    upx = torch.randn(m, cin, generator=g).to(dt)
    bits = torch.randint(0, 256, (m * cin // 8,), dtype=torch.uint8, generator=g).to(DEV)
    stats = torch.zeros(4, cin)
    stats[0] = torch.randn(cin, generator=g) * 0.1
    stats[1] = torch.rand(cin, generator=g) + 0.5
    wtt = krsc(wt.permute(1, 0, 2, 3).contiguous()).to(DEV)
    wtf = _pack_frag(wtt, cin, 9, cout)
    for mode in ("v2", "old"):
        conv_env(IIF_CONV_NO_V2=None if mode == "v2" else "1", IIF_CONV_HALO_FORCE="1", IIF_CONV_V2_FORCE="1", IIF_CONV_NO_REGW="1")
        out = torch.full((n, h, w, cout), float("nan"), dtype=dt, device=DEV)
        partial = torch.full((((m + 127) // 128) * 2 * cout,), float("nan"), device=DEV)
        nt = ops.conv_forward_bnstats(xd, wd, 3, 3, 1, 1, out, partial, w_frag=wf)
        dx = ops.conv_dgrad(nhwc(dy).to(DEV), wtt, 3, 3, 1, 1, (h, w), res=nhwc(rs).to(DEV), w_frag=wtf)
        dx2 = torch.full((n, h, w, cin), float("nan"), dtype=dt, device=DEV)
        partial2 = torch.full(((m + 127) // 128 + 8, 2, cin), float("nan"), device=DEV)
        nt2 = ops.conv_dgrad_bnbwd(nhwc(dy).to(DEV), wtt, 3, 3, 1, 1, (h, w), dx2, upx.view(n, h, w, cin).to(DEV), bits,
                                   stats.to(DEV), partial2.view(-1), w_frag=wtf)
        res[mode] = (out, partial[:nt * 2 * cout].view(nt, 2, cout).sum(0).cpu(), nt, dx, dx2, partial2[:nt2].sum(0).cpu(), nt2)
    out, ps, nt, dx, dx2, ps2, nt2 = res["v2"]
    assert nt == (m + 255) // 256 and nt2 == (m + 255) // 256                     # the fragment kernel ran: 256-pixel tiles
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    st = out.float().cpu().reshape(-1, cout)
    assert not torch.isnan(ps).any()
    assert (ps[0] - st.sum(0)).abs().max().item() <= 1e-5 * max(1.0, st.sum(0).abs().max().item())
    assert (ps[1] - st.square().sum(0)).abs().max().item() <= 1e-5 * max(1.0, st.square().sum(0).abs().max().item())
    refdx = torch.nn.grad.conv2d_input(x.shape, wt.float(), dy.float(), 1, 1) + rs.float()
    assert (dx.float().cpu().permute(0, 3, 1, 2) - refdx).abs().max().item() <= 2.0 ** -7 * refdx.abs().max().item()
    mask = ((bits.cpu().view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, cin).float()
    gq = dx2.float().cpu().view(m, cin) * mask
    s1, s2 = gq.sum(0), (gq * ((upx.float() - stats[0]) * stats[1])).sum(0)
    assert not torch.isnan(ps2).any()
    assert (ps2[0] - s1).abs().max().item() <= 2e-6 * max(1.0, s1.abs().max().item()) * 8
    assert (ps2[1] - s2).abs().max().item() <= 2e-6 * max(1.0, s2.abs().max().item()) * 8
    # against the kernels it replaces: the halo kernel accumulates in the same order (chunk outer, tap inner) -> identical
    # bits; the tap kernel (cout < 128 or a window beyond its 512 halo rows) is tap-major -> equal within the bf16 rounding
    o_out, _, o_nt, o_dx, o_dx2 = res["old"][:5]
    span = (256 + w - 1) // w + 1 + 2 * (256 // (h * w) + 1)
    span1 = (128 + w - 1) // w + 1 + 2 * (128 // (h * w) + 1)
    halo_ok = (span + 2) * (w + 2) <= 512 and (span1 + 2) * (w + 2) <= 304            # use_halo's window conditions
    halo_fwd = cout >= 128 and halo_ok
    halo_bwd = cin >= 128 and halo_ok
    if halo_fwd:
        assert torch.equal(out, o_out)
    else:
        assert (out.float() - o_out.float()).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item()
    if halo_bwd:
        assert torch.equal(dx, o_dx) and torch.equal(dx2, o_dx2)
    else:
        assert (dx.float() - o_dx.float()).abs().max().item() <= 2.0 ** -6 * refdx.abs().max().item()


@pytest.mark.parametrize("case", [(4, 64, 14, 14, 128, 1, 1, True, True), (4, 128, 14, 14, 64, 3, 1, False, False),
                                  (3, 64, 16, 16, 128, 3, 2, False, True), (3, 256, 16, 16, 512, 1, 2, True, False),
                                  (8, 256, 28, 28, 128, 3, 1, False, True), (2, 64, 15, 13, 128, 3, 2, False, True),
                                  # narrow -> wide 1x1 with M % 64 == 0: the weights-in-registers kernel (conv_regw.hip, EPI)
                                  (4, 256, 16, 16, 64, 1, 1, True, True), (8, 512, 16, 16, 128, 1, 1, False, True),
                                  (16, 1024, 16, 16, 256, 1, 1, True, False), (64, 2048, 8, 8, 512, 1, 1, True, True),
                                  (40, 512, 28, 28, 128, 1, 1, True, True)])
def test_dgrad_emits_upstream_bn_backward_sums(case):
    """iif_conv_igemm_dgrad_bnbwd: the data gradient (stride 1 and the 4 parity classes of stride 2, with and without
    residual) plus per-tile (sum g, sum g*xhat) of the upstream unit, g = stored gradient gated by its ReLU bits."""
    from iif_amd import ops
    n, cin, h, w, cout, k, stride, use_res, use_bits = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(cin + cout + k)
    pad = k // 2
    ho, wo = ops.conv_out_hw(h, w, k, k, stride, pad)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dt)
    dy = torch.randn(n, cout, ho, wo, generator=g).to(dt)
    res = torch.randn(n, cin, h, w, generator=g).to(dt)
    upx = torch.randn(n, cin, h, w, generator=g).to(dt)
    pre = torch.randn(n, cin, h, w, generator=g).to(dt)
    stats = torch.zeros(4, cin)
    stats[0] = torch.randn(cin, generator=g) * 0.1
    stats[1] = torch.rand(cin, generator=g) + 0.5
    st_id = torch.zeros(4, cin)
    st_id[2] = 1.0
    y = torch.empty(n * h * w, cin, dtype=dt, device=DEV)
    bits = torch.empty(n * h * w * cin // 8, dtype=torch.uint8, device=DEV)
    ops.bn_apply(nhwc(pre.float()).to(dt).reshape(-1, cin).to(DEV), st_id.to(DEV), y, relu=True, relu_bits=bits)
    wtt = krsc(wt.float().permute(1, 0, 2, 3).contiguous(), 8).to(dt)
    out = torch.empty(n, h, w, cin, dtype=dt, device=DEV)
    partial = torch.full(((n * h * w + 127) // 128 + 8, 2, cin), float("nan"), device=DEV)
    nt = ops.conv_dgrad_bnbwd(nhwc(dy.float()).to(dt).to(DEV), wtt.to(DEV), k, k, stride, pad, (h, w), out,
                              nhwc(upx.float()).to(dt).to(DEV), bits if use_bits else None, stats.to(DEV), partial.view(-1),
                              res=nhwc(res.float()).to(dt).to(DEV) if use_res else None)
    # float64 textbook gradient of the same bf16 operands (the register-weight epilogue variants included: the cases above)
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), stride, pad)
    if use_res:
        ref = ref + res.double()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert (got.double() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    gq = got * ((pre.float() > 0) if use_bits else 1.0)             # the sums are over the STORED values
    xhat = (upx.float() - stats[0][None, :, None, None]) * stats[1][None, :, None, None]
    s1, s2 = gq.sum(dim=(0, 2, 3)), (gq * xhat).sum(dim=(0, 2, 3))
    ps = partial[:nt].sum(0).cpu()
    assert not torch.isnan(ps).any()
    assert (ps[0] - s1).abs().max().item() <= 2e-6 * max(1.0, s1.abs().max().item())
    assert (ps[1] - s2).abs().max().item() <= 2e-6 * max(1.0, s2.abs().max().item())


@pytest.mark.parametrize("case", [(4, 4, 64, 14, 14, True), (3, 2, 64, 9, 11, False), (2, 8, 64, 28, 28, True)],
                         ids=lambda c: "n%d_g%dx%d_%dx%d" % (c[0], c[1], c[2], c[3], c[4]))
def test_grouped_dgrad_emits_upstream_bn_backward_sums(case):
    """The same epilogue on a grouped 3x3 / stride-1 data gradient (ResNeXt's conv2, block-diagonal chunks of 64 channels,
    blockIdx.y = chunk): the stored gradient is bit-identical to the plain grouped data gradient, the partial rows hold
    (sum g, sum g*xhat) of the chunk-offset channels of the upstream unit."""
    from iif_amd import ops
    n, G, cg, h, w, use_bits = case
    C = G * cg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(G * 100 + h)
    k, pad = 3, 1
    ldw = k * k * cg
    wtt = (torch.randn(C, ldw, generator=g) / ldw ** 0.5).to(dt).to(DEV)        # per chunk: [cg rows, 9 * cg] transposed weights
    dy = torch.randn(n, h, w, C, generator=g).to(dt).to(DEV)
    upx = torch.randn(n, h, w, C, generator=g).to(dt).to(DEV)
    bits = torch.randint(0, 256, (n * h * w * C // 8,), dtype=torch.uint8, generator=g).to(DEV)
    stats = torch.zeros(4, C)
    stats[0] = torch.randn(C, generator=g) * 0.1
    stats[1] = torch.rand(C, generator=g) + 0.5
    plain = ops.conv_dgrad(dy, wtt, k, k, 1, pad, (h, w), groups=G)
    out = torch.full((n, h, w, C), float("nan"), dtype=dt, device=DEV)
    m = n * h * w
    partial = torch.full(((m + 127) // 128 + 8, 2, C), float("nan"), device=DEV)
    nt = ops.conv_dgrad_bnbwd(dy, wtt, k, k, 1, pad, (h, w), out, upx, bits if use_bits else None, stats.to(DEV), partial.view(-1),
                              groups=G)
    assert torch.equal(out, plain)
    gq = plain.float().cpu().view(m, C)
    if use_bits:
        gq = gq * ((bits.cpu().view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, C)
    xhat = (upx.float().cpu().view(m, C) - stats[0]) * stats[1]
    s1, s2 = gq.sum(0), (gq * xhat).sum(0)
    ps = partial[:nt].sum(0).cpu()
    assert nt == (m + 127) // 128 and not torch.isnan(ps).any()
    assert (ps[0] - s1).abs().max().item() <= 2e-6 * max(1.0, s1.abs().max().item()) * 8
    assert (ps[1] - s2).abs().max().item() <= 2e-6 * max(1.0, s2.abs().max().item()) * 8


@pytest.mark.parametrize("n,h,w", [(8, 32, 32), (3, 112, 112), (2, 17, 23), (5, 1, 3), (1, 40, 120)])
def test_stem_s2d_weight_gradient_all_taps_per_block(n, h, w):
    """4x4 / stride 1 / pad 2 (top-left) over the 16-channel space-to-depth image with output size = input size:
    conv4x4_s2d_wgrad_kernel (W <= 112; (1, 40, 120) exceeds its ring and takes the tap-per-tile kernel) against an
    fp64 evaluation of the same bf16 operands."""
    import torch.nn.functional as F
    from iif_amd import ops
    g = torch.Generator().manual_seed(n * 100 + w)
    x = torch.randn(n, h, w, 16, generator=g).bfloat16()
    dy = torch.randn(n, h, w, 64, generator=g).bfloat16()
    ws = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
    xp = F.pad(x.double().permute(0, 3, 1, 2), (2, 1, 2, 1))
    wref = torch.zeros(64, 16, 4, 4, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(xp, wref)
    assert tuple(out.shape[-2:]) == (h, w)
    (gw,) = torch.autograd.grad(out, wref, dy.double().permute(0, 3, 1, 2))
    ref = gw.permute(0, 2, 3, 1).reshape(64, 256)
    for splits in (0, 1, 5):
        dw = ops.conv_wgrad(x.to(DEV), dy.to(DEV), 4, 4, 1, 2, workspace=ws, splits=splits)
        assert (dw.double().cpu() - ref).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1e-6), (n, h, w, splits)


@pytest.mark.parametrize("n,h,w", [(2, 112, 112), (3, 16, 16), (5, 96, 96), (1, 128, 128), (2, 80, 80), (40, 112, 112), (2, 17, 23), (1, 256, 256)])
def test_stem_s2d_forward_window_kernel(n, h, w):
    """The stem in its space-to-depth form (4x4 / stride 1 / pad 2 top-left, 16 -> 64 channels, output size = input size;
    classification/resnet_pytorch.py:196-197 after the 2x2 sub-pixel split): stem4x4_kernel (window of a 128-pixel tile in LDS,
    weights in registers, one partial row per block) against an fp64 evaluation of the same bf16 operands, and its batch-norm
    partial sums against the sums of the values it stored.  (2, 17, 23) and (1, 256, 256) do not fit it and run the general
    kernel through the same entry point."""
    import torch.nn.functional as F
    from iif_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + w)
    x = torch.randn(n, h, w, 16, generator=g).bfloat16()
    wt = (torch.randn(64, 256, generator=g) * 0.1).bfloat16()
    xp = F.pad(x.double().permute(0, 3, 1, 2), (2, 1, 2, 1))
    ref = F.conv2d(xp, wt.double().view(64, 4, 4, 16).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    out = torch.full((n, h, w, 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    partial = torch.full((max(4096, (n * h * w + 127) // 128 + 8), 2, 64), float("nan"), device=DEV)
    nt = ops.conv_forward_bnstats(x.to(DEV), wt.to(DEV), 4, 4, 1, 2, out, partial)
    got = out.double().cpu()
    assert not torch.isnan(got).any()
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    ps = partial[:nt].double().sum(0).cpu()
    flat = got.view(-1, 64)
    s1, s2 = flat.sum(0), (flat * flat).sum(0)
    assert nt > 0 and not torch.isnan(ps).any()
    assert (ps[0] - s1).abs().max().item() <= 1e-5 * max(1.0, flat.abs().sum(0).max().item())
    assert (ps[1] - s2).abs().max().item() <= 1e-5 * s2.max().item()
    # without statistics: the same stored values
    out2 = ops.conv_forward(x.to(DEV), wt.to(DEV), 4, 4, 1, 2, out_hw=(h, w))
    assert torch.equal(out2, out)


@pytest.mark.parametrize("case", [(16, 16, 128, 512), (16, 16, 256, 1024), (64, 8, 512, 2048), (16, 16, 512, 128), (16, 16, 1024, 256),
                                  (37, 28, 128, 512), (3, 16, 128, 512),
                                  (16, 16, 128, 256), (16, 16, 256, 512), (32, 8, 512, 1024), (64, 8, 1024, 2048)],   # ResNeXt conv3
                         ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
def test_forward_1x1_weights_in_registers_equals_the_tile_kernel(case, conv_env):
    """conv_regw.hip (persistent blocks, weight fragments in registers, one partial row per tile sequence) against the tile
    kernels through the same entry point: stored values bit-identical (same K order), batch-norm sums equal to 1e-6;
    (3, 16, ..) has too few rows for it and runs the tile kernel on both sides."""
    from iif_amd import ops
    n, hw, k, c = case
    g = torch.Generator().manual_seed(k + c + n)
    x = torch.randn(n, hw, hw, k, generator=g).bfloat16().to(DEV)
    wt = (torch.randn(c, k, generator=g) / k ** 0.5).bfloat16().to(DEV)
    m = n * hw * hw
    res = []
    for off in (False, True):
        conv_env(IIF_CONV_NO_REGW="1" if off else None)
        out = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        partial = torch.full(((m + 127) // 128 + 8, 2, c), float("nan"), device=DEV)
        nt = ops.conv_forward_bnstats(x, wt, 1, 1, 1, 0, out, partial.view(-1))
        plain = ops.conv_forward(x, wt, 1, 1, 1, 0)
        assert torch.equal(plain, out)
        res.append((out, partial[:nt].double().sum(0).cpu(), nt))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][2] <= (m + 127) // 128
    flat = res[0][0].double().cpu().view(m, c)
    ref = torch.stack([flat.sum(0), (flat * flat).sum(0)])
    for _, ps, _ in res:
        assert not torch.isnan(ps).any()
        assert (ps - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("n,hw", [(4, 16), (5, 24), (16, 56), (3, 56), (64, 8), (2, 20)])
def test_conv3x3_64_channels_weights_in_registers(n, hw, conv_env):
    """conv_regw.hip, 3x3 / stride 1 / pad 1 over 64 -> 64 channels on 8 x 8 pixel tiles (weights in registers, window through a
    three-slot LDS ring): forward with batch-norm sums and the data gradient with the upstream BN-backward sums, against fp64
    references of the same bf16 operands and against the window / tile kernels through the same entry points.  (2, 20): the
    image side is not a multiple of 8, the other kernels run on both sides."""
    import torch.nn.functional as F
    from iif_amd import ops
    c = 64
    g = torch.Generator().manual_seed(n * 100 + hw)
    x = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(DEV)
    w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).bfloat16()
    wk = w.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous().to(DEV)                   # [cout][r][s][cin]
    wt = w.permute(1, 2, 3, 0).reshape(c, 9 * c).contiguous().to(DEV)                   # [cin][r][s][cout]: the data gradient's operand
    upx = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(DEV)
    bits = torch.randint(0, 256, (n * hw * hw * c // 8,), dtype=torch.uint8, generator=g).to(DEV)
    stats = torch.zeros(4, c)
    stats[0] = torch.randn(c, generator=g) * 0.1
    stats[1] = torch.rand(c, generator=g) + 0.5
    m = n * hw * hw
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    refd = F.conv_transpose2d(x.double().cpu().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    got = {}
    for off in (False, True):
        conv_env(IIF_CONV_NO_REGW="1" if off else None)
        out = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        partial = torch.full(((m + 127) // 128 + 8, 2, c), float("nan"), device=DEV)
        nt = ops.conv_forward_bnstats(x, wk, 3, 3, 1, 1, out, partial.view(-1))
        dx = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=DEV)
        partial2 = torch.full(((m + 127) // 128 + 8, 2, c), float("nan"), device=DEV)
        nt2 = ops.conv_dgrad_bnbwd(x, wt, 3, 3, 1, 1, (hw, hw), dx, upx, bits, stats.to(DEV), partial2.view(-1))
        assert nt <= (m + 127) // 128 and nt2 <= (m + 127) // 128
        assert (out.double().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
        assert (dx.double().cpu() - refd).abs().max().item() <= 2.0 ** -7 * refd.abs().max().item()
        flat = out.double().cpu().view(m, c)
        ps = partial[:nt].double().sum(0).cpu()
        assert (ps[0] - flat.sum(0)).abs().max().item() <= 1e-6 * flat.abs().sum(0).max().item()
        assert (ps[1] - (flat * flat).sum(0)).abs().max().item() <= 1e-6 * (flat * flat).sum(0).max().item()
        gq = dx.double().cpu().view(m, c) * ((bits.cpu().view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, c)
        xhat = (upx.double().cpu().view(m, c) - stats[0].double()) * stats[1].double()
        ps2 = partial2[:nt2].double().sum(0).cpu()
        assert (ps2[0] - gq.sum(0)).abs().max().item() <= 1e-5 * gq.abs().sum(0).max().item()
        assert (ps2[1] - (gq * xhat).sum(0)).abs().max().item() <= 1e-5 * (gq * xhat).abs().sum(0).max().item()
        got[off] = (out, dx)
    # the two routes order the K loop differently (chunk-major here, tap-major in the tile kernels): equal to a bf16 rounding
    assert (got[False][0].float() - got[True][0].float()).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item()
    assert (got[False][1].float() - got[True][1].float()).abs().max().item() <= 2.0 ** -6 * refd.abs().max().item()


# ------------------------------------------------------------------------------------------- streaming 1x1 kernel
STREAM_CASES = [  # N, Cin, H, W, Cout : every instantiation, ragged last tile, fewer tiles than blocks, many tiles per block
    (3, 64, 20, 23, 256), (2, 256, 17, 19, 64), (2, 256, 24, 24, 128), (1, 64, 9, 7, 64), (2, 32, 40, 40, 256),
    (5, 128, 12, 12, 64), (40, 64, 56, 56, 256), (37, 256, 56, 56, 64),
    # N slices of a tile sequence on neighbouring blocks (S = 2, 4, 8) of the 256-channel resident plan; shapes without a plan
    # (K = 128, K = 512: removed in round 4) run the tile kernel on both sides of the comparison
    (3, 128, 28, 28, 512), (3, 512, 28, 28, 128), (2, 128, 13, 11, 256), (5, 256, 14, 14, 1024), (2, 256, 28, 28, 512),
    (2, 512, 28, 28, 256), (9, 128, 28, 28, 512), (1, 512, 5, 5, 128), (2, 256, 56, 56, 128), (2, 128, 56, 56, 256),
]


@pytest.fixture
def conv_env(monkeypatch):
    """Set / unset convolution switches AND make the library read them again (it caches them when it is loaded)."""
    from iif_amd import _lib

    def set_(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        _lib.check(_lib.lib().iif_conv_reload_env(), "iif_conv_reload_env")
    yield set_
    monkeypatch.undo()
    _lib.lib().iif_conv_reload_env()


@pytest.mark.parametrize("case", STREAM_CASES, ids=["%dx%dx%d_%d_to_%d" % (c[0], c[2], c[3], c[1], c[4]) for c in STREAM_CASES])
def test_stream1x1_forward_stats_and_dgrad_epilogues(case, conv_env):
    """gemm1x1_stream_kernel (persistent, an N slice of the weights resident in LDS, compute / store wave groups, epilogue
    operands fetched a tile ahead) forced onto small grids: forward + fused BN statistics, data gradient with a ReLU-masked
    residual, data gradient with the upstream BN-backward sums, and the conv1 form with BOTH (masked residual + upstream
    sums) — each against torch and BIT-IDENTICAL in the stored tensor to the tile kernels it replaces."""
    from iif_amd import ops
    n, cin, h, w, cout = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(n * 1000 + cin + cout)
    m = n * h * w
    x = torch.randn(m, cin, generator=g).to(dt)
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dt)
    xd = x.view(n, h, w, cin).to(DEV)
    wd = wt.to(DEV)
    ref = x.float() @ wt.float().t()

    def run(force):
        if force:
            conv_env(IIF_CONV_STREAM1X1_FORCE="1", IIF_CONV_NO_STREAM1X1=None, IIF_CONV_NO_REGW="1")    # (this test: streaming against tile kernel)
        else:
            conv_env(IIF_CONV_STREAM1X1_FORCE=None, IIF_CONV_NO_STREAM1X1="1", IIF_CONV_NO_REGW="1")
        out = torch.full((n, h, w, cout), float("nan"), dtype=dt, device=DEV)
        partial = torch.full((((m + 127) // 128) * 2 * cout,), float("nan"), device=DEV)
        nt = ops.conv_forward_bnstats(xd, wd, 1, 1, 1, 0, out, partial)
        return out, partial, nt
    out_s, part_s, nt_s = run(True)
    out_t, part_t, nt_t = run(False)
    assert 0 < nt_s <= (m + 127) // 128               # one partial row per tile SEQUENCE of the persistent grid (round 5; per tile before)
    assert (out_s.float().cpu().view(m, cout) - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    assert torch.equal(out_s, out_t)                                  # same fp32 accumulation order per element
    sums_s = part_s[:nt_s * 2 * cout].view(nt_s, 2, cout).sum(0)
    stored = out_s.float().view(m, cout)
    assert not torch.isnan(sums_s).any()
    assert (sums_s[0] - stored.sum(0)).abs().max().item() <= 2e-6 * max(1.0, stored.sum(0).abs().max().item()) * 8
    assert (sums_s[1] - (stored * stored).sum(0)).abs().max().item() <= 2e-5 * (stored * stored).sum(0).abs().max().item()
    if nt_t == nt_s:
        assert (part_s[:nt_s * 2 * cout] - part_t[:nt_s * 2 * cout]).abs().max().item() <= 1e-4 * part_t[:nt_s * 2 * cout].abs().max().item()

    # data gradient dX[m, cin] = dY[m, cout] @ W, with the masked residual, then with the upstream BN-backward sums
    dy = torch.randn(m, cout, generator=g).to(dt)
    res = torch.randn(m, cin, generator=g).to(dt)
    pre = torch.randn(m, cin, generator=g).to(dt)
    upx = torch.randn(m, cin, generator=g).to(dt)
    st_id = torch.zeros(4, cin); st_id[2] = 1.0
    y = torch.empty(m, cin, dtype=dt, device=DEV)
    bits = torch.empty(m * cin // 8, dtype=torch.uint8, device=DEV)
    ops.bn_apply(pre.to(DEV), st_id.to(DEV), y, relu=True, relu_bits=bits)
    wtt = torch.zeros(cin, (cout + 15) // 16 * 16, dtype=dt, device=DEV)
    ops.weight_transpose(wt.float().to(DEV), cout, cin, 1, wtt)
    dyd = dy.view(n, h, w, cout).to(DEV)
    refdx = dy.float() @ wt.float() + res.float() * (pre.float() > 0)
    stats = torch.zeros(4, cin)
    stats[0] = torch.randn(cin, generator=g) * 0.1
    stats[1] = torch.rand(cin, generator=g) + 0.5
    got = {}
    bits2 = torch.randint(0, 256, (m * cin // 8,), dtype=torch.uint8, generator=g).to(DEV)
    for force in (True, False):
        if force:
            conv_env(IIF_CONV_STREAM1X1_FORCE="1", IIF_CONV_NO_STREAM1X1=None, IIF_CONV_NO_REGW="1")    # (this test: streaming against tile kernel)
        else:
            conv_env(IIF_CONV_STREAM1X1_FORCE=None, IIF_CONV_NO_STREAM1X1="1", IIF_CONV_NO_REGW="1")
        dx = ops.conv_dgrad(dyd, wtt, 1, 1, 1, 0, (h, w), res=res.view(n, h, w, cin).to(DEV), res_bits=bits)
        out2 = torch.full((n, h, w, cin), float("nan"), dtype=dt, device=DEV)
        partial = torch.full(((m + 127) // 128 + 8, 2, cin), float("nan"), device=DEV)
        nt = ops.conv_dgrad_bnbwd(dyd, wtt, 1, 1, 1, 0, (h, w), out2, upx.view(n, h, w, cin).to(DEV), bits, stats.to(DEV),
                                  partial.view(-1))
        # the conv1 form: masked residual (block-output gradient) AND the upstream sums on one launch
        out3 = torch.full((n, h, w, cin), float("nan"), dtype=dt, device=DEV)
        partial3 = torch.full(((m + 127) // 128 + 8, 2, cin), float("nan"), device=DEV)
        nt3 = ops.conv_dgrad_bnbwd(dyd, wtt, 1, 1, 1, 0, (h, w), out3, upx.view(n, h, w, cin).to(DEV), bits2, stats.to(DEV),
                                   partial3.view(-1), res=res.view(n, h, w, cin).to(DEV), res_bits=bits)
        got[force] = (dx, out2, partial[:nt].sum(0).cpu(), nt, out3, partial3[:nt3].sum(0).cpu(), nt3, bits2.cpu())
    dx_s, out2_s, ps, nt = got[True][:4]
    out3_s, ps3, nt3, bits2 = got[True][4:]
    assert torch.equal(out3_s, got[False][4]) and torch.equal(out3_s, dx_s)           # the sums do not touch the stored tensor
    assert 0 < nt3 <= (m + 127) // 128 and not torch.isnan(ps3).any()
    mask2 = ((bits2.view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, cin).float()
    g3 = out3_s.float().cpu().view(m, cin) * mask2
    t1, t2 = g3.sum(0), (g3 * ((upx.float() - stats[0]) * stats[1])).sum(0)
    assert (ps3[0] - t1).abs().max().item() <= 2e-6 * max(1.0, t1.abs().max().item()) * 8
    assert (ps3[1] - t2).abs().max().item() <= 2e-6 * max(1.0, t2.abs().max().item()) * 8
    assert (ps3 - got[False][5]).abs().max().item() <= 1e-4 * max(1.0, got[False][5].abs().max().item())
    assert (dx_s.float().cpu().view(m, cin) - refdx).abs().max().item() <= 2.0 ** -7 * refdx.abs().max().item()
    assert torch.equal(dx_s, got[False][0]) and torch.equal(out2_s, got[False][1])
    gq = out2_s.float().cpu().view(m, cin) * (pre.float() > 0)
    xhat = (upx.float() - stats[0]) * stats[1]
    s1, s2 = gq.sum(0), (gq * xhat).sum(0)
    assert not torch.isnan(ps).any() and 0 < nt <= (m + 127) // 128
    assert (ps[0] - s1).abs().max().item() <= 2e-6 * max(1.0, s1.abs().max().item()) * 8
    assert (ps[1] - s2).abs().max().item() <= 2e-6 * max(1.0, s2.abs().max().item()) * 8


# ------------------------------------------------------------------- BN backward through the expanding 1x1 layer, by algebra
@pytest.mark.parametrize("case", [(2, 14, 64, 256), (3, 9, 128, 512), (1, 20, 64, 256), (2, 7, 256, 1024)],
                         ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
@pytest.mark.parametrize("from_p", [True, False], ids=["sums_from_P", "sums_from_producer"])
def test_bn3_backward_by_algebra(case, from_p):
    """conv3 -> bn3 backward without re-reading conv3's output (csrc/bn3_algebra.hip): the weight gradient, the BN
    parameter gradients and the data gradient from P = g~^T a2, Gram = a2^T a2, colsum(a2) and the stacked-weights GEMM over
    [g~ | a2] — against the textbook BN + convolution backward evaluated in float64 on the same bf16 operands."""
    from iif_amd import ops
    n, hw, c, C = case
    m = n * hw * hw
    g = torch.Generator().manual_seed(c + C + hw)
    dt = torch.bfloat16
    a2 = torch.relu(torch.randn(m, c, generator=g)).to(dt)
    W = (torch.randn(C, c, generator=g) / c ** 0.5).to(dt)
    y = a2.float() @ W.float().t()                                   # what conv3 accumulates (fp32 from bf16 operands)
    mu, var = y.mean(0), y.var(0, unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    gamma = torch.rand(C, generator=g) + 0.5
    gt = (torch.randn(m, C, generator=g) * (torch.rand(m, C, generator=g) > 0.4)).to(dt)      # already ReLU-gated
    # float64 reference
    yd, gd = y.double(), gt.double()
    xhat = (yd - mu.double()) * invstd.double()
    s = gamma.double() * invstd.double()
    dy = s * (gd - gd.mean(0) - xhat * (gd * xhat).mean(0))
    ref_dw, ref_da = dy.t() @ a2.double(), dy @ W.double()
    ref_dgamma, ref_dbeta = (gd * xhat).sum(0), gd.sum(0)
    # HIP chain
    d = lambda t: t.to(DEV)                                          # noqa: E731
    a2d, gtd = d(a2).view(n, hw, hw, c), d(gt).view(n, hw, hw, C)
    ldw = (c + 15) // 16 * 16
    Wd = torch.zeros(C, ldw, dtype=dt, device=DEV); Wd[:, :c] = d(W)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    P = ops.conv_wgrad(a2d, gtd, 1, 1, 1, 0, ldw=ldw, workspace=ws)                       # [C, ldw]
    gram = ops.conv_wgrad(a2d, a2d, 1, 1, 1, 0, ldw=ldw, workspace=ws)                    # [c, ldw]
    sums = torch.empty(2, c, device=DEV)
    ops.bn_stats_sums(a2d.view(m, c), sums, ops.bn_workspace(m, c, DEV))
    npart = 70                                                       # > 64: slices of more than one partial row
    part = torch.zeros(npart, 2, C)
    for r in range(npart):
        part[r, 0] = gt[r::npart].float().sum(0)
    xh32 = ((y.to(dt).float() - mu) * invstd)                        # what the producing epilogue sees: the stored bf16 y
    for r in range(npart):
        part[r, 1] = (gt[r::npart].float() * xh32[r::npart]).sum(0)
    part = d(part)
    stats = torch.zeros(4, C, device=DEV); stats[0] = d(mu); stats[1] = d(invstd)
    coef = torch.empty(3, C, device=DEV)
    dgam, dbet = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    wt = torch.zeros(c, C + c, dtype=dt, device=DEV)
    bias = torch.empty(c, device=DEV)
    if from_p:
        part[:, 1] = 1e30                                            # (summed, never used: sum g~ y comes from rowdot(P, W))
    tickets = torch.zeros(64, dtype=torch.int32, device=DEV)
    # first without, then with the column-sum compensation (colsum(a2) given): the data gradient's column sums - zero in exact
    # arithmetic, what the BN below turns into its d beta - lose the coherent part of the stacked weights' bf16 rounding
    colerr = []
    for csum2 in (None, sums[0].contiguous()):
        for _ in range(2):                                           # twice: the tickets reset themselves
            ops.bn3_algebra_prep(P if from_p else None, Wd, c, part, npart, stats, d(gamma), m, coef, dgam, dbet, wt, bias,
                                 ops.bn3_algebra_prep_scratch(C, c, DEV), tickets, colsum2=csum2)
        assert not tickets.any()
        da = torch.full((n, hw, hw, c), float("nan"), dtype=dt, device=DEV)
        ops.conv_dgrad2_bnbwd(gtd, a2d, wt, bias, da)
        colerr.append((da.float().cpu().view(m, c).double().sum(0) - ref_da.sum(0)).abs().max().item())
    # (what is left with the compensation is the bf16 rounding of the stored elements, independent from pixel to pixel)
    assert colerr[1] <= 0.5 * colerr[0] + 2.0 ** -9 * ref_da.abs().max().item() * m ** 0.5, colerr
    dW = torch.zeros(C, ldw, device=DEV)
    ops.bn3_algebra_dw(P, Wd, c, gram, sums[0].contiguous(), coef, dW)
    rel = lambda a_, b_: (a_.cpu().double() - b_).norm().item() / b_.norm().item()         # noqa: E731
    # sums from the producer: xhat of the STORED (bf16) y, as the standard route has it: 2^-9 |y| / sigma per element
    assert rel(dgam, ref_dgamma) <= (2e-4 if from_p else 1e-2) and rel(dbet, ref_dbeta) <= 1e-5
    assert rel(dW[:, :c], ref_dw) <= (2e-3 if from_p else 5e-3)   # P and Gram carry fp32 sums over m bf16 products
    assert not dW[:, c:].any()
    got = da.float().cpu().view(m, c).double()
    assert (got - ref_da).abs().max().item() <= 2.0 ** -6 * ref_da.abs().max().item()          # bf16 weights + bf16 result
    assert rel(da.view(m, c), ref_da) <= 1e-2


@pytest.mark.parametrize("case", [(2, 14, 64, 256), (3, 11, 128, 512),
                                  # M % 64 == 0: the weights-in-registers kernel (conv_regw.hip, EPI), one to sixteen N slices
                                  (4, 16, 64, 256), (8, 16, 128, 512), (4, 32, 256, 1024), (16, 16, 512, 2048), (33, 32, 128, 512),
                                  (8, 16, 512, 1024), (5, 32, 512, 1024)],      # ResNeXt-101's 14 x 14 producer: the K = 512 instance
                         ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
def test_dgrad_masked_store_and_column_sums(case):
    """iif_conv_igemm_dgrad_masksum: the conv1 data gradient (+ ReLU-gated residual) stored already gated by the upstream
    block's ReLU bits, per-tile column sums of the stored tensor in the partial rows; bit-identical to the plain data gradient
    followed by the gate."""
    from iif_amd import ops
    n, hw, c, C = case
    m = n * hw * hw
    g = torch.Generator().manual_seed(7 * c + hw)
    dt = torch.bfloat16
    dy1 = torch.randn(n, hw, hw, c, generator=g).to(dt).to(DEV)
    w1 = (torch.randn(c, C, generator=g) / C ** 0.5).to(dt)            # conv1: C -> c; its data gradient: c -> C
    wtt = torch.zeros(C, (c + 15) // 16 * 16, dtype=dt, device=DEV)
    ops.weight_transpose(w1.float().to(DEV), c, C, 1, wtt)
    res = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV)
    rbits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV)
    ubits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV)
    plain = ops.conv_dgrad(dy1, wtt, 1, 1, 1, 0, (hw, hw), res=res, res_bits=rbits)
    out = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    partial = torch.full(((m + 127) // 128 + 8, 2, C), float("nan"), device=DEV)
    nt = ops.conv_dgrad_masksum(dy1, wtt, (hw, hw), out, ubits, partial.view(-1), res=res, res_bits=rbits)
    mask = ((ubits.cpu().view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, C).bool()
    expect = torch.where(mask, plain.cpu().view(m, C), torch.zeros((), dtype=dt))
    assert torch.equal(out.cpu().view(m, C), expect)
    ps = partial[:nt].sum(0).cpu()
    assert nt >= 1 and not torch.isnan(ps).any() and not ps[1].any()
    ref = expect.float().sum(0)
    assert (ps[0] - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()) * 8
    # with the upstream BN's input and statistics: same stored tensor, second half = sum of stored * xhat
    upx = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV)
    stats = torch.zeros(4, C, device=DEV)
    stats[0] = torch.randn(C, generator=g).to(DEV) * 0.1
    stats[1] = (torch.rand(C, generator=g) + 0.5).to(DEV)
    out2 = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    partial.fill_(float("nan"))
    nt2 = ops.conv_dgrad_masksum(dy1, wtt, (hw, hw), out2, ubits, partial.view(-1), res=res, res_bits=rbits, up_x=upx, up_stats=stats)
    assert nt2 == nt and torch.equal(out2, out)
    ps2 = partial[:nt2].sum(0).cpu()
    xhat = (upx.float().cpu().view(m, C) - stats[0].cpu()) * stats[1].cpu()
    ref2 = (expect.float() * xhat).sum(0)
    assert (ps2[0] - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()) * 8
    assert (ps2[1] - ref2).abs().max().item() <= 2e-6 * max(1.0, ref2.abs().max().item()) * 8


# ------------------------------------------------------------------- two-pass forward of conv + BN + identity + ReLU
@pytest.mark.parametrize("case", [(2, 14, 64, 256), (3, 11, 128, 512), (1, 30, 64, 256), (2, 7, 256, 1024)],
                         ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
@pytest.mark.parametrize("stream", [False, True], ids=["tile", "streaming"])
def test_two_pass_forward_is_bit_identical_to_conv_then_bn_apply(case, stream, conv_env):
    """iif_conv_igemm_stats_only + iif_conv_igemm_bn_relu against iif_conv_igemm_bnstats + iif_bn_apply (residual, ReLU,
    ReLU bits): the partial rows, the activation and the bit bytes are bit-identical — the raw convolution output is rounded
    to bf16 in the staging tile exactly as the stored one is."""
    from iif_amd import ops
    if stream:      # the persistent streaming kernel (it shares the staged drain): allowed for these options and forced onto small grids
        conv_env(IIF_CONV_STREAM1X1_FORCE="1")
    n, hw, c, C = case
    m = n * hw * hw
    g = torch.Generator().manual_seed(11 * c + hw)
    dt = torch.bfloat16
    x = torch.relu(torch.randn(n, hw, hw, c, generator=g)).to(dt).to(DEV)
    w = (torch.randn(C, c, generator=g) / c ** 0.5).to(dt).to(DEV)
    res = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV)
    rows = (m + 127) // 128 + 8
    y = torch.empty(n, hw, hw, C, dtype=dt, device=DEV)
    p1 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt1 = ops.conv_forward_bnstats(x, w, 1, 1, 1, 0, y, p1.view(-1))
    p2 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt2 = ops.conv_forward_stats_only(x, w, p2.view(-1))
    assert nt1 == nt2 and torch.equal(p1[:nt1], p2[:nt2])
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV) * 0.1
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    stats = torch.zeros(4, C, device=DEV)
    ops.bn_finalize_stats(p1.view(-1), nt1, m, C, gamma, beta, rm, rv, stats, 1e-5, 0.1)
    a_ref = torch.empty_like(y)
    bits_ref = torch.zeros(m * C // 8, dtype=torch.uint8, device=DEV)
    ops.bn_apply(y.view(m, C), stats, a_ref.view(m, C), relu=True, residual=res.view(m, C), relu_bits=bits_ref)
    a_two = torch.full_like(y, float("nan"))
    bits_two = torch.zeros_like(bits_ref)
    ops.conv_forward_bn_relu(x, w, a_two, stats, res=res, relu_bits=bits_two)
    assert torch.equal(a_two, a_ref) and torch.equal(bits_two, bits_ref)
    # without the identity
    ops.bn_apply(y.view(m, C), stats, a_ref.view(m, C), relu=True, relu_bits=bits_ref)
    ops.conv_forward_bn_relu(x, w, a_two, stats, relu_bits=bits_two)
    assert torch.equal(a_two, a_ref) and torch.equal(bits_two, bits_ref)


@pytest.mark.parametrize("case", [(4, 16, 64, 256), (2, 32, 128, 512), (4, 16, 256, 1024), (16, 56, 64, 256), (4, 16, 128, 256),
                                  (2, 24, 256, 512)], ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
def test_never_stored_forward_on_the_register_weight_kernel(case):
    """Round 6: iif_conv_igemm_stats_acc + iif_conv_igemm_bn_relu2 (weights in registers).  Pass 2 with given statistics is
    bit-identical to convolution + iif_bn_apply (plain residual, normalised residual of a convolutional shortcut, none);
    pass 1's rows sum to the column sums of the fp32 product (the unrounded accumulators), 1e-5 of sum |y| / sum y^2."""
    from iif_amd import ops
    n, hw, c, C = case
    m = n * hw * hw
    assert ops.conv_fwdbn_ok(n, hw, hw, c, C, torch.bfloat16)
    g = torch.Generator().manual_seed(13 * c + hw)
    dt = torch.bfloat16
    x = torch.relu(torch.randn(n, hw, hw, c, generator=g)).to(dt).to(DEV)
    w = (torch.randn(C, c, generator=g) / c ** 0.5).to(dt).to(DEV)
    res = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV)
    rows = (m + 127) // 128 + 8
    # pass 1 against the fp32 product
    p = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt = ops.conv_forward_stats_acc(x, w, p.view(-1))
    assert 0 < nt <= (m + 127) // 128
    yf = x.view(m, c).float() @ w.float().t()
    s, q = p[:nt, 0].double().sum(0), p[:nt, 1].double().sum(0)
    assert not torch.isnan(p[:nt]).any()
    assert ((s - yf.double().sum(0)).abs() <= 1e-5 * yf.double().abs().sum(0) + 1e-6).all()
    assert ((q - (yf.double() ** 2).sum(0)).abs() <= 1e-5 * (yf.double() ** 2).sum(0) + 1e-6).all()
    # pass 2 against convolution + bn_apply, same statistics
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV) * 0.1
    stats = torch.zeros(4, C, device=DEV)
    ops.bn_finalize_stats(p.view(-1), nt, m, C, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), stats, 1e-5, 0.1)
    stats2 = torch.zeros(4, C, device=DEV)
    stats2[2] = torch.rand(C, generator=g).to(DEV) + 0.5
    stats2[3] = torch.randn(C, generator=g).to(DEV) * 0.2
    y = ops.conv_forward(x, w, 1, 1, 1, 0)
    a_ref = torch.empty_like(y)
    bits_ref = torch.zeros(m * C // 8, dtype=torch.uint8, device=DEV)
    for r_, rs_ in ((res, None), (res, stats2), (None, None)):
        ops.bn_apply(y.view(m, C), stats, a_ref.view(m, C), relu=True, residual=None if r_ is None else r_.view(m, C),
                     residual_stats=rs_, relu_bits=bits_ref)
        a_two = torch.full_like(y, float("nan"))
        bits_two = torch.full_like(bits_ref, 0xAA)
        ops.conv_forward_bn_relu2(x, w, a_two, stats, bits_two, res=r_, res_stats=rs_)
        assert torch.equal(a_two, a_ref) and torch.equal(bits_two, bits_ref)


@pytest.mark.parametrize("case", [(4, 16, 64, 256), (2, 32, 128, 512), (4, 16, 256, 1024), (16, 56, 64, 256), (8, 14, 256, 1024),
                                  (32, 7, 512, 2048)], ids=lambda c: "%dx%dx%d_%d_%d" % (c[0], c[1], c[1], c[2], c[3]))
def test_bn_relu_prologue_in_the_convolution_operand_path(case):
    """Round 6: iif_conv_igemm_bnstats_pro (previous unit's BN + ReLU applied to each tile in LDS, activation written as a
    by-product) against iif_bn_apply followed by the plain launch: activation, ReLU bits, convolution output and partial rows
    bit-identical; the column-sum rows sum to the activation's column sums."""
    from iif_amd import ops
    n, hw, c, C = case
    m = n * hw * hw
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17 * c + hw)
    x_raw = torch.randn(n, hw, hw, c, generator=g).to(dt).to(DEV)
    w = (torch.randn(C, c, generator=g) / c ** 0.5).to(dt).to(DEV)
    st = torch.zeros(4, c)
    st[2] = torch.rand(c, generator=g) + 0.5
    st[3] = torch.randn(c, generator=g) * 0.3
    st = st.to(DEV)
    a_ref = torch.empty_like(x_raw)
    bits_ref = torch.zeros(m * c // 8, dtype=torch.uint8, device=DEV)
    ops.bn_apply(x_raw.view(m, c), st, a_ref.view(m, c), relu=True, relu_bits=bits_ref)
    rows = (m + 127) // 128 + 8
    for stats_only in (True, False):
        if not ops.conv_pro_ok(n, hw, hw, c, C, dt, stats_only):
            continue
        p_ref = torch.full((rows, 2, C), float("nan"), device=DEV)
        p = torch.full((rows, 2, C), float("nan"), device=DEV)
        act = torch.full_like(x_raw, float("nan"))
        bits = torch.full_like(bits_ref, 0x55)
        csum = torch.full((rows, 2, c), float("nan"), device=DEV)
        if stats_only:
            nt_ref = ops.conv_forward_stats_acc(a_ref, w, p_ref.view(-1))
            nt = ops.conv_forward_bnstats_pro(x_raw, st, act, bits, w, None, p.view(-1), act_csum=csum)
        else:
            y_ref = torch.empty(n, hw, hw, C, dtype=dt, device=DEV)
            nt_ref = ops.conv_forward_bnstats(a_ref, w, 1, 1, 1, 0, y_ref, p_ref.view(-1))
            y = torch.full_like(y_ref, float("nan"))
            nt = ops.conv_forward_bnstats_pro(x_raw, st, act, bits, w, y, p.view(-1), act_csum=csum)
            assert torch.equal(y, y_ref)
        assert nt == nt_ref and nt > 0
        assert torch.equal(act, a_ref) and torch.equal(bits, bits_ref)
        assert torch.equal(p[:nt], p_ref[:nt])
        cs = csum[:nt, 0].double().sum(0)
        assert (csum[:nt, 1] == 0).all()
        ref = a_ref.view(m, c).double().sum(0)
        assert ((cs - ref).abs() <= 1e-5 * a_ref.view(m, c).double().abs().sum(0) + 1e-6).all()


@pytest.mark.parametrize("case", [(4, 16, 64, 256, 64), (4, 16, 128, 256, 64), (2, 32, 128, 512, 128), (4, 16, 256, 512, 128),
                                  (16, 56, 64, 256, 64)], ids=lambda c: "%dx%dx%d_k%d_n%d_c%d" % (c[0], c[1], c[1], c[2], c[3], c[4]))
@pytest.mark.parametrize("with_res", [False, True], ids=["plain", "residual"])
def test_dgrad_masksum_with_the_upstream_output_recomputed(case, with_res):
    """Round 6: iif_conv_igemm_dgrad_masksum_rx (the upstream block's conv3 tile recomputed from its a2 and weights) against
    iif_conv_igemm_dgrad_masksum reading that output from memory: gated data gradient and partial rows bit-identical (the
    recomputed tile is the stored one, value for value: same K order, same bf16 rounding)."""
    from iif_amd import ops
    n, hw, k, C, c2 = case
    m = n * hw * hw
    dt = torch.bfloat16
    assert ops.conv_dgrad_rx_ok(n, hw, hw, k, C, c2, dt)
    g = torch.Generator().manual_seed(7 * k + c2 + hw)
    a2 = torch.relu(torch.randn(n, hw, hw, c2, generator=g)).to(dt).to(DEV)
    w3 = (torch.randn(C, c2, generator=g) / c2 ** 0.5).to(dt).to(DEV)
    y3 = ops.conv_forward(a2, w3, 1, 1, 1, 0)                          # what the forward pass would have stored
    dy = torch.randn(n, hw, hw, k, generator=g).to(dt).to(DEV)          # dL/dx1 of the next block's conv1
    wt = (torch.randn(C, k, generator=g) / k ** 0.5).to(dt).to(DEV)     # its transposed weights [cin = C][cout = k]
    ubits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV)
    res = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV) if with_res else None
    rbits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV) if with_res else None
    stats = torch.zeros(4, C)
    stats[0] = torch.randn(C, generator=g) * 0.1
    stats[1] = torch.rand(C, generator=g) + 0.5
    stats = stats.to(DEV)
    rows = (m + 127) // 128 + 8
    o1 = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    p1 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt1 = ops.conv_dgrad_masksum(dy, wt, (hw, hw), o1, ubits, p1.view(-1), res=res, res_bits=rbits, up_x=y3, up_stats=stats)
    o2 = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    p2 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt2 = ops.conv_dgrad_masksum_rx(dy, wt, (hw, hw), o2, ubits, p2.view(-1), a2, w3, stats, res=res, res_bits=rbits)
    assert nt1 == nt2 and nt1 > 0
    assert torch.equal(o1, o2)
    assert not torch.isnan(p2[:nt2]).any()
    assert torch.equal(p1[:nt1], p2[:nt2])


@pytest.mark.parametrize("case", [(4, 16, 64, 256, 64), (4, 16, 128, 256, 64), (16, 56, 64, 256, 64), (3, 32, 128, 256, 64)],
                         ids=lambda c: "%dx%dx%d_k%d_n%d_c%d" % (c[0], c[1], c[1], c[2], c[3], c[4]))
@pytest.mark.parametrize("with_res", [False, True], ids=["plain", "residual"])
def test_recomputing_producer_leaves_the_algebra_matrices_behind(case, with_res):
    """Round 6: iif_conv_igemm_dgrad_masksum_rx_pg = iif_conv_igemm_dgrad_masksum_rx (stored tensor and partial rows bit-identical)
    plus one fp32 slab [(C + c2), ld] per tile sequence; their sum (iif_slab_sum, slab order) is P = g~^T a2 and Gram = a2^T a2
    of the STORED g~ - compared with the float64 products of the stored tensors (fp32 accumulation error only) and with the
    weight-gradient GEMM that used to compute them (iif_wgrad1x1_stacked)."""
    from iif_amd import ops
    n, hw, k, C, c2 = case
    m = n * hw * hw
    dt = torch.bfloat16
    assert ops.conv_dgrad_rx_pg_ok(n, hw, hw, k, C, c2, dt)
    g = torch.Generator().manual_seed(11 * k + c2 + hw)
    a2 = torch.relu(torch.randn(n, hw, hw, c2, generator=g)).to(dt).to(DEV)
    w3 = (torch.randn(C, c2, generator=g) / c2 ** 0.5).to(dt).to(DEV)
    dy = torch.randn(n, hw, hw, k, generator=g).to(dt).to(DEV)
    wt = (torch.randn(C, k, generator=g) / k ** 0.5).to(dt).to(DEV)
    ubits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV)
    res = torch.randn(n, hw, hw, C, generator=g).to(dt).to(DEV) if with_res else None
    rbits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(DEV) if with_res else None
    stats = torch.zeros(4, C)
    stats[0] = torch.randn(C, generator=g) * 0.1
    stats[1] = torch.rand(C, generator=g) + 0.5
    stats = stats.to(DEV)
    rows = (m + 127) // 128 + 8
    o1 = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    p1 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt1 = ops.conv_dgrad_masksum_rx(dy, wt, (hw, hw), o1, ubits, p1.view(-1), a2, w3, stats, res=res, res_bits=rbits)
    ld = c2
    cap = 256 + 256 // 16 + 1
    slabs = torch.full((cap, C + c2, ld), float("nan"), device=DEV)
    o2 = torch.full((n, hw, hw, C), float("nan"), dtype=dt, device=DEV)
    p2 = torch.full((rows, 2, C), float("nan"), device=DEV)
    nt2, ns = ops.conv_dgrad_masksum_rx_pg(dy, wt, (hw, hw), o2, ubits, p2.view(-1), a2, w3, stats, slabs.view(-1), ld, res=res, res_bits=rbits)
    assert nt1 == nt2 and 0 < ns <= 256
    assert torch.equal(o1, o2) and torch.equal(p1[:nt1], p2[:nt2])
    assert not torch.isnan(slabs[:ns]).any() and torch.isnan(slabs[ns:]).all()
    ext = torch.full((C + c2, ld), float("nan"), device=DEV)
    ops.slab_sum(slabs.view(-1), ns, C + c2, ld, c2, ext)
    assert torch.equal(ext, slabs[:ns].sum(0)) or (ext - slabs[:ns].double().sum(0).float()).abs().max().item() <= 1e-5 * ext.abs().max().item()
    gs, a2d = o2.view(m, C).double().cpu(), a2.view(m, c2).double().cpu()
    ref_p, ref_g = gs.t() @ a2d, a2d.t() @ a2d
    got = ext.double().cpu()
    assert (got[:C] - ref_p).abs().max().item() <= 2e-5 * (gs.abs().t() @ a2d).max().item()
    assert (got[C:] - ref_g).abs().max().item() <= 2e-5 * ref_g.max().item()
    # and the weight-gradient GEMM it replaces (the same products, another summation order)
    old = torch.zeros(C + c2, ld, device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    ops.wgrad1x1_stacked(a2.view(m, c2), o2.view(m, C), a2.view(m, c2), old, ws)
    assert (old - ext).abs().max().item() <= 4e-5 * max(old.abs().max().item(), 1.0)


@pytest.mark.parametrize("case", [(4, 128, 14, 14, 32), (2, 256, 28, 28, 32), (1, 512, 7, 7, 32), (3, 64, 9, 11, 8), (2, 512, 14, 14, 32),
                                  (1, 128, 56, 56, 32)],
                         ids=lambda c: "n%d_w%d_%dx%d_g%d" % c)
def test_grouped_conv3x3_on_the_fragment_kernel(case, conv_env):
    """ResNeXt's grouped 3x3 / stride 1 on the fragment-weights kernel (blockIdx.y = 64-channel chunk): forward with fused
    statistics, data gradient with a ReLU-masked residual and with the upstream BN-backward sums — the stored tensors
    against the tile kernel's (same products, the summation order over K differs: bf16 ulp) and the partial rows against
    sums over the stored values."""
    from iif_amd import ops
    n, width, h, w, groups = case
    cg = width // groups
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(width + groups + h)
    x = torch.randn(n, h, w, width, generator=g).to(dt).to(DEV)
    wt = (torch.randn(width, cg, 3, 3, generator=g) / (9 * cg) ** 0.5).to(dt).float()
    ldm = (9 * cg + 15) // 16 * 16
    master = torch.zeros(width, ldm)
    master[:, :9 * cg] = wt.permute(0, 2, 3, 1).reshape(width, 9 * cg)
    wp = torch.empty(width, 576, dtype=dt, device=DEV)
    wpt = torch.empty(width, 576, dtype=dt, device=DEV)
    ops.group_pack(master.to(DEV), width, cg, 64, 9, wp)
    ops.group_pack(master.to(DEV), width, cg, 64, 9, wpt, transposed=True)
    G = width // 64
    m = n * h * w
    conv_env(IIF_CONV_V2_FORCE="1")
    assert ops.conv3x3_frag_ok(n, h, w, 64, 64, dt, groups=G)
    wf, wtf = _pack_frag(wp, width, 9, 64), _pack_frag(wpt, width, 9, 64)
    modes = ["tile", "frag"]
    if cg <= 16 and 16 % cg == 0:
        # round 6: the 16-channel format (K of an MFMA = two taps x the output tile's own 16 input channels) for narrow groups
        def pack16(w2d):
            tab, blocks = ops.pack_table_g16([(0, 0, width, 9, 64, w2d.shape[1])], DEV)
            return ops.pack_fragments_g16(w2d, tab, 1, blocks, torch.empty(width // 64 * 20 * 512, dtype=w2d.dtype, device=DEV))
        g16f, g16t = (pack16(wp), 1), (pack16(wpt), 1)
        modes.append("g16")
    dy = torch.randn(n, h, w, width, generator=g).to(dt).to(DEV)
    res = torch.randn(n, h, w, width, generator=g).to(dt).to(DEV)
    rbits = torch.randint(0, 256, (m * width // 8,), dtype=torch.uint8, generator=g).to(DEV)
    upx = torch.randn(n, h, w, width, generator=g).to(dt).to(DEV)
    ubits = torch.randint(0, 256, (m * width // 8,), dtype=torch.uint8, generator=g).to(DEV)
    stats = torch.zeros(4, width)
    stats[0] = torch.randn(width, generator=g) * 0.1
    stats[1] = torch.rand(width, generator=g) + 0.5
    out = {}
    for mode in modes:
        f, ft = (wf, wtf) if mode == "frag" else ((g16f, g16t) if mode == "g16" else (None, None))
        y = torch.full((n, h, w, width), float("nan"), dtype=dt, device=DEV)
        p = torch.full(((m + 127) // 128 + 8, 2, width), float("nan"), device=DEV)
        nt = ops.conv_forward_bnstats(x, wp, 3, 3, 1, 1, y, p.view(-1), groups=G, w_frag=f)
        dx = ops.conv_dgrad(dy, wpt, 3, 3, 1, 1, (h, w), res=res, res_bits=rbits, groups=G, w_frag=ft)
        dx2 = torch.full((n, h, w, width), float("nan"), dtype=dt, device=DEV)
        p2 = torch.full(((m + 127) // 128 + 8, 2, width), float("nan"), device=DEV)
        nt2 = ops.conv_dgrad_bnbwd(dy, wpt, 3, 3, 1, 1, (h, w), dx2, upx, ubits, stats.to(DEV), p2.view(-1), groups=G, w_frag=ft)
        out[mode] = (y, p[:nt].sum(0).cpu(), dx, dx2, p2[:nt2].sum(0).cpu())
    for mode in modes[1:]:
        for i in (0, 2, 3):
            a_, b_ = out[mode][i].float(), out["tile"][i].float()
            assert not torch.isnan(a_).any()
            assert (a_ - b_).abs().max().item() <= 2.0 ** -6 * b_.abs().max().item(), (mode, i)
        for i in (1, 4):
            a_, b_ = out[mode][i], out["tile"][i]
            assert (a_ - b_).abs().max().item() <= 2e-3 * max(1.0, b_.abs().max().item()), (mode, i)
    yf = out["frag"][0].float().cpu().view(m, width)
    s = out["frag"][1]
    assert (s[0] - yf.sum(0)).abs().max().item() <= 1e-3 * max(1.0, yf.sum(0).abs().max().item())
    assert (s[1] - (yf * yf).sum(0)).abs().max().item() <= 1e-3 * (yf * yf).sum(0).abs().max().item()
    gq = out["frag"][3].float().cpu().view(m, width) * ((ubits.cpu().view(-1, 1).int() >> torch.arange(8).view(1, 8)) & 1).view(m, width)
    xhat = (upx.float().cpu().view(m, width) - stats[0]) * stats[1]
    s2 = out["frag"][4]
    assert (s2[0] - gq.sum(0)).abs().max().item() <= 1e-3 * max(1.0, gq.sum(0).abs().max().item())
    assert (s2[1] - (gq * xhat).sum(0)).abs().max().item() <= 1e-3 * max(1.0, (gq * xhat).sum(0).abs().max().item())


# ------------------------------------------------------------------- the register-staged fallback (operands >= 2 GiB)
@pytest.mark.parametrize("case", [(3, 64, 14, 14, 128, 1, 1), (2, 32, 13, 11, 64, 3, 1), (2, 64, 12, 12, 96, 3, 2)],
                         ids=lambda c: "%dx%dx%dx%d_to_%d_k%d_s%d" % c)
def test_register_staged_fallback_kernels(case, conv_env):
    """IIF_CONV_REGSTAGE=1 puts every launch on the round-1 register-staged kernels — what operands beyond the 32-bit
    LDS-DMA offsets (>= 2 GiB, batch ~1280 at 224 x 224) fall back to: forward, data gradient and weight gradient against
    torch, in bf16 tolerances, and against the default (LDS-DMA) kernels."""
    import torch.nn.functional as F
    from iif_amd import ops
    n, cin, h, w, cout, k, stride = case
    pad = k // 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(n, cin, h, w, generator=g).to(dt)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * k * cin) ** 0.5).to(dt)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g).to(dt)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()                                    # noqa: E731
    xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    wd = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(DEV)
    ldwt = (k * k * cout + 15) // 16 * 16
    wtt = torch.zeros(cin, ldwt, dtype=dt, device=DEV)
    ops.weight_transpose(wt.permute(0, 2, 3, 1).reshape(cout, -1).float().to(DEV), cout, cin, k * k, wtt)
    ref_y = F.conv2d(x.float(), wt.float(), None, stride, pad)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wt.float(), dy.float(), stride, pad)
    ref_dw = torch.nn.grad.conv2d_weight(x.float(), wt.shape, dy.float(), stride, pad)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    got = {}
    for mode in ("regstage", "default"):
        conv_env(IIF_CONV_REGSTAGE="1" if mode == "regstage" else None)
        y = ops.conv_forward(xd, wd, k, k, stride, pad)
        dx = ops.conv_dgrad(dyd, wtt, k, k, stride, pad, (h, w))
        dw = ops.conv_wgrad(xd, dyd, k, k, stride, pad, workspace=ws)
        got[mode] = (y.float().cpu().permute(0, 3, 1, 2), dx.float().cpu().permute(0, 3, 1, 2),
                     dw.cpu().view(cout, k, k, cin).permute(0, 3, 1, 2))
    for mode, (y, dx, dw) in got.items():
        assert (y - ref_y).abs().max().item() <= 2.0 ** -7 * ref_y.abs().max().item(), mode
        assert (dx - ref_dx).abs().max().item() <= 2.0 ** -7 * ref_dx.abs().max().item(), mode
        assert (dw - ref_dw).norm().item() <= 1e-4 * ref_dw.norm().item(), mode             # fp32 sums of exact bf16 products
    assert (got["regstage"][0] - got["default"][0]).abs().max().item() <= 2.0 ** -7 * ref_y.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_group_pack_batched_equals_the_per_layer_launches(dt):
    """One launch over a device table (ResNeXt: every grouped layer, both orientations) writes the bytes of the
    per-layer iif_group_pack calls (resnet_pytorch.py:137,141 grouped 3x3 weights)."""
    from iif_amd import ops
    g = torch.Generator().manual_seed(11)
    layers = [(128, 4), (256, 8), (512, 16), (1024, 32), (64, 2)]
    ent, want = [], []
    for width, cg in layers:
        ch = max(64, cg)
        ldm = (9 * cg + 15) // 16 * 16
        master = torch.zeros(width, ldm)
        master[:, :9 * cg] = torch.randn(width, 9 * cg, generator=g)
        master = master.to(DEV)
        for tr in (False, True):
            ref = torch.empty(width, 9 * ch, dtype=dt, device=DEV)
            ops.group_pack(master, width, cg, ch, 9, ref, transposed=tr)
            out = torch.full((width, 9 * ch), 7.0, dtype=dt, device=DEV)
            ent.append((master, width, cg, ch, 9, out, tr))
            want.append(ref)
    tab = ops.group_pack_table(ent, DEV)
    ops.group_pack_batched(tab)
    torch.cuda.synchronize()
    for e, ref in zip(ent, want):
        assert torch.equal(e[5], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2 * 28 * 28, 128, 512, 128), (3 * 14 * 14 + 5, 256, 1024, 256), (777, 64, 256, 64), (4 * 196, 128, 128, 128),
                                  (1000, 256, 384, 40)])
def test_wgrad1x1_stacked_equals_two_weight_gradients(case):
    """iif_wgrad1x1_stacked: [dy | dy2]^T x in one pass (P = g~^T a2 and the Gram matrix a2^T a2 of the BN-by-algebra
    backward) against the float64 products of the same bf16 operands; pad columns untouched; splits 0 / 1 / -2."""
    from iif_amd import ops
    m, cs, cd1, cd2 = case
    g = torch.Generator().manual_seed(m + cs + cd1)
    x = torch.randn(m, cs, generator=g).bfloat16()
    dy = torch.randn(m, cd1, generator=g).bfloat16()
    dy2 = x[:, :cd2].contiguous() if cd2 <= cs else torch.randn(m, cd2, generator=g).bfloat16()
    ref = torch.cat([dy.double().t() @ x.double(), dy2.double().t() @ x.double()])
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    ldw = cs + 16
    xd, dyd, dy2d = x.to(DEV), dy.to(DEV), dy2.to(DEV)
    for splits in (0, 1, -2):
        out = torch.full((cd1 + cd2, ldw), 7.0, device=DEV)
        ops.wgrad1x1_stacked(xd, dyd, dy2d, out, ws, splits=splits)
        assert (out[:, cs:] == 7.0).all()
        assert (out[:, :cs].double().cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), (case, splits)
    a = torch.zeros((cd1 + cd2, ldw), device=DEV)
    ops.wgrad1x1_stacked(xd, dyd, dy2d, a, ws)
    assert torch.equal(a, out.new_tensor(a))            # (contiguity)
    b = torch.zeros_like(a)
    ops.wgrad1x1_stacked(xd, dyd, dy2d, b, ws)
    assert torch.equal(a, b)                            # deterministic
