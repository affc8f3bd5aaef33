"""conv_regw.hip (weights in registers) against the tile kernel: bit-identity of the stored output, partial sums, time alone."""
import os, sys
sys.path.insert(0, ".")
import torch
from iif_amd import ops, _lib
dev = "cuda:0"
SHAPES = [(256, 28, 28, 128, 512), (256, 14, 14, 256, 1024), (256, 7, 7, 512, 2048), (256, 28, 28, 512, 128), (256, 14, 14, 1024, 256),
          (64, 28, 28, 128, 512), (8, 14, 14, 256, 1024), (40, 28, 28, 512, 128), (24, 14, 14, 1024, 256), (64, 7, 7, 512, 2048)]


def run(n, h, w, k, c, regw):
    if regw:
        os.environ.pop("IIF_CONV_NO_REGW", None)
    else:
        os.environ["IIF_CONV_NO_REGW"] = "1"
    _lib.lib().iif_conv_reload_env()
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(n, h, w, k, generator=g).bfloat16().to(dev)
    wt = (torch.randn(c, k, generator=g) * 0.1).bfloat16().to(dev)
    out = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((n * h * w + 127) // 128 + 8) * 2 * c, device=dev)
    nt = ops.conv_forward_bnstats(x, wt, 1, 1, 1, 0, out, partial)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ops.conv_forward_bnstats(x, wt, 1, 1, 1, 0, out, partial)
    e0.record()
    for _ in range(50):
        ops.conv_forward_bnstats(x, wt, 1, 1, 1, 0, out, partial)
    e1.record()
    torch.cuda.synchronize()
    ps = partial[:nt * 2 * c].view(nt, 2, c).double().sum(0)
    return out, ps, e0.elapsed_time(e1) / 50 * 1e3, nt


for s in SHAPES:
    o1, p1, t1, n1 = run(*s, True)
    o0, p0, t0, n0 = run(*s, False)
    same = torch.equal(o1, o0)
    flat = o0.double().view(-1, s[4])
    ref = torch.stack([flat.sum(0), (flat * flat).sum(0)])
    e1 = ((p1 - ref).abs().max() / ref.abs().max()).item()
    e0 = ((p0 - ref).abs().max() / ref.abs().max()).item()
    mb = s[0] * s[1] * s[2] * (s[3] + s[4]) * 2 / 1e6
    print("%s  regw %6.1f us (%4.0f GB/s, rows %d)  tile %6.1f us (%4.0f GB/s, rows %d)  identical %s  sums err %.1e / %.1e" % (
        s, t1, mb / t1 * 1e3, n1, t0, mb / t0 * 1e3, n0, same, e1, e0))


# ---- data gradients with the fused epilogue (conv1 dgrad = producer of the block-output gradient)
def run_dgrad(n, hw, c, C, regw, with_x):
    if regw:
        os.environ.pop("IIF_CONV_NO_REGW", None)
    else:
        os.environ["IIF_CONV_NO_REGW"] = "1"
    _lib.lib().iif_conv_reload_env()
    m = n * hw * hw
    g = torch.Generator().manual_seed(c + hw)
    dy = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(dev)
    wtt = (torch.randn(C, c, generator=g) / C ** 0.5).bfloat16().to(dev)
    res = torch.randn(n, hw, hw, C, generator=g).bfloat16().to(dev)
    ubits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(dev)
    upx = torch.randn(n, hw, hw, C, generator=g).bfloat16().to(dev) if with_x else None
    stats = torch.rand(4, C, generator=g).to(dev) + 0.5
    out = torch.empty(n, hw, hw, C, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * C, device=dev)
    f = lambda: ops.conv_dgrad_masksum(dy, wtt, (hw, hw), out, ubits, partial, res=res, up_x=upx, up_stats=stats if with_x else None)  # noqa: E731
    nt = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        f()
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    ps = partial[:nt * 2 * C].view(nt, 2, C).double().sum(0)
    return out, ps, e0.elapsed_time(e1) / 50 * 1e3, nt


for (n, hw, c, C, wx) in [(256, 56, 64, 256, False), (256, 28, 128, 512, True), (256, 14, 256, 1024, True), (256, 7, 512, 2048, True),
                          (256, 28, 128, 512, False)]:
    o1, p1, t1, n1 = run_dgrad(n, hw, c, C, True, wx)
    o0, p0, t0, n0 = run_dgrad(n, hw, c, C, False, wx)
    mb = n * hw * hw * (c + C * (3 if wx else 2)) * 2 / 1e6
    err = ((p1 - p0).abs().max() / p0.abs().max()).item()
    print("dgrad masksum %s x=%s  regw %6.1f us (%4.0f GB/s, rows %d)  tile %6.1f us (%4.0f GB/s, rows %d)  identical %s  sums %.1e" % (
        (n, hw, c, C), wx, t1, mb / t1 * 1e3, n1, t0, mb / t0 * 1e3, n0, torch.equal(o1, o0), err))
