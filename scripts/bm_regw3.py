"""3x3 with the weights in registers (conv_regw.hip) against the window kernels: bit-identity, sums, time alone."""
import os, sys
sys.path.insert(0, ".")
import torch
from iif_amd import ops, _lib
dev = "cuda:0"


def setup(regw):
    if regw:
        os.environ.pop("IIF_CONV_NO_REGW", None)
    else:
        os.environ["IIF_CONV_NO_REGW"] = "1"
    _lib.lib().iif_conv_reload_env()


def timeit(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


for (n, hw, c) in [(256, 56, 64), (32, 56, 64), (5, 24, 64), (256, 28, 128)]:
    g = torch.Generator().manual_seed(c + hw)
    x = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(dev)
    wt = (torch.randn(c, 9 * c, generator=g) / (9 * c) ** 0.5).bfloat16().to(dev)
    upx = torch.randn(n, hw, hw, c, generator=g).bfloat16().to(dev)
    bits = torch.randint(0, 256, (n * hw * hw * c // 8,), dtype=torch.uint8, generator=g).to(dev)
    stats = (torch.rand(4, c, generator=g) + 0.5).to(dev)
    m = n * hw * hw
    res = {}
    for regw in (True, False):
        setup(regw)
        out = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=dev)
        partial = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
        nt = ops.conv_forward_bnstats(x, wt, 3, 3, 1, 1, out, partial)
        ps = partial[:nt * 2 * c].view(nt, 2, c).double().sum(0).clone()
        tf = timeit(lambda: ops.conv_forward_bnstats(x, wt, 3, 3, 1, 1, out, partial))
        out2 = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=dev)
        partial2 = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
        nt2 = ops.conv_dgrad_bnbwd(x, wt, 3, 3, 1, 1, (hw, hw), out2, upx, bits, stats, partial2)
        ps2 = partial2[:nt2 * 2 * c].view(nt2, 2, c).double().sum(0).clone()
        td = timeit(lambda: ops.conv_dgrad_bnbwd(x, wt, 3, 3, 1, 1, (hw, hw), out2, upx, bits, stats, partial2))
        res[regw] = (out.clone(), ps, tf, nt, out2.clone(), ps2, td, nt2)
    a, b = res[True], res[False]
    e1 = ((a[1] - b[1]).abs().max() / b[1].abs().max()).item()
    e2 = ((a[5] - b[5]).abs().max() / b[5].abs().max()).item()
    gf = 2 * m * c * 9 * c / 1e9
    print("%s fwd regw %6.1f us (%4.0f TF, rows %d) window %6.1f us (rows %d) same %s sums %.1e | dgrad+sums regw %6.1f window %6.1f same %s sums %.1e" % (
        (n, hw, c), a[2], gf / a[2] * 1e3, a[3], b[2], b[3], torch.equal(a[0], b[0]), e1, a[6], b[6], torch.equal(a[4], b[4]), e2))
