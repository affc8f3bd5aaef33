"""A/B of the streaming 1x1 kernel against the tile kernels on the ResNet50 bs-256 layer shapes it takes over
(forward with fused BN statistics; data gradient with the upstream BN-backward sums; the conv1 data gradient with the masked
residual as well), each alone on the GPU.  Modes: k64 = the 4-blocks-per-CU tile kernel, r2 = the round-2 streaming kernel's
coverage (forward, whole weight matrix resident), stream = this round's (N slices, epilogue operands fetched a tile ahead)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops, _lib
dev = "cuda:0"
dt = torch.bfloat16
N = int(os.environ.get("BM_BATCH", "256"))


def timed(f, it=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


MODES = ("k64", "stream")


def mode(m):
    for k in ("IIF_CONV_NO_STREAM1X1", "IIF_CONV_STREAM1X1_FORCE"):
        os.environ.pop(k, None)
    if m == "k64":
        os.environ["IIF_CONV_NO_STREAM1X1"] = "1"
    elif m == "r2":
        pass                                            # the default coverage
    else:
        os.environ["IIF_CONV_STREAM1X1_FORCE"] = "1"
    _lib.check(_lib.lib().iif_conv_reload_env(), "reload")


# (hw, cin, cout, launches per step forward, dgrad-with-sums, conv1-form dgrad)   [forward cin -> cout; data gradient cout -> cin]
SHAPES = ((56, 64, 256, 4, 4, 0), (56, 256, 64, 2, 0, 2), (56, 256, 128, 1, 0, 1), (28, 128, 512, 4, 4, 0), (28, 512, 128, 3, 0, 3),
          (28, 512, 256, 1, 0, 1), (14, 256, 1024, 6, 6, 0), (14, 1024, 256, 5, 0, 5))
tot = {k: 0.0 for k in MODES}
for (hw, cin, cout, nf, nd, nd1) in SHAPES:
    m = N * hw * hw
    x = torch.randn(N, hw, hw, cin, device=dev).to(dt)
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(dt)
    out = torch.empty(N, hw, hw, cout, device=dev, dtype=dt)
    partial = torch.empty(((m + 127) // 128 + 8) * 2 * max(cin, cout), device=dev)
    byt = 2 * m * (cin + cout)
    res = {}
    for s in MODES:
        mode(s)
        res[s] = timed(lambda: ops.conv_forward_bnstats(x, w, 1, 1, 1, 0, out, partial))
        tot[s] += nf * res[s]
    print("fwd+stats   %2dx%-2d %4d->%-4d " % (hw, hw, cin, cout) + "   ".join("%s %.3f ms (%4.0f GB/s)" % (k, res[k], byt / res[k] / 1e6) for k in MODES), flush=True)
    # data gradient of the same layer: dy [m, cout] -> dx [m, cin], + upstream BN-backward sums (+ masked residual)
    dy = torch.randn(N, hw, hw, cout, device=dev).to(dt)
    wtt = torch.zeros(cin, (cout + 15) // 16 * 16, dtype=dt, device=dev)
    ops.weight_transpose(w.float(), cout, cin, 1, wtt)
    dx = torch.empty(N, hw, hw, cin, device=dev, dtype=dt)
    upx = torch.randn(N, hw, hw, cin, device=dev).to(dt)
    rs = torch.randn(N, hw, hw, cin, device=dev).to(dt)
    bits = torch.randint(0, 255, (m * cin // 8,), device=dev, dtype=torch.uint8)
    bits2 = torch.randint(0, 255, (m * cin // 8,), device=dev, dtype=torch.uint8)
    stats = torch.rand(4, cin, device=dev)
    byt2 = 2 * m * (cin + cout) + 2 * m * cin + m * cin // 8
    byt3 = byt2 + 2 * m * cin + m * cin // 8
    for s in MODES:
        mode(s)
        res[s] = timed(lambda: ops.conv_dgrad_bnbwd(dy, wtt, 1, 1, 1, 0, (hw, hw), dx, upx, bits, stats, partial))
        tot[s] += nd * res[s]
    print("dgrad+bw    %2dx%-2d %4d->%-4d " % (hw, hw, cout, cin) + "   ".join("%s %.3f ms (%4.0f GB/s)" % (k, res[k], byt2 / res[k] / 1e6) for k in MODES), flush=True)
    for s in MODES:
        mode(s)
        res[s] = timed(lambda: ops.conv_dgrad_bnbwd(dy, wtt, 1, 1, 1, 0, (hw, hw), dx, upx, bits, stats, partial, res=rs, res_bits=bits2))
        tot[s] += nd1 * res[s]
    print("dgrad+bw+rs %2dx%-2d %4d->%-4d " % (hw, hw, cout, cin) + "   ".join("%s %.3f ms (%4.0f GB/s)" % (k, res[k], byt3 / res[k] / 1e6) for k in MODES), flush=True)
    del x, out, dy, dx, upx, rs
print("weighted by launches per step: " + "  ".join("%s %.3f ms" % (k, tot[k]) for k in MODES))
