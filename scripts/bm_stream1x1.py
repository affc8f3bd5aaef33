"""A/B of the streaming 1x1 kernel against the tile kernels on the ResNet50 bs-256 layer shapes it takes over
(forward with fused BN statistics, data gradient with the upstream BN-backward sums), each alone on the GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops
dev = "cuda:0"
dt = torch.bfloat16
N = 256


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


MODES = ("tile", "k64", "stream")


def mode(m):
    os.environ.pop("IIF_CONV_NO_STREAM1X1", None); os.environ.pop("IIF_CONV_NO_SHORTK", None)
    os.environ["IIF_CONV_STREAM1X1_FORCE"] = "1"
    if m != "stream":
        os.environ["IIF_CONV_NO_STREAM1X1"] = "1"; os.environ.pop("IIF_CONV_STREAM1X1_FORCE", None)
    if m == "tile":
        os.environ["IIF_CONV_NO_SHORTK"] = "1"


for (hw, cin, cout) in ((56, 64, 256), (56, 256, 64), (56, 64, 64), (56, 256, 128), (28, 64, 256), (28, 256, 64)):
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
    print("fwd+stats %dx%d %3d->%3d  " % (hw, hw, cin, cout) + "   ".join("%s %.3f ms (%4.0f GB/s)" % (k, res[k], byt / res[k] / 1e6) for k in MODES))
    # data gradient of the same layer: dy [m, cout] -> dx [m, cin], + residual + upstream BN-backward sums
    dy = torch.randn(N, hw, hw, cout, device=dev).to(dt)
    wtt = torch.zeros(cin, (cout + 15) // 16 * 16, dtype=dt, device=dev)
    ops.weight_transpose(w.float(), cout, cin, 1, wtt)
    dx = torch.empty(N, hw, hw, cin, device=dev, dtype=dt)
    upx = torch.randn(N, hw, hw, cin, device=dev).to(dt)
    bits = torch.randint(0, 255, (m * cin // 8,), device=dev, dtype=torch.uint8)
    stats = torch.rand(4, cin, device=dev)
    byt2 = 2 * m * (cin + cout) + 2 * m * cin + m * cin // 8
    for s in MODES:
        mode(s)
        res[s] = timed(lambda: ops.conv_dgrad_bnbwd(dy, wtt, 1, 1, 1, 0, (hw, hw), dx, upx, bits, stats, partial))
    print("dgrad+bw  %dx%d %3d->%3d  " % (hw, hw, cout, cin) + "   ".join("%s %.3f ms (%4.0f GB/s)" % (k, res[k], byt2 / res[k] / 1e6) for k in MODES))
