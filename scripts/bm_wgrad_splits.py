"""Split-K sweep of the 1x1 / 3x3 weight gradients with few output tiles (14x14 / 7x7 stages): time of wgrad + slab reduce
against the number of splits (0 = the library's choice)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops
dev = "cuda:0"
dt = torch.bfloat16


def timed(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
big = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
for (n, hw, cin, cout, k) in ((256, 14, 256, 1024, 1), (256, 14, 1024, 256, 1), (256, 7, 512, 2048, 1), (256, 7, 2048, 512, 1), (256, 14, 256, 256, 3),
                              (128, 14, 512, 1024, 1), (128, 14, 1024, 512, 1), (256, 28, 128, 512, 1), (256, 28, 512, 128, 1)):
    x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
    dy = torch.randn(n, hw, hw, cout, device=dev).to(dt)
    out = torch.zeros(cout, k * k * cin, device=dev)
    res = []
    for sp in (0, 4, 8, 16, 32, 64, 128):
        def f():
            big.zero_() if os.environ.get("FLUSH") else None
            ops.conv_wgrad(x, dy, k, k, 1, k // 2, out=out, workspace=ws, splits=sp)
        res.append((sp, timed(f)))
    byt = 2 * n * hw * hw * (cin + cout)
    print("wgrad n%d %dx%d %4d->%4d k%d (operands %.0f MB, dW %.1f MB): " % (n, hw, hw, cin, cout, k, byt / 1e6, cout * k * k * cin * 4 / 1e6)
          + "  ".join("%d: %.3f" % r for r in res), flush=True)
