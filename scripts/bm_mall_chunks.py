"""Does the 256 MiB Infinity Cache serve a consumer that follows its producer closely enough?  bn_apply (x -> y) followed by
the 1x1 convolution that reads y, on the whole batch and in batch chunks (y chunk << 256 MiB, read back right after it is
written).  Also: bn_apply -> two consumers (the backward pattern dx -> data gradient + weight gradient)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops
dev = "cuda:0"
dt = torch.bfloat16
N = 256


def timed(f, it=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for (hw, cin, cout) in ((56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128)):
    m = N * hw * hw
    x = torch.randn(N, hw, hw, cin, device=dev).to(dt)
    y = torch.empty_like(x)
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(dt)
    out = torch.empty(N, hw, hw, cout, device=dev, dtype=dt)
    stats = torch.rand(4, cin, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def run(chunks):
        per = N // chunks
        for c in range(chunks):
            sl = slice(c * per, (c + 1) * per)
            ops.bn_apply(x[sl].view(-1, cin), stats, y[sl].view(-1, cin), relu=True)
            ops.conv_forward(y[sl], w, 1, 1, 1, 0, out=out[sl])
    t_bn = timed(lambda: ops.bn_apply(x.view(-1, cin), stats, y.view(-1, cin), relu=True))
    t_cv = timed(lambda: ops.conv_forward(y, w, 1, 1, 1, 0, out=out))
    res = {c: timed(lambda: run(c)) for c in (1, 2, 4, 8, 16)}
    print("%dx%d %d->%d  y %.0f MB out %.0f MB: bn_apply alone %.3f, conv alone %.3f; pair by chunks: " % (hw, hw, cin, cout, m * cin * 2 / 1e6, m * cout * 2 / 1e6, t_bn, t_cv)
          + "  ".join("%d: %.3f" % (c, res[c]) for c in res), flush=True)
