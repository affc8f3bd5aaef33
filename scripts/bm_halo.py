"""The 256-pixel 3x3 window kernel alone at the benchmark's 14x14 / 7x7 shapes: forward with statistics and the data gradient.
Run once as is (three taps per hand-over) and once with IIF_CONV_HALO_TPB1=1 (a hand-over per tap).   python scripts/bm_halo.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iif_amd import ops
dev = "cuda:0"


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(1)
R = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)   # noqa: E731
print("IIF_CONV_HALO_TPB1 =", os.environ.get("IIF_CONV_HALO_TPB1"))
for B, hw, c in ((256, 14, 256), (256, 7, 512), (128, 14, 256)):
    m = B * hw * hw
    x, w = R(B, hw, hw, c), R(c, 9 * c) * 0.02
    out = torch.empty(B, hw, hw, c, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
    t0 = timeit(lambda: ops.conv_forward_bnstats(x, w, 3, 3, 1, 1, out, partial))
    fl = 2.0 * m * c * 9 * c
    print("  B %3d %2dx%2d %4d ch   forward + sums %6.1f us (%5.1f TF/s)" % (B, hw, hw, c, t0, fl / t0 / 1e6))
