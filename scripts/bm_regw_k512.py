"""The K = 512 shapes of the register-weight 1x1 kernel, alone: conv3 of the 7x7 bottlenecks (512 -> 2048, batch 256) and of
ResNeXt-101's 14x14 ones (512 -> 1024, batch 128), plain and with the BN + ReLU prologue.  Run once as is (32 columns per
wave, 32-row tiles) and once with IIF_REGW_K512_CW16=1 (round 5: 16 columns, 64-row tiles).
   python scripts/bm_regw_k512.py"""
import os
import sys
sys.path.insert(0, ".")
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
print("IIF_REGW_K512_CW16 =", os.environ.get("IIF_REGW_K512_CW16"))
for B, hw, c, C in ((256, 7, 512, 2048), (128, 14, 512, 1024), (256, 14, 512, 1024)):
    m = B * hw * hw
    raw, w3 = R(B, hw, hw, c), R(C, c) * 0.05
    a2 = torch.relu(raw)
    out = torch.empty(B, hw, hw, C, dtype=torch.bfloat16, device=dev)
    act = torch.empty_like(raw)
    bits = torch.empty(m * c // 8, dtype=torch.uint8, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * C, device=dev)
    stats = torch.rand(4, c, generator=g).to(dev) + 0.5
    t0 = timeit(lambda: ops.conv_forward_bnstats(a2, w3, 1, 1, 1, 0, out, partial))
    t1 = float("nan")
    if ops.conv_pro_ok(B, hw, hw, c, C, torch.bfloat16, False):
        t1 = timeit(lambda: ops.conv_forward_bnstats_pro(raw, stats, act, bits, w3, out, partial))
    fl = 2.0 * m * c * C
    print("  B %3d %2dx%2d %4d -> %4d   forward + sums %6.1f us (%5.1f TF/s)   with prologue %6.1f us (%5.1f TF/s)" % (
        B, hw, hw, c, C, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
