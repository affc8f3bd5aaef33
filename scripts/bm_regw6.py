"""Round 6 modes of conv_regw.hip, each alone at the benchmark's shapes (batch 256): the statistics pass, the BN-epilogue pass, the
data-gradient producer in its three forms (sums of g~ only / reading the stored upstream output / recomputing it).
   python scripts/bm_regw6.py [batch]"""
import sys
sys.path.insert(0, ".")
import torch
from iif_amd import ops
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(f, n=30):
    for _ in range(3):
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
print("forward: conv3 c -> C at hw (statistics pass | BN epilogue pass, identity | normalised shortcut)")
for hw, c, C in ((56, 64, 256), (28, 128, 512), (14, 256, 1024)):
    m = B * hw * hw
    a2, w3, res = torch.relu(R(B, hw, hw, c)), R(C, c) * 0.1, R(B, hw, hw, C)
    out = torch.empty(B, hw, hw, C, dtype=torch.bfloat16, device=dev)
    bits = torch.empty(m * C // 8, dtype=torch.uint8, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * C, device=dev)
    stats = torch.rand(4, C, generator=g).to(dev) + 0.5
    t1 = timeit(lambda: ops.conv_forward_stats_acc(a2, w3, partial))
    t2 = timeit(lambda: ops.conv_forward_bn_relu2(a2, w3, out, stats, bits, res=res))
    t3 = timeit(lambda: ops.conv_forward_bn_relu2(a2, w3, out, stats, bits, res=res, res_stats=stats))
    mb1, mb2 = m * c * 2 / 1e6, m * (c + 2 * C) * 2 / 1e6
    print("  %2dx%2d %4d -> %4d   stats %6.1f us (%4.0f GB/s, %5.1f TF/s)   bn+id %6.1f us (%4.0f GB/s)   bn+shortcut %6.1f us" % (
        hw, hw, c, C, t1, mb1 / t1 * 1e3, 2.0 * m * c * C / t1 / 1e6, t2, mb2 / t2 * 1e3, t3))
print("backward: producer conv1 dgrad k -> C with residual (gated store + sums | + stored upstream x | + recomputed upstream x over c2)")
for hw, k, C, c2 in ((56, 64, 256, 64), (56, 128, 256, 64), (28, 128, 512, 128), (28, 256, 512, 128), (14, 256, 1024, 256)):
    m = B * hw * hw
    dy, wt, res = R(B, hw, hw, k), R(C, k) * 0.1, R(B, hw, hw, C)
    ub = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(dev)
    a2, w3 = torch.relu(R(B, hw, hw, c2)), R(C, c2) * 0.1
    y3 = ops.conv_forward(a2, w3, 1, 1, 1, 0)
    out = torch.empty(B, hw, hw, C, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * C, device=dev)
    stats = torch.rand(4, C, generator=g).to(dev) + 0.5
    t1 = timeit(lambda: ops.conv_dgrad_masksum(dy, wt, (hw, hw), out, ub, partial, res=res))
    t2 = timeit(lambda: ops.conv_dgrad_masksum(dy, wt, (hw, hw), out, ub, partial, res=res, up_x=y3, up_stats=stats))
    t3 = float("nan")
    if ops.conv_dgrad_rx_ok(B, hw, hw, k, C, c2, torch.bfloat16):
        t3 = timeit(lambda: ops.conv_dgrad_masksum_rx(dy, wt, (hw, hw), out, ub, partial, a2, w3, stats, res=res))
    t4 = float("nan")
    if ops.conv_dgrad_rx_pg_ok(B, hw, hw, k, C, c2, torch.bfloat16):
        slabs = torch.empty(273 * (C + c2) * c2, device=dev)
        t4 = timeit(lambda: ops.conv_dgrad_masksum_rx_pg(dy, wt, (hw, hw), out, ub, partial, a2, w3, stats, slabs, c2, res=res))
    mb = m * (k + 2 * C) * 2 / 1e6
    print("  %2dx%2d %4d -> %4d (c2 %3d)   sums %6.1f us (%4.0f GB/s)   stored x %6.1f us (%4.0f GB/s)   recomputed x %6.1f us (%4.0f GB/s)"
          "   + P and Gram %6.1f us" % (
              hw, hw, k, C, c2, t1, mb / t1 * 1e3, t2, (mb + m * C * 2 / 1e6) / t2 * 1e3, t3, (mb + m * c2 * 2 / 1e6) / t3 * 1e3, t4))
