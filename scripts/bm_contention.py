"""What two kernels of the step do to each other when they run side by side on two HIP streams.

For every pair (A on the compute stream, B on the weight-gradient stream) of the backward pass's most common overlaps:
time of A alone, of B alone, and of na launches of A next to nb launches of B (na, nb chosen so that both streams are busy
for about the same time alone), all as algorithmic bytes per second.  "sum alone" is what the pair would take serialised;
"paired" what it takes overlapped: paired / sum-alone = 1 means the overlap bought nothing.

    python scripts/bm_contention.py            # timings
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cont_kt -- python3 scripts/bm_contention.py --once
                                               # one launch of each kernel: LDS / VGPR / grid per kernel in the trace
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iif_amd import ops

dev = "cuda:0"
once = "--once" in sys.argv
g = torch.Generator().manual_seed(0)


def bf(*s):
    return torch.randn(*s, generator=g).bfloat16().to(dev)


def mk_bn_bwd_apply(n, hw, c):
    """bn_bwd_apply through the fused-sums entry (partial rows from the producer): reads g, x, bits; writes dx."""
    m = n * hw * hw
    gy, x, dx = bf(m, c), bf(m, c), torch.empty(m, c, dtype=torch.bfloat16, device=dev)
    bits = torch.randint(0, 256, (m * c // 8,), dtype=torch.uint8, generator=g).to(dev)
    stats = (torch.rand(4, c, generator=g) + 0.5).to(dev)
    gamma = torch.ones(c, device=dev)
    dgam, dbet = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    nt = (m + 127) // 128
    partial = torch.zeros(nt * 2 * c, device=dev)
    ws = ops.bn_workspace(m, c, dev)
    tk = torch.zeros(64, dtype=torch.int32, device=dev)
    f = lambda: ops.bn_backward_partials(gy, bits, x, stats, gamma, partial, nt, dgam, dbet, dx, ws, tickets=tk)  # noqa: E731
    return "bn_bwd_apply %dx%d^2x%d" % (n, hw, c), f, m * c * 6 + m * c // 8


def mk_bn_apply(n, hw, c):
    m = n * hw * hw
    x, res, y = bf(m, c), bf(m, c), torch.empty(m, c, dtype=torch.bfloat16, device=dev)
    bits = torch.empty(m * c // 8, dtype=torch.uint8, device=dev)
    stats = (torch.rand(4, c, generator=g) + 0.5).to(dev)
    f = lambda: ops.bn_apply(x, stats, y, relu=True, residual=res, relu_bits=bits)  # noqa: E731
    return "bn_apply+res %dx%d^2x%d" % (n, hw, c), f, m * c * 6 + m * c // 8


def mk_dgrad_masksum(n, hw, c, C):
    """conv1 data gradient c -> C with the gated store and the upstream sums (gemm1x1_regw_kernel<.., true>)."""
    m = n * hw * hw
    dy = bf(n, hw, hw, c)
    wtt = (torch.randn(C, c, generator=g) / C ** 0.5).bfloat16().to(dev)
    res, upx = bf(n, hw, hw, C), bf(n, hw, hw, C)
    ubits = torch.randint(0, 256, (m * C // 8,), dtype=torch.uint8, generator=g).to(dev)
    stats = (torch.rand(4, C, generator=g) + 0.5).to(dev)
    out = torch.empty(n, hw, hw, C, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * C, device=dev)
    f = lambda: ops.conv_dgrad_masksum(dy, wtt, (hw, hw), out, ubits, partial, res=res, up_x=upx, up_stats=stats)  # noqa: E731
    return "dgrad1x1+epi %d^2 %d->%d" % (hw, c, C), f, m * (c + 3 * C) * 2 + m * C // 8


def mk_dgrad2(n, hw, C, c):
    """conv3 data gradient on the algebra route: [g~ | a2] (K = C + c) -> c, with bn2's backward sums."""
    m = n * hw * hw
    gt, a2 = bf(n, hw, hw, C), bf(n, hw, hw, c)
    wt = (torch.randn(c, C + c, generator=g) / C ** 0.5).bfloat16().to(dev)
    bias = torch.zeros(c, device=dev)
    upx = bf(n, hw, hw, c)
    ubits = torch.randint(0, 256, (m * c // 8,), dtype=torch.uint8, generator=g).to(dev)
    stats = (torch.rand(4, c, generator=g) + 0.5).to(dev)
    out = torch.empty(n, hw, hw, c, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
    f = lambda: ops.conv_dgrad2_bnbwd(gt, a2, wt, bias, out, upx, ubits, stats, partial)  # noqa: E731
    return "dgrad2 %d^2 %d+%d->%d" % (hw, C, c, c), f, m * (C + 3 * c) * 2


def mk_dgrad3x3(n, hw, c):
    m = n * hw * hw
    dy = bf(n, hw, hw, c)
    wt = (torch.randn(c, 9 * c, generator=g) / (9 * c) ** 0.5).bfloat16().to(dev)
    upx = bf(n, hw, hw, c)
    ubits = torch.randint(0, 256, (m * c // 8,), dtype=torch.uint8, generator=g).to(dev)
    stats = (torch.rand(4, c, generator=g) + 0.5).to(dev)
    out = torch.empty(n, hw, hw, c, dtype=torch.bfloat16, device=dev)
    partial = torch.zeros(((m + 127) // 128 + 8) * 2 * c, device=dev)
    f = lambda: ops.conv_dgrad_bnbwd(dy, wt, 3, 3, 1, 1, (hw, hw), out, upx, ubits, stats, partial)  # noqa: E731
    return "dgrad3x3 %d^2 %d" % (hw, c), f, m * c * 3 * 2


def mk_wgrad1x1(n, hw, ci, co):
    x, dy = bf(n, hw, hw, ci), bf(n, hw, hw, co)
    out = torch.zeros(co, ci, device=dev)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    f = lambda: ops.conv_wgrad(x, dy, 1, 1, 1, 0, ldw=ci, out=out, workspace=ws)  # noqa: E731
    return "wgrad1x1 %d^2 %dx%d" % (hw, ci, co), f, n * hw * hw * (ci + co) * 2


def mk_wgrad3x3(n, hw, c):
    x, dy = bf(n, hw, hw, c), bf(n, hw, hw, c)
    out = torch.zeros(c, 9 * c, device=dev)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    f = lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1, ldw=9 * c, out=out, workspace=ws)  # noqa: E731
    return "wgrad3x3 %d^2 %d" % (hw, c), f, n * hw * hw * 2 * c * 2


def time_alone(f, it):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3          # us


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def time_pair(fa, na, fb, nb):
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    s1.wait_event(start)
    s2.wait_event(start)
    # interleave the enqueues so that neither stream starts far ahead of the other
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na and ia * nb <= ib * na:
            with torch.cuda.stream(s1):
                fa()
            ia += 1
        else:
            with torch.cuda.stream(s2):
                fb()
            ib += 1
    ea.record(s1)
    eb.record(s2)
    torch.cuda.synchronize()
    return max(start.elapsed_time(ea), start.elapsed_time(eb)) * 1e3, start.elapsed_time(ea) * 1e3, start.elapsed_time(eb) * 1e3


N = 256
PAIRS = [
    (lambda: mk_bn_bwd_apply(N, 56, 64), lambda: mk_wgrad1x1(N, 56, 64, 256)),
    (lambda: mk_bn_bwd_apply(N, 28, 128), lambda: mk_wgrad1x1(N, 14, 256, 1024)),
    (lambda: mk_dgrad_masksum(N, 56, 64, 256), lambda: mk_wgrad1x1(N, 56, 256, 64)),
    (lambda: mk_dgrad_masksum(N, 14, 256, 1024), lambda: mk_wgrad1x1(N, 14, 1024, 256)),
    (lambda: mk_dgrad2(N, 14, 1024, 256), lambda: mk_wgrad1x1(N, 14, 256, 1024)),
    (lambda: mk_dgrad2(N, 28, 512, 128), lambda: mk_wgrad3x3(N, 28, 128)),
    (lambda: mk_dgrad3x3(N, 14, 256), lambda: mk_wgrad1x1(N, 14, 256, 1024)),
    (lambda: mk_dgrad3x3(N, 28, 128), lambda: mk_wgrad3x3(N, 28, 128)),
    (lambda: mk_bn_bwd_apply(N, 56, 64), lambda: mk_wgrad3x3(N, 56, 64)),
    (lambda: mk_bn_apply(N, 56, 256), lambda: mk_bn_bwd_apply(N, 56, 64)),
]

print("%-28s %-24s | A alone      B alone      | paired: wall / serialised   TB/s paired (alone A, B)" % ("A (compute stream)", "B (side stream)"))
for (ma, mb) in PAIRS:
    na_, fa, ba = ma()
    nb_, fb, bb = mb()
    if once:
        fa(); fb()
        torch.cuda.synchronize()
        continue
    ta, tb = time_alone(fa, 20), time_alone(fb, 20)
    # ~3 ms of work per stream
    na, nb = max(2, int(round(3000.0 / ta))), max(2, int(round(3000.0 / tb)))
    time_pair(fa, na, fb, nb)
    wall, wa, wb = time_pair(fa, na, fb, nb)
    ser = na * ta + nb * tb
    tot = na * ba + nb * bb
    print("%-28s %-24s | %6.1f us %4.2f  %6.1f us %4.2f | %7.0f / %7.0f = %.2f    %.2f (%.2f, %.2f)   [A done %.0f, B done %.0f]" % (
        na_, nb_, ta, ba / ta / 1e6, tb, bb / tb / 1e6, wall, ser, wall / ser, tot / wall / 1e6, ba / ta / 1e6, bb / tb / 1e6, wa, wb))
    del fa, fb
    torch.cuda.empty_cache()
