"""HBM rate of the fused IIF loss kernel (iif_ce_fwd_bwd) and of the BN / SGD streaming kernels at sizes large
enough to be bandwidth-bound; algorithmic bytes / HIP-event time.  Prints one line per kernel."""
import sys
sys.path.insert(0, '.')
import torch
from iif_amd import ops
from iif_amd.custom import IIFLoss

dev = 'cuda:0'


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class DS:
    def __init__(self, c): self.c = c
    def get_cls_num_list(self): return self.c


for (B, C, dt) in ((256, 1000, torch.float32), (1024, 1204, torch.float32), (2048, 1204, torch.float32), (65536, 1000, torch.float32), (65536, 1000, torch.bfloat16),
                   (16384, 8142, torch.bfloat16)):
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    crit = IIFLoss(DS(counts), variant="raw", device=dev)
    x = torch.randn(B, C, device=dev).to(dt).requires_grad_(True)
    y = torch.randint(0, C, (B,), device=dev)

    def f():
        x.grad = None
        crit(x, y).backward()
    from iif_amd import custom
    # the kernel alone: logits read once, gradient written once (fp32 grad)
    tab = crit.iif["raw"]
    def k():
        custom._launch_ce(x.detach(), tab, y, None, 1.0, None, None, -100, 1.0 / B, True)
    ms = timed(k)
    byt = B * C * (x.element_size() + x.element_size()) + 8 * B + 4 * C
    print("iif_ce_fwd_bwd  B=%6d C=%5d %-8s %8.3f ms  %7.1f GB/s (algorithmic %6.1f MB)" % (B, C, str(dt).split('.')[-1], ms, byt / ms / 1e6, byt / 1e6))

M, Cc = 256 * 56 * 56, 256
for dt in (torch.bfloat16,):
    x = torch.randn(M, Cc, device=dev).to(dt)
    y = torch.empty_like(x)
    r = torch.randn(M, Cc, device=dev).to(dt)
    stats = torch.randn(4, Cc, device=dev)
    bits = torch.empty(M * Cc // 8, dtype=torch.uint8, device=dev)
    ms = timed(lambda: ops.bn_apply(x, stats, y, relu=True, residual=r, relu_bits=bits))
    byt = M * Cc * 2 * 3 + M * Cc // 8
    print("bn_apply(+res,+relu,+bits) [%d,%d] bf16   %8.3f ms  %7.1f GB/s" % (M, Cc, ms, byt / ms / 1e6))
    gam = torch.ones(Cc, device=dev); dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
    ws = ops.bn_workspace(M, Cc, dev)
    dx = torch.empty_like(x)
    ms = timed(lambda: ops.bn_backward(r, None, x, stats, gam, dg, db, dx, ws, relu_bits=bits))
    byt = M * Cc * 2 * 5 + 2 * (M * Cc // 8)
    print("bn_backward(reduce+apply)  [%d,%d] bf16   %8.3f ms  %7.1f GB/s" % (M, Cc, ms, byt / ms / 1e6))
n = 25_600_000
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev); m = torch.zeros(n, device=dev)
ms = timed(lambda: ops.sgd_step(p, g, m, 0.1, 0.9, 1e-4))
print("sgd_step %d params                         %8.3f ms  %7.1f GB/s" % (n, ms, 20.0 * n / ms / 1e6))
