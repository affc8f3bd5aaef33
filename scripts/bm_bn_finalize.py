"""Latency of the batch-norm statistics finalisation (partial rows -> mean / invstd / running statistics) on the ResNet50
bs-256 shapes: (tile rows, channels).  47 such launches sit on the compute stream's critical path of every step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops
dev = "cuda:0"


def timed(f, it=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


empty = torch.zeros(1, device=dev)
t0 = timed(lambda: empty.add_(1.0))
print("smallest torch kernel back to back: %.1f us" % t0)
for (nt, c) in ((25088, 64), (6272, 64), (6272, 256), (1568, 128), (1568, 512), (392, 256), (392, 1024), (98, 512), (98, 2048)):
    partial = torch.randn(nt, 2, c, device=dev).abs()
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    stats = torch.empty(4, c, device=dev)
    scratch = torch.empty(128 * c, device=dev)
    tickets = torch.zeros(64, dtype=torch.int32, device=dev)
    m = nt * 128
    t_f = timed(lambda: ops.bn_finalize_stats(partial, nt, m, c, gamma, beta, rm, rv, stats, scratch=scratch, tickets=tickets))
    t_2 = timed(lambda: ops.bn_finalize_stats(partial, nt, m, c, gamma, beta, rm, rv, stats, scratch=scratch))
    print("rows %5d C %4d (%.1f MB): one launch %.1f us, two launches %.1f us" % (nt, c, nt * 2 * c * 4 / 1e6, t_f, t_2), flush=True)
