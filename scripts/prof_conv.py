"""Run one convolution shape repeatedly (for rocprofv3 counter collection)."""
import sys
sys.path.insert(0, '.')
import torch
from iif_amd import ops
kind, n, h, cin, cout, k, stride = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
pad = k // 2
dev = 'cuda:0'
x = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5).to(torch.bfloat16)
ho = (h + 2 * pad - k) // stride + 1
dy = torch.randn(n, ho, ho, cout, device=dev).to(torch.bfloat16)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(reps + 1):
    if it == 1:
        e0.record()
    if kind == 'fwd':
        ops.conv_forward(x, w, k, k, stride, pad)
    elif kind == 'wgrad':
        ops.conv_wgrad(x, dy, k, k, stride, pad, workspace=ws)
    else:
        wt = (torch.randn(cin, k * k * cout, device=dev)).to(torch.bfloat16)
        ops.conv_dgrad(dy, wt, k, k, stride, pad, (h, h))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * ho * ho * cout * k * k * cin
print("%s n%d h%d %d->%d k%d s%d: %.3f ms  %.1f TFLOP/s" % (kind, n, h, cin, cout, k, stride, ms, fl / ms / 1e9))
