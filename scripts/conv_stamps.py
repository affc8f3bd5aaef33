"""Phase shares of the LDS-DMA convolution kernel from the diagnostic build (`make -C iif_amd/csrc stamps`):
per wave, cycles spent (a) waiting for the DMA of the step + the block barrier, (b) reading fragments from LDS,
(c) issuing the refill DMA, (d) issuing the step's MFMAs, (e) in the epilogue.
    python scripts/conv_stamps.py fwd 256 14 256 256 3 1
"""
import os
import sys
sys.path.insert(0, '.')
import ctypes
import torch
from iif_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "_debug", "libiif_amd_stamps.so")
from iif_amd import ops

kind, n, h, cin, cout, k, stride = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
pad = k // 2
dev = 'cuda:0'
x = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5).to(torch.bfloat16)
ho = (h + 2 * pad - k) // stride + 1
dy = torch.randn(n, ho, ho, cout, device=dev).to(torch.bfloat16)
wt = torch.randn(cin, k * k * cout, device=dev).to(torch.bfloat16)
stamps = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.iif_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.iif_debug_set_stamps.restype = ctypes.c_int
assert lib.iif_debug_set_stamps(stamps.data_ptr()) == 0
lib.iif_debug_set_wgrad_stamps.argtypes = [ctypes.c_void_p]
lib.iif_debug_set_wgrad_stamps.restype = ctypes.c_int
assert lib.iif_debug_set_wgrad_stamps(stamps.data_ptr()) == 0
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def run():
    if kind == 'fwd':
        ops.conv_forward(x, w, k, k, stride, pad)
    elif kind == 'wgrad':          # columns: wait = vmcnt+barrier, DMA issue = address arithmetic + DMA, MFMA issue = fragments + MFMA
        ops.conv_wgrad(x, dy, k, k, stride, pad, workspace=ws)
    else:
        ops.conv_dgrad(dy, wt, k, k, stride, pad, (h, h))


run(); torch.cuda.synchronize()
stamps.zero_()
run(); torch.cuda.synchronize()
s = stamps.view(512, 4, 8).cpu().double()
s = s[s[:, :, 5] > 0]
tot = s[:, 5].mean().item()
nk = s[:, 6].mean().item()
names = ["wait(vmcnt+barrier)", "LDS fragment reads", "refill DMA issue", "MFMA issue", "epilogue"]
print("%s n%d h%d %d->%d k%d s%d: %d waves sampled, %.0f K steps, %.0f cycles per wave (100 MHz s_memtime ticks x?)" % (kind, n, h, cin, cout, k, stride, s.shape[0], nk, tot))
for i, nm in enumerate(names):
    v = s[:, i].mean().item()
    print("  %-22s %9.0f ticks  %5.1f %%   %7.1f per K step" % (nm, v, 100 * v / tot, v / max(nk, 1)))
acc = sum(s[:, i].mean().item() for i in range(5))
print("  %-22s %9.0f ticks  %5.1f %%" % ("(prologue, unaccounted)", tot - acc, 100 * (tot - acc) / tot))
