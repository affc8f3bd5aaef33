"""Phase shares of gemm1x1_stream_kernel from the diagnostic build (make -C iif_amd/csrc stamps).
    python scripts/stream_stamps.py fwd|dgrad N H Cin Cout"""
import os, sys, ctypes
sys.path.insert(0, '.')
import torch
from iif_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "_debug", "libiif_amd_stamps.so")
from iif_amd import ops
kind, n, h, cin, cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev, dt = 'cuda:0', torch.bfloat16
m = n * h * h
x = torch.randn(n, h, h, cin, device=dev).to(dt)
w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(dt)
out = torch.empty(n, h, h, cout, device=dev, dtype=dt)
partial = torch.empty(((m + 127) // 128 + 8) * 2 * max(cin, cout), device=dev)
dy = torch.randn(n, h, h, cout, device=dev).to(dt)
wtt = torch.zeros(cin, (cout + 15) // 16 * 16, dtype=dt, device=dev)
ops.weight_transpose(w.float(), cout, cin, 1, wtt)
dx = torch.empty(n, h, h, cin, device=dev, dtype=dt)
upx = torch.randn(n, h, h, cin, device=dev).to(dt)
bits = torch.randint(0, 255, (m * cin // 8,), device=dev, dtype=torch.uint8)
stats = torch.rand(4, cin, device=dev)
stamps = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.iif_debug_set_stamps.argtypes = [ctypes.c_void_p]; lib.iif_debug_set_stamps.restype = ctypes.c_int
assert lib.iif_debug_set_stamps(stamps.data_ptr()) == 0


def run():
    if kind == 'fwd':
        ops.conv_forward_bnstats(x, w, 1, 1, 1, 0, out, partial)
    else:
        ops.conv_dgrad_bnbwd(dy, wtt, 1, 1, 1, 0, (h, h), dx, upx, bits, stats, partial)


run(); torch.cuda.synchronize(); stamps.zero_(); run(); torch.cuda.synchronize()
s = stamps.view(512, 4, 8).cpu().double()
c = s[:, 0][s[:, 0, 5] > 0]; st = s[:, 1][s[:, 1, 5] > 0]
tiles = c[:, 6].mean().item()
print("%s n%d h%d %d->%d: %d blocks, %.1f tiles per block; ticks are 10 ns" % (kind, n, h, cin, cout, c.shape[0], tiles))
tot = c[:, 5].mean().item()
for i, nm in enumerate(["DMA wait", "frag reads + MFMA + refill issue", "barrier A wait (drain of the previous tile)", "staging write", "barrier B wait"]):
    v = c[:, i].mean().item()
    print("  compute wave: %-46s %8.0f ticks %5.1f %%  %6.2f us per tile" % (nm, v, 100 * v / tot, v / tiles / 100))
print("  compute wave total %.0f ticks = %.1f us; per tile %.2f us" % (tot, tot / 100, tot / tiles / 100))
tot2 = st[:, 5].mean().item()
for i, nm in enumerate(["barrier waits (A..B)", "drain loop"]):
    v = st[:, i].mean().item()
    print("  store wave:   %-46s %8.0f ticks %5.1f %%  %6.2f us per tile" % (nm, v, 100 * v / tot2, v / tiles / 100))
print("  store wave total %.0f ticks" % tot2)
print("  block start spread: %.1f us" % ((c[:, 7].max() - c[:, 7].min()).item() / 100))
