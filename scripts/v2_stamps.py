"""Phase shares of the 3x3 fragment kernel from the diagnostic build (`make -C iif_amd/csrc stamps`): per wave, cycles in the tile
prologue (geometry + first halo window + barrier), inside the tap loops, at the chunk boundaries (wait + barrier) and in the
epilogue (staging + drain).    IIF_CONV_V2_WIDE=1 python scripts/v2_stamps.py 256 14 256"""
import os, sys, ctypes
sys.path.insert(0, '.')
import torch
from iif_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "_debug", os.environ.get("IIF_STAMPS_LIB", "libiif_amd_stamps.so"))
from iif_amd import ops
n, h, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = 'cuda:0'
dt = torch.bfloat16
x = torch.randn(n, h, h, c, device=dev).to(dt)
w = (torch.randn(c, 9 * c, device=dev) / (9 * c) ** 0.5).to(dt)
tab, blocks = ops.pack_table([(0, 0, c, 9, c, 9 * c)], dev)
wf = ops.pack_fragments(w, tab, 1, blocks, torch.empty(c * 9 * c, dtype=dt, device=dev))
out = torch.empty(n, h, h, c, device=dev, dtype=dt)
m = n * h * h
partial = torch.empty(((m + 127) // 128 + 8) * 2 * c, device=dev)
stamps = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.iif_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.iif_debug_set_stamps(stamps.data_ptr()) == 0
f = lambda: ops.conv_forward_bnstats(x, w, 3, 3, 1, 1, out, partial, w_frag=wf)
for _ in range(3):
    f()
torch.cuda.synchronize(); stamps.zero_(); f(); torch.cuda.synchronize()
s = stamps.view(512, 4, 8).cpu().double()
s = s[s[:, :, 5] > 0]
tot = s[:, 5].mean().item()
tiles = s[:, 4].mean().item(); steps = s[:, 6].mean().item()
print("3x3 n%d %dx%d %d ch: %d waves, %.2f tiles per block, %d taps x chunks per tile, %.0f cycles per wave" % (n, h, h, c, s.shape[0], tiles, steps, tot))
for i, nm in enumerate(["prologue (geometry, halo 0, barrier)", "tap loops (MFMA pipeline)", "chunk boundaries (wait + barrier)", "epilogue (stage + drain)"]):
    v = s[:, i].mean().item()
    print("  %-38s %9.0f cycles %5.1f %%  %8.0f per tile" % (nm, v, 100 * v / tot, v / tiles))
loop = s[:, 1].mean().item()
print("  tap loop: %.0f cycles per tap (ideal 512 = 32 MFMAs x 16)" % (loop / tiles / steps))
