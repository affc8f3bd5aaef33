"""A/B of the generation-2 3x3 kernel (weights as MFMA fragments) against the round-1/2 kernels (halo / tap kernels) on the
ResNet50 bs-256 3x3 / stride-1 layers: forward with fused BN statistics and data gradient with the upstream BN-backward sums,
each alone on the GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import ops, _lib
dev = "cuda:0"
dt = torch.bfloat16
N = int(os.environ.get("BM_BATCH", "256"))


def timed(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def mode(m):
    os.environ.pop("IIF_CONV_NO_V2", None)
    if m == "old":
        os.environ["IIF_CONV_NO_V2"] = "1"
    _lib.check(_lib.lib().iif_conv_reload_env(), "reload")


def pack(w2d, rows, k):
    tab, blocks = ops.pack_table([(0, 0, rows, 9, k, w2d.shape[1])], dev)
    return ops.pack_fragments(w2d, tab, 1, blocks, torch.empty(rows * 9 * k, dtype=dt, device=dev))


tot = {"old": 0.0, "v2": 0.0}
ONLY = [int(v) for v in os.environ.get("BM_ONLY", "").split(",") if v]
for (hw, c, cnt) in ((56, 64, 3), (28, 128, 3), (14, 256, 5), (7, 512, 2)):
    if ONLY and hw not in ONLY:
        continue
    m = N * hw * hw
    x = torch.randn(N, hw, hw, c, device=dev).to(dt)
    w = (torch.randn(c, 9 * c, device=dev) / (9 * c) ** 0.5).to(dt)
    wf = pack(w, c, c)
    out = torch.empty(N, hw, hw, c, device=dev, dtype=dt)
    partial = torch.empty(((m + 127) // 128 + 8) * 2 * c, device=dev)
    upx = torch.randn(N, hw, hw, c, device=dev).to(dt)
    bits = torch.randint(0, 255, (m * c // 8,), device=dev, dtype=torch.uint8)
    stats = torch.rand(4, c, device=dev)
    fl = 2.0 * m * c * 9 * c
    r = {}
    for s in ("old", "v2"):
        mode(s)
        r[s] = (timed(lambda: ops.conv_forward_bnstats(x, w, 3, 3, 1, 1, out, partial, w_frag=wf)),
                timed(lambda: ops.conv_dgrad_bnbwd(x, w, 3, 3, 1, 1, (hw, hw), out, upx, bits, stats, partial, w_frag=wf)))
        tot[s] += cnt * (r[s][0] + r[s][1])
    print("3x3 %2dx%-2d %3d ch   fwd+stats: old %.3f ms (%4.0f TF)  v2 %.3f ms (%4.0f TF)    dgrad+bw: old %.3f ms (%4.0f TF)  v2 %.3f ms (%4.0f TF)" % (
        hw, hw, c, r["old"][0], fl / r["old"][0] / 1e9, r["v2"][0], fl / r["v2"][0] / 1e9,
        r["old"][1], fl / r["old"][1] / 1e9, r["v2"][1], fl / r["v2"][1] / 1e9), flush=True)
print("weighted by launches per step (fwd + dgrad): old %.3f ms   v2 %.3f ms" % (tot["old"], tot["v2"]))
