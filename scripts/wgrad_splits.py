"""Weight-gradient time against the split-K count (slabs written + reduced vs. blocks in flight)."""
import sys
sys.path.insert(0, '.')
import torch
from iif_amd import ops
dev = 'cuda:0'
ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (n, h, cin, cout, k, stride) in ((256, 7, 512, 2048, 1, 1), (256, 7, 2048, 512, 1, 1), (256, 14, 256, 1024, 1, 1), (256, 14, 1024, 256, 1, 1),
                                     (256, 28, 128, 512, 1, 1), (256, 28, 512, 128, 1, 1), (256, 14, 256, 256, 3, 1), (256, 7, 512, 512, 3, 1)):
    pad = k // 2
    x = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, h, h, cout, device=dev).to(torch.bfloat16)
    out = []
    for splits in (0, 4, 8, 16, 32, 64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(21):
            if it == 1:
                e0.record()
            ops.conv_wgrad(x, dy, k, k, stride, pad, workspace=ws, splits=splits)
        e1.record(); torch.cuda.synchronize()
        out.append("%s:%.3f" % ("auto" if splits == 0 else splits, e0.elapsed_time(e1) / 20))
    print("h%d %d->%d k%d  " % (h, cin, cout, k) + "  ".join(out))
