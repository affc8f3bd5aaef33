import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iif_amd import _lib
dev = "cuda:0"
L = _lib.lib()
for (B, C) in ((256, 1000), (128, 100), (1024, 1204), (2048, 1204), (8192, 1204)):
    x = torch.randn(B, C, device=dev); tab = torch.rand(C, device=dev) * 5 + 0.5
    y = torch.randint(0, C, (B,), device=dev)
    rows = torch.empty(B, device=dev); loss = torch.empty((), device=dev); d = torch.empty_like(x)
    ws = torch.zeros(2049, dtype=torch.int32, device=dev)
    def f():
        L.iif_ce_fwd_bwd(x.data_ptr(), 0, C, tab.data_ptr(), y.data_ptr(), 0, 1.0, 0, 0, -100, 1.0 / B, B, C, rows.data_ptr(), loss.data_ptr(), d.data_ptr(), C, 0, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    g = torch.cuda.CUDAGraph()
    f(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(50):
            f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print("[%5d,%5d] %.2f us per launch (50 back-to-back launches in a graph)" % (B, C, a.elapsed_time(b) * 20))
