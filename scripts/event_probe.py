"""What does a cross-stream event record cost the stream it is recorded on?  (probe, round 5)
A chain of N short kernels on one stream, with between every two of them: nothing / a torch event record / the same with
a second stream waiting for it / HIP events created with hipEventDisableSystemFence or hipEventReleaseToDevice.
Per-link cost = (chain time - plain chain time) / N."""
import ctypes
import sys
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")
DISABLE_TIMING, NO_SYS_FENCE, REL_DEVICE, REL_SYSTEM = 0x2, 0x20000000, 0x40000000, 0x80000000
dev = torch.device("cuda", 0)
N = 400
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.zeros(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev)
y = torch.zeros(4 * 1024 * 1024, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()


def hip_events(flags, n):
    evs = []
    for _ in range(n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
        evs.append(e)
    return evs


def chain(kind, evs=None, waiter=False):
    cur = torch.cuda.current_stream()
    for i in range(N):
        x.add_(1)
        if kind == "torch":
            e = torch.cuda.Event()
            e.record()
            if waiter:
                side.wait_event(e)
                with torch.cuda.stream(side):
                    y.add_(1)
        elif kind == "hip":
            e = evs[i]
            assert hip.hipEventRecord(e, ctypes.c_void_p(cur.cuda_stream)) == 0
            if waiter:
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), e, 0) == 0
                with torch.cuda.stream(side):
                    y.add_(1)


def timeit(tag, *a, **k):
    chain(*a, **k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chain(*a, **k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e6
    print("%-62s %7.2f us per link" % (tag, dt), flush=True)
    return dt


print("kernel: in-place add over %d MB" % mb)
timeit("plain chain", "none")
timeit("torch event record between kernels", "torch")
timeit("torch event record + side stream waits and runs a kernel", "torch", waiter=True)
for name, fl in (("default flags", 0), ("disable timing", DISABLE_TIMING), ("disable timing | no system fence", DISABLE_TIMING | NO_SYS_FENCE),
                 ("disable timing | release to device", DISABLE_TIMING | REL_DEVICE), ("disable timing | release to system", DISABLE_TIMING | REL_SYSTEM)):
    evs = hip_events(fl, N)
    timeit("hip event (%s)" % name, "hip", evs)
    timeit("hip event (%s) + side stream waits" % name, "hip", evs, waiter=True)
timeit("plain chain again", "none")
