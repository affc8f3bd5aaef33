"""Probe: the side streams (weight gradients, shortcut branch) confined to a subset of the CUs by a CU mask
(hipExtStreamCreateWithCUMask), the compute stream on all of them.   python scripts/cumask_probe.py"""
import ctypes, sys, time
sys.path.insert(0, '.')
import torch
from iif_amd import resnet_pytorch
from iif_amd.custom import IIFLoss
import bench as B
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device('cuda', 0)
C, bs = 1000, 256
counts = B.lt_counts(C, 1280)
crit = IIFLoss(B._Counts(counts), variant="raw", reduction="mean", device=dev)
x = torch.randn(bs, 3, 224, 224).to(dev); y = torch.randint(0, C, (bs,)).to(dev)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def run(tag, wg_bits=None, ds_bits=None):
    net = resnet_pytorch.resnet50(num_classes=C, use_norm="None", pretrained="None", device=dev, compute_dtype=torch.bfloat16)
    net.train()
    plan = net._plan(bs, 224, 224)
    if wg_bits is not None:
        plan.wg_stream = masked_stream(wg_bits)
    if ds_bits is not None:
        plan.ds_stream = masked_stream(ds_bits)
    def step():
        net.loss_and_backward(x, y, crit)
        net.sgd_step(1e-4, 0.9, 1e-4)
    for _ in range(6): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (tag, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del net


ALL = (1 << 256) - 1
def every(k, off=0):      # every k-th CU
    return sum(1 << i for i in range(256) if i % k == off)
def low(n):
    return (1 << n) - 1
run("no masks")
run("side streams: all CUs (masked API)", ALL, ALL)
run("wgrad: low 128 CUs", low(128), low(128))
run("wgrad: low 64 CUs", low(64), low(64))
run("wgrad: every 2nd CU", every(2), every(2))
run("wgrad: every 4th CU", every(4), every(4))
run("wgrad: every 2nd, shortcut: other half", every(2), every(2, 1))
run("wgrad: low 96 CUs", low(96), low(96))
run("no masks")
