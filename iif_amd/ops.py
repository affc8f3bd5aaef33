"""Thin tensor-level wrappers over the C ABI (one call = one kernel launch on
torch's current HIP stream).  Layout everywhere: activations NHWC, weights
[Cout][R][S][Cin] rows with pitch ``ldw``.  No fallbacks: GPU tensors only."""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check, dtype_code, lib, ptr, require_gpu, stream_ptr


def _desc(n, hs, ws, cs, hd, wd, cd, r, s, stride, pad, transposed, ldw, dtype, dst_dtype):
    return ConvDesc(n, hs, ws, cs, hd, wd, cd, r, s, stride, pad, transposed, ldw, dtype, dst_dtype)


def conv_out_hw(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


def conv_forward(x, w, r, s, stride, pad, out=None, out_dtype=None, bias=None, res=None):
    """x: [N,H,W,Cin] NHWC; w: [Cout, ldw] (rows = r*s*Cin K-contiguous, KRSC)."""
    require_gpu(x, w, bias, res)
    n, h, wd_, cin = x.shape
    cout, ldw = w.shape
    ho, wo = conv_out_hw(h, wd_, r, s, stride, pad)
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=odt, device=x.device)
    d = _desc(n, h, wd_, cin, ho, wo, cout, r, s, stride, pad, 0, ldw, dtype_code(x), dtype_code(out))
    check(lib().iif_conv_igemm(ctypes.byref(d), ptr(x), ptr(w), ptr(out), ptr(res), ptr(bias), stream_ptr()), "iif_conv_igemm")
    return out


def conv_dgrad(dy, wt, r, s, stride, pad, in_hw, out=None, res=None):
    """dy: [N,Ho,Wo,Cout]; wt: [Cin, ldw] rows of r*s*Cout (the CRSK transpose);
    returns dx [N,H,W,Cin] (+ res)."""
    require_gpu(dy, wt, res)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    if out is None:
        out = torch.empty((n, h, w_, cin), dtype=dy.dtype, device=dy.device)
    d = _desc(n, ho, wo, cout, h, w_, cin, r, s, stride, pad, 1, ldw, dtype_code(dy), dtype_code(out))
    check(lib().iif_conv_igemm(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), 0, stream_ptr()), "iif_conv_igemm(dgrad)")
    return out


def conv_wgrad(x, dy, r, s, stride, pad, ldw=None, out=None, workspace=None, splits=0):
    """x: [N,H,W,Cin], dy: [N,Ho,Wo,Cout] -> dw float32 [Cout, ldw] (KRSC rows)."""
    require_gpu(x, dy, out, workspace)
    n, h, w_, cin = x.shape
    _, ho, wo, cout = dy.shape
    k = r * s * cin
    ldw = ldw or k
    if out is None:
        out = torch.zeros((cout, ldw), dtype=torch.float32, device=x.device)
    d = _desc(n, h, w_, cin, ho, wo, cout, r, s, stride, pad, 0, ldw, dtype_code(x), _lib.IIF_F32)
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(lib().iif_conv_wgrad(ctypes.byref(d), ptr(x), ptr(dy), ptr(out), ptr(workspace), wsb, splits, stream_ptr()),
          "iif_conv_wgrad")
    return out
