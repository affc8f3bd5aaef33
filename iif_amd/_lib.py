"""ctypes binding of libiif_amd.so (the C ABI declared in include/iif_amd.h).

The library is built in-tree (``iif_amd/csrc/libiif_amd.so``) by
``__graft_entry__.build()`` / ``make -C iif_amd/csrc``.  There is no CPU or
PyTorch fallback: if the library is missing, or a tensor is not on the GPU,
the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IIF_AMD_LIB: another build of the same C ABI (the stamped diagnostic build, A/B experiments)
LIB_PATH = os.environ.get("IIF_AMD_LIB") or os.path.join(_HERE, "csrc", "libiif_amd.so")

IIF_F32, IIF_BF16 = 0, 1
VARIANT_CODE = {"raw": 0, "smooth": 1, "rel": 2, "normit": 3, "gombit": 4, "base2": 5, "base10": 6}
_ERR = {-1: "IIF_EINVAL (bad argument)", -2: "IIF_EUNSUPPORTED (shape/alignment)", -3: "IIF_ELAUNCH (HIP error)"}

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> argtypes; must list every symbol of include/iif_amd.h (tests check this)
SIGNATURES = {
    "iif_build_table": [_P, _I, _I, _I, _P],
    "iif_set_cu_budget": [_I],
    "iif_get_cu_budget": [],
    "iif_ce_fwd_bwd": [_P, _I, _L, _P, _P, _P, _F, _P, _P, _L, _F, _I, _I, _P, _P, _P, _L, _P, _P, _P],
    "iif_scale_logits": [_P, _I, _L, _P, _I, _I, _P, _L, _P],
    "iif_softmax": [_P, _I, _L, _P, _I, _I, _P, _L, _P],
    "iif_topk_hits": [_P, _I, _L, _P, _P, _I, _I, _P, _I, _P, _P],
    "iif_scale_by_device_scalar": [_P, _I, _L, _P, _P, _P],
    "iif_mix_rows": [_P, _I, _P, _F, _I, _L, _P, _P],
    "iif_conv_igemm": [_P, _P, _P, _P, _P, _P, _P],
    "iif_conv_wgrad": [_P, _P, _P, _P, _P, _L, _I, _P],
    "iif_conv_igemm_bnstats": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "iif_bn_finalize_stats": [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _L, _P],
    "iif_bn_finalize_stats_fused": [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _L, _P, _P],
    "iif_bn_finalize_stats_sums": [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _L, _P, _P, _I, _I, _P, _P],
    "iif_bn_backward_partials_fused": [_P, _P, _P, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P, _P],
    "iif_bn_partial_sums": [_P, _I, _I, _P, _P],
    "iif_bn_stats_sums": [_P, _I, _L, _I, _P, _P, _L, _P],
    "iif_bn_backward_sums": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _L, _P],
    "iif_bn_backward_apply_sums": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _c.c_double, _P, _P, _P, _P, _P, _P],
    "iif_bn_workspace_bytes": [_L, _I],
    "iif_bn_forward_stats": [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _L, _P],
    "iif_bn_apply": [_P, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P],
    "iif_bn_backward": [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    "iif_maxpool_forward": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "iif_maxpool_backward": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_avgpool_forward": [_P, _I, _I, _I, _I, _P, _P],
    "iif_avgpool_backward": [_P, _I, _I, _I, _I, _P, _P],
    "iif_im2col_nchw": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_cast": [_P, _I, _P, _I, _L, _P],
    "iif_weight_transpose": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_shortcut_a_forward": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_shortcut_a_backward_acc": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_colsum_f32": [_P, _I, _I, _L, _P, _P],
    "iif_sgd_step": [_P, _P, _P, _L, _F, _P, _F, _F, _I, _F, _P],
    "iif_group_pack": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_group_pack_batched": [_P, _I, _I, _I, _P],
    "iif_wgrad1x1_stacked": [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _L, _I, _P],
    "iif_group_unpack_grad": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_space_to_depth_nchw": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_stem_s2d_pack": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "iif_stem_s2d_unpack_grad": [_P, _I, _I, _I, _I, _I, _P, _P],
    "iif_se_squeeze": [_P, _I, _I, _I, _I, _P, _P],
    "iif_se_apply": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "iif_se_backward_sums": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "iif_se_backward_form": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "iif_se_excite_forward": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P],
    "iif_se_excite_backward": [_P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P],
    "iif_rownorm_forward": [_P, _P, _I, _I, _L, _F, _F, _F, _P, _L, _P, _P],
    "iif_rownorm_backward": [_P, _P, _P, _P, _I, _I, _L, _L, _F, _F, _F, _P, _L, _P],
    "iif_weight_transpose_batched": [_P, _P, _I, _I, _I, _P, _P],
    "iif_conv_igemm_masked_res": [_P, _P, _P, _P, _P, _P, _P],
    "iif_conv_reload_env": [],
    "iif_conv_igemm_stats_only": [_P, _P, _P, _P, _L, _P, _P],
    "iif_conv_igemm_bn_relu": [_P, _P, _P, _P, _P, _P, _P, _P],
    "iif_conv_fwdbn_ok": [_P],
    "iif_conv_igemm_stats_acc": [_P, _P, _P, _P, _L, _P, _P],
    "iif_conv_igemm_bn_relu2": [_P, _P, _P, _P, _P, _P, _P, _P, _P],
    "iif_conv_igemm_dgrad_masksum": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "iif_conv_pro_ok": [_P, _I],
    "iif_conv_igemm_bnstats_pro": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "iif_conv_dgrad_rx_ok": [_P, _I],
    "iif_conv_igemm_dgrad_masksum_rx": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _L, _P, _P],
    "iif_conv_dgrad_rx_pg_ok": [_P, _I],
    "iif_conv_igemm_dgrad_masksum_rx_pg": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _L, _P, _P, _L, _I, _P, _P],
    "iif_slab_sum": [_P, _L, _I, _I, _I, _I, _P, _P],
    "iif_conv_igemm_dgrad2_bnbwd": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "iif_bn3_algebra_prep": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _L, _P, _P, _P, _P, _I, _P, _P, _L, _P, _P, _P],
    "iif_bn3_algebra_prep_scratch_floats": [_I, _I],
    "iif_bn3_algebra_dw": [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _P],
    "iif_conv_pack_fragments": [_P, _P, _I, _I, _P, _P],
    "iif_conv_pack_fragments_g16": [_P, _P, _I, _I, _P, _P],
    "iif_conv3x3_frag_ok": [_P],
    "iif_mask_gather": [_P, _I, _P, _I, _I, _I, _P, _P, _P],
    "iif_mask_bce_fwd_bwd": [_P, _I, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "iif_class_accumulate": [_P, _P, _I, _I, _P, _P, _P],
    "iif_fasa_update": [_P, _P, _I, _I, _L, _I, _F, _P, _P, _P, _P],
    "iif_fasa_generate": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "iif_conv_igemm_dgrad_bnbwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "iif_bn_backward_partials": [_P, _P, _P, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P],
    "iif_maxpool_bn_forward": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "iif_bn_backward_relu_recompute": [_P, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _L, _P],
    "iif_bn_backward_relu_recompute_pooled": [_P, _P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _L, _P],
    "iif_bn_backward_pool_fused": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P],
    "iif_rowmap_forward": [_P, _I, _I, _I, _L, _I, _F, _F, _P, _I, _L, _P, _P],
    "iif_rowmap_backward": [_P, _I, _P, _P, _I, _I, _I, _L, _L, _I, _F, _F, _P, _I, _L, _P],
    "iif_transpose_f32": [_P, _I, _I, _L, _P, _L, _P],
    "iif_dot_window_f32": [_P, _P, _I, _I, _L, _L, _F, _P, _P, _P],
}



class ConvDesc(ctypes.Structure):
    """Mirror of ``iif_conv_desc`` (include/iif_amd.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ("n", "hs", "ws", "cs", "hd", "wd", "cd", "r", "s", "stride", "pad",
                                               "transposed", "ldw", "dtype", "dst_dtype", "groups")] + [("wgt_frag", ctypes.c_void_p),
                                                                                                          ("wgt_frag_kind", ctypes.c_int32)]


class PackDesc(ctypes.Structure):
    """Mirror of ``iif_pack_desc`` (include/iif_amd.h)."""
    _fields_ = [("src_off", ctypes.c_int64), ("dst_off", ctypes.c_int64)] + [(k, ctypes.c_int32) for k in (
        "rows", "taps", "k", "ld", "block_start", "reserved")]


_lib = None


class IIFNativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IIFNativeError(
                "libiif_amd.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C iif_amd/csrc`; iif_amd has no CPU/PyTorch fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        l.iif_version.restype = _c.c_char_p
        l.iif_version.argtypes = []
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = _L if name in ("iif_bn_workspace_bytes", "iif_bn3_algebra_prep_scratch_floats") else _I
        _lib = l
    return _lib


def check(rc, what, tickets=None):
    """Raise on a non-zero status.  ``tickets``: the ticket words of a single-launch reduction this call was given — a
    launch that failed part-way may leave them non-zero, and a non-zero ticket would silently stop every later
    finalisation, so they are re-zeroed before the error propagates."""
    if rc != 0:
        if tickets is not None:
            try:
                tickets.zero_()
            except Exception:       # the device itself may be gone; the original error is the one to report
                pass
        raise IIFNativeError("%s failed: %s" % (what, _ERR.get(rc, rc)))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The HIP stream torch is currently enqueuing on (works under graph capture).  Every launch asks: the two C entry points
    (what torch's own compiled-kernel launcher uses) cost ~0.3 us against ~6 us for torch.cuda.current_stream().cuda_stream -
    a third of the host time of a step that is bound by the launch rate (the CIFAR step: scripts/host_profile.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t):
    if t.dtype == torch.float32:
        return IIF_F32
    if t.dtype == torch.bfloat16:
        return IIF_BF16
    raise IIFNativeError("unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise IIFNativeError(
                "iif_amd runs on MI355X only: got a %s tensor on %s (no CPU fallback; the CPU restatement "
                "lives in oracle/ and is test infrastructure)" % (tuple(t.shape), t.device))


def ptr(t):
    return 0 if t is None else t.data_ptr()
