"""Mask-side class-channel selection of the Mask R-CNN head on the gfx950 kernels.

Mirrors ``mask_cross_entropy`` (instance_segmentation/mmdet/models/losses/cross_entropy_loss.py:112-162)
and the per-RoI channel pick of ``FCNMaskHead.get_seg_masks`` (fcn_mask_head.py:289-290).  The labels are
the ones the IIF classifier produced; nothing else of the mask head changes.
"""
import torch

from . import _lib


def _prep(pred, label):
    _lib.require_gpu(pred, label)
    if pred.dim() < 3:
        raise ValueError("pred must be [N, C, *]")
    if pred.dtype not in (torch.float32, torch.bfloat16):
        pred = pred.float()
    pred = pred.contiguous()
    n, c = pred.shape[0], pred.shape[1]
    hw = pred[0, 0].numel() if n else int(torch.tensor(pred.shape[2:]).prod())
    label = label.reshape(-1).to(torch.int64).contiguous()
    if label.numel() != n:
        raise ValueError("one label per RoI expected")
    return pred, label, n, c, hw


def gather_class_masks(mask_pred, labels):
    """``mask_pred[range(N), labels]`` -> [N, *] fp32 (fcn_mask_head.py:289-290)."""
    pred, label, n, c, hw = _prep(mask_pred, labels)
    out = torch.empty((n,) + tuple(pred.shape[2:]), dtype=torch.float32, device=pred.device)
    if n == 0:
        return out
    status = torch.zeros(1, dtype=torch.int32, device=pred.device)
    _lib.check(_lib.lib().iif_mask_gather(_lib.ptr(pred), _lib.dtype_code(pred), _lib.ptr(label), n, c, hw, _lib.ptr(out),
                                          _lib.ptr(status), _lib.stream_ptr()), "iif_mask_gather")
    return out


class _MaskBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, label):
        p, lb, n, c, hw = _prep(pred.detach(), label)
        t = target.detach().reshape(n, hw).float().contiguous()
        dev = p.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        rows = torch.empty(n, dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        need = pred.requires_grad
        dpred = torch.zeros(p.shape, dtype=torch.float32, device=dev) if need else None
        _lib.check(_lib.lib().iif_mask_bce_fwd_bwd(_lib.ptr(p), _lib.dtype_code(p), _lib.ptr(t), _lib.ptr(lb), n, c, hw, 1.0,
                                                   _lib.ptr(rows), _lib.ptr(loss), _lib.ptr(dpred), _lib.ptr(status),
                                                   _lib.stream_ptr()), "iif_mask_bce_fwd_bwd")
        ctx.dpred = dpred
        ctx.in_dtype = pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        d = ctx.dpred
        if d is None:
            return None, None, None
        d = d * g.reshape(()).to(d.dtype)
        return d.to(ctx.in_dtype), None, None


def mask_cross_entropy(pred, target, label, reduction="mean", avg_factor=None, class_weight=None, ignore_index=None):
    """cross_entropy_loss.py:112-162: BCE-with-logits on the class channel of every RoI, mean over all
    pixels, returned with shape ``(1,)``.  Loss and gradient come out of one pass over the selected channels."""
    assert ignore_index is None, "BCE loss does not support ignore_index"
    assert reduction == "mean" and avg_factor is None
    if class_weight is not None:
        raise NotImplementedError("class_weight is broadcast against [N, H, W] by the reference; not supported natively")
    if pred.size(0) == 0:
        return pred.sum()[None] * 0 + float("nan")        # mean over an empty slice, as the reference
    return _MaskBCE.apply(pred, target, label)
