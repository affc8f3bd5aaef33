"""IIF classifier loss and mixup — host-side mirror of the reference's
``classification/custom.py`` (IIFLoss :6-39, Mixup :91-117) over the fused
gfx950 kernel ``iif_ce_fwd_bwd``.

Same constructor, attributes (``.iif`` dict of float32 ``[1, C]`` tables,
``.variant``, ``.reduction``) and call signature as the reference, so
``train.py`` / ``per_shot_acc.py``-style callers work unchanged; tensors must
live on the MI355X (no CPU path — see ``iif_amd._lib``).
"""
import numpy as np
import torch
import torch.nn as nn
from scipy.special import ndtri

from . import _lib

VARIANTS = ("raw", "smooth", "rel", "normit", "gombit", "base2", "base10")


def build_tables(cls_num_list, iif_norm=0):
    """float32 ``[1, C]`` host tensors for the seven variants.

    Host arithmetic, once per run, in the reference's own number formats
    (custom.py:14-26): float64 numpy on the integer counts, one cast to
    float32, optional division by the float32 p-norm.
    """
    f = np.asarray(cls_num_list)
    tot = f.sum()
    ratio = tot / f
    frac = f / tot
    host = {
        "raw": np.log(ratio),
        "smooth": np.log((tot + 1) / (f + 1)) + 1,
        "rel": np.log((tot - f) / f),
        "normit": -ndtri(frac),
        "gombit": -np.log(-np.log(1 - frac)),
        "base2": np.log2(ratio),
        "base10": np.log10(ratio),
    }
    tabs = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(torch.float32).unsqueeze(0)
            for k, v in host.items()}
    if iif_norm > 0:
        tabs = {k: v / torch.norm(v, p=iif_norm) for k, v in tabs.items()}
    return tabs


def _launch_ce(pred, table, ta, tb, lam, row_weight, class_weight, ignore_index, scale, want_grad, keep_rows=False):
    """One launch of the fused kernel.  Returns (loss, rows, dlogits-or-None)."""
    _lib.require_gpu(pred, table, ta, tb, row_weight, class_weight)
    if pred.dim() != 2:
        raise ValueError("logits must be [B, C], got %s" % (tuple(pred.shape),))
    if pred.stride(1) != 1:
        pred = pred.contiguous()
    B, C = pred.shape
    if table.numel() != C:
        raise ValueError("IIF table has %d classes, logits have %d" % (table.numel(), C))
    dlogits = torch.empty((B, C), dtype=pred.dtype, device=pred.device) if want_grad else None
    loss = torch.empty((), dtype=torch.float32, device=pred.device)          # written by the kernel (0 for B == 0)
    rows, ticket, status = _workspace(pred.device, B, keep_rows)
    ta = ta.to(torch.int64).contiguous()
    tb = None if tb is None else tb.to(torch.int64).contiguous()
    rw = None if row_weight is None else row_weight.to(torch.float32).contiguous()
    cw = None if class_weight is None else class_weight.to(torch.float32).contiguous()
    tab = table.to(torch.float32).contiguous()
    rc = _lib.lib().iif_ce_fwd_bwd(
        _lib.ptr(pred), _lib.dtype_code(pred), pred.stride(0) if B else C, _lib.ptr(tab), _lib.ptr(ta),
        _lib.ptr(tb), float(lam), _lib.ptr(rw), _lib.ptr(cw), int(ignore_index), float(scale), B, C,
        _lib.ptr(rows), _lib.ptr(loss), _lib.ptr(dlogits), C, _lib.ptr(status), _lib.ptr(ticket), _lib.stream_ptr())
    _lib.check(rc, "iif_ce_fwd_bwd", ticket[:1])
    return loss, rows, dlogits


_WS = {}


def _workspace(device, B, own_rows):
    """Per (device, stream) scratch of the fused launch: the per-row loss vector (grown on demand; a fresh tensor
    when the caller hands the rows out, reduction='none'), the ticket word of the in-kernel loss reduce (zero
    between launches) and the sticky label-status word (see ``check_label_status``)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = {"rows": torch.empty(0, dtype=torch.float32, device=device),
                         "ticket": torch.zeros(1 + 2048, dtype=torch.int32, device=device),      # IIF_CE_WORKSPACE_BYTES
                         "status": torch.zeros(1, dtype=torch.int32, device=device)}
    if own_rows:
        rows = torch.empty(B, dtype=torch.float32, device=device)
    else:
        if ws["rows"].numel() < B:
            ws["rows"] = torch.empty(max(B, 2 * ws["rows"].numel()), dtype=torch.float32, device=device)
        rows = ws["rows"]
    return rows, ws["ticket"], ws["status"]


def check_label_status():
    """Raise if any target seen by the fused loss since the last check was outside [0, C) and not the ignore
    index (the reference asserts inside nll_loss; here such rows contribute zero and set a device flag).  One
    host sync per call: use it where the loop synchronises anyway (classification/train.py:87-92)."""
    for ws in _WS.values():
        if int(ws["status"].item()) != 0:
            ws["status"].zero_()
            raise IndexError("IIF loss: a target label is outside [0, C) and is not the ignore index")


class _FusedIIFCrossEntropy(torch.autograd.Function):
    """Scalar loss from ONE pass of the fused kernel; the gradient w.r.t. the
    logits is produced in that same pass and kept for backward, where it is only
    multiplied by the upstream scalar (device-side, no host sync)."""

    @staticmethod
    def forward(ctx, pred, table, ta, tb, lam, row_weight, class_weight, ignore_index, scale):
        loss, _, dlogits = _launch_ce(pred, table, ta, tb, lam, row_weight, class_weight, ignore_index, scale,
                                      ctx.needs_input_grad[0])
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        (dlogits,) = ctx.saved_tensors
        if dlogits is None:
            return (None,) * 9
        g = g_loss.to(torch.float32).contiguous()
        out = torch.empty_like(dlogits)           # the saved gradient stays intact: backward may run twice
        rc = _lib.lib().iif_scale_by_device_scalar(_lib.ptr(dlogits), _lib.dtype_code(dlogits), dlogits.numel(),
                                                   _lib.ptr(g), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "iif_scale_by_device_scalar")
        return (out,) + (None,) * 8


def fused_iif_cross_entropy(pred, table, targets, targets_b=None, lam=1.0, row_weight=None, class_weight=None,
                            ignore_index=-100, reduction="mean", avg_factor=None, loss_weight=1.0):
    """Functional form shared by the classification and the mmdet surfaces.

    reduction 'mean' -> sum/B (or sum/avg_factor), 'sum' -> sum, 'none' -> the
    per-row vector (its gradient path scales rows by the upstream vector).
    """
    B = pred.shape[0]
    if reduction == "mean":
        scale = loss_weight / (float(avg_factor) if avg_factor is not None else float(max(B, 1)))
    elif reduction == "sum":
        if avg_factor is not None:
            raise ValueError('avg_factor can not be used with reduction="sum"')
        scale = loss_weight
    elif reduction == "none":
        return _rows_loss(pred, table, targets, targets_b, lam, row_weight, class_weight, ignore_index) * loss_weight
    else:
        raise ValueError("unknown reduction %r" % (reduction,))
    loss = _FusedIIFCrossEntropy.apply(pred, table, targets, targets_b, lam, row_weight, class_weight,
                                       ignore_index, scale)
    if B == 0 and reduction == "mean" and avg_factor is None:
        return loss * float("nan")          # torch: mean of an empty tensor
    return loss


class _FusedIIFRows(torch.autograd.Function):
    """reduction='none': rows out; backward = per-row scaling of the unit-scale gradient."""

    @staticmethod
    def forward(ctx, pred, table, ta, tb, lam, row_weight, class_weight, ignore_index):
        _, rows, dlogits = _launch_ce(pred, table, ta, tb, lam, row_weight, class_weight, ignore_index, 1.0,
                                      ctx.needs_input_grad[0], keep_rows=True)
        ctx.save_for_backward(dlogits)
        return rows

    @staticmethod
    def backward(ctx, g_rows):
        (dlogits,) = ctx.saved_tensors
        if dlogits is None:
            return (None,) * 8
        return (dlogits * g_rows.to(dlogits.dtype).unsqueeze(1),) + (None,) * 7


def _rows_loss(pred, table, ta, tb, lam, row_weight, class_weight, ignore_index):
    return _FusedIIFRows.apply(pred, table, ta, tb, lam, row_weight, class_weight, ignore_index)


class IIFLoss(nn.Module):
    """Drop-in for ``custom.IIFLoss`` (classification/custom.py:6-39).

    ``dataset`` only needs ``get_cls_num_list() -> list[int]``.  ``weight`` is
    the optional per-class weight of deferred re-weighting
    (initialisers.py:16-28); as in the reference the 'mean' reduction divides
    by the batch size, not by the weight sum.
    """

    def __init__(self, dataset, variant="raw", iif_norm=0, reduction="mean", device="cuda", weight=None):
        super().__init__()
        if variant not in VARIANTS:
            raise KeyError(variant)
        self.reduction = reduction
        self.variant = variant
        self.iif_norm = iif_norm
        self.weight = weight
        tabs = build_tables(dataset.get_cls_num_list(), iif_norm)
        self.iif = {k: v.to(device, non_blocking=True) for k, v in tabs.items()}
        # True only for the plain-CE criterion built by initialisers.get_criterion: 'mean' then divides by the sum
        # of the targets' class weights (nn.CrossEntropyLoss semantics) instead of the batch size
        self.weighted_mean = False

    def mean_denominator(self, targets):
        """Device scalar sum_i weight[t_i] (ignored / out-of-range targets count 0), or None for the plain 1/B."""
        if not (self.weighted_mean and self.weight is not None and self.reduction == "mean"):
            return None
        w = self.weight.to(device=targets.device, dtype=torch.float32)
        ok = (targets >= 0) & (targets < w.numel())
        return (w[targets.clamp(0, w.numel() - 1)] * ok).sum()

    def _table(self, like):
        t = self.iif[self.variant]
        if t.device != like.device:
            t = t.to(like.device)
            self.iif[self.variant] = t
        return t

    def forward(self, pred, targets=None, infer=False):
        _lib.require_gpu(pred)
        table = self._table(pred)
        if infer is False:
            red = self.reduction if self.reduction in ("mean", "sum") else "none"
            w = self.weight
            if w is not None and w.device != pred.device:
                w = w.to(pred.device)
            den = self.mean_denominator(targets)
            if den is not None:            # weighted mean: sum of the weighted rows over the sum of the weights
                return fused_iif_cross_entropy(pred, table, targets, class_weight=w, reduction="sum") / den
            return fused_iif_cross_entropy(pred, table, targets, class_weight=w, reduction=red)
        return scale_logits(pred, table)

    def mixup_loss(self, pred, y_a, y_b, lam):
        """``lam*L(pred,y_a) + (1-lam)*L(pred,y_b)`` from a single softmax pass."""
        _lib.require_gpu(pred)
        red = self.reduction if self.reduction in ("mean", "sum") else "none"
        w = self.weight
        if w is not None and w.device != pred.device:
            w = w.to(pred.device)
        if self.mean_denominator(y_a) is not None:          # two different denominators: two passes, as the reference
            return lam * self(pred, y_a) + (1 - lam) * self(pred, y_b)
        return fused_iif_cross_entropy(pred, self._table(pred), y_a, targets_b=y_b, lam=lam, class_weight=w,
                                       reduction=red)


def scale_logits(pred, table):
    """``pred * iif`` for evaluation (custom.py:37-39), native kernel, no autograd."""
    _lib.require_gpu(pred, table)
    x = pred.detach()
    if x.stride(-1) != 1:
        x = x.contiguous()
    B, C = x.shape
    out = torch.empty((B, C), dtype=x.dtype, device=x.device)
    rc = _lib.lib().iif_scale_logits(_lib.ptr(x), _lib.dtype_code(x), x.stride(0) if B else C,
                                     _lib.ptr(table.contiguous()), B, C, _lib.ptr(out), C, _lib.stream_ptr())
    _lib.check(rc, "iif_scale_logits")
    return out


class Mixup(object):
    """Mirror of ``custom.Mixup`` (classification/custom.py:91-117).

    ``__call__`` draws ``lam ~ Beta(alpha, alpha)`` from numpy's global RNG and a
    device permutation, as the reference does, and blends the images with one
    native pass.  ``mixup_criterion`` uses the fused two-target kernel when the
    criterion is an :class:`IIFLoss`.
    """

    def __init__(self, criterion, alpha=1):
        self.alpha = alpha
        self.criterion = criterion

    def __call__(self, x, y, use_cuda=True):
        lam = np.random.beta(self.alpha, self.alpha) if self.alpha > 0 else 1
        index = torch.randperm(x.size(0), device=x.device)
        mixed = mix_rows(x, index, lam)
        return mixed, y, y[index], lam

    def mixup_criterion(self, pred, y_a, y_b, lam):
        if isinstance(self.criterion, IIFLoss):
            return self.criterion.mixup_loss(pred, y_a, y_b, lam)
        return lam * self.criterion(pred, y_a) + (1 - lam) * self.criterion(pred, y_b)


def mix_rows(x, index, lam):
    """``lam*x + (1-lam)*x[index]`` (custom.py:112) by the native blend kernel."""
    _lib.require_gpu(x, index)
    xc = x.contiguous()
    out = torch.empty_like(xc)
    B = xc.shape[0]
    n = xc.numel() // max(B, 1)
    rc = _lib.lib().iif_mix_rows(_lib.ptr(xc), _lib.dtype_code(xc), _lib.ptr(index.to(torch.int64).contiguous()),
                                 float(lam), B, n, _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "iif_mix_rows")
    return out
