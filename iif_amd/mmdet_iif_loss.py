"""mmdet ``IIFLoss`` plugin over the fused gfx950 kernels.

Mirror of instance_segmentation/mmdet/models/losses/iif_loss.py:12-202: same
constructor keywords, ``custom_cls_channels / custom_activation /
custom_accuracy`` attributes and ``forward / get_activation / get_cls_channels /
get_accuracy`` methods that ``BBoxHead`` reads (bbox_head.py:58,68-69,96-106,
269-281,349-350).  When mmdet is importable the class registers itself in its
``LOSSES`` registry under the name ``IIFLoss`` (``force=True`` replaces the
stock one); without mmdet it is a plain ``nn.Module``.
"""
import csv

import torch
import torch.nn as nn

from . import _lib
from .custom import fused_iif_cross_entropy
from .utils import topk_hit_counts


def read_iif_csv(path, variant):
    """CSV column ``variant`` -> float32 ``[1, C+1]``: the first data row (an
    all-ones placeholder) is dropped and 1.0 is appended for the background
    class (iif_loss.py:47-50)."""
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    try:
        col = rows[0].index(variant)
    except ValueError:
        raise KeyError(variant)
    vals = [float(r[col]) for r in rows[1:]]
    return torch.tensor(vals[1:] + [1.0], dtype=torch.float32).unsqueeze(0)


class IIFLoss(nn.Module):

    def __init__(self, use_sigmoid=False, reduction="mean", class_weight=None, ignore_index=None,
                 loss_weight=1.0, num_classes=1203, path="./lvis_files/idf_1204.csv", variant="raw",
                 device="cuda"):
        super().__init__()
        assert use_sigmoid is False
        self.use_sigmoid = use_sigmoid
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.class_weight = class_weight
        self.ignore_index = ignore_index
        self.num_classes = num_classes
        self.variant = variant
        self.iif_weights = read_iif_csv(path, variant).to(device)
        self.cls_criterion = self.cross_entropy
        self.custom_cls_channels = True
        self.custom_activation = True
        self.custom_accuracy = True

    # --- plugin protocol -------------------------------------------------
    def get_cls_channels(self, num_classes):
        assert num_classes == self.num_classes
        return num_classes + 1

    def get_activation(self, cls_score):
        """softmax(iif * cls_score) (iif_loss.py:65-78), one native pass."""
        _lib.require_gpu(cls_score)
        x = cls_score.detach()
        if x.stride(-1) != 1:
            x = x.contiguous()
        N, C = x.shape
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        rc = _lib.lib().iif_softmax(_lib.ptr(x), _lib.dtype_code(x), x.stride(0) if N else C,
                                    _lib.ptr(self._table(x)), N, C, _lib.ptr(out), C, _lib.stream_ptr())
        _lib.check(rc, "iif_softmax")
        return out

    def get_accuracy(self, cls_score, labels):
        """{'acc_classes': top-1 % on the RAW cls_score} (iif_loss.py:92-107,
        accuracy.py:7-51: one-element tensor, 0. for an empty batch)."""
        if cls_score.size(0) == 0:
            return dict(acc_classes=cls_score.new_tensor(0.0))
        hits = topk_hit_counts(cls_score, labels, (1,))
        return dict(acc_classes=hits.to(torch.float32) * (100.0 / cls_score.size(0)))

    # --- loss ------------------------------------------------------------
    def _table(self, like):
        if self.iif_weights.device != like.device:
            self.iif_weights = self.iif_weights.to(like.device)
        return self.iif_weights

    def cross_entropy(self, pred, label, weight=None, reduction="mean", avg_factor=None, class_weight=None,
                      ignore_index=-100):
        """iif_loss.py:157-202 + losses/utils.py:29-55 in one fused launch."""
        ignore_index = -100 if ignore_index is None else ignore_index
        if avg_factor is not None and reduction == "sum":
            raise ValueError('avg_factor can not be used with reduction="sum"')
        if reduction == "none":
            avg_factor = None            # 'none' ignores avg_factor (utils.py:50-52)
        return fused_iif_cross_entropy(pred, self._table(pred), label, row_weight=weight,
                                       class_weight=class_weight, ignore_index=ignore_index,
                                       reduction=reduction, avg_factor=avg_factor)

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None,
                **kwargs):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if ignore_index is None:
            ignore_index = self.ignore_index
        class_weight = None
        if self.class_weight is not None:
            class_weight = cls_score.new_tensor(self.class_weight, device=cls_score.device)
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=class_weight,
                                                     reduction=reduction, avg_factor=avg_factor,
                                                     ignore_index=ignore_index, **kwargs)


def register_into_mmdet():
    """Register the native class as mmdet's ``IIFLoss`` if mmdet is importable."""
    try:
        from mmdet.models.builder import LOSSES
    except Exception:
        return False
    LOSSES.register_module(name="IIFLoss", force=True, module=IIFLoss)
    return True


register_into_mmdet()
