"""FASA (feature augmentation and sampling adaptation) pieces that sit on the IIF classifier path, on the
gfx950 kernels.

Mirrors, from instance_segmentation/mmdet:
  * ``FasaIIFLoss`` (models/losses/fasa_iif_loss.py:12-208): the IIF cross entropy plus, while ``use_cums`` is
    open, per-class accumulators of the row losses and label counts (:154-160);
  * the feature bank of ``ConvFCFASABBoxHead`` (models/roi_heads/bbox_heads/fasa_bbox_head.py:35-66,118-215):
    ``fa_update`` (per-class running mean / variance of positive embeddings), ``fa_generate`` (virtual
    embeddings for under-sampled classes) and the once-per-epoch ``dynamic_sampling`` (host-side,
    AffinityPropagation over class means).

The reference walks ``torch.unique(labels)`` on the host (one device sync per class and step); here each of
those is one launch with no host round trip, except the row count of ``fa_generate`` (one sync, as the
reference's ``len(embedding_list)``).
"""
import torch
import torch.nn as nn

from . import _lib
from .mmdet_iif_loss import IIFLoss


class FasaIIFLoss(IIFLoss):
    """fasa_iif_loss.py:12-208.  ``use_sigmoid`` / ``use_mask`` select stock mmdet criteria in the reference
    and are not part of the IIF path."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, loss_weight=1.0,
                 use_cums=False, num_classes=1203, path="./lvis_files/idf_1204.csv", variant="raw", device="cuda"):
        assert (use_sigmoid is False) or (use_mask is False)
        if use_sigmoid or use_mask:
            raise NotImplementedError("FasaIIFLoss(use_sigmoid/use_mask) dispatches to stock mmdet criteria")
        super().__init__(use_sigmoid=False, reduction=reduction, class_weight=class_weight, ignore_index=None,
                         loss_weight=loss_weight, num_classes=num_classes, path=path, variant=variant, device=device)
        self.use_mask = use_mask
        self._device = device
        self.use_cums = use_cums
        if self.use_cums:
            self.open_cums()

    def open_cums(self):
        self.use_cums = True
        self.reduction_old = self.reduction
        self.reduction = "none"
        self.cum_losses = torch.zeros(self.num_classes + 1, device=self._device)
        self.cum_labels = torch.zeros(self.num_classes + 1, device=self._device)

    def close_cums(self):
        self.use_cums = False
        self.reduction = self.reduction_old
        self.cum_losses = torch.zeros(self.num_classes + 1, device=self._device)
        self.cum_labels = torch.zeros(self.num_classes + 1, device=self._device)

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        loss_cls = super().forward(cls_score, label, weight=weight, avg_factor=avg_factor,
                                   reduction_override=reduction_override, **kwargs)
        if self.use_cums:
            rows = loss_cls.detach().float().contiguous()
            lb = label.reshape(-1).to(torch.int64).contiguous()
            _lib.check(_lib.lib().iif_class_accumulate(_lib.ptr(rows), _lib.ptr(lb), rows.numel(), self.num_classes + 1,
                                                       _lib.ptr(self.cum_losses), _lib.ptr(self.cum_labels),
                                                       _lib.stream_ptr()), "iif_class_accumulate")
            loss_cls = loss_cls.mean()
        return loss_cls


class FasaFeatureBank(nn.Module):
    """State and methods ``ConvFCFASABBoxHead`` adds to the stock bbox head (fasa_bbox_head.py:35-66,118-215).
    ``instance_counts``: per-class training instance counts (the reference's LVIS_INSTANCES table)."""

    def __init__(self, num_classes, feat_dim, instance_counts, fasa_cfg=None, device="cuda"):
        super().__init__()
        cfg = fasa_cfg or {}
        self.num_classes, self.feat_dim = num_classes, feat_dim
        self.feature_mean = nn.Parameter(torch.zeros(num_classes, feat_dim, device=device), requires_grad=False)
        self.feature_std = nn.Parameter(torch.zeros(num_classes, feat_dim, device=device), requires_grad=False)   # a variance
        self.feature_used = nn.Parameter(torch.zeros(num_classes, device=device), requires_grad=False)
        self.decay_ratio = cfg.get("decay_ratio", 0.1)
        self.loss_aug_weight = cfg.get("loss_aug_weight", 0.1)
        self.dynamic_up = cfg.get("dynamic_up", 1.1)
        self.dynamic_down = cfg.get("dynamic_down", 0.9)
        counts = torch.as_tensor(instance_counts, dtype=torch.float32, device=device)
        prob = 1 / counts
        prob = cfg.get("instance_prob_scale", 1) * torch.pow(prob / prob.sum(), cfg.get("instance_prob_power", 1))
        self.prob_list = nn.Parameter(prob.clamp(0, 1), requires_grad=False)
        self.cum_loss_perclass_t0 = torch.zeros(num_classes + 1, device=device)
        self.cum_loss_perclass_t1 = torch.zeros(num_classes + 1, device=device)
        self.group_cluster_list = []
        self.epoch = 0
        self._slots = torch.zeros(num_classes, dtype=torch.int32, device=device)
        self._count = torch.zeros(1, dtype=torch.int32, device=device)

    def fa_update(self, embedding, labels):
        """fasa_bbox_head.py:118-147, all classes in one launch."""
        if len(labels) == 0:
            return
        _lib.require_gpu(embedding, labels)
        e = embedding.detach().float()
        if e.stride(1) != 1:
            e = e.contiguous()
        lb = labels.reshape(-1).to(torch.int64).contiguous()
        _lib.check(_lib.lib().iif_fasa_update(_lib.ptr(e), _lib.ptr(lb), e.shape[0], e.shape[1], e.stride(0), self.num_classes,
                                              float(self.decay_ratio), _lib.ptr(self.feature_mean.data),
                                              _lib.ptr(self.feature_std.data), _lib.ptr(self.feature_used.data),
                                              _lib.stream_ptr()), "iif_fasa_update")

    def fa_generate(self, rand=None, normal=None):
        """fasa_bbox_head.py:149-172.  ``rand`` [C] / ``normal`` [C, D] default to fresh torch draws (the
        reference draws ``torch.rand(C)`` and one ``torch.normal`` per selected class)."""
        dev = self.feature_mean.device
        if rand is None:
            rand = torch.rand(self.num_classes, device=dev)
        if normal is None:
            normal = torch.randn(self.num_classes, self.feat_dim, device=dev)
        out = torch.empty(self.num_classes, self.feat_dim, device=dev)
        labels = torch.empty(self.num_classes, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().iif_fasa_generate(_lib.ptr(rand.float().contiguous()), _lib.ptr(self.prob_list.data),
                                                _lib.ptr(self.feature_used.data), _lib.ptr(self.feature_mean.data),
                                                _lib.ptr(self.feature_std.data), _lib.ptr(normal.float().contiguous()),
                                                self.num_classes, self.feat_dim, _lib.ptr(self._slots), _lib.ptr(self._count),
                                                _lib.ptr(out), _lib.ptr(labels), _lib.stream_ptr()), "iif_fasa_generate")
        k = int(self._count.item())
        if k == 0:
            return [], []
        return out[:k], labels[:k]

    def dynamic_sampling(self, loss_cls, training=False):
        """fasa_bbox_head.py:174-215: once per epoch, in eval mode.  Host-side (AffinityPropagation)."""
        if training:
            return
        import torch.distributed as dist
        from sklearn.cluster import AffinityPropagation

        def reduce_mean(t):                       # mmdet/core/utils/dist_utils.py:67-73
            if not (dist.is_available() and dist.is_initialized()):
                return t
            t = t.clone()
            dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
            return t
        cum_labels = reduce_mean(loss_cls.cum_labels)
        cum_losses = reduce_mean(loss_cls.cum_losses)
        self.cum_loss_perclass_t1 = cum_losses / cum_labels.sum()
        if self.cum_loss_perclass_t0.sum() == 0:
            self.cum_loss_perclass_t0[:] = self.cum_loss_perclass_t1[:]
        fm = self.feature_mean.data
        mean_xy = torch.matmul(fm, fm.T)
        sq = torch.sum(fm.square(), dim=1)
        distance = (sq.unsqueeze(1) - 2 * mean_xy + sq.unsqueeze(0)).cpu().numpy()
        clustering = AffinityPropagation(random_state=1, affinity="precomputed").fit(distance)
        self.group_cluster_list = [[i for i, v in enumerate(clustering.labels_) if v == g]
                                   for g in range(max(clustering.labels_) + 1)]
        for group in self.group_cluster_list:
            delta = self.cum_loss_perclass_t1[group].sum() - self.cum_loss_perclass_t0[group].sum()
            if delta > 0:
                self.prob_list.data[group] = (self.prob_list.data[group] * self.dynamic_down).clamp(0, 1)
            if delta < 0:
                self.prob_list.data[group] = (self.prob_list.data[group] * self.dynamic_up).clamp(0, 1)
        self.cum_loss_perclass_t0[:] = self.cum_loss_perclass_t1[:]


def register_into_mmdet():
    try:
        from mmdet.models.builder import LOSSES
    except Exception:
        return False
    LOSSES.register_module(name="FasaIIFLoss", force=True, module=FasaIIFLoss)
    return True


register_into_mmdet()
