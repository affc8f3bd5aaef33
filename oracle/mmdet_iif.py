"""CPU restatement of the mmdet ``IIFLoss`` plugin arithmetic.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Parity PINNED: every function below is checked against
vectors produced by executing the reference's own files (tests/golden/make_golden_mmdet.py runs
mmdet/models/losses/{utils,accuracy,cross_entropy_loss,iif_loss,fasa_iif_loss}.py, models/utils/normed_predictor.py
and roi_heads/bbox_heads/fasa_bbox_head.py under placeholder mmcv / registry modules; fixtures g11..g15), by
tests/test_mmdet_golden.py on the CPU.  Additional cross-checks: with weight=None, avg_factor=None,
class_weight=None, loss_weight=1 the loss equals ``oracle.iif_oracle.iif_ce(reduction='mean')`` (golden G4), and the
CE known answers of instance_segmentation/tests/test_metrics/test_losses.py:8-32 (table of ones).
Citations are relative to /root/reference/instance_segmentation/.
"""
import csv

import torch
import torch.nn.functional as F


def read_table(path, variant, dtype=torch.float32):
    """mmdet/models/losses/iif_loss.py:47-50 — CSV column ``variant``, drop the
    first data row (an all-ones placeholder), append 1.0 for background,
    cast to float32, shape ``[1, C+1]``.  (The reference parses with pandas;
    python's ``float()`` parses the same decimal strings to the same doubles.)
    """
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    col = rows[0].index(variant)
    vals = [float(r[col]) for r in rows[1:]]
    vals = vals[1:] + [1.0]
    return torch.tensor(vals, dtype=dtype).unsqueeze(0)


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    """mmdet/models/losses/utils.py:29-55."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == "mean":
            return loss.mean()
        if reduction == "sum":
            return loss.sum()
        return loss
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def iif_cross_entropy(pred, label, table, weight=None, reduction="mean", avg_factor=None,
                      class_weight=None, ignore_index=None, loss_weight=1.0):
    """mmdet/models/losses/iif_loss.py:132-152,184-202."""
    ignore_index = -100 if ignore_index is None else ignore_index
    loss = F.cross_entropy(pred * table, label, weight=class_weight, reduction="none",
                           ignore_index=ignore_index)
    if weight is not None:
        weight = weight.float()
    return loss_weight * weight_reduce_loss(loss, weight=weight, reduction=reduction,
                                            avg_factor=avg_factor)


def get_activation(cls_score, table):
    """mmdet/models/losses/iif_loss.py:65-78."""
    return torch.softmax(table * cls_score, dim=-1)


def accuracy_top1(pred, target):
    """mmdet/models/losses/accuracy.py:7-51 with topk=1, thresh=None: a
    one-element tensor holding the top-1 hit rate in percent (0. for N=0)."""
    if pred.size(0) == 0:
        return pred.new_tensor(0.0)
    _, lbl = pred.topk(1, dim=1)
    correct = lbl.t().eq(target.view(1, -1))
    return correct[:1].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0))


def normed_linear(x, weight, bias=None, temperature=20.0, power=1.0, eps=1e-6, iif_rows=None):
    """mmdet/models/utils/normed_predictor.py: NormedLinear.forward :34-40; with ``iif_rows`` ([C] or [C,1])
    IIFNormedLinear.forward :67-73 (rows of W scaled by the class's IIF weight before normalising)."""
    w = weight if iif_rows is None else iif_rows.reshape(-1, 1) * weight
    weight_ = w / (w.norm(dim=1, keepdim=True).pow(power) + eps)
    x_ = x / (x.norm(dim=1, keepdim=True).pow(power) + eps)
    x_ = x_ * temperature
    return F.linear(x_, weight_, bias)


def normed_conv2d(x, weight, bias=None, temperature=20.0, power=1.0, eps=1e-6, norm_over_kernel=False, stride=1, padding=0):
    """normed_predictor.py: NormedConv2d.forward :104-124.  The input is normalised over the channel dimension of every
    pixel; the filter over its channel dimension per tap (:105-108), or over the whole filter with norm_over_kernel
    (:109-113)."""
    if not norm_over_kernel:
        weight_ = weight / (weight.norm(dim=1, keepdim=True).pow(power) + eps)
    else:
        weight_ = weight / (weight.view(weight.size(0), -1).norm(dim=1, keepdim=True).pow(power)[..., None, None] + eps)
    x_ = x / (x.norm(dim=1, keepdim=True).pow(power) + eps)
    x_ = x_ * temperature
    return F.conv2d(x_, weight_, bias, stride, padding)


def normed_conv2d_1x1(x, weight, bias=None, temperature=20.0, power=1.0, eps=1e-6):
    """The 1x1 predictor the mask head builds (both normalisations coincide)."""
    return normed_conv2d(x, weight, bias, temperature, power, eps)


def mask_cross_entropy(pred, target, label):
    """mmdet/models/losses/cross_entropy_loss.py:158-162 (reduction 'mean', no avg_factor, no class_weight)."""
    num_rois = pred.size()[0]
    inds = torch.arange(0, num_rois, dtype=torch.long, device=pred.device)
    pred_slice = pred[inds, label].squeeze(1)
    return F.binary_cross_entropy_with_logits(pred_slice, target, reduction="mean")[None]


def gather_class_masks(mask_pred, labels):
    """mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:289-290."""
    return mask_pred[range(mask_pred.shape[0]), labels]


def fasa_accumulate(loss_rows, label, cum_losses, cum_labels):
    """mmdet/models/losses/fasa_iif_loss.py:154-160 (in place on the two accumulators); returns the mean."""
    for u_l in label.unique():
        inds_ = torch.where(label == u_l)[0]
        cum_labels[int(u_l)] += len(inds_)
        cum_losses[int(u_l)] += loss_rows[inds_].sum()
    return loss_rows.mean()


def fasa_update(embedding, labels, feature_mean, feature_var, feature_used, decay_ratio):
    """fasa_bbox_head.py:118-147 (fa_update + fa_update_push), in place."""
    for c in torch.unique(labels):
        c = int(c)
        e = embedding[torch.nonzero(labels == c, as_tuple=False).squeeze(1)]
        mean = e.mean(dim=0)
        var = e.var(dim=0, unbiased=False)
        n = e.numel() / e.size(1)
        if n > 1:
            var = var * n / (n - 1)
        if feature_used[c] > 0:
            feature_mean[c] = decay_ratio * mean + (1 - decay_ratio) * feature_mean[c]
            feature_var[c] = decay_ratio * var + (1 - decay_ratio) * feature_var[c]
        else:
            feature_mean[c] = mean
            feature_var[c] = var
            feature_used[c] += 1


def fasa_generate(rand, prob_list, feature_used, feature_mean, feature_var, normal):
    """fasa_bbox_head.py:149-172 with the random draws passed in (``normal[c]`` is the draw of class c)."""
    emb, lab = [], []
    for c in torch.where(rand < prob_list)[0]:
        c = int(c)
        if feature_used[c] == 0:
            continue
        emb.append((feature_mean[c] + torch.sqrt(feature_var[c]) * normal[c]).unsqueeze(0))
        lab.append(c)
    if emb:
        return torch.cat(emb, 0), torch.tensor(lab)
    return [], []
