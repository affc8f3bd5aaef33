"""CPU restatement of the IIF classifier head (classification half).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  numpy float64 for the table
arithmetic, torch CPU fp32 ops for the loss (the reference *is* those torch
ops), explicit closed-form gradient in float64 for an autograd-free check.
All ``file:line`` citations are relative to ``/root/reference/``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.special import ndtri

VARIANTS = ("raw", "smooth", "rel", "normit", "gombit", "base2", "base10")


# --------------------------------------------------------------------------- a1
def img_num_per_cls(cls_num, n_images, imb_type="exp", imb_factor=0.01):
    """Class counts of the long-tailed CIFAR subsets.

    Follows classification/imbalanced_dataset.py:23-37: ``img_max`` is a
    python float, the power is a python float power and ``int()`` truncates.
    Integer outputs must be bit-exact.
    """
    img_max = n_images / cls_num
    if imb_type == "exp":
        return [int(img_max * (imb_factor ** (i / (cls_num - 1.0)))) for i in range(cls_num)]
    if imb_type == "step":
        half = cls_num // 2
        return [int(img_max)] * half + [int(img_max * imb_factor)] * half
    return [int(img_max)] * cls_num


# --------------------------------------------------------------------------- a2
def lt_class_map(targets, num_classes, kind=None):
    """Rank classes by descending count and remap the labels.

    Follows classification/imbalanced_dataset.py:112-127 (and the eval-side
    remap at :161).  The reference calls ``np.argsort(-counts)`` with numpy's
    default (non-stable introsort); ``kind=None`` reproduces that call,
    ``kind='stable'`` is the documented tie rule of the product
    (ties keep ascending original class id).
    Returns (class_map, remapped_targets, cls_num_list).
    """
    t = np.asarray(targets, dtype=np.int64)
    old = np.array([int(np.sum(t == i)) for i in range(num_classes)])
    order = np.argsort(-old) if kind is None else np.argsort(-old, kind=kind)
    class_map = [0] * num_classes
    for rank in range(num_classes):
        class_map[int(order[rank])] = rank
    new_t = np.array(class_map, dtype=np.int64)[t]
    cls_num_list = [int(np.sum(new_t == i)) for i in range(num_classes)]
    return class_map, new_t.tolist(), cls_num_list


# --------------------------------------------------------------------------- a3
def iif_tables(cls_num_list, iif_norm=0):
    """The seven per-class weight tables, float32 ``[1, C]``.

    Follows classification/custom.py:14-26: float64 numpy arithmetic on the
    integer counts, one cast to float32, then optional division by the
    p-norm computed in float32 (``torch.norm(v, p=iif_norm)``).
    """
    f = np.array(cls_num_list)
    s = f.sum()
    t64 = {
        "raw": np.log(s / f),
        "smooth": np.log((s + 1) / (f + 1)) + 1,
        "rel": np.log((s - f) / f),
        "normit": -ndtri(f / s),
        "gombit": -np.log(-np.log(1 - (f / s))),
        "base2": np.log2(s / f),
        "base10": np.log10(s / f),
    }
    out = {k: torch.from_numpy(np.asarray(v, dtype=np.float64)).to(torch.float32).reshape(1, -1)
           for k, v in t64.items()}
    if iif_norm > 0:
        out = {k: v / torch.norm(v, p=iif_norm) for k, v in out.items()}
    return out


def deferred_class_weight(cls_num_list):
    """classification/initialisers.py:16-19 — ``sum/count`` on an int64 tensor.

    ``per_cls_weights.sum()/per_cls_weights`` on integer tensors is true
    division and yields float32.
    """
    c = torch.tensor(cls_num_list)
    return c.sum() / c


# --------------------------------------------------------------------------- a4
def iif_ce(pred, targets, table, class_weight=None, reduction="mean"):
    """Training loss, torch CPU ops.  classification/custom.py:10,28-36.

    ``CrossEntropyLoss(reduction='none', weight=w)`` gives ``w[t_i]*nll_i``; the
    outer ``mean`` divides by B (not by the weight sum).
    """
    z = pred * table
    per_row = F.cross_entropy(z, targets, weight=class_weight, reduction="none")
    if reduction == "mean":
        return per_row.mean()
    if reduction == "sum":
        return per_row.sum()
    return per_row


def iif_ce_closed_form(pred, targets, table, class_weight=None, reduction="mean"):
    """Same loss and its gradient from the closed form, float64, no autograd.

    loss_i = w[t_i] * (logsumexp(z_i) - z_i[t_i]),  z = pred * table
    dpred[i,c] = table_c * w[t_i] * (softmax(z_i)_c - [c == t_i]) / denom
    with denom = B for 'mean' (custom.py:32-33) and 1 for 'sum'.
    Returns (loss float64 scalar, dpred float64 [B,C], per_row float64 [B]).
    """
    p = pred.detach().to(torch.float64)
    tb = table.detach().to(torch.float64).reshape(1, -1)
    z = p * tb
    m = z.max(dim=1, keepdim=True).values
    e = torch.exp(z - m)
    ssum = e.sum(dim=1, keepdim=True)
    lse = (m + torch.log(ssum)).squeeze(1)
    B = p.shape[0]
    idx = torch.arange(B)
    w = torch.ones(B, dtype=torch.float64)
    if class_weight is not None:
        w = class_weight.detach().to(torch.float64)[targets]
    per_row = w * (lse - z[idx, targets])
    soft = e / ssum
    onehot = torch.zeros_like(soft)
    onehot[idx, targets] = 1.0
    denom = float(B) if reduction == "mean" else 1.0
    dpred = tb * (soft - onehot) * (w / denom).unsqueeze(1)
    loss = per_row.sum() / denom
    return loss, dpred, per_row


# --------------------------------------------------------------------------- a5
def iif_infer(pred, table):
    """Scaled logits for evaluation.  classification/custom.py:37-39."""
    return pred * table


# --------------------------------------------------------------------------- a6
def mixup_criterion(pred, y_a, y_b, lam, table, class_weight=None, reduction="mean"):
    """classification/custom.py:116-117 — convex blend of two CE passes."""
    return (lam * iif_ce(pred, y_a, table, class_weight, reduction)
            + (1 - lam) * iif_ce(pred, y_b, table, class_weight, reduction))


def mixup_inputs(x, index, lam):
    """classification/custom.py:112 — ``lam*x + (1-lam)*x[index]``."""
    return lam * x + (1 - lam) * x[index, :]


# --------------------------------------------------------------------------- a7
def accuracy(output, target, topk=(1,)):
    """Top-k hit rate in percent.  classification/utils.py:165-179."""
    maxk = max(topk)
    n = target.shape[0]
    _, idx = output.topk(maxk, 1, True, True)
    hit = idx.t().eq(target[None])
    return [hit[:k].flatten().sum(dtype=torch.float32) * (100.0 / n) for k in topk]


# --------------------------------------------------------------------------- a12
def warmup_factor(it, warmup_iters, warmup_factor0=1.0 / 1000):
    """LR multiplier of the first epoch.  classification/utils.py:182-189."""
    if it >= warmup_iters:
        return 1
    alpha = float(it) / warmup_iters
    return warmup_factor0 * (1 - alpha) + alpha


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4, nesterov=False):
    """One torch.optim.SGD step (dampening 0), in place, as used at
    classification/train.py:199-204: g += wd*p; buf = g on the first step,
    else buf = m*buf + g; p -= lr*(g + m*buf if nesterov else buf).
    ``bufs[i] is None`` marks the first step.
    """
    for i, (p, g) in enumerate(zip(params, grads)):
        d = g
        if weight_decay != 0:
            d = d.add(p, alpha=weight_decay)
        if momentum != 0:
            if bufs[i] is None:
                bufs[i] = d.clone()
            else:
                bufs[i].mul_(momentum).add_(d)
            d = d.add(bufs[i], alpha=momentum) if nesterov else bufs[i]
        p.add_(d, alpha=-lr)
    return bufs


def cosine_lr(base_lr, epoch, t_max, eta_min=0.0):
    """Closed form of CosineAnnealingLR(optimizer, epochs, 0) (train.py:223-225)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def multistep_lr(base_lr, epoch, milestones, gamma):
    """MultiStepLR(optimizer, milestones, gamma) (train.py:226-228)."""
    return base_lr * (gamma ** sum(1 for m in milestones if epoch >= m))
