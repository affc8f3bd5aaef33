"""Many / median / low-shot accuracy report (what classification/per_shot_acc.py:62-106 returns), as three
``np.bincount`` passes over the label space instead of a per-class scan: host-side integer counting."""
import numpy as np
import torch


def _as_int_array(v, what):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    elif not isinstance(v, (np.ndarray, list, tuple)):
        raise TypeError("Type ({}) of {} not supported".format(type(v), what))
    return np.asarray(v).astype(np.int64).reshape(-1)


def shot_acc(preds, labels, train_targets, many_shot_thr=100, low_shot_thr=20, acc_per_cls=False):
    """Mean per-class accuracy over the classes PRESENT in ``labels``, split by how often the class occurs in the
    training set: many (> many_shot_thr), low (< low_shot_thr), median (the rest).  An empty split reports 0.
    With ``acc_per_cls`` the per-class accuracies (ascending class id, present classes only) are appended."""
    if not isinstance(preds, (torch.Tensor, np.ndarray)):
        raise TypeError("Type ({}) of preds not supported".format(type(preds)))
    pred, gt, seen = _as_int_array(preds, "preds"), _as_int_array(labels, "labels"), _as_int_array(train_targets, "train_targets")
    width = int(max(gt.max(initial=-1), seen.max(initial=-1))) + 1
    n_test = np.bincount(gt, minlength=width)
    n_hit = np.bincount(gt[pred == gt], minlength=width)
    n_train = np.bincount(seen, minlength=width)
    present = n_test > 0
    acc = n_hit[present] / n_test[present]
    freq = n_train[present]
    splits = (freq > many_shot_thr, (freq <= many_shot_thr) & (freq >= low_shot_thr), freq < low_shot_thr)
    many, median, low = (acc[sel].mean() if sel.any() else np.float64(0) for sel in splits)
    if acc_per_cls:
        return many, median, low, list(acc)
    return many, median, low
