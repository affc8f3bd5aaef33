"""Many / median / low-shot accuracy (classification/per_shot_acc.py:62-106): host-side integer counting."""
import numpy as np
import torch


def shot_acc(preds, labels, train_targets, many_shot_thr=100, low_shot_thr=20, acc_per_cls=False):
    training_labels = np.array(train_targets).astype(int)
    if isinstance(preds, torch.Tensor):
        preds = preds.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    elif not isinstance(preds, np.ndarray):
        raise TypeError("Type ({}) of preds not supported".format(type(preds)))
    train_class_count, test_class_count, class_correct = [], [], []
    for l in np.unique(labels):
        train_class_count.append(len(training_labels[training_labels == l]))
        test_class_count.append(len(labels[labels == l]))
        class_correct.append((preds[labels == l] == labels[labels == l]).sum())
    many_shot, median_shot, low_shot = [], [], []
    for i in range(len(train_class_count)):
        acc = class_correct[i] / test_class_count[i]
        if train_class_count[i] > many_shot_thr:
            many_shot.append(acc)
        elif train_class_count[i] < low_shot_thr:
            low_shot.append(acc)
        else:
            median_shot.append(acc)
    many_shot = many_shot or [0]
    median_shot = median_shot or [0]
    low_shot = low_shot or [0]
    if acc_per_cls:
        class_accs = [c / cnt for c, cnt in zip(class_correct, test_class_count)]
        return np.mean(many_shot), np.mean(median_shot), np.mean(low_shot), class_accs
    return np.mean(many_shot), np.mean(median_shot), np.mean(low_shot)
