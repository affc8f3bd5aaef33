"""Criterion / data factories with the surface of classification/initialisers.py.
Only the IIF and plain cross-entropy criteria are on the hot path; the datasets
are the synthetic long-tailed sets of ``iif_amd.imbalanced_dataset`` (no
torchvision / network here)."""
import torch

from . import custom, imbalanced_dataset


def get_weights(dataset, device="cuda"):
    """Deferred re-weighting class weights ``sum/count`` (initialisers.py:16-19)."""
    c = torch.tensor(dataset.get_cls_num_list(), device=device)
    return c.sum() / c


class _UniformTable(object):
    def __init__(self, n):
        self.n = n

    def get_cls_num_list(self):
        return [1] * self.n


def get_criterion(args, dataset, model, num_classes):
    """initialisers.py:22-48.  'iif' -> fused IIFLoss; 'ce' -> the same fused kernel
    with an all-ones table built by hand (plain softmax cross-entropy); the focal / BCE
    branches of the reference are outside the hot path."""
    device = getattr(args, "device", "cuda")
    weight = get_weights(dataset, device) if getattr(args, "deffered", False) else None
    if args.classif == "iif":
        return custom.IIFLoss(dataset, variant=args.iif, iif_norm=args.iif_norm, reduction=args.reduction,
                              device=device, weight=weight)
    if args.classif == "ce":
        crit = custom.IIFLoss(dataset, variant="raw", reduction=args.reduction, device=device, weight=weight)
        ones = torch.ones(1, num_classes, device=device)
        crit.iif = {k: ones for k in crit.iif}
        crit.is_plain_ce = True
        # nn.CrossEntropyLoss(weight=w, reduction='mean') divides by the sum of the targets' weights
        # (initialisers.py:43-46), unlike IIFLoss whose .mean() divides by the batch size (custom.py:32-33)
        crit.weighted_mean = weight is not None and args.reduction == "mean"
        return crit
    raise NotImplementedError("criterion %r is outside the IIF hot path (SURVEY §2a: FocalLoss hard-codes CUDA tensors "
                              "and is used by no config)" % (args.classif,))


def get_data(args):
    """initialisers.py:51-112: returns (dataset, num_classes, train_loader, test_loader, train_sampler).
    With ``--data-path`` the long-tailed sets are read from the reference's list files (``--train-txt`` /
    ``--eval-txt`` default to the paths hard-coded at initialisers.py:83-100) through ``LT_Dataset`` /
    ``LT_Dataset_Eval``; without it they are synthetic sets of the same shape (no dataset ships with the image)."""
    name = args.dset_name.lower()
    key = {"imagenet": "imagenet_lt", "imagenet_lt": "imagenet_lt", "places_lt": "places_lt", "inat18": "inat18"}.get(name)
    if key is not None and getattr(args, "data_path", ""):
        C, train_txt, eval_txt = imbalanced_dataset.LT_LISTS[key]
        ds, ds_test = imbalanced_dataset.get_dataset_lt(args, C, getattr(args, "train_txt", None) or train_txt,
                                                        getattr(args, "eval_txt", None) or eval_txt)
        ds.num_classes = len(ds.cls_num_list)
    elif name.startswith("cifar"):
        C = 100 if "100" in name else 10
        ds = imbalanced_dataset.synthetic_cifar_lt(C, args.imb_type, args.imb_factor, args.rand_number, True)
        ds_test = imbalanced_dataset.synthetic_cifar_lt(C, args.imb_type, args.imb_factor, args.rand_number, False)
    else:
        if key is None:
            raise KeyError("unknown dataset %r" % (args.dset_name,))
        ds = imbalanced_dataset.synthetic_lt(key, args.rand_number, True, getattr(args, "synthetic_scale", 1.0))
        ds_test = imbalanced_dataset.synthetic_lt(key, args.rand_number, False)
    sampler = test_sampler = None
    mode = getattr(args, "sampler", "random")
    if mode != "random":                     # initialisers.py:154-171: class-balanced index stream
        from .samplers import BalanceClassSampler, DistributedSamplerWrapper
        sampler = BalanceClassSampler(ds.targets, mode=mode)
        if getattr(args, "distributed", False):
            sampler = DistributedSamplerWrapper(sampler)
            test_sampler = torch.utils.data.distributed.DistributedSampler(ds_test, shuffle=False)
    elif getattr(args, "distributed", False):
        sampler = torch.utils.data.distributed.DistributedSampler(ds)
        test_sampler = torch.utils.data.distributed.DistributedSampler(ds_test, shuffle=False)
    pin = torch.cuda.is_available()
    loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                                         num_workers=args.workers, pin_memory=pin, drop_last=True)
    loader_test = torch.utils.data.DataLoader(ds_test, batch_size=args.batch_size, shuffle=False, sampler=test_sampler,
                                              num_workers=args.workers, pin_memory=pin)
    return ds, ds.num_classes, loader, loader_test, sampler
