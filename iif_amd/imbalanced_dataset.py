"""Class-count / class-map arithmetic of the reference's long-tailed datasets
(the part of classification/imbalanced_dataset.py that feeds ``IIFLoss``) plus
synthetic long-tailed datasets of the same shapes, and the list-file datasets
``LT_Dataset`` / ``LT_Dataset_Eval``.  Augmentation is out of scope and
image decoding needs PIL (absent here; pluggable ``loader``); there is no network in
this environment, so the datasets the training entry point builds are generated from a seed.
"""
import numpy as np
import torch
from torch.utils.data import Dataset


def img_num_per_cls(cls_num, n_images, imb_type="exp", imb_factor=0.01):
    """Per-class image counts of the imbalanced CIFAR subsets
    (imbalanced_dataset.py:23-37): python-float power, ``int()`` truncation."""
    img_max = n_images / cls_num
    if imb_type == "exp":
        return [int(img_max * (imb_factor ** (i / (cls_num - 1.0)))) for i in range(cls_num)]
    if imb_type == "step":
        return [int(img_max)] * (cls_num // 2) + [int(img_max * imb_factor)] * (cls_num // 2)
    return [int(img_max)] * cls_num


def lt_class_map(targets, num_classes):
    """Rank classes by descending count and remap labels (imbalanced_dataset.py:112-127).

    The reference's ``np.argsort(-counts)`` is not stable, so its order among
    equal-count classes is unspecified; here ties keep ascending original class
    id (stable sort).  Returns (class_map, remapped_targets, cls_num_list).
    """
    t = np.asarray(targets, dtype=np.int64)
    counts = np.bincount(t, minlength=num_classes)[:num_classes]
    order = np.argsort(-counts, kind="stable")
    class_map = np.empty(num_classes, dtype=np.int64)
    class_map[order] = np.arange(num_classes)
    new_t = class_map[t]
    cls_num_list = np.bincount(new_t, minlength=num_classes)[:num_classes]
    return class_map.tolist(), new_t.tolist(), [int(v) for v in cls_num_list]


def lt_profile(num_classes, top, bottom=5):
    """Exponential head-to-tail profile used for the synthetic ImageNet-LT / Places-LT shapes."""
    return [max(int(top * (bottom / top) ** (i / max(num_classes - 1.0, 1.0))), 1) for i in range(num_classes)]


class SyntheticLT(Dataset):
    """Long-tailed synthetic image classification set: labels follow
    ``cls_num_list`` exactly, images are class-dependent Gaussian blobs generated
    from (seed, index) — already 'normalised', as the reference normalises on the
    host (initialisers.py:59,132)."""

    def __init__(self, cls_num_list, image_size, seed=0, length=None):
        self.cls_num_list = [int(c) for c in cls_num_list]
        self.num_classes = len(self.cls_num_list)
        self.image_size = image_size
        self.seed = seed
        t = np.repeat(np.arange(self.num_classes), self.cls_num_list)
        rng = np.random.RandomState(seed)
        rng.shuffle(t)
        if length is not None:
            t = t[:length]
        self.targets = t.tolist()
        g = torch.Generator().manual_seed(seed)
        self._means = torch.randn(self.num_classes, 3, 1, 1, generator=g) * 0.5

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        y = self.targets[i]
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        x = torch.randn(3, self.image_size, self.image_size, generator=g) + self._means[y]
        return x, y

    def get_cls_num_list(self):
        return self.cls_num_list


def synthetic_cifar_lt(num_classes=100, imb_type="exp", imb_factor=0.01, seed=0, train=True):
    counts = img_num_per_cls(num_classes, 50000, imb_type, imb_factor)
    if not train:
        counts = [10000 // num_classes // 10] * num_classes       # small balanced eval split
    return SyntheticLT(counts, 32, seed if train else seed + 1)


def synthetic_lt(name, seed=0, train=True, scale=1.0):
    shapes = {"imagenet_lt": (1000, 1280), "places_lt": (365, 4980), "inat18": (8142, 1000)}
    C, top = shapes[name]
    counts = lt_profile(C, top) if train else [2] * C
    counts = [max(int(c * scale), 1) for c in counts]
    return SyntheticLT(counts, 224, seed if train else seed + 1)


# ---------------------------------------------------------------------------------------------------
# Real long-tailed list files (ImageNet-LT / Places-LT / iNaturalist: "relative/path label" per line).
# Parsing, class counting and the descending-frequency class map are the reference's host arithmetic
# (classification/imbalanced_dataset.py:100-174); decoding an image needs PIL, which the build image lacks:
# pass ``loader`` (path -> image) or install PIL.
def _default_loader(path):
    if path.endswith(".npy"):                    # pre-decoded HWC uint8 arrays need no image library
        import numpy as np
        return np.load(path)
    try:
        from PIL import Image
    except Exception as e:                       # pragma: no cover - depends on the deployment image
        raise RuntimeError("decoding %s needs PIL (not in this image); pass loader= to LT_Dataset" % path) from e
    with open(path, "rb") as f:
        return Image.open(f).convert("RGB")


def _read_list(root, txt):
    import os
    paths, targets = [], []
    with open(txt) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            paths.append(os.path.join(root, parts[0]))
            targets.append(int(parts[1]))
    return paths, targets


class LT_Dataset(Dataset):
    """imbalanced_dataset.py:100-144: classes are renumbered by descending training frequency."""

    def __init__(self, root, txt, num_classes, transform=None, loader=None):
        import numpy as np
        self.num_classes = num_classes
        self.transform = transform
        self.loader = loader or _default_loader
        self.img_path, raw = _read_list(root, txt)
        self.class_map, self.targets, self.cls_num_list = lt_class_map(raw, num_classes)
        self.classes = np.unique(np.array(self.targets))
        self.class_data = [[] for _ in range(num_classes)]
        for i, j in enumerate(self.targets):
            self.class_data[j].append(i)

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, index):
        sample = self.loader(self.img_path[index])
        if self.transform is not None:
            sample = self.transform(sample)
        return sample, self.targets[index]

    def get_cls_num_list(self):
        return self.cls_num_list


class LT_Dataset_Eval(Dataset):
    """imbalanced_dataset.py:148-174: evaluation list remapped with the TRAINING set's class map."""

    def __init__(self, root, txt, class_map, num_classes, transform=None, loader=None):
        import numpy as np
        self.num_classes = num_classes
        self.transform = transform
        self.loader = loader or _default_loader
        self.class_map = class_map
        self.img_path, raw = _read_list(root, txt)
        self.targets = np.array(self.class_map)[raw].tolist() if raw else []

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, index):
        sample = self.loader(self.img_path[index])
        if self.transform is not None:
            sample = self.transform(sample)
        return sample, self.targets[index]


# ---------------------------------------------------------------------------------------------------
# Host-side tensor transforms for the list datasets (torchvision is not in this image).  Same geometry and
# statistics as the reference's pipelines (imbalanced_dataset.py:189-233: RandomResizedCrop(224) + horizontal
# flip for training, Resize(256) + CenterCrop(224) for evaluation, per-dataset mean / std), its ColorJitter(0.4, 0.4, 0.4,
# hue 0.25 for iNaturalist / 0 otherwise) and, with ``auto_augment`` = "imagenet" / "randaugment" / "cifar", the policy that
# REPLACES the jitter there (imbalanced_dataset.py:210-225) - restated on tensors in iif_amd/augment.py (parity unpinned: no PIL here).
LT_LISTS = {   # initialisers.py:83-100: (classes, train list, eval list) relative to the reference's working directory
    "imagenet_lt": (1000, "../../../datasets/ImageNet-LT/ImageNet_LT_train.txt", "../../../datasets/ImageNet-LT/ImageNet_LT_test.txt"),
    "inat18": (8142, "../../../datasets/train_val2018/iNaturalist18_train.txt", "../../../datasets/train_val2018/iNaturalist18_val.txt"),
    "places_lt": (365, "../../../datasets/places365_standard/Places_LT_train.txt", "../../../datasets/places365_standard/Places_LT_test.txt"),
}


class TensorTransform(object):
    def __init__(self, dset_name, train, size=224, seed=0, auto_augment=None, color_jitter=True):
        from . import augment
        inat = dset_name == "inat18"
        self.colour = None
        if train:
            if auto_augment == "imagenet":
                self.colour = augment.AutoAugmentPolicy("imagenet")
            elif auto_augment == "randaugment":
                self.colour = augment.RandAugment()
            elif auto_augment in ("cifar", "cifar10"):
                self.colour = augment.AutoAugmentPolicy("cifar10")
            else:
                if auto_augment not in (None, "", "None"):
                    # classification/imbalanced_dataset.py:210-225 knows 'imagenet' and 'randaugment' for the list datasets and keeps
                    # its ColorJitter for every other value, silently; so does this, with a warning
                    import warnings
                    warnings.warn("--auto-augment %r is not a policy of the list datasets (imagenet / randaugment / cifar): "
                                  "ColorJitter stays, as in the reference" % (auto_augment,))
                if color_jitter:
                    self.colour = augment.ColorJitter(0.4, 0.4, 0.4, 0.25 if inat else 0.0)
        self.mean = torch.tensor([0.466, 0.471, 0.380] if inat else [0.485, 0.456, 0.406]).view(3, 1, 1)
        self.std = torch.tensor([0.195, 0.194, 0.192] if inat else [0.229, 0.224, 0.225]).view(3, 1, 1)
        self.train, self.size = train, size
        # The generator is created lazily, per PROCESS, from torch.initial_seed(): the DataLoader sets that to
        # base_seed + worker_id in every worker and draws a new base_seed every epoch, so workers, epochs and DDP ranks get
        # different crop / flip sequences (as torchvision transforms on the global RNG do in the reference).  A generator
        # seeded here once would be forked identically into every worker of every epoch.
        self.seed = int(seed)
        self.gen, self._gen_key = None, None

    def _generator(self):
        import os
        key = (os.getpid(), torch.initial_seed())
        if self._gen_key != key:
            self.gen = torch.Generator().manual_seed((torch.initial_seed() + self.seed) % (1 << 63))
            self._gen_key = key
        return self.gen

    def _to_chw(self, img):
        import numpy as np
        a = np.asarray(img)
        if a.ndim == 2:
            a = np.stack([a] * 3, -1)
        t = torch.from_numpy(np.ascontiguousarray(a[..., :3])).permute(2, 0, 1).float()
        return t / 255.0 if a.dtype == np.uint8 else t

    def _resize(self, t, h, w):
        return torch.nn.functional.interpolate(t[None], size=(h, w), mode="bilinear", align_corners=False, antialias=True)[0]

    def __call__(self, img):
        import math
        t = self._to_chw(img)
        _, h, w = t.shape
        s = self.size
        if self.train:                                   # RandomResizedCrop(scale 0.08-1, ratio 3/4-4/3), 10 tries, then flip
            gen = self._generator()
            r = lambda: torch.rand((), generator=gen).item()      # noqa: E731
            for _ in range(10):
                area = h * w * (0.08 + 0.92 * r())
                logr = math.log(3 / 4) + (math.log(4 / 3) - math.log(3 / 4)) * r()
                cw, ch = int(round(math.sqrt(area * math.exp(logr)))), int(round(math.sqrt(area / math.exp(logr))))
                if 0 < cw <= w and 0 < ch <= h:
                    top, left = int(r() * (h - ch + 1)), int(r() * (w - cw + 1))
                    break
            else:
                ch = cw = min(h, w); top, left = (h - ch) // 2, (w - cw) // 2
            t = self._resize(t[:, top:top + ch, left:left + cw], s, s)
            if r() < 0.5:
                t = t.flip(-1)
            if self.colour is not None:                  # on the [0, 1] image, before normalisation, as the reference composes it
                t = self.colour(t.clamp(0.0, 1.0), gen)
        else:                                            # Resize(256 * s / 224) on the short side + CenterCrop(s)
            short = int(round(s * 256 / 224))
            nh, nw = (short, max(int(round(w * short / h)), short)) if h <= w else (max(int(round(h * short / w)), short), short)
            t = self._resize(t, nh, nw)
            top, left = (nh - s) // 2, (nw - s) // 2
            t = t[:, top:top + s, left:left + s]
        return (t - self.mean) / self.std


def get_dataset_lt(args, num_classes, train_txt, eval_txt, loader=None):
    """imbalanced_dataset.py:177-259 without its samplers (built by the caller): the two list datasets, the
    evaluation one remapped with the training class map."""
    size = getattr(args, "image_size", 224)
    train = LT_Dataset(args.data_path, train_txt, num_classes,
                       transform=TensorTransform(args.dset_name, True, size, args.rand_number, auto_augment=getattr(args, "auto_augment", None)),
                       loader=loader)
    ev = LT_Dataset_Eval(args.data_path, eval_txt, train.class_map, num_classes, transform=TensorTransform(args.dset_name, False, size),
                         loader=loader)
    return train, ev
