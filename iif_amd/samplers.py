"""Class-balanced sampling used by ``--sampler upsampling|downsampling`` (classification/initialisers.py:154-171).

The reference imports ``BalanceClassSampler`` / ``DistributedSamplerWrapper`` from the third-party package
``catalyst`` (un-pinned: README.md:46 ``pip install catalyst``), which is absent from this image.  This is a
restatement of catalyst's published algorithm (catalyst/data/sampler.py): every class contributes
``samples_per_class`` indices per epoch — the smallest class size for ``downsampling``, the largest for
``upsampling``, or an explicit integer — drawn with replacement only when the class is smaller, then the
concatenation is shuffled.  Randomness comes from numpy's global generator, as in catalyst.
"""
import numpy as np
from torch.utils.data import Sampler
from torch.utils.data.distributed import DistributedSampler


class BalanceClassSampler(Sampler):
    def __init__(self, labels, mode="downsampling"):
        labels = np.array(labels)
        samples_per_class = {label: int((labels == label).sum()) for label in set(labels.tolist())}
        self.lbl2idx = {label: np.arange(len(labels))[labels == label].tolist() for label in set(labels.tolist())}
        if isinstance(mode, str):
            assert mode in ("downsampling", "upsampling")
        if isinstance(mode, int) or mode == "upsampling":
            samples_per_class = mode if isinstance(mode, int) else max(samples_per_class.values())
        else:
            samples_per_class = min(samples_per_class.values())
        self.labels = labels
        self.samples_per_class = samples_per_class
        self.length = self.samples_per_class * len(set(labels.tolist()))

    def __iter__(self):
        indices = []
        for key in sorted(self.lbl2idx):
            replace_flag = self.samples_per_class > len(self.lbl2idx[key])
            indices += np.random.choice(self.lbl2idx[key], self.samples_per_class, replace=replace_flag).tolist()
        assert len(indices) == self.length
        np.random.shuffle(indices)
        return iter(indices)

    def __len__(self):
        return self.length


class DistributedSamplerWrapper(DistributedSampler):
    """Shard the index stream of another sampler across ranks (catalyst's wrapper): every epoch the inner
    sampler is re-drawn and rank r takes the elements r, r+world, ... of the (padded) stream."""

    def __init__(self, sampler, num_replicas=None, rank=None, shuffle=True):
        self.sampler = sampler
        super().__init__(list(range(len(sampler))), num_replicas=num_replicas, rank=rank, shuffle=shuffle)

    def __iter__(self):
        inner = list(self.sampler)
        return iter(inner[i] for i in super().__iter__())
