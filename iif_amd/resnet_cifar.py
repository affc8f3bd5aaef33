"""CIFAR ResNet-s constructors (classification/resnet_cifar.py:174-235: 3x3 stem,
three stages of basic blocks with the option-A shortcut, average pool, linear)
on the native MI355X engine.  Same names / keywords as the reference
(``resnet32(num_classes=10, use_norm=None)``); the classifier is exposed as
``model.linear`` (train.py:128-131 relies on that)."""
import torch

from .resnet_engine import NativeResNet

__all__ = ["ResNet_s", "resnet20", "resnet32", "resnet44", "resnet56", "resnet110", "resnet1202", "se_resnet32"]


def ResNet_s(block, num_blocks, num_classes=10, use_norm=None, device="cuda", compute_dtype=torch.bfloat16):
    """``block``: "basic" (BasicBlock) or "se" (Se_Block, resnet_cifar.py:140-171)."""
    un = use_norm if use_norm in ("cosine", "lr_cosine", "norm") else None
    return NativeResNet("cifar", "basic", list(num_blocks), num_classes, device=device, compute_dtype=compute_dtype,
                        use_norm=un, se=block == "se")


def resnet20(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [3, 3, 3], num_classes, use_norm, **kw)


def resnet32(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [5, 5, 5], num_classes, use_norm, **kw)


def resnet44(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [7, 7, 7], num_classes, use_norm, **kw)


def resnet56(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [9, 9, 9], num_classes, use_norm, **kw)


def resnet110(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [18, 18, 18], num_classes, use_norm, **kw)


def resnet1202(num_classes=10, use_norm=None, **kw):
    return ResNet_s("basic", [200, 200, 200], num_classes, use_norm, **kw)


def se_resnet32(num_classes=10, use_norm=None, **kw):
    """resnet_cifar.py:221-222."""
    return ResNet_s("se", [5, 5, 5], num_classes, use_norm, **kw)
