"""ImageNet-style ResNet constructors with the reference's names and keyword
surface (classification/resnet_pytorch.py:421-551), built on the native MI355X
engine.  ``train.py`` selects a model with
``eval(f'resnet_pytorch.{name}(num_classes=..., use_norm="...", pretrained="...")')``
(train.py:184) — ``use_norm`` / ``pretrained`` therefore arrive as strings and
``"None"`` means absent.

Extra keywords (ignored by the reference surface, defaulted here):
``device`` and ``compute_dtype`` (torch.bfloat16 = performance mode,
torch.float32 = exact-fp32 parity mode).
"""
import torch

from .resnet_engine import NativeResNet

__all__ = ["ResNet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152", "wide_resnet50_2",
           "wide_resnet101_2", "resnext50_32x4d", "resnext101_32x4d", "resnext101_32x8d", "se_resnet50", "se_resnet152",
           "se_resnext50_32x4d"]


def _absent(v):
    return v is None or v == "None" or v is False


def ResNet(block, layers, use_norm=None, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
           device="cuda", compute_dtype=torch.bfloat16):
    """``block``: "basic", "bottleneck" or "se_bottleneck" (SEBottleneck, resnet_pytorch.py:320-381)."""
    se = block == "se_bottleneck"
    block = "bottleneck" if se else block
    un = None if _absent(use_norm) else use_norm
    if un not in (None, "cosine", "lr_cosine", "norm"):
        un = None                       # the reference falls through to nn.Linear for any other string
    return NativeResNet("imagenet", block, list(layers), num_classes, groups=groups, width_per_group=width_per_group,
                        device=device, compute_dtype=compute_dtype, zero_init_residual=zero_init_residual, use_norm=un,
                        se=se)


def _resnet(block, layers, pretrained, use_norm, **kwargs):
    model = ResNet(block, layers, use_norm, **kwargs)
    if not _absent(pretrained):
        if pretrained == "pytorch":
            raise RuntimeError("no network access: pass a local checkpoint path as `pretrained`")
        _load_mismatched(model, pretrained)
    return model


def _load_mismatched(model, path):
    """Backbone weights from a checkpoint whose classifier has a different width
    (resnet_pytorch.py:383-397 swaps a 1000-way fc in and out; here the classifier
    entries are simply skipped)."""
    sd = torch.load(path, map_location="cpu", weights_only=False)    # the reference's checkpoint dict pickles its argparse Namespace
    sd = sd.get("model", sd)
    own = model.state_dict()
    keep = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
    model.load_state_dict(keep, strict=False)


def resnet18(pretrained=None, progress=True, use_norm=None, **kw):
    return _resnet("basic", [2, 2, 2, 2], pretrained, use_norm, **kw)


def resnet34(pretrained=None, progress=True, use_norm=None, **kw):
    return _resnet("basic", [3, 4, 6, 3], pretrained, use_norm, **kw)


def resnet50(pretrained=None, progress=True, use_norm=None, **kw):
    return _resnet("bottleneck", [3, 4, 6, 3], pretrained, use_norm, **kw)


def resnet101(pretrained=None, progress=True, use_norm=None, **kw):
    return _resnet("bottleneck", [3, 4, 23, 3], pretrained, use_norm, **kw)


def resnet152(pretrained=None, progress=True, use_norm=None, **kw):
    return _resnet("bottleneck", [3, 8, 36, 3], pretrained, use_norm, **kw)


def wide_resnet50_2(pretrained=None, progress=True, use_norm=None, **kw):
    kw["width_per_group"] = 128
    return _resnet("bottleneck", [3, 4, 6, 3], pretrained, use_norm, **kw)


def wide_resnet101_2(pretrained=None, progress=True, use_norm=None, **kw):
    kw["width_per_group"] = 128
    return _resnet("bottleneck", [3, 4, 23, 3], pretrained, use_norm, **kw)


def resnext50_32x4d(pretrained=None, progress=True, use_norm=None, **kw):
    kw["groups"], kw["width_per_group"] = 32, 4
    return _resnet("bottleneck", [3, 4, 6, 3], pretrained, use_norm, **kw)


def resnext101_32x4d(pretrained=None, progress=True, use_norm=None, **kw):
    """Not defined by the reference (only 50_32x4d and 101_32x8d are); BASELINE config 4 names it."""
    kw["groups"], kw["width_per_group"] = 32, 4
    return _resnet("bottleneck", [3, 4, 23, 3], pretrained, use_norm, **kw)


def resnext101_32x8d(pretrained=None, progress=True, use_norm=None, **kw):
    kw["groups"], kw["width_per_group"] = 32, 8
    return _resnet("bottleneck", [3, 4, 23, 3], pretrained, use_norm, **kw)


def se_resnet50(pretrained=None, progress=True, use_norm=None, **kw):
    """resnet_pytorch.py:537-539."""
    return _resnet("se_bottleneck", [3, 4, 6, 3], pretrained, use_norm, **kw)


def se_resnet152(pretrained=None, progress=True, use_norm=None, **kw):
    """resnet_pytorch.py:472-480."""
    return _resnet("se_bottleneck", [3, 8, 36, 3], pretrained, use_norm, **kw)


def se_resnext50_32x4d(pretrained=None, progress=True, use_norm=None, **kw):
    """resnet_pytorch.py:542-551."""
    kw["groups"] = 32
    kw["width_per_group"] = 4
    return _resnet("se_bottleneck", [3, 4, 6, 3], pretrained, use_norm, **kw)
