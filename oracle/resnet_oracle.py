"""CPU restatement of the ResNet forward/backward + SGD training step.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Functional style: a model is
a flat ``dict[str, Tensor]`` that uses the reference's ``state_dict`` key names,
so a state dict taken from the reference model drives this code unchanged (this
is how ``tests/golden/make_golden.py`` pins it).  The arithmetic is torch CPU
fp32 ``conv2d`` / ``batch_norm`` / ``relu`` — the same third-party kernels the
reference dispatches to.  Citations are relative to
/root/reference/classification/.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import iif_oracle

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# name -> (block kind, blocks per stage, groups, width_per_group)
# resnet_pytorch.py:421-551; resnext101_32x4d has no constructor in the
# reference and is built from the same class with [3,4,23,3], 32 groups x 4.
IMAGENET_ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2), 1, 64),
    "resnet34": ("basic", (3, 4, 6, 3), 1, 64),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 1, 64),
    "resnet101": ("bottleneck", (3, 4, 23, 3), 1, 64),
    "resnet152": ("bottleneck", (3, 8, 36, 3), 1, 64),
    "resnext50_32x4d": ("bottleneck", (3, 4, 6, 3), 32, 4),
    "resnext101_32x4d": ("bottleneck", (3, 4, 23, 3), 32, 4),
    "resnext101_32x8d": ("bottleneck", (3, 4, 23, 3), 32, 8),
    "wide_resnet50_2": ("bottleneck", (3, 4, 6, 3), 1, 128),
    # squeeze-and-excitation variants (SEBottleneck, resnet_pytorch.py:320-381; ctors :472-551)
    "se_resnet50": ("bottleneck", (3, 4, 6, 3), 1, 64),
    "se_resnet152": ("bottleneck", (3, 8, 36, 3), 1, 64),
    "se_resnext50_32x4d": ("bottleneck", (3, 4, 6, 3), 32, 4),
}
CIFAR_ARCHS = {"resnet20": (3, 3, 3), "resnet32": (5, 5, 5), "resnet44": (7, 7, 7),
               "resnet56": (9, 9, 9), "resnet110": (18, 18, 18),
               "se_resnet32": (5, 5, 5)}          # Se_Block, resnet_cifar.py:140-171, ctor :221-222
SE_REDUCTION = 16                                 # SE_Block(c, r=16), resnet_pytorch.py:303
SE_REDUCTION_CIFAR = 4                            # SE_Block(c, r=4), resnet_cifar.py:91


def is_se(arch):
    return arch.startswith("se_")


# ------------------------------------------------------------------ parameters
def _bn_entries(sd, prefix, c):
    sd[prefix + ".weight"] = torch.ones(c)
    sd[prefix + ".bias"] = torch.zeros(c)
    sd[prefix + ".running_mean"] = torch.zeros(c)
    sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _conv_entry(sd, name, cout, cin_g, k, gen, mode):
    w = torch.empty(cout, cin_g, k, k)
    fan = (cout if mode == "fan_out" else cin_g) * k * k
    std = math.sqrt(2.0 / fan)
    sd[name] = w.normal_(0, std, generator=gen)


def init_imagenet(arch, num_classes, seed=0):
    """Random-init state dict with the reference's keys and init law.

    resnet_pytorch.py:172-226: kaiming-normal (fan_out, relu) conv weights,
    BN weight 1 / bias 0, ``nn.Linear`` default init for ``fc``.
    """
    kind, layers, groups, wpg = IMAGENET_ARCHS[arch]
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    _conv_entry(sd, "conv1.weight", 64, 3, 7, g, "fan_out")
    _bn_entries(sd, "bn1", 64)
    inpl = 64
    exp = 4 if kind == "bottleneck" else 1
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), layers)):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 0) else 1
            p = "layer%d.%d" % (li + 1, b)
            if kind == "bottleneck":
                width = int(planes * (wpg / 64.0)) * groups
                _conv_entry(sd, p + ".conv1.weight", width, inpl, 1, g, "fan_out")
                _bn_entries(sd, p + ".bn1", width)
                _conv_entry(sd, p + ".conv2.weight", width, width // groups, 3, g, "fan_out")
                _bn_entries(sd, p + ".bn2", width)
                _conv_entry(sd, p + ".conv3.weight", planes * 4, width, 1, g, "fan_out")
                _bn_entries(sd, p + ".bn3", planes * 4)
            else:
                _conv_entry(sd, p + ".conv1.weight", planes, inpl, 3, g, "fan_out")
                _bn_entries(sd, p + ".bn1", planes)
                _conv_entry(sd, p + ".conv2.weight", planes, planes, 3, g, "fan_out")
                _bn_entries(sd, p + ".bn2", planes)
            if b == 0 and (stride != 1 or inpl != planes * exp):
                _conv_entry(sd, p + ".downsample.0.weight", planes * exp, inpl, 1, g, "fan_out")
                _bn_entries(sd, p + ".downsample.1", planes * exp)
            if is_se(arch):
                # nn.Linear default init (bias=False); the init loop at :221-226 touches conv and BN only
                c, h = planes * exp, planes * exp // SE_REDUCTION
                sd[p + ".se.excitation.0.weight"] = torch.empty(h, c).uniform_(-1 / math.sqrt(c), 1 / math.sqrt(c), generator=g)
                sd[p + ".se.excitation.2.weight"] = torch.empty(c, h).uniform_(-1 / math.sqrt(h), 1 / math.sqrt(h), generator=g)
            inpl = planes * exp
    bound = 1.0 / math.sqrt(inpl)
    sd["fc.weight"] = torch.empty(num_classes, inpl).uniform_(-bound, bound, generator=g)
    sd["fc.bias"] = torch.empty(num_classes).uniform_(-bound, bound, generator=g)
    return sd


def init_cifar(arch, num_classes, seed=0):
    """resnet_cifar.py:174-193 with ``_weights_init`` (:33-36): kaiming-normal
    (fan_in) for conv and linear weights; linear bias keeps nn.Linear default."""
    nbs = CIFAR_ARCHS[arch]
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    _conv_entry(sd, "conv1.weight", 16, 3, 3, g, "fan_in")
    _bn_entries(sd, "bn1", 16)
    inpl = 16
    for li, (planes, nb) in enumerate(zip((16, 32, 64), nbs)):
        for b in range(nb):
            p = "layer%d.%d" % (li + 1, b)
            _conv_entry(sd, p + ".conv1.weight", planes, inpl, 3, g, "fan_in")
            _bn_entries(sd, p + ".bn1", planes)
            _conv_entry(sd, p + ".conv2.weight", planes, planes, 3, g, "fan_in")
            _bn_entries(sd, p + ".bn2", planes)
            if is_se(arch):     # _weights_init (:33-36) re-initialises every nn.Linear: kaiming-normal, fan_in
                h = planes // SE_REDUCTION_CIFAR
                sd[p + ".se.excitation.0.weight"] = torch.empty(h, planes).normal_(0, math.sqrt(2.0 / planes), generator=g)
                sd[p + ".se.excitation.2.weight"] = torch.empty(planes, h).normal_(0, math.sqrt(2.0 / h), generator=g)
            inpl = planes
    sd["linear.weight"] = torch.empty(num_classes, 64).normal_(0, math.sqrt(2.0 / 64), generator=g)
    bound = 1.0 / math.sqrt(64)
    sd["linear.bias"] = torch.empty(num_classes).uniform_(-bound, bound, generator=g)
    return sd


def trainable_keys(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


# --------------------------------------------------------------------- forward
def bf16_storage(t):
    """Round to bf16 and back (straight-through for autograd): emulates the
    product's bf16 STORAGE of a tensor whose arithmetic stays fp32."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _id(t):
    return t


class ReluMasks(object):
    """Optional externally supplied ReLU decisions (NCHW bool tensors, in forward
    order).  ``relu(t)`` becomes ``t * mask``: identical to ``F.relu`` wherever the
    mask equals ``t > 0`` and, where it does not, ``t`` is within rounding of zero
    (tests assert that), so values agree while the BACKWARD uses exactly the given
    decisions.  Lets a test compare gradients of two fp32 implementations without
    the ambiguity of pre-activations that round to different signs."""

    def __init__(self, masks):
        self.masks, self.i, self.disagree, self.total, self.worst = list(masks), 0, 0, 0, 0.0

    def __call__(self, t):
        m = self.masks[self.i]
        self.i += 1
        own = t.detach() > 0
        diff = own != m
        self.disagree += int(diff.sum())
        self.total += m.numel()
        if diff.any():
            scale = float(t.detach().abs().max())
            self.worst = max(self.worst, float(t.detach()[diff].abs().max()) / max(scale, 1e-30))
        return t * m.to(t.dtype)


_RELU = [F.relu]


def _relu(t):
    return _RELU[0](t)


def _bn(sd, prefix, x, training):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training,
                        BN_MOMENTUM, BN_EPS)


def _count_bn(sd, prefix, training):
    if training:
        sd[prefix + ".num_batches_tracked"] += 1


def head_forward(sd, prefix, x, head="linear", q=_id):
    """Classifier heads.  linear: resnet_pytorch.py:219 / resnet_cifar.py:192.
    cosine / lr_cosine: CosNorm_Classifier.forward, resnet_cifar.py:68-78
    (scale 16, or the learnable ``scale`` parameter squared).  norm:
    NormedLinear.forward, resnet_cifar.py:46-48 (weight stored [in, out])."""
    w = sd[prefix + ".weight"]
    if head == "linear":
        return F.linear(x, q(w), sd[prefix + ".bias"])
    if head in ("cosine", "lr_cosine"):
        norm_x = torch.norm(x, 2, 1, keepdim=True)
        ex = (norm_x / (1 + norm_x)) * (x / norm_x)
        ew = w / torch.norm(w, 2, 1, keepdim=True)
        scale = sd[prefix + ".scale"] ** 2 if head == "lr_cosine" else 16
        return torch.mm(scale * ex, ew.t())
    if head == "norm":
        return F.normalize(x, dim=1).mm(F.normalize(w, dim=0))
    raise ValueError(head)


def _se(sd, p, o):
    """SE_Block.forward, resnet_pytorch.py:313-317: squeeze (global mean), two bias-free linears with
    ReLU / sigmoid, channel-wise rescale."""
    y = o.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[p + ".se.excitation.0.weight"]))
    y = torch.sigmoid(F.linear(y, sd[p + ".se.excitation.2.weight"]))
    return o * y[:, :, None, None]


def forward_imagenet(sd, x, arch, training=True, q=_id, head="linear"):
    """resnet_pytorch.py:279-295 (stem, 4 stages, GAP, fc); blocks :95-111 and
    :149-169 (stride on the 3x3 = v1.5).  ``q`` (identity by default) is applied
    wherever the MI355X product STORES a tensor (input, weights, conv outputs,
    activated outputs, pooled features): ``q=bf16_storage`` gives the bf16-storage
    variant of the same arithmetic; with the default this is the reference."""
    kind, layers, groups, _ = IMAGENET_ARCHS[arch]
    x = q(F.conv2d(q(x), q(sd["conv1.weight"]), None, 2, 3))
    x = q(_relu(_bn(sd, "bn1", x, training))); _count_bn(sd, "bn1", training)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(layers):
        for b in range(nb):
            p = "layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            idt = x
            if kind == "bottleneck":
                o = q(F.conv2d(x, q(sd[p + ".conv1.weight"])))
                o = q(_relu(_bn(sd, p + ".bn1", o, training))); _count_bn(sd, p + ".bn1", training)
                o = q(F.conv2d(o, q(sd[p + ".conv2.weight"]), None, stride, 1, 1, groups))
                o = q(_relu(_bn(sd, p + ".bn2", o, training))); _count_bn(sd, p + ".bn2", training)
                o = q(F.conv2d(o, q(sd[p + ".conv3.weight"])))
                o = _bn(sd, p + ".bn3", o, training); _count_bn(sd, p + ".bn3", training)
            else:
                o = q(F.conv2d(x, q(sd[p + ".conv1.weight"]), None, stride, 1))
                o = q(_relu(_bn(sd, p + ".bn1", o, training))); _count_bn(sd, p + ".bn1", training)
                o = q(F.conv2d(o, q(sd[p + ".conv2.weight"]), None, 1, 1))
                o = _bn(sd, p + ".bn2", o, training); _count_bn(sd, p + ".bn2", training)
            if is_se(arch):
                o = _se(sd, p, o)
            if (p + ".downsample.0.weight") in sd:
                idt = q(F.conv2d(x, q(sd[p + ".downsample.0.weight"]), None, stride))
                idt = _bn(sd, p + ".downsample.1", idt, training)
                _count_bn(sd, p + ".downsample.1", training)
            x = q(_relu(o + idt))
    x = q(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))
    return head_forward(sd, "fc", x, head, q)


def forward_cifar(sd, x, arch="resnet32", training=True, q=_id, head="linear"):
    """resnet_cifar.py:204-212; blocks :133-138; option-A shortcut :125-126
    (spatial ::2 subsample, planes//4 zero channels on each side).  ``q``: see
    forward_imagenet."""
    nbs = CIFAR_ARCHS[arch]
    x = q(F.conv2d(q(x), q(sd["conv1.weight"]), None, 1, 1))
    x = q(_relu(_bn(sd, "bn1", x, training))); _count_bn(sd, "bn1", training)
    inpl = 16
    for li, (planes, nb) in enumerate(zip((16, 32, 64), nbs)):
        for b in range(nb):
            p = "layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            o = q(F.conv2d(x, q(sd[p + ".conv1.weight"]), None, stride, 1))
            o = q(_relu(_bn(sd, p + ".bn1", o, training))); _count_bn(sd, p + ".bn1", training)
            o = q(F.conv2d(o, q(sd[p + ".conv2.weight"]), None, 1, 1))
            o = _bn(sd, p + ".bn2", o, training); _count_bn(sd, p + ".bn2", training)
            if is_se(arch):
                o = _se(sd, p, o)
            sc = x
            if stride != 1 or inpl != planes:
                sc = F.pad(x[:, :, ::2, ::2], (0, 0, 0, 0, planes // 4, planes // 4))
            x = q(_relu(o + sc))
            inpl = planes
    x = q(F.avg_pool2d(x, x.size(3)).view(x.size(0), -1))
    return head_forward(sd, "linear", x, head, q)


def forward(sd, x, arch, training=True, q=_id, head="linear"):
    if arch in CIFAR_ARCHS:
        return forward_cifar(sd, x, arch, training, q, head)
    return forward_imagenet(sd, x, arch, training, q, head)


def set_head(sd, arch, num_classes, head, seed=0):
    """Replace the linear classifier entries by the ones of another head type
    (init laws of resnet_cifar.py:42-44 and :63-65)."""
    prefix = "linear" if arch in CIFAR_ARCHS else "fc"
    d = sd[prefix + ".weight"].shape[1]
    g = torch.Generator().manual_seed(seed + 17)
    if head == "norm":
        w = torch.empty(d, num_classes).uniform_(-1, 1, generator=g)
        sd[prefix + ".weight"] = w.renorm_(2, 1, 1e-5).mul_(1e5)
        sd[prefix + ".bias"] = torch.randn(num_classes, generator=g)
    elif head in ("cosine", "lr_cosine"):
        bound = 1.0 / math.sqrt(d)
        sd[prefix + ".weight"] = torch.empty(num_classes, d).uniform_(-bound, bound, generator=g)
        del sd[prefix + ".bias"]
        if head == "lr_cosine":
            sd[prefix + ".scale"] = 5.0 * torch.ones(1)
    return sd


# ----------------------------------------------------------------- train step
def loss_and_grads(sd, x, y, table, arch, class_weight=None, reduction="mean", q=_id, relu_masks=None,
                   head="linear"):
    """Forward (train mode, running stats updated in ``sd``), IIF loss
    (custom.py:28-36) and autograd gradients for every trainable key.
    Returns (loss, logits, {key: grad})."""
    keys = trainable_keys(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    if relu_masks is not None:
        _RELU[0] = relu_masks
    try:
        logits = forward(work, x, arch, training=True, q=q, head=head)
    finally:
        _RELU[0] = F.relu
    for k in sd:                                   # running stats / counters
        if k not in leaves:
            sd[k] = work[k]
    loss = iif_oracle.iif_ce(logits, y, table, class_weight, reduction)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = [torch.zeros_like(leaves[k]) if g is None else g for k, g in zip(keys, grads)]   # e.g. NormedLinear.bias
    return loss.detach(), logits.detach(), dict(zip(keys, grads))


def train_step(sd, bufs, x, y, table, arch, lr, momentum=0.9, weight_decay=1e-4,
               nesterov=False, class_weight=None, reduction="mean", q=_id, relu_masks=None, head="linear"):
    """One iteration of classification/train.py:60-78 without the logging:
    forward, loss, backward, SGD.  ``bufs`` is a dict key -> momentum buffer
    (missing = first step).  Updates ``sd`` and ``bufs`` in place."""
    loss, logits, grads = loss_and_grads(sd, x, y, table, arch, class_weight, reduction, q, relu_masks, head)
    keys = list(grads.keys())
    params = [sd[k] for k in keys]
    blist = [bufs.get(k) for k in keys]
    with torch.no_grad():
        iif_oracle.sgd_step(params, [grads[k] for k in keys], blist, lr, momentum,
                            weight_decay, nesterov)
    for k, b in zip(keys, blist):
        bufs[k] = b
    return loss, logits
