"""Native ResNet forward/backward engine for MI355X.

One ``NativeResNet`` module owns
  * a flat fp32 PARAMETER ARENA (all weights of the model, each tensor 64-byte
    aligned, conv weights stored [Cout][R][S][Cin] with a 16-element row pitch
    padding) and a parallel GRADIENT ARENA and MOMENTUM ARENA — one fused SGD
    launch and contiguous all-reduce buckets fall out of the layout;
  * ``nn.Parameter`` views into the arena that carry the REFERENCE's names and
    OIHW shapes (channels-last strided views), so ``state_dict`` /
    ``load_state_dict`` / ``model.fc`` / ``model.linear`` interoperate with
    checkpoints of classification/resnet_pytorch.py and resnet_cifar.py;
  * a static per-input-shape PLAN: every activation (NHWC, bf16 or fp32), BN
    statistic block and gradient buffer is allocated once; a step is a fixed
    sequence of C-ABI kernel launches on the current HIP stream (capturable
    into a hipGraph) — no autograd tape, no per-op allocation.

Data flow per conv unit:  x = conv(src)  ->  stats = bn_stats(x)  ->
y = relu(a*x + b [+ residual]).  Backward walks the units in reverse:
bn_backward (mask from the stored y) -> wgrad (x-operand = stored src) ->
dgrad (residual gradients are added in the dgrad epilogue).

Compute dtype: ``torch.bfloat16`` (performance mode: bf16 storage, fp32
accumulate on v_mfma_f32_16x16x32_bf16) or ``torch.float32`` (parity mode: exact
fp32 MFMA, used to match the CPU reference to 1e-4).
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib, ops

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# channels of the space-to-depth stem image: 4 * 3 real, zero padded to 16 (measured: stem forward 0.56 -> 0.26 ms
# and weight gradient 0.44 -> 0.25 ms against a padding of 32, which would keep one tap per K step)
S2D_CPAD = 16


def _round_up(v, m):
    return (v + m - 1) // m * m


# ----------------------------------------------------------------- holders
class ConvParam(nn.Module):
    """Holds one convolution's weight (reference name ``<prefix>.weight``)."""

    def __init__(self, cin, cout, k, stride, pad, groups=1):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad, self.groups = cin, cout, k, stride, pad, groups
        self.cg = cin // groups                   # input channels per group
        self.kdim = k * k * self.cg               # GEMM K of one output channel's row
        self.ldw = _round_up(self.kdim, 16)       # row pitch of the stored [cout][ldw] matrix
        if groups > 1:
            assert cin == cout and cin % groups == 0, "grouped convolutions are width -> width (ResNeXt)"
            self.chunk = max(64, self.cg)         # channels per dense chunk on the MFMA path
            assert cin % self.chunk == 0 and self.chunk % self.cg == 0
        self.weight = nn.Parameter(torch.empty(cout, self.cg, k, k))

    def forward(self, *a, **k):
        raise RuntimeError("ConvParam is a parameter holder; NativeResNet.forward runs the network")


class BNParam(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_features = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class LinearParam(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.out_padded = _round_up(out_features, 8)
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))


class CosNormParam(nn.Module):
    """Cosine classifier (resnet_cifar.py:50-78): weight [out, in]; ``scale`` is 16, or a learnable
    parameter initialised to 5 and used squared (``lr_scale``)."""

    def __init__(self, in_dims, out_dims, scale=16, lr_scale=False):
        super().__init__()
        self.in_features, self.out_features, self.out_dims = in_dims, out_dims, out_dims
        self.out_padded = _round_up(out_dims, 8)
        self.lr_scale = lr_scale
        self.weight = nn.Parameter(torch.empty(out_dims, in_dims))
        if lr_scale:
            self.scale = nn.Parameter(5.0 * torch.ones(1))
        else:
            self.scale = scale


class NormedLinearParam(nn.Module):
    """NormedLinear (resnet_cifar.py:38-48): weight [in, out], column-normalised in forward; the
    reference also registers an (unused) bias."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.out_padded = _round_up(out_features, 8)
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.empty(out_features))


class SELinearParam(nn.Module):
    """Bias-free linear of an SE block (nn.Linear(c, c // r, bias=False), resnet_pytorch.py:306-310)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))


class SEParam(nn.Module):
    """SE_Block (resnet_pytorch.py:301-317, r=16; resnet_cifar.py:89-106, r=4): parameters under the
    reference's names ``se.excitation.0.weight`` / ``se.excitation.2.weight``."""

    def __init__(self, c, r):
        super().__init__()
        self.c, self.hidden = c, c // r
        self.excitation = nn.Sequential(SELinearParam(c, c // r), nn.Identity(), SELinearParam(c // r, c), nn.Identity())


class BlockParam(nn.Module):
    """conv/bn pairs of one residual block, registered in the reference's order."""

    def __init__(self, kind, inplanes, planes, stride, width, out_planes, downsample, shortcut_a=False, groups=1,
                 se_reduction=0):
        super().__init__()
        self.kind, self.stride, self.shortcut_a = kind, stride, shortcut_a
        self.inplanes, self.out_planes = inplanes, out_planes
        if kind == "bottleneck":
            self.conv1 = ConvParam(inplanes, width, 1, 1, 0); self.bn1 = BNParam(width)
            self.conv2 = ConvParam(width, width, 3, stride, 1, groups); self.bn2 = BNParam(width)
            self.conv3 = ConvParam(width, out_planes, 1, 1, 0); self.bn3 = BNParam(out_planes)
        else:
            self.conv1 = ConvParam(inplanes, planes, 3, stride, 1); self.bn1 = BNParam(planes)
            self.conv2 = ConvParam(planes, planes, 3, 1, 1); self.bn2 = BNParam(planes)
        if downsample:
            self.downsample = nn.Sequential(ConvParam(inplanes, out_planes, 1, stride, 0), BNParam(out_planes))
        else:
            self.downsample = None
        self.se = SEParam(out_planes, se_reduction) if se_reduction else None

    def units(self):
        if self.kind == "bottleneck":
            return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


# --------------------------------------------------------------- the network
class NativeResNet(nn.Module):
    """ImageNet-style (``style='imagenet'``) or CIFAR-style (``style='cifar'``) ResNet."""

    def __init__(self, style, block, layers, num_classes, groups=1, width_per_group=64, device="cuda",
                 compute_dtype=torch.bfloat16, zero_init_residual=False, use_norm=None, se=False):
        super().__init__()
        assert compute_dtype in (torch.bfloat16, torch.float32)
        self.style, self.block_kind = style, block
        self.compute_dtype = compute_dtype
        self.num_classes = num_classes
        self._plans = {}
        self._saved = None
        self._head_only = False
        self._sync_bn = None            # (process group, world size) once enable_sync_bn() was called
        if style == "imagenet":
            self.conv1 = ConvParam(3, 64, 7, 2, 3); self.bn1 = BNParam(64)
            exp = 4 if block == "bottleneck" else 1
            inpl = 64
            stages = []
            for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), layers)):
                blocks = []
                for b in range(nb):
                    stride = 2 if (b == 0 and li > 0) else 1
                    width = int(planes * (width_per_group / 64.0)) * groups
                    ds = b == 0 and (stride != 1 or inpl != planes * exp)
                    blocks.append(BlockParam(block, inpl, planes, stride, width, planes * exp, ds,
                                             groups=groups if block == "bottleneck" else 1, se_reduction=16 if se else 0))
                    inpl = planes * exp
                stages.append(nn.Sequential(*blocks))
            self.layer1, self.layer2, self.layer3, self.layer4 = stages
            self.fc = self._make_head(use_norm, inpl, num_classes)
            self._stages = stages
        else:
            self.conv1 = ConvParam(3, 16, 3, 1, 1); self.bn1 = BNParam(16)
            inpl = 16
            stages = []
            for li, (planes, nb) in enumerate(zip((16, 32, 64), layers)):
                blocks = []
                for b in range(nb):
                    stride = 2 if (b == 0 and li > 0) else 1
                    sa = stride != 1 or inpl != planes
                    blocks.append(BlockParam("basic", inpl, planes, stride, planes, planes, False, shortcut_a=sa,
                                             se_reduction=4 if se else 0))
                    inpl = planes
                stages.append(nn.Sequential(*blocks))
            self.layer1, self.layer2, self.layer3 = stages
            self.linear = self._make_head(use_norm, inpl, num_classes)
            self._stages = stages
        # the stem runs as a GEMM over gathered patches: K padded to a multiple of 32
        self.conv1.ldw = _round_up(self.conv1.kdim, 32)
        self._init_parameters(zero_init_residual)
        self._flatten(torch.device(device))

    @staticmethod
    def _make_head(use_norm, in_features, num_classes):
        """resnet_pytorch.py:212-219 / resnet_cifar.py:185-192."""
        if use_norm == "cosine":
            return CosNormParam(in_features, num_classes)
        if use_norm == "lr_cosine":
            return CosNormParam(in_features, num_classes, lr_scale=True)
        if use_norm == "norm":
            return NormedLinearParam(in_features, num_classes)
        return LinearParam(in_features, num_classes)

    @property
    def _head(self):
        return self.fc if self.style == "imagenet" else self.linear

    # ------------------------------------------------------------ init / arena
    def _init_parameters(self, zero_init_residual):
        for m in self.modules():
            if isinstance(m, ConvParam):
                if self.style == "imagenet":      # resnet_pytorch.py:221-223 kaiming_normal_(fan_out, relu)
                    std = math.sqrt(2.0 / (m.cout * m.k * m.k))
                else:                             # resnet_cifar.py:33-36 kaiming_normal_ (fan_in)
                    std = math.sqrt(2.0 / (m.cg * m.k * m.k))
                nn.init.normal_(m.weight, 0.0, std)
            elif isinstance(m, CosNormParam):     # resnet_cifar.py:63-65 uniform(+-1/sqrt(in))
                bound = 1.0 / math.sqrt(m.in_features)
                nn.init.uniform_(m.weight, -bound, bound)
            elif isinstance(m, NormedLinearParam):  # resnet_cifar.py:42-44
                with torch.no_grad():
                    m.weight.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)
                    m.bias.normal_()
            elif isinstance(m, SELinearParam):
                if self.style == "imagenet":      # nn.Linear default (the init loop :221-226 skips Linear)
                    nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                else:                             # _weights_init, resnet_cifar.py:33-36
                    nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, LinearParam):
                if self.style == "imagenet":      # nn.Linear default
                    nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                else:
                    nn.init.kaiming_normal_(m.weight)
                bound = 1.0 / math.sqrt(m.in_features)
                nn.init.uniform_(m.bias, -bound, bound)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BlockParam):
                    nn.init.constant_((m.bn3 if m.kind == "bottleneck" else m.bn2).weight, 0)

    def _param_specs(self):
        """[(module, attr, rows, pitch)] in registration (= reference) order."""
        specs = []
        for m in self.modules():
            if isinstance(m, ConvParam):
                specs.append((m, "weight", m.cout, m.ldw))
            elif isinstance(m, BNParam):
                specs.append((m, "weight", 1, m.num_features))
                specs.append((m, "bias", 1, m.num_features))
            elif isinstance(m, SELinearParam):
                specs.append((m, "weight", m.out_features, m.in_features))
            elif isinstance(m, LinearParam):
                specs.append((m, "weight", m.out_padded, m.in_features))
                specs.append((m, "bias", 1, m.out_padded))
            elif isinstance(m, CosNormParam):
                specs.append((m, "weight", m.out_padded, m.in_features))
                if m.lr_scale:
                    specs.append((m, "scale", 1, 16))
            elif isinstance(m, NormedLinearParam):
                specs.append((m, "weight", m.in_features, m.out_padded))
                specs.append((m, "bias", 1, m.out_padded))
        return specs

    def _flatten(self, device):
        """(Re)build the arenas on ``device`` and point every Parameter at its view."""
        specs = self._param_specs()
        off, offs = 0, []
        for (_, _, rows, pitch) in specs:
            offs.append(off)
            off += _round_up(rows * pitch, 16)
        total = off
        arena = torch.zeros(total, dtype=torch.float32, device=device)
        self._offsets = {}
        with torch.no_grad():
            for (m, attr, rows, pitch), o in zip(specs, offs):
                flat = arena[o:o + rows * pitch].view(rows, pitch)
                old = getattr(m, attr)
                if isinstance(m, ConvParam):
                    view = flat[:, :m.kdim].view(m.cout, m.k, m.k, m.cg).permute(0, 3, 1, 2)
                    m._w2d = flat
                elif isinstance(m, SELinearParam):
                    view = flat
                    m._w2d = flat
                elif isinstance(m, LinearParam):
                    if attr == "weight":
                        view = flat[:m.out_features]
                        m._w2d = flat
                    else:
                        view = flat.view(-1)[:m.out_features]
                        m._b1d = flat.view(-1)
                elif isinstance(m, CosNormParam):
                    if attr == "weight":
                        view = flat[:m.out_features]
                        m._w2d = flat
                    else:
                        view = flat.view(-1)[:1]
                        m._s1d = flat.view(-1)
                elif isinstance(m, NormedLinearParam):
                    if attr == "weight":
                        view = flat[:, :m.out_features]
                        m._w2d = flat
                    else:
                        view = flat.view(-1)[:m.out_features]
                else:
                    view = flat.view(-1)
                view.copy_(old.detach().to(device))
                old.data = view
                self._offsets[(id(m), attr)] = (o, rows, pitch)
        self._arena = arena
        self._grad_arena = torch.zeros_like(arena)
        self._mom_arena = torch.zeros_like(arena)
        # running statistics / counters: flat buffers too
        bns = [m for m in self.modules() if isinstance(m, BNParam)]
        ctot = sum(_round_up(b.num_features, 16) for b in bns)
        self._rstat = torch.zeros(2 * ctot, dtype=torch.float32, device=device)
        self._nbt = torch.zeros(len(bns), dtype=torch.long, device=device)
        o = 0
        with torch.no_grad():
            for i, b in enumerate(bns):
                c = b.num_features
                rm, rv = self._rstat[o:o + c], self._rstat[ctot + o:ctot + o + c]
                rm.copy_(b.running_mean.to(device)); rv.copy_(b.running_var.to(device))
                b.running_mean, b.running_var = rm, rv
                nb = self._nbt[i]
                nb.copy_(b.num_batches_tracked.to(device))
                b.num_batches_tracked = nb
                o += _round_up(c, 16)
        self._device = device
        self._plans = {}
        self._saved = None
        # gradient views (KRSC storage seen as OIHW), same order as parameters()
        self._grad_views = []
        for (m, attr, rows, pitch), o2 in zip(specs, offs):
            gflat = self._grad_arena[o2:o2 + rows * pitch].view(rows, pitch)
            if isinstance(m, ConvParam):
                gv = gflat[:, :m.kdim].view(m.cout, m.k, m.k, m.cg).permute(0, 3, 1, 2)
                m._g2d = gflat
            elif isinstance(m, SELinearParam):
                gv = gflat
                m._g2d = gflat
            elif isinstance(m, LinearParam):
                if attr == "weight":
                    gv = gflat[:m.out_features]; m._g2d = gflat
                else:
                    gv = gflat.view(-1)[:m.out_features]; m._gb1d = gflat.view(-1)
            elif isinstance(m, CosNormParam):
                if attr == "weight":
                    gv = gflat[:m.out_features]; m._g2d = gflat
                else:
                    gv = gflat.view(-1)[:1]; m._gs1d = gflat.view(-1)
            elif isinstance(m, NormedLinearParam):
                if attr == "weight":
                    gv = gflat[:, :m.out_features]; m._g2d = gflat
                else:
                    gv = gflat.view(-1)[:m.out_features]
            else:
                gv = gflat.view(-1)
                if attr == "weight":
                    m._dgamma = gv
                else:
                    m._dbeta = gv
            self._grad_views.append(gv)
        self._param_list = [getattr(m, attr) for (m, attr, _, _) in specs]

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        dev = next(self.parameters()).device
        if next(self.parameters()).dtype != torch.float32:
            raise RuntimeError("NativeResNet keeps fp32 master weights; choose bf16 compute with compute_dtype=")
        self._flatten(dev)
        return out

    # ------------------------------------------------------------------ plan
    def _units(self):
        """Flat list of descriptors in forward order."""
        seq = [("stem", self.conv1, self.bn1)]
        for st in self._stages:
            for blk in st:
                seq.append(("block", blk))
        seq.append(("head", self._head))
        return seq

    def _plan(self, n, h, w):
        key = (n, h, w, self.compute_dtype)
        if key in self._plans:
            return self._plans[key]
        P = _Plan(self, n, h, w)
        self._plans[key] = P
        return P

    # --------------------------------------------------------------- forward
    def forward(self, x):
        """Drop-in ``model(x)``: returns its OWN [N, num_classes] fp32 tensor (the plan's logits buffer is reused by
        the next same-shaped call, so the module surface never hands out a view of it)."""
        _lib.require_gpu(x)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected an NCHW image batch [N,3,H,W], got %s" % (tuple(x.shape),))
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list)
        if need_grad:
            if not self.training:
                raise RuntimeError("NativeResNet: an eval-mode forward under autograd is not supported (the native backward "
                                   "is the training-mode one: batch statistics, transposed weights); wrap evaluation in "
                                   "torch.no_grad() as classification/train.py:97 does, or call model.train()")
            return _NetFunction.apply(self, x, *self._param_list)
        return self.run_forward(x, self.training).clone()

    def run_forward(self, x, training):
        """Native forward.  Returns fp32 logits [N, num_classes] — a VIEW of the plan's buffer, valid until the next
        forward of the same shape (``loss_and_backward`` consumes it in place; ``forward`` clones it)."""
        self._join_side_work()
        plan = self._plan(x.shape[0], x.shape[2], x.shape[3])
        plan.forward(x, training)
        if training and not torch.is_grad_enabled():
            self._join_side_work()          # no backward will follow (BN recalibration under no_grad): nothing stays un-joined
        plan.generation += 1
        self._saved = plan if training else None
        return plan.logits[:, :self.num_classes]

    def _join_side_work(self):
        """A training-mode forward leaves the transposed weight copies of its step in flight on the weight-gradient stream (backward
        waits for them).  If no backward follows, whoever next touches the master arena or the BN buffers on the compute stream -
        sgd_step, another forward, state_dict / load_state_dict, a graph capture's end - joins that work first (round-5 advice)."""
        for plan in self._plans.values():
            ev = getattr(plan, "_prep_bwd_done", None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                plan._prep_bwd_done = None

    def state_dict(self, *args, **kwargs):
        self._join_side_work()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._join_side_work()
        return super().load_state_dict(*args, **kwargs)

    def run_backward(self, dlogits=None, reducer=None, generation=None):
        """Native backward from d(loss)/d(logits); fills the gradient arena.
        ``dlogits`` None means the plan's own dlogits buffer was already filled
        (fused loss path).  ``generation``: the plan generation the caller's forward produced; a mismatch means the
        saved activations belong to a later forward (two forwards before one backward) and raises instead of
        back-propagating the wrong batch."""
        plan = self._saved
        if plan is None:
            raise RuntimeError("run_backward() without a preceding training-mode forward")
        if generation is not None and generation != (id(plan), plan.generation):
            raise RuntimeError("NativeResNet keeps the activations of ONE forward: backward() was called for a forward whose "
                               "activations a later forward has overwritten (gradient accumulation must run "
                               "forward -> backward per micro-batch)")
        if dlogits is not None:
            plan.dlogits[:, :self.num_classes].copy_(dlogits)
        plan.backward(reducer)

    # -------------------------------------------------- fused training step
    def sgd_step(self, lr, momentum=0.9, weight_decay=1e-4, nesterov=False, grad_scale=1.0):
        """ONE launch over the whole parameter arena (classification/train.py:199-204,78); with a
        frozen backbone only the classifier's slice (the tail of the arena) is stepped, as
        torch.optim.SGD skips parameters without gradients."""
        self._join_side_work()
        lo = self.block_offsets()["head"] if self._head_only else 0
        ops.sgd_step(self._arena[lo:], self._grad_arena[lo:], self._mom_arena[lo:], lr, momentum, weight_decay, nesterov,
                     grad_scale)

    # ------------------------------------------------ optimizer-state interop
    def optimizer_state_dict(self, lr, momentum=0.9, weight_decay=1e-4, nesterov=False, initial_lr=None):
        """The ``torch.optim.SGD.state_dict()`` the reference would have saved (classification/train.py:266-271):
        one param group over ``model.parameters()`` order, ``momentum_buffer`` per parameter taken from the momentum
        arena (as OIHW / reference-shaped tensors), so a reference run can resume from a native checkpoint."""
        views = self._arena_views(self._mom_arena)
        state = {i: {"momentum_buffer": v.detach().clone().contiguous().cpu()} for i, v in enumerate(views)}
        group = {"lr": lr, "momentum": momentum, "dampening": 0, "weight_decay": weight_decay, "nesterov": nesterov,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(views)))}
        if initial_lr is not None:
            group["initial_lr"] = initial_lr
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """Momentum buffers of a reference checkpoint's ``optimizer`` entry -> momentum arena.  Parameters without
        state (never stepped) keep a zero buffer, which is what SGD's lazy initialisation amounts to after step 1
        only; so a state-less parameter is reported by name."""
        views = self._arena_views(self._mom_arena)
        missing = []
        with torch.no_grad():
            for i, v in enumerate(views):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is None or st.get("momentum_buffer") is None:
                    v.zero_(); missing.append(i)
                else:
                    v.copy_(st["momentum_buffer"].to(v.device))
        return missing

    def _arena_views(self, arena):
        """Reference-shaped views of ``arena`` (same layout as the parameter arena), in parameters() order."""
        out = []
        for (m, attr, rows, pitch) in self._param_specs():
            o = self._offsets[(id(m), attr)][0]
            flat = arena[o:o + rows * pitch].view(rows, pitch)
            p = getattr(m, attr)
            if isinstance(m, ConvParam):
                out.append(flat[:, :m.kdim].view(m.cout, m.k, m.k, m.cg).permute(0, 3, 1, 2))
            elif p.dim() == 2:
                out.append(flat[:p.shape[0], :p.shape[1]])
            else:
                out.append(flat.view(-1)[:p.numel()])
        return out

    def select_training_param(self):
        """Decoupled classifier stage (classification/train.py:123-145): freeze everything, re-initialise
        the classifier (xavier-uniform weight, bias 0.01) and train only it.  BN layers keep running in
        training mode, exactly as in the reference (it never switches the backbone to eval)."""
        for p in self.parameters():
            p.requires_grad = False
        head = self._head
        with torch.no_grad():
            nn.init.xavier_uniform_(head.weight)
            if isinstance(head, LinearParam):
                head.bias.fill_(0.01)
        head.weight.requires_grad = True
        if getattr(head, "bias", None) is not None and isinstance(head, LinearParam):
            head.bias.requires_grad = True
        self._head_only = True
        return self

    def enable_sync_bn(self, process_group=None):
        """Batch statistics over all ranks (``--sync-bn``, classification/train.py:190-191: nn.SyncBatchNorm).  Forward:
        per-rank (sum, sum of squares) all-reduced before the finalisation; backward: the mean terms of the BN gradient from
        all-reduced sums, dgamma / dbeta local (averaged by the gradient all-reduce like every other gradient).  One small
        blocking collective per BN layer and direction, as in the reference.  Plans built before the call are dropped."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("enable_sync_bn() needs an initialised process group")
        self._sync_bn = (process_group, dist.get_world_size(process_group))
        self._plans = {}
        self._saved = None
        return self

    def make_reducer(self, bucket_bytes=32 << 20, process_group=None, mode="allreduce", cu_budget=None):
        """Bucketed, backward-overlapped all-reduce of the gradient arena (see iif_amd.ddp)."""
        from .ddp import ArenaReducer
        bounds = [o for (o, _, _) in self._offsets.values()]
        return ArenaReducer(self._grad_arena, bounds, bucket_bytes, process_group, mode=mode, cu_budget=cu_budget)

    def block_offsets(self):
        """Arena offset of the first parameter of the stem, of every block (forward order) and of the head."""
        first = lambda mod: min(self._offsets[(id(m), a)][0] for m in mod.modules()      # noqa: E731
                                for a in ("weight", "bias") if (id(m), a) in self._offsets)
        blocks = [first(b) for st in self._stages for b in st]
        return {"stem": 0, "blocks": blocks, "head": first(self._head)}

    def loss_and_backward(self, x, targets, criterion, targets_b=None, lam=1.0, reducer=None):
        """forward -> fused IIF loss (writes dlogits in the same pass) -> backward.
        Returns (loss 0-dim tensor, logits view).  No autograd involved.  With a
        ``reducer`` the gradient all-reduce is launched bucket by bucket while
        backward is still running."""
        from .custom import IIFLoss
        if not isinstance(criterion, IIFLoss):
            raise TypeError("loss_and_backward needs an iif_amd.custom.IIFLoss criterion")
        logits = self.run_forward(x, True)
        plan = self._saved
        B, C = x.shape[0], self.num_classes
        table = criterion._table(plan.logits)
        scale = 1.0 / B if criterion.reduction == "mean" else 1.0
        cw = criterion.weight
        den = criterion.mean_denominator(targets)
        if den is not None and targets_b is not None:
            # plain CE with class weights, 'mean', under mixup: lam * L_a / sum w[y_a] + (1 - lam) * L_b / sum w[y_b]
            # (custom.py:116-117 on nn.CrossEntropyLoss(weight=w)).  Two denominators fold into the kernel's (lam, scale)
            # pair; they are read back once per step (this rare recipe pays one host sync).
            ka = float(lam) / float(den.item())
            kb = (1.0 - float(lam)) / float(criterion.mean_denominator(targets_b).item())
            scale, lam, den = ka + kb, ka / (ka + kb), None
        rc = _lib.lib().iif_ce_fwd_bwd(
            _lib.ptr(plan.logits), _lib.IIF_F32, plan.logits.stride(0), _lib.ptr(table), _lib.ptr(targets),
            _lib.ptr(targets_b), float(lam), 0, _lib.ptr(cw), -100, scale, B, C, _lib.ptr(plan.loss_rows),
            _lib.ptr(plan.loss), _lib.ptr(plan.dlogits), plan.dlogits.stride(0), _lib.ptr(plan.label_status),
            _lib.ptr(plan.loss_ticket), _lib.stream_ptr())
        _lib.check(rc, "iif_ce_fwd_bwd", plan.loss_ticket[:1])
        if den is not None:
            # plain CE with class weights, 'mean': the launch above used 1/B; rescale loss and dlogits by B / sum w[t]
            plan.loss_rescale.copy_((float(B) / den).reshape(1))
            plan.loss.mul_(plan.loss_rescale[0])
            _lib.check(_lib.lib().iif_scale_by_device_scalar(_lib.ptr(plan.dlogits), _lib.IIF_F32, plan.dlogits.numel(),
                                                             _lib.ptr(plan.loss_rescale), _lib.ptr(plan.dlogits),
                                                             _lib.stream_ptr()), "iif_scale_by_device_scalar")
        plan.backward(reducer)
        return plan.loss, logits

    def check_labels(self):
        """Raise if any target handed to ``loss_and_backward`` since the last check was outside [0, num_classes)
        (the reference asserts on the device in nll_loss).  One host sync: call it where the training loop syncs
        anyway (classification/train.py:87-92, the metric ``.item()`` calls)."""
        for plan in self._plans.values():
            if int(plan.label_status.item()) != 0:
                plan.label_status.zero_()
                raise IndexError("IIF loss: a target label is outside [0, %d) and is not the ignore index" % self.num_classes)

    @property
    def grad_arena(self):
        return self._grad_arena

    @property
    def param_arena(self):
        return self._arena


class _NetFunction(torch.autograd.Function):
    """Autograd bridge: the whole network is one node, so ``loss.backward()`` and
    ``torch.optim`` / DDP work on the drop-in surface (classification/train.py:66-78)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.net = net
        out = net.run_forward(x, True).clone()
        ctx.generation = (id(net._saved), net._saved.generation)
        return out

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        net.run_backward(g, generation=ctx.generation)
        return (None, None) + tuple(net._grad_views)


# ------------------------------------------------------------------- the plan
class _ConvUnit(object):
    """Buffers of one conv+BN unit inside a plan."""
    __slots__ = ("conv", "bn", "src", "x", "stats", "y", "w", "wt", "n", "hi", "wi", "ho", "wo", "is_patch_gemm",
                 "groups", "dwp", "bits", "geom", "s2d", "wf", "wtf")


class _Plan(object):
    def __init__(self, net, n, h, w):
        self.net = net
        self.n, self.h, self.w = n, h, w
        self.generation = 0      # bumped by every forward; autograd nodes remember theirs
        self.dt = net.compute_dtype
        self.dev = net._device
        dt, dev = self.dt, self.dev
        E = lambda *s: torch.empty(s, dtype=dt, device=dev)   # noqa: E731
        self.units = []
        self.steps = []          # structural description used by forward/backward
        # low-precision copy of the whole parameter arena: ONE cast launch per step; a dense convolution's
        # bf16 weights are a view of it (same offsets, same [cout][ldw] rows)
        self.lp_arena = torch.empty_like(net._arena, dtype=dt) if dt != torch.float32 else None
        # ---- stem
        c1 = net.conv1
        ho, wo = ops.conv_out_hw(h, w, c1.k, c1.k, c1.stride, c1.pad)
        # 7x7/2 stem on even images: 4x4/1 convolution over the 2x2 space-to-depth image (no patch matrix);
        # anything else (CIFAR 3x3 stem, odd sizes): GEMM over gathered patches
        self.stem_s2d = c1.k == 7 and c1.stride == 2 and c1.pad == 3 and h % 2 == 0 and w % 2 == 0
        if self.stem_s2d:
            self.patches = E(n, h // 2, w // 2, S2D_CPAD)
        else:
            self.patches = E(n, ho, wo, c1.ldw)
        u = self._unit(c1, net.bn1, self.patches, n, ho, wo, patch=True)
        if self.stem_s2d:
            u.s2d, u.geom = True, (4, 1, 2)
            u.w = torch.empty((c1.cout, 16 * S2D_CPAD), dtype=dt, device=dev)
            u.dwp = torch.empty((c1.cout, 16 * S2D_CPAD), dtype=torch.float32, device=dev)
        self.stem = u
        self.pool_fused = net.style == "imagenet" and net._sync_bn is None
        if net.style == "imagenet":
            self.pool_hw = ((ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1)
            self.pool_out = E(n, self.pool_hw[0], self.pool_hw[1], c1.cout)
            self.pool_idx = torch.empty((n, self.pool_hw[0], self.pool_hw[1], c1.cout), dtype=torch.uint8, device=dev)
            # the raw stem output at each window's arg max: what bn1's backward column sums are taken from (bf16, fused pool)
            self.pool_x = E(n, self.pool_hw[0], self.pool_hw[1], c1.cout) if (self.pool_fused and dt == torch.bfloat16) else None
            cur = self.pool_out
        else:
            cur = u.y
        # ---- blocks
        self.blocks = []
        for st in net._stages:
            for blk in st:
                b = {"blk": blk, "inp": cur}
                units = []
                src = cur
                for (cv, bn) in blk.units():
                    hh, ww = src.shape[1], src.shape[2]
                    oh, ow = ops.conv_out_hw(hh, ww, cv.k, cv.k, cv.stride, cv.pad)
                    uu = self._unit(cv, bn, src, n, oh, ow)
                    units.append(uu)
                    src = uu.y
                b["units"] = units
                last = units[-1]
                if blk.downsample is not None:
                    dcv, dbn = blk.downsample[0], blk.downsample[1]
                    du = self._unit(dcv, dbn, cur, n, last.ho, last.wo, need_y=False)
                    b["ds"] = du
                elif blk.shortcut_a:
                    b["sc"] = E(n, last.ho, last.wo, blk.out_planes)
                if blk.se is not None:
                    Z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)   # noqa: E731
                    c, hid = blk.se.c, blk.se.hidden
                    b["se"] = {"sums": Z(n, c), "q": Z(n, c), "h": Z(n, hid), "e": Z(n, c), "s1": Z(n, c), "s2": Z(n, c),
                               "o": Z(n, c), "w2t": Z(hid, c), "dz2": Z(n, c), "dz1": Z(n, hid)}
                self.blocks.append(b)
                cur = last.y
        self.final = cur
        # ---- head
        head = net._head
        self.pooled = E(n, head.in_features)
        self.logits = torch.zeros((n, head.out_padded), dtype=torch.float32, device=dev)
        self.dlogits = torch.zeros((n, head.out_padded), dtype=torch.float32, device=dev)
        self.dlogits_t = torch.zeros((n, head.out_padded), dtype=dt, device=dev)
        self.loss_rows = torch.zeros(n, dtype=torch.float32, device=dev)
        self.loss_rescale = torch.ones(1, dtype=torch.float32, device=dev)
        self.loss_ticket = torch.zeros(1 + 2048, dtype=torch.int32, device=dev)      # IIF_CE_WORKSPACE_BYTES: ticket + block partials
        self.label_status = torch.zeros(1, dtype=torch.int32, device=dev)     # sticky: 1 once a label was out of range
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        op, D = head.out_padded, head.in_features
        self.head_kind = ("linear" if isinstance(head, LinearParam) else
                          "norm" if isinstance(head, NormedLinearParam) else "cosine")
        F32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)   # noqa: E731
        if self.head_kind == "linear":
            self.head_wsrc = head._w2d                       # fp32 [op, D] matrix the GEMM weights come from
        else:
            self.head_wsrc = F32(op, D)                      # normalised weights (fp32)
            self.head_wnorm = F32(op)
            self.head_ex = E(n, D)                           # mapped features fed to the GEMM
            self.head_xnorm = F32(n)
            self.head_dwn = F32(op, D)                       # gradient w.r.t. the normalised weights
            self.head_dex = E(n, D)
            if self.head_kind == "norm":
                self.head_wT = F32(op, D)                    # W^T: [out, in] rows of the [in, out] parameter
                self.head_dwT = F32(op, D)
            if self.head_kind == "cosine" and head.lr_scale:
                self.head_s2 = F32(1)
        if self.head_kind == "linear" and dt != torch.float32:
            ho_, hr_, hp_ = net._offsets[(id(head), "weight")]
            self.head_w = self.lp_arena[ho_:ho_ + hr_ * hp_].view(hr_, hp_)
        else:
            self.head_w = self.head_wsrc if dt == torch.float32 else torch.empty((op, D), dtype=dt, device=dev)
        self.head_wt = None if self.head_kind == "linear" else torch.zeros((D, _round_up(op, 16)), dtype=dt, device=dev)
        self._finish_weight_plan()
        # ---- scratch
        cmax = max(u.conv.cout for u in self.units)
        mmax = max(u.n * u.ho * u.wo for u in self.units)
        self.bn_ws = ops.bn_workspace(mmax, cmax, dev)
        # ticket words of the single-launch two-stage BN reductions (self-resetting; one set per stream that finalises)
        self.bn_tickets = torch.zeros(64, dtype=torch.int32, device=dev)
        self.bn_tickets_side = torch.zeros(64, dtype=torch.int32, device=dev) if self.bn_tickets is not None else None
        self.bn_partial = torch.empty(max(((u.n * u.ho * u.wo + 127) // 128) * 2 * u.conv.cout for u in self.units),
                                      dtype=torch.float32, device=dev)
        self.bn_scratch = torch.empty(128 * cmax, dtype=torch.float32, device=dev)
        # the convolutional shortcut of a stage's first block runs on the side stream next to the main branch
        ds_units = [b["ds"] for b in self.blocks if "ds" in b]
        self.fwd_side = bool(ds_units) and dt == torch.bfloat16
        if self.fwd_side:
            self.bn_partial_side = torch.empty(max(((u.n * u.ho * u.wo + 127) // 128) * 2 * u.conv.cout for u in ds_units),
                                               dtype=torch.float32, device=dev)
            self.bn_scratch_side = torch.empty(128 * max(u.conv.cout for u in ds_units), dtype=torch.float32, device=dev)
        # BN-backward partial sums written by the data-gradient epilogue that produces a unit's output gradient
        self.bw_partial = torch.empty(max(((u.n * u.ho * u.wo + 127) // 128 + 8) * 2 * u.conv.cout for u in self.units),
                                      dtype=torch.float32, device=dev)
        self.fuse_bwd = dt == torch.bfloat16 and not os.environ.get("IIF_NO_BWD_FUSE")
        self._bw_ready = None
        # BN backward through the expanding 1x1 layer of a bottleneck by algebra (csrc/bn3_algebra.hip): the block-output
        # gradient arrives already gated by the block's ReLU (the producing data gradient stores it so), and conv3's output is
        # never re-read by a BN-backward pass.  Units: the last conv+BN of every bottleneck with <= 256 input channels.
        # Two variants (measured): "pure" takes sum g~ y from P = g~^T a2 (conv3's output is not read at all in
        # backward, 4 passes over that tensor saved, but P sits on the compute stream before the data gradient) - it wins
        # where the tensor is large (>= 1.5e8 elements: the 56 x 56 stage at batch 256); otherwise the producing data gradient
        # still reads conv3's output once for sum g~ xhat (3 passes saved) and P moves to the weight-gradient stream.
        self.wg_lag = int(os.environ.get("IIF_WG_LAG", "2"))      # blocks the weight-gradient stream may lag (explained where the side streams are created; 3 / 4 / 6 measured level)
        self.alg3_units = set()
        # (round 6: 9e7 = the 56 x 56 and 28 x 28 stages at batch 256 - "sums from P" is what lets the forward pass never store
        # conv3's output, see nostore_units below; 1.5e8 = the 56 x 56 stage only, as in rounds 3-5)
        self.a3_pure_min = float(os.environ.get("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "9e7"))
        if self.fuse_bwd and net._sync_bn is None and not os.environ.get("IIF_NO_BN3_ALGEBRA"):
            for bi, b in enumerate(self.blocks[:-1]):
                if "se" in b or "sc" in b or len(b["units"]) != 3:
                    continue
                u3 = b["units"][-1]
                cv3 = u3.conv
                # the route is taken only if the producer of this block's output gradient - the next block's conv1 data
                # gradient - can store it gated and emit the sums (decided here, so that the Gram matrix of a unit is only
                # ever computed for a unit that will use it)
                nxt = self.blocks[bi + 1]
                f = nxt["units"][0]
                producer = ("se" not in nxt and f.conv.k == 1 and f.conv.stride == 1 and f.conv.groups == 1 and _dma_ok(f.x)
                            and _dma_ok(nxt["inp"]))
                if (producer and cv3.k == 1 and cv3.stride == 1 and cv3.groups == 1 and cv3.cin in (64, 128, 256) and cv3.cout % 64 == 0
                        and cv3.cout <= 4096 and _dma_ok(u3.x) and u3.n * u3.ho * u3.wo < (1 << 30)):
                    self.alg3_units.add(u3)
        # Two-pass forward (conv3's raw output is never stored): pass 1 = statistics only, pass 2 = the same convolution with
        # BN + identity + ReLU in its epilogue (bit-identical to conv + bn_apply).  Only where backward never needs that
        # output: "sums from P" algebra units of identity blocks whose gradient producer (the next block's conv1 data gradient)
        # takes the gated-store route.  Measured level with conv + bn_apply (339 against 341 us per 56 x 56 block: the statistics
        # pass costs a whole convolution launch although it stores nothing), so it is opt-in: IIF_TWOPASS=1.
        self.twopass_units = set()
        if self.alg3_units and os.environ.get("IIF_TWOPASS"):
            for bi, b in enumerate(self.blocks[:-1]):
                u3 = b["units"][-1]
                nxt = self.blocks[bi + 1]
                f = nxt["units"][0]
                if (u3 in self.alg3_units and self._a3_is_pure(u3) and "ds" not in b and "se" not in nxt and f.conv.k == 1
                        and f.conv.stride == 1 and f.groups == 1 and _dma_ok(f.y) and _dma_ok(u3.src)):
                    self.twopass_units.add(u3)
        # Never-stored forward on the register-weight kernel (round 6): pass 1 = the convolution's statistics straight from the
        # accumulators (iif_conv_igemm_stats_acc: nothing staged, nothing stored, ~the time of reading a2 once), finalisation,
        # pass 2 = the convolution again with bn3 + identity (or the NORMALISED output of the block's convolutional shortcut) +
        # ReLU + ReLU bits in its epilogue (iif_conv_igemm_bn_relu2).  conv3's raw output - the widest tensor of the block - is
        # neither written nor read: -2 passes over [M, C] per block for one more pass over [M, c].  Units: every "sums from P"
        # algebra unit whose shape the kernel takes, downsample blocks included.  IIF_NO_NOSTORE=1 switches it off.
        # "sums from the producer" WITHOUT the stored output (round 6, iif_conv_igemm_dgrad_masksum_rx): the data gradient that
        # produces g~ recomputes conv3's tile from a2 and W3 on the matrix pipe for its sum g~ xhat.  P leaves the compute stream again
        # (7 launches, 1.4 ms at batch 256: profiles/r6 timeline) and nothing reads conv3's output.  Preferred over "sums from P"
        # wherever the register-weight kernel has the (K, c) instance.  IIF_NO_RX=1 switches it off.
        self.rx_units = set()
        if self.alg3_units and not os.environ.get("IIF_NO_RX"):
            for bi, b in enumerate(self.blocks[:-1]):
                u3 = b["units"][-1]
                f = self.blocks[bi + 1]["units"][0]
                if u3 in self.alg3_units and ops.conv_dgrad_rx_ok(f.n, f.hi, f.wi, f.conv.cout, f.conv.cin, u3.conv.cin, dt):
                    self.rx_units.add(u3)
        self.nostore_units = set()
        if self.alg3_units and not os.environ.get("IIF_NO_NOSTORE"):
            for b in self.blocks[:-1]:
                u3 = b["units"][-1]
                if (u3 in self.alg3_units and (self._a3_is_pure(u3) or u3 in self.rx_units) and u3 not in self.twopass_units
                        and ops.conv_fwdbn_ok(u3.n, u3.ho, u3.wo, u3.conv.cin, u3.conv.cout, dt) and _dma_ok(u3.src)):
                    self.nostore_units.add(u3)
        # bn2 + ReLU in conv3's operand path (round 6, iif_conv_igemm_bnstats_pro): conv3 (or its statistics pass) reads conv2's RAW
        # output, normalises each tile in LDS and writes a2 / its ReLU bits / its column sums as by-products: one launch and one pass
        # over a2 less per bottleneck.  pro_units: conv3 unit -> (conv2 unit, rows of column sums or None).  IIF_NO_PROLOGUE=1: off.
        self.pro_units = {}
        # (K = 512 - the 7 x 7 stage, ResNeXt-101's 14 x 14 one - stays with the separate bn_apply launch: every N slice of the
        # register-weight kernel normalises the whole [rows x K] tile again, 22 us on top of a 39 us launch against a 10 us bn_apply;
        # ResNeXt-101 21.28 -> 21.14 ms, ResNet-50 level, profiles/r6_ab.txt ab11)
        pro_max_k = int(os.environ.get("IIF_PRO_MAX_K", "256"))
        if dt == torch.bfloat16 and net._sync_bn is None and not os.environ.get("IIF_NO_PROLOGUE"):
            for b in self.blocks:
                if "se" in b or len(b["units"]) != 3:
                    continue
                u2, u3 = b["units"][1], b["units"][2]
                cv3 = u3.conv
                if not (cv3.k == 1 and cv3.stride == 1 and cv3.groups == 1 and _dma_ok(u2.x)) or u3 in self.twopass_units:
                    continue
                if cv3.cin > pro_max_k:
                    continue
                if ops.conv_pro_ok(u3.n, u3.ho, u3.wo, cv3.cin, cv3.cout, dt, u3 in self.nostore_units):
                    rows = None
                    if u3 in self.alg3_units:
                        rows = torch.zeros(((u3.n * u3.ho * u3.wo + 127) // 128 + 8) * 2 * cv3.cin, dtype=torch.float32, device=dev)
                    fin = torch.zeros((2, cv3.cin), dtype=torch.float32, device=dev) if rows is not None else None
                    # [conv2 unit, partial rows of a2's column sums, their count in the last forward, instance kind, the sums themselves]
                    self.pro_units[u3] = [u2, rows, 0, u3 in self.nostore_units, fin]
        self.a3 = None
        if self.alg3_units:
            cm = max(u.conv.cin for u in self.alg3_units)
            Cm = max(u.conv.cout for u in self.alg3_units)
            ldm = max(u.conv.ldw for u in self.alg3_units)
            F = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)   # noqa: E731
            # per-block scratch rotates over wg_lag slots (the weight-gradient stream finishes block b before block b - wg_lag
            # starts: _wgrad_fence); Gram / colsum are issued one block AHEAD on another stream, so they rotate over wg_lag + 1
            # "P" holds P = g~^T a2 [C, ldw] and, behind it, the Gram matrix a2^T a2 [c, ldw] where one stacked launch writes both
            self.a3 = [{"P": F(Cm + cm, ldm), "coef": F(3, Cm), "bias": F(cm),
                        "wt": torch.zeros(cm * (Cm + cm), dtype=dt, device=dev),
                        "scr": torch.empty(max(ops.lib().iif_bn3_algebra_prep_scratch_floats(u.conv.cout, u.conv.cin)
                                               for u in self.alg3_units), dtype=torch.float32, device=dev),
                        "tickets": torch.zeros(64, dtype=torch.int32, device=dev)}
                       for _ in range(self.wg_lag)]
            self.a3g = [{"gram": F(cm, ldm), "csum": F(2, cm), "ws_gram": torch.empty(64 << 20, dtype=torch.uint8, device=dev),
                         "ws_sum": ops.bn_workspace(max(u.n * u.ho * u.wo for u in self.alg3_units), cm, dev), "ev": None, "ev_csum": None,
                         "csum_fwd": None}
                        for _ in range(self.wg_lag + 1)]
            self.a3_ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)          # split-K slabs of P on the compute stream
        wmax = max(max(u.conv.cout * (u.dwp.shape[1] if u.dwp is not None else u.conv.ldw) for u in self.units),
                   head.out_padded * head.in_features)
        self.wg_ws = torch.empty(min(16 * wmax * 4, 512 << 20), dtype=torch.uint8, device=dev)
        self._grad_pool = {}
        self._bwd_ready = False
        # Side streams (weight gradients, shortcut branch) pay when the kernels are long enough to hide the events that tie the
        # streams together; a small step is bound by the host's enqueue time and each cross-stream event adds to it.  Measured
        # (scripts/ab_side_small.sh, ms per step with / without): ResNet32 32x32 bs 128 3.24 / 2.52, bs 256 3.24 / 3.33, bs 512
        # 4.20 / 5.05; ResNet50 224 bs 16 6.74 / 6.01, bs 64 7.85 / 9.01; ResNet18 224 bs 64 2.94 / 3.19.  The crossover sits at
        # ~5e6 convolution-output elements per layer and step.  IIF_SIDE_STREAMS=1 / 0 (or IIF_NO_WGRAD_STREAM=1) force either.
        side = os.environ.get("IIF_SIDE_STREAMS")
        if os.environ.get("IIF_NO_WGRAD_STREAM"):
            side = "0"
        if side is None:
            per_unit = sum(u.n * u.ho * u.wo * u.conv.cout for u in self.units) / max(len(self.units), 1)
            side = "1" if per_unit >= 5e6 else "0"
        self.wg_stream = None
        if dev.type == "cuda" and side != "0":
            self.wg_stream = torch.cuda.Stream(device=dev)
        self._wg_events = {}
        # IIF_WG_STREAMS=n: n weight-gradient streams taken round robin, each launch sized for 1/n of the device.  Every block of
        # a split-K weight gradient writes its accumulator tile once, so a round over 256 CUs writes (and the reduction reads
        # back) 32-75 MB of slabs per launch, ~4 GB of the step's 73; n part rounds side by side write 1/n of that.  Measured
        # (round 5, same call, ms per step): 1 stream 17.65, 2 streams 18.74 (the part rounds take 1.3x their share of time:
        # the two streams together are busy 13.2 ms against 10.0, profiles/r5_two_wgrad_streams.txt), 3 streams 29.7 (five
        # streams on four hardware queues).  Default 1.
        self.wg_streams = []
        if self.wg_stream is not None:
            nws = int(os.environ.get("IIF_WG_STREAMS", "1"))
            self.wg_streams = [self.wg_stream] + [torch.cuda.Stream(device=dev) for _ in range(max(nws, 1) - 1)]
        self.wg_wss = [self.wg_ws] + [torch.empty_like(self.wg_ws) for _ in self.wg_streams[1:]]
        self._wg_rr = 0
        self.stem_wgrad_main = self.wg_stream is not None
        # own split-K workspace of the stem's weight gradient (it runs on the compute stream next to the side stream's):
        # allocated here, never inside backward (a lazy allocation there would land inside a hipGraph capture)
        self.wg_ws_stem = (torch.empty(64 << 20, dtype=torch.uint8, device=dev)
                           if (self.stem_wgrad_main and self.stem_s2d) else None)
        # blocks the weight-gradient stream may lag behind the compute stream: dx buffers rotate over `wg_lag` slots,
        # block-input gradients over wg_lag + 1, and block b waits for the weight gradients of the blocks >= b + wg_lag
        # backward of the convolutional shortcut (BN backward + dgrad + wgrad of 4 blocks) on a third stream
        self.ds_stream = None
        if self.wg_stream is not None and ds_units and dt == torch.bfloat16 and not os.environ.get("IIF_NO_BWD_SIDE"):
            self.ds_stream = torch.cuda.Stream(device=dev)
            self.bn_ws_ds = ops.bn_workspace(max(u.n * u.ho * u.wo for u in ds_units), max(u.conv.cout for u in ds_units), dev)
        # BN backward of a stride-1 convolutional shortcut by the same algebra as bn3 (round 5): the shortcut's BN sees the SAME
        # gated block-output gradient g~ as bn3, and its input x_in is the narrow block input, so
        #     d x_in (shortcut part) = [g~ | x_in] [A o Wd ; Wd^T diag(B) Wd]^T + D Wd,   sum g~ y_d = rowdot(g~^T x_in, Wd)
        # replaces the reduction pass, the normalisation pass and the data gradient over the C-wide shortcut output
        # (3.1 -> 1.3 GB at 56 x 56, all of it on the shortcut stream, which the compute stream used to wait ~120 us for at the
        # end of layer1.0).  Units: 1 x 1 / stride 1 shortcuts of a block whose last unit takes the bn3 algebra (ResNet-50: layer1.0).
        self.ds_alg = {}
        self._alg_rows = {}
        if self.ds_stream is not None and self.alg3_units and not os.environ.get("IIF_NO_DS_ALGEBRA"):
            for b in self.blocks:
                du = b.get("ds")
                if du is None or "se" in b or b["units"][-1] not in self.alg3_units:
                    continue
                cv = du.conv
                if (cv.k == 1 and cv.stride == 1 and cv.groups == 1 and cv.cin in (64, 128, 256) and cv.cout % 64 == 0 and cv.cout <= 4096
                        and _dma_ok(du.x) and _dma_ok(b["inp"]) and du.n * du.ho * du.wo < (1 << 30)):
                    F = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)   # noqa: E731
                    C, c = cv.cout, cv.cin
                    self.ds_alg[id(du)] = {
                        "P": F(C, cv.ldw), "coef": F(3, C), "bias": F(c), "wt": torch.zeros((c, C + c), dtype=dt, device=dev),
                        "scr": torch.empty(ops.lib().iif_bn3_algebra_prep_scratch_floats(C, c), dtype=torch.float32, device=dev),
                        "tickets": torch.zeros(64, dtype=torch.int32, device=dev), "gram": F(c, cv.ldw), "csum": F(2, c),
                        "rows": torch.empty(self.bw_partial.numel(), dtype=torch.float32, device=dev),
                        "ws": torch.empty(64 << 20, dtype=torch.uint8, device=dev),
                        "ws_sum": ops.bn_workspace(du.n * du.ho * du.wo, c, dev)}
                    # the data gradient that feeds this block's last unit writes its per-tile sums of g~ straight into "rows":
                    # both branches read them, and the compute stream's next fused data gradient does not overwrite them
                    # (round 5: the copy out of bw_partial was a 65 us blit on the compute stream)
                    self._alg_rows[id(b["units"][-1])] = self.ds_alg[id(du)]["rows"]
        # P = g~^T a2 and Gram = a2^T a2 of an algebra unit as BY-PRODUCTS of the recomputing producer (round 6,
        # iif_conv_igemm_dgrad_masksum_rx_pg): the block that forms a tile of g~ holds it in its staging buffers and a2's tile in LDS;
        # one fp32 slab per tile sequence, summed on the weight-gradient stream (iif_slab_sum).  Replaces the stacked weight-gradient
        # launch that re-read g~ and a2 (0.5 GB per bottleneck at 56 x 56).  Units: c = 64 (the 56 x 56 stage).  IIF_NO_PG=1: off.
        self.pg_units = {}
        if self.rx_units and self.wg_stream is not None and not os.environ.get("IIF_NO_PG"):
            for bi, b in enumerate(self.blocks[:-1]):
                u3 = b["units"][-1]
                f = self.blocks[bi + 1]["units"][0]
                cv3 = u3.conv
                if (u3 in self.rx_units and self._a3_gram_stacked(u3) and cv3.ldw % 4 == 0 and cv3.ldw >= cv3.cin
                        and ops.conv_dgrad_rx_pg_ok(f.n, f.hi, f.wi, f.conv.cout, f.conv.cin, cv3.cin, dt)):
                    # up to 256 slabs (one per resident block) + the second reduction stage's ceil(256 / 16); [slabs, count of the last launch]
                    self.pg_units[u3] = [torch.empty((256 + 17) * (cv3.cout + cv3.cin) * cv3.ldw, dtype=torch.float32, device=dev), 0]

    def _finish_weight_plan(self):
        """One arena for every dense transposed weight copy ([cin][k*k*cout], the data-gradient operand) and
        the device table that lets ONE launch fill it from the fp32 parameter arena."""
        net, entries, views, off = self.net, [], [], 0
        for u in self.units:
            cv = u.conv
            if cv.groups > 1 or u.is_patch_gemm:
                continue
            ldwt = _round_up(cv.k * cv.k * cv.cout, 16)
            entries.append((net._offsets[(id(cv), "weight")][0], off, cv.cout, cv.cin, cv.k * cv.k, cv.ldw, ldwt))
            views.append((u, off, cv.cin, ldwt))
            off += _round_up(cv.cin * ldwt, 64)
        head = net._head
        head_view = None
        if self.head_kind == "linear":
            ldwt = _round_up(head.out_padded, 16)
            entries.append((net._offsets[(id(head), "weight")][0], off, head.out_padded, head.in_features, 1, head.in_features, ldwt))
            head_view = (off, head.in_features, ldwt)
            off += _round_up(head.in_features * ldwt, 64)
        self.wt_arena = torch.zeros(max(off, 1), dtype=self.dt, device=self.dev)
        for (u, o, rows, ld) in views:
            u.wt = self.wt_arena[o:o + rows * ld].view(rows, ld)
        if head_view is not None:
            o, rows, ld = head_view
            self.head_wt = self.wt_arena[o:o + rows * ld].view(rows, ld)
        self.wt_n = len(entries)
        self.wt_table, self.wt_blocks = ops.wt_table(entries, self.dev) if entries else (None, 0)
        # 3x3 / stride-1 layers: the same weights once more as MFMA fragments (forward from the bf16 parameter arena, data
        # gradient from the transposed arena), ONE pack launch per source arena and step; the fragment kernel reads them
        # straight into registers (csrc/conv_igemm.hip, conv3x3_v2_body)
        self.frag_arena, self.frag_fwd, self.frag_bwd = None, None, None
        if self.dt == torch.bfloat16 and not os.environ.get("IIF_CONV_NO_V2"):
            fwd, bwd, off = [], [], 0
            for (u, o, rows, ld) in views:
                cv = u.conv
                if cv.k != 3 or cv.stride != 1 or cv.pad != 1 or cv.cin % 32 or cv.cout % 32:
                    continue
                if ops.conv3x3_frag_ok(u.n, u.hi, u.wi, cv.cin, cv.cout, self.dt):
                    po = net._offsets[(id(cv), "weight")]
                    fwd.append((po[0], off, cv.cout, 9, cv.cin, po[2]))
                    u.wf = off
                    off += cv.cout * 9 * cv.cin
                if ops.conv3x3_frag_ok(u.n, u.ho, u.wo, cv.cout, cv.cin, self.dt):
                    bwd.append((o, off, cv.cin, 9, cv.cout, ld))
                    u.wtf = off
                    off += cv.cout * 9 * cv.cin
            if off:
                self.frag_arena = torch.empty(off, dtype=self.dt, device=self.dev)
                for u in self.units:
                    if isinstance(u.wf, int):
                        u.wf = self.frag_arena[u.wf:]
                    if isinstance(u.wtf, int):
                        u.wtf = self.frag_arena[u.wtf:]
                if fwd:
                    self.frag_fwd = ops.pack_table(fwd, self.dev) + (len(fwd),)
                if bwd:
                    self.frag_bwd = ops.pack_table(bwd, self.dev) + (len(bwd),)
        # grouped 3x3 / stride-1 layers (ResNeXt): the block-diagonal chunk matrices move into ONE arena so that one more pack
        # launch per step turns them into fragments too; the fragment kernel then takes blockIdx.y as the chunk
        self.gfrag = None
        self.g16frag = None
        gunits = [u for u in self.units if u.groups > 1]
        if gunits and self.dt == torch.bfloat16 and not os.environ.get("IIF_CONV_NO_V2"):
            gw = torch.empty(sum(u.w.numel() + u.wt.numel() for u in gunits), dtype=self.dt, device=self.dev)
            entries, off, foff = [], 0, 0
            # round 6: groups of <= 16 channels (ResNeXt 32x4d at 56 / 28 / 14: 4 / 8 / 16 per group) take the 16-channel fragment
            # format - K of an MFMA = two taps x the output tile's own 16 input channels, 80 MFMAs per wave and tile instead of 288
            # (csrc/conv_igemm.hip, conv3x3_v2_body<..., G16>); 20 KB of fragments per 64-channel chunk and orientation
            g16_entries, g16_off = [], 0
            no_g16 = bool(os.environ.get("IIF_NO_G16"))
            for u in gunits:
                cv = u.conv
                nw = u.w.numel()
                ow, owt = off, off + nw
                u.w = gw[ow:ow + nw].view(u.w.shape)
                u.wt = gw[owt:owt + nw].view(u.wt.shape)
                off += 2 * nw
                ok = cv.k == 3 and cv.stride == 1 and cv.pad == 1 and cv.chunk == 64 and cv.cin == cv.cout
                if (ok and not no_g16 and cv.cg <= 16 and 16 % cv.cg == 0
                        and ops.conv3x3_frag_ok(u.n, u.hi, u.wi, 64, 64, self.dt, groups=u.groups)):
                    per = (cv.cout // 64) * 20 * 512                      # bf16 elements of one orientation's fragments
                    g16_entries.append((ow, g16_off, cv.cout, 9, 64, u.w.shape[1]))
                    g16_entries.append((owt, g16_off + per, cv.cin, 9, 64, u.wt.shape[1]))
                    u.wf, u.wtf = ("g16", g16_off), ("g16", g16_off + per)
                    g16_off += 2 * per
                    continue
                if ok and ops.conv3x3_frag_ok(u.n, u.hi, u.wi, 64, 64, self.dt, groups=u.groups):
                    entries.append((ow, foff, cv.cout, 9, 64, u.w.shape[1]))
                    entries.append((owt, foff + nw, cv.cin, 9, 64, u.wt.shape[1]))
                    u.wf, u.wtf = foff, foff + nw
                    foff += 2 * nw
            self.gw_arena = gw
            self.g16frag = None
            if g16_entries:
                self.g16_arena = torch.empty(g16_off, dtype=self.dt, device=self.dev)
                for u in gunits:
                    if isinstance(u.wf, tuple) and u.wf[0] == "g16":
                        u.wf, u.wtf = (self.g16_arena[u.wf[1]:], 1), (self.g16_arena[u.wtf[1]:], 1)
                self.g16frag = ops.pack_table_g16(g16_entries, self.dev) + (len(g16_entries),)
            if entries:
                self.gfrag_arena = torch.empty(foff, dtype=self.dt, device=self.dev)
                for u in gunits:
                    if isinstance(u.wf, int):
                        u.wf, u.wtf = self.gfrag_arena[u.wf:], self.gfrag_arena[u.wtf:]
                self.gfrag = ops.pack_table(entries, self.dev) + (len(entries),)

    def _unit(self, conv, bn, src, n, ho, wo, patch=False, need_y=True):
        dt, dev = self.dt, self.dev
        u = _ConvUnit()
        u.conv, u.bn, u.src = conv, bn, src
        u.n, u.ho, u.wo = n, ho, wo
        u.hi, u.wi = src.shape[1], src.shape[2]
        u.is_patch_gemm = patch
        u.s2d = False
        u.geom = (1, 1, 0) if patch else (conv.k, conv.stride, conv.pad)   # geometry the MFMA kernels see
        u.x = torch.empty((n, ho, wo, conv.cout), dtype=dt, device=dev)
        u.y = torch.empty((n, ho, wo, conv.cout), dtype=dt, device=dev) if need_y else None
        u.stats = torch.empty((4, conv.cout), dtype=torch.float32, device=dev)
        vec = 8 if dt == torch.bfloat16 else 4                     # channels per 16-byte vector
        u.bits = torch.empty(n * ho * wo * conv.cout // vec, dtype=torch.uint8, device=dev) if need_y else None
        u.groups, u.dwp = 1, None
        u.wf = u.wtf = None
        if conv.groups > 1:
            # grouped conv: dense inside chunks of conv.chunk channels, block-diagonal packed weights
            ldp = conv.k * conv.k * conv.chunk
            u.groups = conv.cin // conv.chunk
            u.w = torch.empty((conv.cout, ldp), dtype=dt, device=dev)
            u.wt = torch.empty((conv.cin, ldp), dtype=dt, device=dev)
            u.dwp = torch.empty((conv.cout, ldp), dtype=torch.float32, device=dev)
        else:
            o, rows, pitch = self.net._offsets[(id(conv), "weight")]
            u.w = conv._w2d if dt == torch.float32 else self.lp_arena[o:o + rows * pitch].view(rows, pitch)
            u.wt = None        # dense transposed copies live in one arena, filled by one launch (_finish_weight_plan)
        self.units.append(u)
        return u

    # ---------------------------------------------------------------- weights
    def _group_pack_table(self, transposed):
        key = bool(transposed)
        tabs = self.__dict__.setdefault("_gp_tables", {})
        if key not in tabs:
            ent = []
            for u in self.units:
                cv = u.conv
                if not u.s2d and cv.groups > 1:
                    ent.append((cv._w2d, cv.cout, cv.cg, cv.chunk, cv.k * cv.k, u.w, False))
                    if transposed:
                        ent.append((cv._w2d, cv.cout, cv.cg, cv.chunk, cv.k * cv.k, u.wt, True))
            tabs[key] = ops.group_pack_table(ent, self.dev) if ent else None
        return tabs[key]

    def prepare_weights(self, need_transposed, part="all"):
        """Low-precision / transposed / fragment-packed copies of the fp32 master weights.  part = "stem": only the stem's
        packed matrix; "rest": everything else (forward() runs that on the idle weight-gradient stream under the stem), or its
        two halves: "rest_fwd" (what the forward pass reads) and "rest_bwd" (the dense transposed copies and their fragments:
        data-gradient operands, first read in backward)."""
        net = self.net
        if part not in ("rest", "rest_fwd", "rest_bwd"):
            for u in self.units:
                if u.s2d:
                    ops.stem_s2d_pack(u.conv._w2d, u.conv.cout, u.conv.cin, u.conv.k, S2D_CPAD, u.w)
            if part == "stem":
                return
        if part != "rest_fwd":
            if need_transposed and self.wt_n:
                ops.weight_transpose_batched(net._arena, self.wt_table, self.wt_n, self.wt_blocks, self.wt_arena)
            if need_transposed and self.frag_bwd is not None:
                ops.pack_fragments(self.wt_arena, self.frag_bwd[0], self.frag_bwd[2], self.frag_bwd[1], self.frag_arena)
            if part == "rest_bwd":
                return
        if self.lp_arena is not None:
            ops.cast(net._arena, self.lp_arena)
        if self.frag_fwd is not None:
            ops.pack_fragments(self.lp_arena, self.frag_fwd[0], self.frag_fwd[2], self.frag_fwd[1], self.frag_arena)
        # grouped layers: block-diagonal chunk weights, both orientations, ONE launch (the fragment table packs both, so the
        # transposed chunks are filled in evaluation too)
        tab = self._group_pack_table(need_transposed or self.gfrag is not None or getattr(self, "g16frag", None) is not None)
        if tab is not None:
            ops.group_pack_batched(tab)
        if self.gfrag is not None:
            ops.pack_fragments(self.gw_arena, self.gfrag[0], self.gfrag[2], self.gfrag[1], self.gfrag_arena)
        if getattr(self, "g16frag", None) is not None:
            ops.pack_fragments_g16(self.gw_arena, self.g16frag[0], self.g16frag[2], self.g16frag[1], self.g16_arena)
        head = net._head
        if self.head_kind == "linear":
            return
        if self.head_kind == "cosine":           # ew = W / |W_row|   (resnet_cifar.py:73)
            ops.rowmap_forward(head._w2d, 1, 1.0, self.head_wsrc, self.head_wnorm, eps=0.0)
        else:                                    # ew = W / |W_col|, W is [in, out]   (resnet_cifar.py:47)
            ops.transpose_f32(head._w2d, self.head_wT)
            ops.rowmap_forward(self.head_wT, 1, 1.0, self.head_wsrc, self.head_wnorm)
        if self.dt != torch.float32:
            ops.cast(self.head_wsrc, self.head_w)
        if need_transposed:
            ops.weight_transpose(self.head_wsrc, head.out_padded, head.in_features, 1, self.head_wt)

    # ------------------------------------------------- cross-replica statistics
    def _sync_sums(self, c, side=False, which=0):
        key = ("syncsums", c, bool(side), which)
        t = self._grad_pool.get(key)
        if t is None:
            t = self._grad_pool[key] = torch.empty((2, c), dtype=torch.float32, device=self.dev)
        return t

    def _sync_coef(self, c, side=False):
        key = ("synccoef32", c, bool(side))
        t = self._grad_pool.get(key)
        if t is None:
            t = self._grad_pool[key] = torch.empty((3, c), dtype=torch.float32, device=self.dev)
        return t

    def _sync_finalize(self, u, sums, m, sync):
        """SyncBatchNorm forward (classification/train.py:190-191): this rank's (sum x, sum x^2) -> all-reduce -> the
        ordinary finalisation on one "partial row" with the global count."""
        group, world = sync
        torch.distributed.all_reduce(sums, group=group)
        ops.bn_finalize_stats(sums, 1, m * world, u.conv.cout, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var,
                              u.stats, BN_EPS, BN_MOMENTUM)

    # ---------------------------------------------------------------- forward
    def _conv_bn(self, u, training, side=False, pro=None):
        cv = u.conv
        k, st, pd = u.geom
        m = u.n * u.ho * u.wo
        x2 = u.x.view(m, cv.cout)
        sync = self.net._sync_bn if training else None
        if training and self.dt == torch.bfloat16 and cv.cout % 8 == 0 and _dma_ok(u.src):
            # statistics come out of the convolution's epilogue: no extra pass over x
            partial, scratch = (self.bn_partial_side, self.bn_scratch_side) if side else (self.bn_partial, self.bn_scratch)
            if pro is not None:              # the previous unit's BN + ReLU in this launch's operand path (pro_units)
                u2 = pro[0]
                nt = ops.conv_forward_bnstats_pro(u2.x, u2.stats, u2.y, u2.bits, u.w, u.x, partial, act_csum=pro[1])
                pro[2] = nt
            else:
                nt = ops.conv_forward_bnstats(u.src, u.w, k, k, st, pd, u.x, partial, groups=u.groups, w_frag=u.wf)
            if sync is not None:
                self._sync_finalize(u, ops.bn_partial_sums(partial, nt, cv.cout, self._sync_sums(cv.cout, side)), m, sync)
                return x2
            # (with the prologue's column-sum rows: reduced by this same launch, ops.bn_finalize_stats)
            extra = (pro[1], nt, pro[0].conv.cout, pro[4]) if (pro is not None and pro[1] is not None) else None
            ops.bn_finalize_stats(partial, nt, m, cv.cout, u.bn.weight, u.bn.bias, u.bn.running_mean,
                                  u.bn.running_var, u.stats, BN_EPS, BN_MOMENTUM, scratch=scratch,
                                  tickets=self.bn_tickets_side if side else self.bn_tickets, extra_sums=extra)
            return x2
        # the unfused statistics below use the compute stream's BN workspace: a shortcut convolution on the side stream
        # must have taken the fused-statistics path above (bf16, DMA-addressable source), which needs none
        assert not side, "side-stream convolution fell off the fused-statistics path"
        ops.conv_forward(u.src, u.w, k, k, st, pd, out=u.x, groups=u.groups, out_hw=(u.ho, u.wo), w_frag=u.wf)
        if sync is not None:
            self._sync_finalize(u, ops.bn_stats_sums(x2, self._sync_sums(cv.cout, side), self.bn_ws), m, sync)
            return x2
        if training:
            ops.bn_forward_stats(x2, u.bn.weight, u.bn.bias, u.bn.running_mean, u.bn.running_var, u.stats, self.bn_ws,
                                 BN_EPS, BN_MOMENTUM)
        else:
            _eval_stats(u.bn, u.stats)
        return x2

    def forward(self, img, training):
        net = self.net
        prep_done = None
        if training and self.wg_stream is not None and self.stem_s2d:
            # The stem needs only its own packed matrix: the cast / transposes / fragment packs of every other layer
            # (~90 us of small launches) run on the weight-gradient stream, idle at the start of a step, under the stem.
            self.prepare_weights(training, "stem")
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.wg_stream):
                self.wg_stream.wait_event(ev)
                split = not os.environ.get("IIF_PREP_ONE_PART")
                self.prepare_weights(training, "rest_fwd" if split else "rest")
                net._nbt += 1                  # (a 10 us launch nothing in the step reads: not on the compute stream)
                prep_done = torch.cuda.Event()
                prep_done.record()
                if split:
                    # the transposed copies (~170 us of the ~260) are data-gradient operands: backward waits for them, the
                    # forward pass does not (round 5: the compute stream idled ~50 us behind the stem convolution for them)
                    self.prepare_weights(training, "rest_bwd")
                    self._prep_bwd_done = torch.cuda.Event()
                    self._prep_bwd_done.record()
        else:
            self.prepare_weights(training)
            if training:
                net._nbt += 1
        c1 = net.conv1
        if self.stem_s2d:
            ops.space_to_depth_nchw(img, S2D_CPAD, self.patches)
        else:
            ops.im2col_nchw(img, c1.k, c1.k, c1.stride, c1.pad, c1.ldw, self.dt, out=self.patches)
        u = self.stem
        x2 = self._conv_bn(u, training)
        if prep_done is not None:
            torch.cuda.current_stream().wait_event(prep_done)
        if self.pool_fused:
            # bn1 + relu + maxpool in one pass over the raw stem output: the stem's activation is never stored
            _lib.check(_lib.lib().iif_maxpool_bn_forward(_lib.ptr(u.x), _lib.dtype_code(u.x), _lib.ptr(u.stats), u.n, u.ho, u.wo,
                                                         u.conv.cout, 3, 2, 1, _lib.ptr(self.pool_out), _lib.ptr(self.pool_idx),
                                                         _lib.ptr(self.pool_x if training else None), _lib.stream_ptr()),
                       "iif_maxpool_bn_forward")
        else:
            ops.bn_apply(x2, u.stats, u.y.view(x2.shape), relu=True, relu_bits=u.bits)
            if net.style == "imagenet":
                self._maxpool_fwd(u.y)
        for b in self.blocks:
            units = b["units"]
            ds_done = None
            if "ds" in b and "se" not in b and training and self.fwd_side and self.wg_stream is not None:
                # shortcut convolution + its statistics: independent of the main branch until the block's last bn_apply
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self.wg_stream):
                    self.wg_stream.wait_event(ev)
                    self._conv_bn(b["ds"], training, side=True)
                    ds_done = torch.cuda.Event()
                    ds_done.record()
            last = units[-1]
            nostore = training and last in self.nostore_units and last in self.alg3_units and self.fuse_bwd
            pro = self.pro_units.get(last) if (training and self.fuse_bwd) else None
            if pro is not None and pro[3] != nostore:
                pro = None                                   # (the instance was chosen for the other route: tests switch routes on a live plan)
            if last in self.pro_units and pro is None:
                self.pro_units[last][2] = 0                  # no column-sum rows from this forward
            for uu in units[:-1]:
                x2 = self._conv_bn(uu, training)
                if pro is not None and uu is pro[0]:
                    continue                                 # its normalisation happens in conv3's operand path
                ops.bn_apply(x2, uu.stats, uu.y.view(x2.shape), relu=True, relu_bits=uu.bits)
            if nostore:
                cv = last.conv
                m = last.n * last.ho * last.wo
                if pro is not None:
                    u2 = pro[0]
                    nt = ops.conv_forward_bnstats_pro(u2.x, u2.stats, u2.y, u2.bits, last.w, None, self.bn_partial, act_csum=pro[1])
                    pro[2] = nt
                else:
                    nt = ops.conv_forward_stats_acc(last.src, last.w, self.bn_partial)
                extra = (pro[1], nt, cv.cin, pro[4]) if (pro is not None and pro[1] is not None) else None
                ops.bn_finalize_stats(self.bn_partial, nt, m, cv.cout, last.bn.weight, last.bn.bias, last.bn.running_mean,
                                      last.bn.running_var, last.stats, BN_EPS, BN_MOMENTUM, scratch=self.bn_scratch,
                                      tickets=self.bn_tickets, extra_sums=extra)
                if "ds" in b:
                    du = b["ds"]
                    if ds_done is not None:
                        torch.cuda.current_stream().wait_event(ds_done)
                    else:
                        self._conv_bn(du, training)
                    ops.conv_forward_bn_relu2(last.src, last.w, last.y, last.stats, last.bits, res=du.x, res_stats=du.stats)
                else:
                    ops.conv_forward_bn_relu2(last.src, last.w, last.y, last.stats, last.bits, res=b["inp"])
                continue
            if training and last in self.twopass_units:
                cv = last.conv
                m = last.n * last.ho * last.wo
                nt = ops.conv_forward_stats_only(last.src, last.w, self.bn_partial)
                ops.bn_finalize_stats(self.bn_partial, nt, m, cv.cout, last.bn.weight, last.bn.bias, last.bn.running_mean,
                                      last.bn.running_var, last.stats, BN_EPS, BN_MOMENTUM, scratch=self.bn_scratch,
                                      tickets=self.bn_tickets)
                ops.conv_forward_bn_relu(last.src, last.w, last.y, last.stats, res=b["inp"], relu_bits=last.bits)
                continue
            x2 = self._conv_bn(last, training, pro=pro)
            if "se" in b:
                self._se_forward(b, last, training)
                continue
            if "ds" in b:
                du = b["ds"]
                if ds_done is not None:
                    torch.cuda.current_stream().wait_event(ds_done)
                    xd = du.x.view(x2.shape)
                else:
                    xd = self._conv_bn(du, training)
                ops.bn_apply(x2, last.stats, last.y.view(x2.shape), relu=True, residual=xd, residual_stats=du.stats,
                             relu_bits=last.bits)
            elif "sc" in b:
                ops.shortcut_a_forward(b["inp"], b["blk"].out_planes, out=b["sc"])
                ops.bn_apply(x2, last.stats, last.y.view(x2.shape), relu=True, residual=b["sc"].view(x2.shape),
                             relu_bits=last.bits)
            else:
                ops.bn_apply(x2, last.stats, last.y.view(x2.shape), relu=True, residual=b["inp"].view(x2.shape),
                             relu_bits=last.bits)
        ops.avgpool_forward(self.final, out=self.pooled)
        head = net._head
        feat, bias = self.pooled, None
        if self.head_kind == "linear":
            bias = head._b1d
        elif self.head_kind == "norm":           # F.normalize(x, dim=1)
            feat = ops.rowmap_forward(self.pooled, 1, 1.0, self.head_ex, self.head_xnorm)
        else:                                    # scale * x / (1 + |x|)
            if head.lr_scale:                    # learnable scale, used squared; applied on the device
                ops.rowmap_forward(self.pooled, 0, 1.0, self.head_ex, self.head_xnorm)
                torch.mul(head._s1d[:1], head._s1d[:1], out=self.head_s2)
                _lib.check(_lib.lib().iif_scale_by_device_scalar(_lib.ptr(self.head_ex), _lib.dtype_code(self.head_ex),
                                                                 self.head_ex.numel(), _lib.ptr(self.head_s2),
                                                                 _lib.ptr(self.head_ex), _lib.stream_ptr()), "iif_scale_by_device_scalar")
            else:
                ops.rowmap_forward(self.pooled, 0, float(head.scale), self.head_ex, self.head_xnorm)
            feat = self.head_ex
        ops.conv_forward(feat.view(self.n, 1, 1, head.in_features), self.head_w, 1, 1, 1, 0,
                         out=self.logits.view(self.n, 1, 1, head.out_padded), bias=bias)

    def _se_forward(self, b, last, training):
        """y = relu(bn(x) * e + identity), e = sigmoid(W2 relu(W1 mean_hw(bn(x)))) — SEBottleneck.forward
        (resnet_pytorch.py:358-381) / Se_Block.forward (resnet_cifar.py:163-169).  The squeeze and the
        rescale are native streaming kernels; the [N, C]-sized excitation is one native fp32 launch (csrc/se.hip)."""
        se, P = b["blk"].se, b["se"]
        w1, w2 = se.excitation[0]._w2d, se.excitation[2]._w2d
        ops.se_squeeze(last.x, P["sums"])
        # one launch: q = a * mean_hw(x) + b (the BN affine commutes with the mean), h = relu(W1 q), e = sigmoid(W2 h);
        # W2 is read through its transpose so that both matrices stream along the channel axis
        ops.transpose_f32(w2[:, :se.hidden], P["w2t"])
        ops.se_excite_forward(P["sums"], last.stats, last.ho * last.wo, w1, P["w2t"], P["q"], P["h"], P["e"])
        res, rstats = b["inp"], None
        if "ds" in b:
            du = b["ds"]
            self._conv_bn(du, training)
            res, rstats = du.x, du.stats
        elif "sc" in b:
            ops.shortcut_a_forward(b["inp"], b["blk"].out_planes, out=b["sc"])
            res = b["sc"]
        ops.se_apply(last.x, last.stats, P["e"], last.y, last.bits, residual=res, residual_stats=rstats)

    def _se_backward(self, b, last, g, par):
        """g: gradient w.r.t. the block output.  Masks g in place (it then is the identity-path gradient)
        and returns G = d(loss)/d(bn output) = g*e + d(squeeze)/HW, to be fed to the BN backward."""
        se, P = b["blk"].se, b["se"]
        l1, l2 = se.excitation[0], se.excitation[2]
        hw = last.ho * last.wo
        ops.se_backward_sums(g, last.bits, last.x, P["s1"], P["s2"])
        # d e[n,c] = sum_hw g * (a*x + b) = a*S2 + b*S1;  back through sigmoid / linear / relu / linear, the two weight
        # gradients and the offset (dz1 W1) / HW in three native launches (no vendor GEMM on the path)
        ops.se_excite_backward(P["s1"], P["s2"], last.stats, hw, l1._w2d, P["w2t"], P["e"], P["h"], P["q"], P["dz2"], P["dz1"],
                               P["o"], l1._g2d, l2._g2d)
        m = last.n * hw
        G = self._gbuf(("dx", m, last.conv.cout, par), (m, last.conv.cout)).view(last.n, last.ho, last.wo, last.conv.cout)
        return ops.se_backward_form(g, P["e"], P["o"], G)

    def _maxpool_fwd(self, y):
        n, h, w, c = y.shape
        _lib.check(_lib.lib().iif_maxpool_forward(_lib.ptr(y), _lib.dtype_code(y), n, h, w, c, 3, 2, 1,
                                                  _lib.ptr(self.pool_out), _lib.ptr(self.pool_idx), _lib.stream_ptr()),
                   "iif_maxpool_forward")

    # --------------------------------------------------------------- backward
    def _gbuf(self, key, shape):
        t = self._grad_pool.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=self.dt, device=self.dev)
            self._grad_pool[key] = t
        return t

    # ---- weight gradients on a side stream ---------------------------------------------------------
    # After bn_backward the weight gradient (reads x-operand + dx, writes only the gradient arena) and the
    # data gradient are independent.  wgrad is issued on a second HIP stream so that it overlaps the
    # bandwidth-bound BN-backward / dgrad kernels of the following units.  A pending wgrad of block b+1 still
    # reads that block's dx buffers and its g (= the block-input gradient G[b+2]); so dx buffers rotate over
    # 2 slots, block-input gradients over 3, and before block b starts the main stream waits for every wgrad
    # of the blocks >= b+2 (the fence recorded when block b+1 started).
    def _wgrad_async(self, fn, after=None):
        """``fn(workspace, splits)`` on the next weight-gradient stream (round robin); ``splits`` is the value for
        ops.conv_wgrad: 0 with one stream, -n with n (each launch sized for 1/n of the device, include/iif_amd.h)."""
        if self.wg_stream is None:
            fn(self.wg_ws, 0)
            return
        ev = torch.cuda.Event()
        ev.record()
        k = self._wg_rr
        self._wg_rr = (k + 1) % len(self.wg_streams)
        st = self.wg_streams[k]
        with torch.cuda.stream(st):
            st.wait_event(ev)
            if after is not None:
                st.wait_event(after)
            fn(self.wg_wss[k], -len(self.wg_streams) if len(self.wg_streams) > 1 else 0)

    def _wgrad_fence(self, tag):
        """Record where the side streams are after block ``tag``; wait for the fence of block tag+2."""
        if self.wg_stream is None:
            return
        evs = []
        for st in self.wg_streams:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        self._wg_events[tag] = evs
        old = self._wg_events.pop(tag + self.wg_lag - 1, None)
        if old is not None:
            for ev in old:
                torch.cuda.current_stream().wait_event(ev)

    def _stem_wgrad(self, u, dx4):
        cv = u.conv
        if u.s2d:
            if self.stem_wgrad_main:
                # last kernel of backward: on the compute stream (own split-K workspace) it runs next to the weight
                # gradients the side stream still owes, instead of queueing behind them while the compute stream idles
                ops.conv_wgrad(u.src, dx4, 4, 4, 1, 2, ldw=u.dwp.shape[1], out=u.dwp, workspace=self.wg_ws_stem)
                ops.stem_s2d_unpack_grad(u.dwp, cv.cout, cv.cin, cv.k, S2D_CPAD, cv._g2d)
                return

            def stem(ws, sp):
                ops.conv_wgrad(u.src, dx4, 4, 4, 1, 2, ldw=u.dwp.shape[1], out=u.dwp, workspace=ws, splits=sp)
                ops.stem_s2d_unpack_grad(u.dwp, cv.cout, cv.cin, cv.k, S2D_CPAD, cv._g2d)
            self._wgrad_async(stem)
        else:
            self._wgrad_async(lambda ws, sp: ops.conv_wgrad(u.src, dx4, 1, 1, 1, 0, ldw=cv.ldw, out=cv._g2d, workspace=ws, splits=sp))

    def stem_activation(self):
        """The stem's activated output (tests replay its ReLU decisions): stored, or — when bn1/relu/maxpool run
        fused — recomputed by the same bn_apply arithmetic."""
        u = self.stem
        if self.pool_fused:
            x2 = u.x.view(-1, u.conv.cout)
            ops.bn_apply(x2, u.stats, u.y.view(x2.shape), relu=True)
        return u.y

    def _unit_backward(self, u, gy, mask, gmasked=None, dgrad_out=None, dgrad_res=None, need_dgrad=True, par=0,
                       keep_gy=False, mask_bits=None, dgrad_res_bits=None, fuse_up=None, ws=None, dxkey="dx"):
        """gy: grad w.r.t. the unit's activated output (NHWC).  Computes in place
        dx (into gy's storage unless gmasked is requested), the weight / BN
        gradients, and (optionally) the data gradient w.r.t. the unit's source."""
        cv, bn = u.conv, u.bn
        m = u.n * u.ho * u.wo
        g2 = gy.view(m, cv.cout)
        bits = mask_bits if mask_bits is not None else (None if mask is None else u.bits)
        ws = self.bn_ws if ws is None else ws
        ready = None
        if ws is self.bn_ws:                     # (the shortcut branch on its own stream never consumes fused sums)
            ready, self._bw_ready = self._bw_ready, None
        sync = self.net._sync_bn
        if sync is not None:
            # SyncBatchNorm backward: local (sum g, sum g*xhat) -> dgamma / dbeta; all-reduced sums + global count -> dx
            group, world = sync
            side = ws is not self.bn_ws
            local = self._sync_sums(cv.cout, side, 1)
            if ready is not None and ready[0] is u and gmasked is None:
                ops.bn_partial_sums(self.bw_partial, ready[1], cv.cout, local)
            else:
                ops.bn_backward_sums(g2, None if (mask is None or bits is not None) else mask.view(m, cv.cout), u.x.view(m, cv.cout), u.stats,
                                     local, ws, relu_bits=bits)
            total = self._sync_sums(cv.cout, side, 2)
            total.copy_(local)
            torch.distributed.all_reduce(total, group=group)
            dx = self._gbuf((dxkey, m, cv.cout, par), (m, cv.cout)) if (gmasked is not None or keep_gy) else g2
            coef = self._sync_coef(cv.cout, side)
            ops.bn_backward_apply_sums(g2, None if (mask is None or bits is not None) else mask.view(m, cv.cout), u.x.view(m, cv.cout),
                                       u.stats, bn.weight, local, total, float(m) * world, bn._dgamma, bn._dbeta, dx, coef,
                                       gmasked=None if gmasked is None else gmasked.view(m, cv.cout), relu_bits=bits)
        elif ready is not None and ready[0] is u and len(ready) == 3:
            return self._bn3_algebra(u, gy, ready[1], par, self._cur_block, dgrad_out, fuse_up, ready[2])
        elif u in self.twopass_units or (u in self.nostore_units and u in self.alg3_units):
            raise RuntimeError("two-pass unit reached the standard BN backward: its convolution output was never stored")
        elif ready is not None and ready[0] is u and gmasked is None:
            # the data gradient that wrote gy already reduced (sum g, sum g*xhat) per tile: no reduction pass
            dx = self._gbuf((dxkey, m, cv.cout, par), (m, cv.cout)) if keep_gy else g2
            ops.bn_backward_partials(g2, bits, u.x.view(m, cv.cout), u.stats, bn.weight, self.bw_partial, ready[1],
                                     bn._dgamma, bn._dbeta, dx, ws, tickets=self.bn_tickets)
        elif gmasked is not None or keep_gy:
            # dx goes to its own buffer; gy is either overwritten by its masked copy (gmasked) or left as is
            dx = self._gbuf((dxkey, m, cv.cout, par), (m, cv.cout))
            ops.bn_backward(g2, None if mask is None else mask.view(m, cv.cout), u.x.view(m, cv.cout), u.stats, bn.weight,
                            bn._dgamma, bn._dbeta, dx, ws,
                            gmasked=None if gmasked is None else gmasked.view(m, cv.cout), relu_bits=bits)
        else:
            dx = g2
            ops.bn_backward(g2, None if mask is None else mask.view(m, cv.cout), u.x.view(m, cv.cout), u.stats, bn.weight,
                            bn._dgamma, bn._dbeta, dx, ws, relu_bits=bits)
        dx4 = dx.view(u.n, u.ho, u.wo, cv.cout)
        if u.is_patch_gemm:
            self._stem_wgrad(u, dx4)
            return None
        if u.groups > 1:
            def grouped(ws, sp):
                ops.conv_wgrad(u.src, dx4, cv.k, cv.k, cv.stride, cv.pad, ldw=u.dwp.shape[1], out=u.dwp,
                               workspace=ws, groups=u.groups, splits=sp)
                ops.group_unpack_grad(u.dwp, cv.cout, cv.cg, cv.chunk, cv.k * cv.k, cv._g2d)
            self._wgrad_async(grouped)
        else:
            self._wgrad_async(lambda ws, sp: ops.conv_wgrad(u.src, dx4, cv.k, cv.k, cv.stride, cv.pad, ldw=cv.ldw, out=cv._g2d,
                                                            workspace=ws, splits=sp))
        if not need_dgrad:
            return None
        if (fuse_up is not None and self.fuse_bwd and (u.groups == 1 or cv.stride == 1) and dgrad_out is not None
                and _dma_ok(dx4)):
            up, up_bits = fuse_up
            if up in self.alg3_units and cv.k == 1 and cv.stride == 1 and up_bits is not None:
                # the upstream unit's BN backward runs by algebra: store the gradient gated by its block's ReLU, emit its
                # column sums only (conv3's output is not read)
                rows = self._alg_rows.get(id(up), self.bw_partial)
                if up in self.rx_units:
                    pg = self.pg_units.get(up)
                    if pg is not None:
                        nt, pg[1] = ops.conv_dgrad_masksum_rx_pg(dx4, u.wt, (u.hi, u.wi), dgrad_out, up_bits, rows, up.src, up.w, up.stats,
                                                                 pg[0], up.conv.ldw, res=dgrad_res, res_bits=dgrad_res_bits)
                    else:
                        nt = ops.conv_dgrad_masksum_rx(dx4, u.wt, (u.hi, u.wi), dgrad_out, up_bits, rows, up.src, up.w, up.stats,
                                                       res=dgrad_res, res_bits=dgrad_res_bits)
                    self._bw_ready = (up, nt, rows)
                    return dgrad_out
                nt = ops.conv_dgrad_masksum(dx4, u.wt, (u.hi, u.wi), dgrad_out, up_bits, rows, res=dgrad_res,
                                            res_bits=dgrad_res_bits, up_x=None if self._a3_is_pure(up) else up.x,
                                            up_stats=None if self._a3_is_pure(up) else up.stats)
                self._bw_ready = (up, nt, rows)
                return dgrad_out
            nt = ops.conv_dgrad_bnbwd(dx4, u.wt, cv.k, cv.k, cv.stride, cv.pad, (u.hi, u.wi), dgrad_out, up.x, up_bits, up.stats,
                                      self.bw_partial, res=dgrad_res, res_bits=dgrad_res_bits, w_frag=u.wtf, groups=u.groups)
            self._bw_ready = (up, nt)
            return dgrad_out
        return ops.conv_dgrad(dx4, u.wt, cv.k, cv.k, cv.stride, cv.pad, (u.hi, u.wi), out=dgrad_out, res=dgrad_res,
                              groups=u.groups, res_bits=dgrad_res_bits, w_frag=u.wtf)

    def _a3_is_pure(self, u):
        """"Sums from P" (P = g~^T a2 on the compute stream before the coefficients); never where the producer recomputes x."""
        return u not in getattr(self, "rx_units", ()) and u.n * u.ho * u.wo * u.conv.cout >= self.a3_pure_min

    def _a3_gram_stacked(self, u):
        """The Gram matrix a2^T a2 rides in the launch that forms P on the weight-gradient stream ([g~ | a2]^T a2,
        iif_wgrad1x1_stacked: the extra channel tile re-reads rows of a2 the launch streams anyway) instead of a launch and a
        slab reduction of its own a block ahead.  Not where P is formed on the compute stream ("pure" units): there the extra
        tile would lengthen the critical path."""
        return (not self._a3_is_pure(u)) and u.conv.cout % 128 == 0 and self.wg_stream is not None and not os.environ.get("IIF_NO_GRAM_STACKED")

    def _bn3_gram_async(self, u, bi):
        """Gram = a2^T a2 and colsum(a2) of an algebra unit's input: forward data only, so it is issued a block ahead on the
        shortcut stream (idle outside the four downsample blocks) and never waited for in practice."""
        A = self.a3g[bi % (self.wg_lag + 1)]
        cv = u.conv
        a2 = u.src
        A["csum_fwd"] = None
        pro = self.pro_units.get(u) if self.fuse_bwd else None
        if pro is not None and pro[1] is not None and pro[2] > 0:
            # the forward pass already left the column sums of a2 (bn2's prologue in conv3's launch, reduced by that unit's
            # finalisation): nothing to compute, and with the Gram matrix stacked behind P nothing to launch or wait for at all
            A["csum_fwd"] = pro[4]
            if self._a3_gram_stacked(u):
                A["ev"] = A["ev_csum"] = None
                return

        def csum():
            if A["csum_fwd"] is None:
                ops.bn_stats_sums(a2.view(-1, cv.cin), A["csum"].view(-1)[:2 * cv.cin].view(2, cv.cin), A["ws_sum"])

        def gram():
            if not self._a3_gram_stacked(u):
                ops.conv_wgrad(a2, a2, 1, 1, 1, 0, ldw=cv.ldw, out=A["gram"].view(-1)[:cv.cin * cv.ldw].view(cv.cin, cv.ldw),
                               workspace=A["ws_gram"])
        st = self.ds_stream if self.ds_stream is not None else self.wg_stream
        if st is None:
            csum(); gram()
            A["ev"] = A["ev_csum"] = None
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(st):
            st.wait_event(ev)
            # the column sums first, with an event of their own: they are all the compute stream's bn3_algebra_prep needs; the Gram
            # matrix (and its slab reduction) only feeds the weight gradient on the weight-gradient stream (round-5 advice)
            csum()
            A["ev_csum"] = torch.cuda.Event()
            A["ev_csum"].record()
            gram()
            done = torch.cuda.Event()
            done.record()
        A["ev"] = done

    def _bn3_algebra(self, u, gt, nt, par, bi, dgrad_out, fuse_up, rows):
        """Backward of conv3 + bn3 from the gated block-output gradient ``gt`` without reading conv3's output
        (csrc/bn3_algebra.hip).  Returns the gradient w.r.t. the unit's source (a2)."""
        cv, bn = u.conv, u.bn
        A, Ag = self.a3[par % self.wg_lag], self.a3g[bi % (self.wg_lag + 1)]
        C, c = cv.cout, cv.cin
        m = u.n * u.ho * u.wo
        g4 = gt.view(u.n, u.ho, u.wo, C)
        wb = u.w                                                   # the bf16 weights the forward multiplied with, [C, ldw]
        P = A["P"].view(-1)[:C * cv.ldw].view(C, cv.ldw)          # contiguous [C, ldw]: conv_wgrad writes with pitch ldw
        pure = self._a3_is_pure(u)
        if pure:
            ops.conv_wgrad(u.src, g4, 1, 1, 1, 0, ldw=cv.ldw, out=P, workspace=self.a3_ws)
        wt = A["wt"][:c * (C + c)].view(c, C + c)
        coef = A["coef"].view(-1)[:3 * C].view(3, C)
        # colsum(a2) (issued a block ahead with the Gram matrix) keeps the data gradient's column sums at zero through the bf16
        # rounding of the stacked weights (bn3_gm_finish_kernel); the weight gradient needs Gram / colsum too, off the critical path
        csum_a2 = (Ag["csum_fwd"] if Ag.get("csum_fwd") is not None else Ag["csum"]).view(-1)[:c]
        gram_ev = Ag["ev"]
        if Ag.get("ev_csum") is not None:
            torch.cuda.current_stream().wait_event(Ag["ev_csum"])
        ops.bn3_algebra_prep(P if pure else None, wb, c, rows, nt, u.stats, bn.weight, m, coef, bn._dgamma, bn._dbeta, wt,
                             A["bias"][:c], A["scr"], A["tickets"], colsum2=csum_a2)

        stacked = self._a3_gram_stacked(u)

        def finish_dw(ws, sp):
            gram = Ag["gram"].view(-1)[:c * cv.ldw].view(c, cv.ldw)
            pg = self.pg_units.get(u)
            if stacked and pg is not None and pg[1] > 0:
                # the producer of g~ left P and Gram behind, one slab per tile sequence: add them up (no pass over g~ and a2)
                ext = A["P"].view(-1)[:(C + c) * cv.ldw].view(C + c, cv.ldw)
                ops.slab_sum(pg[0], pg[1], C + c, cv.ldw, c, ext)
                pg[1] = 0
                gram = ext[C:]
            elif stacked:
                ext = A["P"].view(-1)[:(C + c) * cv.ldw].view(C + c, cv.ldw)
                ops.wgrad1x1_stacked(u.src.view(m, c), gt.view(m, C), u.src.view(m, c), ext, ws, splits=sp)
                gram = ext[C:]
            elif not pure:
                ops.conv_wgrad(u.src, g4, 1, 1, 1, 0, ldw=cv.ldw, out=P, workspace=ws, splits=sp)
            ops.bn3_algebra_dw(P, wb, c, gram, csum_a2, coef, cv._g2d)
        if self.wg_stream is None:
            finish_dw(self.a3_ws, 0)
        else:
            self._wgrad_async(finish_dw, after=gram_ev)
        up, up_bits = fuse_up
        nt2 = ops.conv_dgrad2_bnbwd(g4, u.src, wt, A["bias"][:c], dgrad_out, up.x, up_bits, up.stats, self.bw_partial)
        self._bw_ready = (up, nt2)
        return dgrad_out

    def _ds_algebra(self, du, D, gt, x_in, nt, gin_ds):
        """Backward of a stride-1 convolutional shortcut (1x1 conv + BN, no ReLU of its own) from the gated block-output gradient
        ``gt`` without reading the shortcut's output (see ``ds_alg`` in __init__).  Runs on the shortcut stream; ``gin_ds`` receives
        the gradient w.r.t. the block input through the shortcut, the first unit's data gradient adds it as its residual."""
        cv, bn = du.conv, du.bn
        C, c = cv.cout, cv.cin
        m = du.n * du.ho * du.wo
        g4 = gt.view(du.n, du.ho, du.wo, C)
        P = D["P"].view(-1)[:C * cv.ldw].view(C, cv.ldw)
        ops.bn_stats_sums(x_in.view(-1, c), D["csum"], D["ws_sum"])
        ops.conv_wgrad(x_in, g4, 1, 1, 1, 0, ldw=cv.ldw, out=P, workspace=D["ws"])
        coef = D["coef"]
        ops.bn3_algebra_prep(P, du.w, c, D["rows"], nt, du.stats, bn.weight, m, coef, bn._dgamma, bn._dbeta, D["wt"], D["bias"],
                             D["scr"], D["tickets"], colsum2=D["csum"].view(-1)[:c])
        ops.conv_dgrad2_bnbwd(g4, x_in, D["wt"], D["bias"], gin_ds)
        done = torch.cuda.Event()
        done.record()                       # what the compute stream waits for: the data gradient
        # the weight gradient: dWd = A P + B Wd Gram(x_in) + D (x) colsum(x_in); nothing waits for it but the gradient
        # reduction (the reducer waits for this stream too) and the end of backward
        gram = D["gram"].view(-1)[:c * cv.ldw].view(c, cv.ldw)
        ops.conv_wgrad(x_in, x_in, 1, 1, 1, 0, ldw=cv.ldw, out=gram, workspace=D["ws"])
        ops.bn3_algebra_dw(P, du.w, c, gram, D["csum"].view(-1)[:c], coef, cv._g2d)
        return done

    def backward(self, reducer=None):
        offs = self.net.block_offsets() if reducer is not None else None
        self._bw_ready = None
        budget = 0
        if reducer is not None:
            reducer.begin()
            reducer.extra_streams = self.wg_streams + ([self.ds_stream] if self.ds_alg and self.ds_stream is not None else [])
            if reducer.world > 1 or reducer.force:
                budget = int(getattr(reducer, "cu_budget", 0))
        if budget:
            # grids are sized when a launch is enqueued: every persistent kernel of this backward leaves the reduction its CUs
            _lib.check(_lib.lib().iif_set_cu_budget(budget), "iif_set_cu_budget")
        try:
            self._backward(reducer, offs)
        finally:
            if budget:
                _lib.lib().iif_set_cu_budget(0)

    def _backward(self, reducer, offs):
        net = self.net
        head = net._head
        n = self.n
        pb = getattr(self, "_prep_bwd_done", None)
        if pb is not None:
            torch.cuda.current_stream().wait_event(pb)        # transposed weight copies of this step (forward())
            self._prep_bwd_done = None
        if self.wg_stream is not None:
            # fork the side streams off the compute stream before anything is recorded on them: under hipGraph capture
            # every event of the step then belongs to the capture (the first fence used to be recorded on a stream that
            # had not joined it yet), and all of them are joined again at the end of backward
            ev = torch.cuda.Event()
            ev.record()
            for st in self.wg_streams:
                st.wait_event(ev)
            if self.ds_stream is not None:
                self.ds_stream.wait_event(ev)
        # ---- head: dlogits (fp32, pad columns are zero) -> head grads -> pooled grad -> final activation grad
        op, D = head.out_padded, head.in_features
        if self.dt == torch.float32:
            dl = self.dlogits
        else:
            dl = ops.cast(self.dlogits, self.dlogits_t)
        dpooled = self._gbuf(("dpooled",), (n, 1, 1, D))
        if self.head_kind == "linear":
            def head_grads(ws, sp):          # only the data gradient is on the chain
                ops.colsum_f32(self.dlogits, n, op, op, head._gb1d)
                ops.conv_wgrad(self.pooled.view(n, 1, 1, D), dl.view(n, 1, 1, op), 1, 1, 1, 0, ldw=D, out=head._g2d,
                               workspace=ws, splits=sp)
            self._wgrad_async(head_grads)
            ops.conv_dgrad(dl.view(n, 1, 1, op), self.head_wt, 1, 1, 1, 0, (1, 1), out=dpooled)
        else:
            # d(normalised weights) = dlogits^T @ ex, then back through the row normalisation
            ops.conv_wgrad(self.head_ex.view(n, 1, 1, D), dl.view(n, 1, 1, op), 1, 1, 1, 0, ldw=D, out=self.head_dwn,
                           workspace=self.wg_ws)
            if self.head_kind == "cosine":
                ops.rowmap_backward(head._w2d, self.head_wnorm, self.head_dwn, 1, 1.0, head._g2d, eps=0.0)
            else:
                ops.rowmap_backward(self.head_wT, self.head_wnorm, self.head_dwn, 1, 1.0, self.head_dwT)
                ops.transpose_f32(self.head_dwT, head._g2d)
            ops.conv_dgrad(dl.view(n, 1, 1, op), self.head_wt, 1, 1, 1, 0, (1, 1), out=self.head_dex.view(n, 1, 1, D))
            dp2 = dpooled.view(n, D)
            if self.head_kind == "norm":
                ops.rowmap_backward(self.pooled, self.head_xnorm, self.head_dex, 1, 1.0, dp2)
            elif head.lr_scale:
                # logits = s^2 * L0  ->  dL/ds = (2/s) * <dlogits, logits>;  d(ex0) = s^2 * d(ex)
                ops.dot_window_f32(self.dlogits, self.logits, n, head.out_features, 2.0, head._gs1d, alpha_div=head._s1d)
                _lib.check(_lib.lib().iif_scale_by_device_scalar(_lib.ptr(self.head_dex), _lib.dtype_code(self.head_dex),
                                                                 self.head_dex.numel(), _lib.ptr(self.head_s2),
                                                                 _lib.ptr(self.head_dex), _lib.stream_ptr()), "iif_scale_by_device_scalar")
                ops.rowmap_backward(self.pooled, self.head_xnorm, self.head_dex, 0, 1.0, dp2)
            else:
                ops.rowmap_backward(self.pooled, self.head_xnorm, self.head_dex, 0, float(head.scale), dp2)
        fh, fw, fc = self.final.shape[1], self.final.shape[2], self.final.shape[3]
        if net._head_only:              # frozen backbone: the classifier's gradients are all that is needed
            if self.wg_stream is not None:
                for st in self.wg_streams:
                    torch.cuda.current_stream().wait_stream(st)
                if self.ds_stream is not None:
                    torch.cuda.current_stream().wait_stream(self.ds_stream)
            if reducer is not None:
                reducer.finish_tail(offs["head"])
            return
        g = self._gbuf(("g", self.final.shape), self.final.shape)
        ops.avgpool_backward(dpooled.view(n, D), fh * fw, out=g.view(n, fh * fw, fc))
        if reducer is not None:
            reducer.gradients_ready_from(offs["head"])
        # ---- blocks in reverse
        for bi in range(len(self.blocks) - 1, -1, -1):
            b = self.blocks[bi]
            units = b["units"]
            last = units[-1]
            inp = b["inp"]
            # g is the gradient w.r.t. the block output (pre-mask).  After this call g holds the masked
            # gradient (the residual-branch gradient) and dx of the last conv has been consumed.
            par = bi % self.wg_lag
            self._wgrad_fence(bi)
            self._cur_block = bi
            gin = self._gbuf(("gin", tuple(inp.shape), bi % (self.wg_lag + 1)), inp.shape)
            dkey = ("d", tuple(last.src.shape), len(units) - 1, par)
            # The block's ReLU gates g on both paths.  With a convolutional shortcut or a plain identity no masked
            # copy of g is written: the shortcut's BN backward and the identity add read g through the ReLU bits.
            lazy_mask = "se" not in b and "sc" not in b
            # the dgrad that writes a unit's output gradient also reduces that unit's BN-backward sums: inside the
            # block each unit feeds the previous one; the dgrad that completes the block-input gradient feeds the
            # previous block's last unit (gated by that block's ReLU bits)
            prev = self.blocks[bi - 1] if bi > 0 else None
            up_in = None
            if prev is not None and "se" not in prev and "sc" not in prev:
                up_in = (prev["units"][-1], prev["units"][-1].bits)
            inner = lambda ui: (units[ui - 1], units[ui - 1].bits)       # noqa: E731
            ds_done = gin_ds = None
            if "ds" in b and lazy_mask and self.ds_stream is not None:
                du = b["ds"]
                gin_ds = self._gbuf(("ginds", tuple(inp.shape), par), inp.shape)
                rdy = self._bw_ready
                D = self.ds_alg.get(id(du))
                by_algebra = D is not None and rdy is not None and rdy[0] is last and len(rdy) == 3
                by_algebra = by_algebra and rdy[2] is D["rows"]        # (the rows live where nothing on the compute stream overwrites them)
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self.ds_stream):
                    self.ds_stream.wait_event(ev)
                    if by_algebra:
                        ds_done = self._ds_algebra(du, D, g, inp, rdy[1], gin_ds)
                    else:
                        self._unit_backward(du, g, last.y, mask_bits=last.bits, keep_gy=True, dgrad_out=gin_ds, par=par,
                                            ws=self.bn_ws_ds, dxkey="dxds")
                        ds_done = torch.cuda.Event()
                        ds_done.record()
            # Gram / column sums of the PREVIOUS block's a2 (forward data, needed by that block's weight gradient only): queued on the
            # shortcut stream BEHIND this block's shortcut branch, which the compute stream waits for at the end of the block (round 5:
            # in front of it, it delayed the branch by its ~100-150 us: the compute stream idled 120-230 us at every downsample block,
            # profiles/r5_b_step_timeline.txt)
            if bi > 0 and self.blocks[bi - 1]["units"][-1] in self.alg3_units:
                self._bn3_gram_async(self.blocks[bi - 1]["units"][-1], bi - 1)
            if "se" in b:
                G = self._se_backward(b, last, g, par)
                d = self._unit_backward(last, G, None, par=par, dgrad_out=self._gbuf(dkey, last.src.shape),
                                        fuse_up=inner(len(units) - 1))
            elif lazy_mask:
                d = self._unit_backward(last, g, last.y, keep_gy=True, par=par, dgrad_out=self._gbuf(dkey, last.src.shape),
                                        fuse_up=inner(len(units) - 1))
            else:
                d = self._unit_backward(last, g, last.y, gmasked=g, par=par, dgrad_out=self._gbuf(dkey, last.src.shape),
                                        fuse_up=inner(len(units) - 1))
            for ui in range(len(units) - 2, 0, -1):
                uu = units[ui]
                d = self._unit_backward(uu, d, uu.y, par=par, fuse_up=inner(ui),
                                        dgrad_out=self._gbuf(("d", tuple(uu.src.shape), ui, par), uu.src.shape))
            first = units[0]
            if "ds" in b and ds_done is not None:
                # the shortcut's gradient was computed next to the main branch: the first unit's dgrad adds it
                torch.cuda.current_stream().wait_event(ds_done)
                self._unit_backward(first, d, first.y, dgrad_out=gin, dgrad_res=gin_ds, par=par, fuse_up=up_in)
            elif "ds" in b and lazy_mask:
                # no shortcut stream: the same order on the compute stream - the shortcut's backward first (its dx in a buffer
                # of its own: g is still read by the gated identity add and, on the algebraic route, by the P GEMM of the
                # weight-gradient stream), then the first unit's data gradient adds it and carries the upstream sums, so the
                # previous block's last unit takes the same route as with the stream
                du = b["ds"]
                gin_ds = self._gbuf(("ginds", tuple(inp.shape), par), inp.shape)
                pending, self._bw_ready = self._bw_ready, None          # the fused sums waiting for `first` are not the shortcut's
                self._unit_backward(du, g, last.y, mask_bits=last.bits, keep_gy=True, dgrad_out=gin_ds, par=par, dxkey="dxds")
                self._bw_ready = pending
                self._unit_backward(first, d, first.y, dgrad_out=gin, dgrad_res=gin_ds, par=par, fuse_up=up_in)
            elif "ds" in b:
                self._unit_backward(first, d, first.y, dgrad_out=gin, par=par)
                self._unit_backward(b["ds"], g, None, dgrad_out=gin, dgrad_res=gin, par=par, fuse_up=up_in)
            elif "sc" in b:
                self._unit_backward(first, d, first.y, dgrad_out=gin, par=par)
                ops.shortcut_a_backward_acc(g, gin)
            elif lazy_mask:
                self._unit_backward(first, d, first.y, dgrad_out=gin, dgrad_res=g, dgrad_res_bits=last.bits, par=par,
                                    fuse_up=up_in)
            else:
                self._unit_backward(first, d, first.y, dgrad_out=gin, dgrad_res=g, par=par, fuse_up=up_in)
            g = gin
            if reducer is not None:
                reducer.gradients_ready_from(offs["blocks"][bi])
        # ---- stem
        u = self.stem
        g_pooled = g
        # bf16 with the fused forward pool: the max-pool backward is gathered inside the BN-backward normalisation pass; the
        # scattered gradient at the stem's resolution (411 MB at batch 256, written and read back) is never formed
        pool_bwd_fused = (net.style == "imagenet" and self.pool_fused and self.dt != torch.float32
                          and not os.environ.get("IIF_NO_POOL_BWD_FUSED"))
        if net.style == "imagenet" and not pool_bwd_fused:
            dy0 = self._gbuf(("dy0",), u.y.shape)
            _lib.check(_lib.lib().iif_maxpool_backward(_lib.ptr(g), _lib.ptr(self.pool_idx), _lib.dtype_code(g), u.n, u.ho,
                                                       u.wo, u.conv.cout, 3, 2, 1, _lib.ptr(dy0), _lib.stream_ptr()),
                       "iif_maxpool_backward")
            g = dy0
        if pool_bwd_fused:
            cv, bn = u.conv, u.bn
            dy0 = self._gbuf(("dy0",), u.y.shape)
            ph, pw = g_pooled.shape[1], g_pooled.shape[2]
            _lib.check(_lib.lib().iif_bn_backward_pool_fused(
                _lib.ptr(g_pooled), _lib.ptr(self.pool_idx), _lib.ptr(self.pool_x), _lib.ptr(u.x), _lib.dtype_code(u.x), u.n, u.ho, u.wo,
                cv.cout, ph, pw, _lib.ptr(u.stats), _lib.ptr(bn.weight), _lib.ptr(bn._dgamma), _lib.ptr(bn._dbeta), _lib.ptr(dy0),
                _lib.ptr(self.bn_ws), self.bn_ws.numel(), _lib.stream_ptr()), "iif_bn_backward_pool_fused")
            self._stem_wgrad(u, dy0)
        elif self.pool_fused:
            cv, bn = u.conv, u.bn
            m = u.n * u.ho * u.wo
            g2 = g.view(m, cv.cout)
            # bf16: the two column sums come from the pooled gradient and the raw stem output at the arg max (pool_x, stored by the
            # fused forward pool): a quarter of the elements, no pass over the scattered gradient and the stem output (190 -> ~45 us
            # at batch 256), term for term the standard pass's sums; the normalisation pass is unchanged.
            # fp32 (parity mode) keeps the reduction pass (the parity tests hold the stem's weight gradient to 2e-4 against a
            # reference that sums in that order).
            if self.dt == torch.float32:
                _lib.check(_lib.lib().iif_bn_backward_relu_recompute(_lib.ptr(g2), _lib.ptr(u.x), _lib.dtype_code(u.x), m, cv.cout,
                                                                     _lib.ptr(u.stats), _lib.ptr(bn.weight), _lib.ptr(bn._dgamma),
                                                                     _lib.ptr(bn._dbeta), _lib.ptr(g2), _lib.ptr(self.bn_ws),
                                                                     self.bn_ws.numel(), _lib.stream_ptr()),
                           "iif_bn_backward_relu_recompute")
            else:
                _lib.check(_lib.lib().iif_bn_backward_relu_recompute_pooled(
                    _lib.ptr(g2), _lib.ptr(u.x), _lib.dtype_code(u.x), m, cv.cout, _lib.ptr(u.stats), _lib.ptr(bn.weight),
                    _lib.ptr(bn._dgamma), _lib.ptr(bn._dbeta), _lib.ptr(g2), _lib.ptr(self.bn_ws), self.bn_ws.numel(), _lib.ptr(g_pooled),
                    _lib.ptr(self.pool_x), self.pool_x.numel() // cv.cout, _lib.stream_ptr()),
                    "iif_bn_backward_relu_recompute_pooled")
            self._stem_wgrad(u, g2.view(u.n, u.ho, u.wo, cv.cout))
        else:
            self._unit_backward(u, g, u.y, need_dgrad=False)
        if self.wg_stream is not None:
            for st in self.wg_streams:
                torch.cuda.current_stream().wait_stream(st)
            if self.ds_stream is not None:
                torch.cuda.current_stream().wait_stream(self.ds_stream)
            self._wg_events.clear()
        if reducer is not None:
            reducer.finish()


_DMA_LIMIT = int(os.environ.get("IIF_DMA_LIMIT", str(0x7f000000)))      # lowered in tests to walk the unfused path


def _dma_ok(t):
    """The pipelined kernels address their operands with 32-bit LDS-DMA offsets (< 2 GiB); beyond that the
    register-staged fallback runs and the fused epilogue options are not available."""
    return t.numel() * t.element_size() < _DMA_LIMIT


def _eval_stats(bn, stats):
    """Inference-mode BN: the affine from running statistics (host-free vector math on [C] tensors)."""
    invstd = torch.rsqrt(bn.running_var + BN_EPS)
    a = bn.weight.detach() * invstd
    stats[0].copy_(bn.running_mean); stats[1].copy_(invstd); stats[2].copy_(a)
    stats[3].copy_(bn.bias.detach() - bn.running_mean * a)
