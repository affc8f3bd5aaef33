"""End-to-end parity of the native ResNet training step against the CPU oracle
(which is pinned to the reference by tests/golden/g7_nets.npz).

fp32 parity mode: logits / loss / gradients / loss curve within 1e-4 relative
(BASELINE.json north_star tolerance).  bf16 performance mode: same checks with
the tolerance of bf16 storage (stated per assertion)."""
import pytest
import torch

from oracle import iif_oracle as O
from oracle import resnet_oracle as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class DS:
    def __init__(self, c):
        self.c = list(c)

    def get_cls_num_list(self):
        return self.c


def _build(arch, num_classes, dt):
    from iif_amd import resnet_cifar, resnet_pytorch
    if arch in R.CIFAR_ARCHS:
        sd = R.init_cifar(arch, num_classes, seed=3)
        net = getattr(resnet_cifar, arch)(num_classes=num_classes, use_norm="None", compute_dtype=dt)
    else:
        sd = R.init_imagenet(arch, num_classes, seed=3)
        net = getattr(resnet_pytorch, arch)(num_classes=num_classes, use_norm="None", pretrained="None", compute_dtype=dt)
    net.load_state_dict(sd)
    return net, sd


def _data(B, hw, counts, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, hw, hw, generator=g)
    prior = torch.tensor(counts, dtype=torch.float64)
    y = torch.multinomial(prior / prior.sum(), B, replacement=True, generator=g)
    return x, y


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def gpu_relu_masks(net, rows=None):
    """The ReLU decisions the GPU step actually took (stored activated outputs > 0),
    in forward order, as NCHW bool tensors for oracle.ReluMasks (``rows``: the first images only)."""
    plan = net._saved
    ys = [plan.stem_activation()] + [u.y for b in plan.blocks for u in b["units"]]
    return [(y[:rows] > 0).permute(0, 3, 1, 2).cpu() for y in ys]


def damp_residual_branches(sd, arch, factor=0.25):
    """Scale the last BN gain of every residual block.  A freshly initialised deep
    ResNet evaluated on a handful of images amplifies fp32 rounding ~300x (measured:
    the reference's own fp32 CPU gradients are 1-2e-2 away from an fp64 evaluation);
    trained networks, and this damped init, are well conditioned, so 1e-4 is testable."""
    last = "bn3.weight" if R.IMAGENET_ARCHS.get(arch, ("basic",))[0] == "bottleneck" else "bn2.weight"
    for k in sd:
        if k.endswith(last) and k.startswith("layer"):
            sd[k] = sd[k] * factor
    return sd


CASES = [("resnet32", 100, 8, 32), ("resnet20", 10, 5, 32), ("resnet50", 1000, 8, 64), ("resnet18", 365, 4, 96),
         ("resnext50_32x4d", 365, 8, 64), ("se_resnet32", 100, 8, 32), ("se_resnet50", 1000, 8, 64)]


@pytest.fixture(params=["side_streams", "one_stream"])
def stream_mode(request, monkeypatch):
    """The engine leaves its side streams off for small steps (resnet_engine.py, "Side streams": CIFAR ResNet32 bs 128 - BASELINE
    config 1 - and every batch <= 16-32 run on ONE stream, with the shortcut backward first and the pending fused sums set
    aside); tests/conftest.py forces them on.  The parity and loss-curve tests that take this fixture run in both
    configurations (round-4 advice)."""
    monkeypatch.setenv("IIF_SIDE_STREAMS", "1" if request.param == "side_streams" else "0")
    return request.param


@pytest.mark.parametrize("arch,C,B,hw", CASES)
def test_fp32_forward_backward_parity(arch, C, B, hw):
    """Drop-in surface (model(x) -> criterion -> loss.backward()) in exact-fp32 mode
    against the CPU oracle: logits, loss and EVERY parameter gradient within 1e-4
    (2e-4 L2 per tensor), given the same ReLU decisions.  Two fp32 implementations
    can round a pre-activation lying within ~1e-7 of zero to different signs; the
    oracle therefore replays the decisions the GPU took, and the test asserts that
    they differ from the oracle's own only on such near-zero elements."""
    from iif_amd.custom import IIFLoss
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.float32)
    if arch not in R.CIFAR_ARCHS:
        net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts)
    table = O.iif_tables(counts)["raw"]
    net.train()
    crit = IIFLoss(DS(counts), variant="raw")
    logits = net(x.to(DEV))
    loss = crit(logits, y.to(DEV))
    loss.backward()
    masks = R.ReluMasks(gpu_relu_masks(net))
    ref_sd = {k: v.clone() for k, v in sd.items()}
    ref_loss, ref_logits, ref_grads = R.loss_and_grads(ref_sd, x, y, table, arch, relu_masks=masks)
    assert masks.i == len(masks.masks)
    assert masks.disagree <= 1e-4 * masks.total and masks.worst <= 1e-4, (masks.disagree, masks.total, masks.worst)
    assert relerr(logits, ref_logits) <= 1e-4
    assert relerr(loss, ref_loss) <= 1e-4
    for k, p in net.named_parameters():
        a, b = p.grad.double().cpu(), ref_grads[k].double()
        e = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        assert e <= 2e-4, (k, e)
    # running statistics / counters follow the reference
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert relerr(v.float(), ref_sd[k].float()) <= 1e-5, k
    # eval mode uses the running statistics
    net.eval()
    with torch.no_grad():
        ev = net(x.to(DEV))
    ref_ev = R.forward(ref_sd, x, arch, training=False)
    assert relerr(ev, ref_ev) <= 1e-4


@pytest.mark.parametrize("arch,C,B,hw", [("resnet32", 100, 8, 32), ("resnet50", 1000, 8, 64), ("resnext50_32x4d", 365, 8, 64),
                                         ("se_resnet32", 100, 8, 32), ("se_resnext50_32x4d", 365, 8, 64)])
def test_fp32_loss_curve_fused_step(arch, C, B, hw, stream_mode):
    """forward -> fused IIF loss -> backward -> ONE fused SGD launch, 4 steps with
    the first-epoch warm-up (train.py:52-56): the loss sequence matches the CPU
    reference to 1e-4 and the final weights to 2e-4 (same ReLU decisions, see above)."""
    from iif_amd.custom import IIFLoss
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.float32)
    if arch not in R.CIFAR_ARCHS:
        net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=9)
    table = O.iif_tables(counts)["raw"]
    crit = IIFLoss(DS(counts), variant="raw")
    ref_sd = {k: v.clone() for k, v in sd.items()}
    bufs = {}
    xd, yd = x.to(DEV), y.to(DEV)
    net.train()
    for it in range(4):
        lr = 0.1 * O.warmup_factor(it, 1000)
        loss, _ = net.loss_and_backward(xd, yd, crit)
        masks = R.ReluMasks(gpu_relu_masks(net))
        net.sgd_step(lr, 0.9, 1e-4)
        ref_loss, _ = R.train_step(ref_sd, bufs, x, y, table, arch, lr, relu_masks=masks)
        assert masks.disagree <= 1e-4 * masks.total and masks.worst <= 1e-4
        assert relerr(loss, ref_loss) <= 1e-4, (it, loss.item(), ref_loss.item())
    for k, v in net.state_dict().items():
        if v.is_floating_point():
            assert relerr(v, ref_sd[k]) <= 2e-4, (k, relerr(v, ref_sd[k]))


@pytest.mark.parametrize("arch,C,B,hw", [("resnet50", 1000, 8, 64), ("resnet32", 100, 16, 32), ("resnext50_32x4d", 365, 8, 64),
                                         ("se_resnet50", 1000, 8, 64)])
def test_bf16_mode_against_bf16_storage_oracle(arch, C, B, hw, stream_mode):
    """Performance mode (bf16 storage, fp32 accumulate).  The oracle run with
    ``q=bf16_storage`` rounds the same tensors at the same places, so what is left
    is accumulation order and 1-ulp bf16 rounding flips: at step 0 logits 3e-2 of
    their range and loss 5e-3; 8e-2 / 2e-2 after SGD steps.  Against the pure-fp32 reference the same
    run must stay within bf16 noise (loss 3e-2)."""
    from iif_amd.custom import IIFLoss
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.bfloat16)
    if arch not in R.CIFAR_ARCHS:
        net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=11)
    table = O.iif_tables(counts)["raw"]
    crit = IIFLoss(DS(counts), variant="raw")
    q_sd = {k: v.clone() for k, v in sd.items()}
    f_sd = {k: v.clone() for k, v in sd.items()}
    qb, fb = {}, {}
    net.train()
    for it in range(3):
        lr = 0.1 * O.warmup_factor(it, 1000)
        q_loss, q_logits = R.train_step(q_sd, qb, x, y, table, arch, lr, q=R.bf16_storage)
        f_loss, _ = R.train_step(f_sd, fb, x, y, table, arch, lr)
        loss, logits = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
        net.sgd_step(lr, 0.9, 1e-4)
        # step 0: same weights, only forward rounding differs.  Later steps also carry the bf16
        # storage of the GPU's gradient tensors, which the oracle's autograd keeps in fp32.
        assert relerr(logits, q_logits) <= (3e-2 if it == 0 else 8e-2), (it, relerr(logits, q_logits))
        assert relerr(loss, q_loss) <= (5e-3 if it == 0 else 2e-2), (it, loss.item(), q_loss.item())
        assert relerr(loss, f_loss) <= 3e-2, (it, loss.item(), f_loss.item())


def test_state_dict_roundtrip_and_device_move():
    from iif_amd import resnet_cifar
    net = resnet_cifar.resnet20(num_classes=10, device="cpu", compute_dtype=torch.float32)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    assert net._arena.is_cuda and net.conv1.weight.is_cuda
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    assert net.linear.weight.data_ptr() >= net._arena.data_ptr()
    x = torch.randn(2, 3, 32, 32, device=DEV)
    out = net(x)
    assert out.shape == (2, 10) and torch.isfinite(out).all()


def test_train_cli_runs_natively(tmp_path, capsys):
    """`python -m iif_amd.train` surface: 1 epoch x 4 iterations of ResNet20 + IIF with mixup on the
    synthetic CIFAR10-LT set, evaluation, checkpoint, then --resume/--test-only."""
    from iif_amd import train
    argv = ["--model", "resnet20", "--dset_name", "cifar10", "--classif", "iif", "--iif", "raw", "-b", "32", "--epochs", "1",
            "--max-iters", "4", "-j", "0", "--print-freq", "2", "--mixup", "1.0", "--output-dir", str(tmp_path),
            "--compute-dtype", "f32"]
    args = train.get_args_parser().parse_args(argv)
    train.main(args)
    out = capsys.readouterr().out
    assert "Acc@1" in out and "best acc is" in out
    ckpt = torch.load(tmp_path / "checkpoint.pth", map_location="cpu", weights_only=False)
    assert ckpt["epoch"] == 0 and "linear.weight" in ckpt["model"]
    args2 = train.get_args_parser().parse_args(argv + ["--resume", str(tmp_path / "checkpoint.pth"), "--test-only"])
    train.main(args2)


@pytest.mark.parametrize("head", ["cosine", "lr_cosine", "norm"])
@pytest.mark.parametrize("arch,C,B,hw", [("resnet20", 10, 6, 32), ("resnet18", 100, 4, 64)])
def test_fp32_cosine_and_normed_heads(head, arch, C, B, hw):
    """--classif_norm {cosine, lr_cosine, norm} (resnet_cifar.py:38-78): logits, loss and every
    gradient (incl. the learnable scale) against the oracle, fp32 parity mode, same ReLU decisions."""
    from iif_amd import resnet_cifar, resnet_pytorch
    from iif_amd.custom import IIFLoss
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    cifar = arch in R.CIFAR_ARCHS
    sd = (R.init_cifar if cifar else R.init_imagenet)(arch, C, seed=3)
    sd = R.set_head(sd, arch, C, head, seed=1)
    if not cifar:
        damp_residual_branches(sd, arch)
    mod = resnet_cifar if cifar else resnet_pytorch
    kw = {} if cifar else {"pretrained": "None"}
    net = getattr(mod, arch)(num_classes=C, use_norm=head, compute_dtype=torch.float32, **kw)
    assert list(net.state_dict().keys()) == list(sd.keys()) or set(net.state_dict().keys()) == set(sd.keys())
    net.load_state_dict(sd)
    x, y = _data(B, hw, counts, seed=21)
    table = O.iif_tables(counts)["raw"]
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    logits = net(x.to(DEV))
    loss = crit(logits, y.to(DEV))
    loss.backward()
    masks = R.ReluMasks(gpu_relu_masks(net))
    ref_sd = {k: v.clone() for k, v in sd.items()}
    ref_loss, ref_logits, ref_grads = R.loss_and_grads(ref_sd, x, y, table, arch, relu_masks=masks, head=head)
    assert masks.disagree <= 1e-4 * masks.total and masks.worst <= 1e-4
    assert relerr(logits, ref_logits) <= 1e-4 and relerr(loss, ref_loss) <= 1e-4
    for k, p in net.named_parameters():
        a, b = p.grad.double().cpu(), ref_grads[k].double()
        e = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        assert e <= 2e-4 or b.norm().item() == 0 and a.norm().item() == 0, (k, e)
    # two fused steps keep tracking the oracle
    bufs = {}
    ref2 = {k: v.clone() for k, v in sd.items()}
    net.load_state_dict(sd)
    net._mom_arena.zero_()
    for it in range(2):
        l, _ = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
        m2 = R.ReluMasks(gpu_relu_masks(net))
        net.sgd_step(0.01, 0.9, 1e-4)
        rl, _ = R.train_step(ref2, bufs, x, y, table, arch, 0.01, relu_masks=m2, head=head)
        assert relerr(l, rl) <= 1e-4, (it, l.item(), rl.item())


def test_bf16_cosine_head_runs():
    from iif_amd import resnet_pytorch
    from iif_amd.custom import IIFLoss
    C = 1000
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    net = resnet_pytorch.resnet50(num_classes=C, use_norm="cosine", pretrained="None")
    x, y = _data(8, 64, counts, seed=2)
    crit = IIFLoss(DS(counts))
    net.train()
    l0, _ = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
    net.sgd_step(0.01, 0.9, 1e-4)
    l1, lg = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
    assert torch.isfinite(l0).item() and torch.isfinite(l1).item() and lg.abs().max().item() <= 16.0 + 1e-2


def test_decoupled_classifier_stage():
    """--decoup (classification/train.py:123-145): only the classifier trains; the backbone's weights and
    momentum stay untouched, BN keeps using batch statistics, the loss sequence follows the oracle."""
    from iif_amd import resnet_cifar
    from iif_amd.custom import IIFLoss
    arch, C, B = "resnet20", 10, 8
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.float32)
    net.select_training_param()
    assert [k for k, p in net.named_parameters() if p.requires_grad] == ["linear.weight", "linear.bias"]
    assert (net.linear.bias == 0.01).all()
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    ref = {k: v.cpu() for k, v in ref.items()}
    frozen_before = net.param_arena[:net.block_offsets()["head"]].clone()
    x, y = _data(B, 32, counts, seed=4)
    table = O.iif_tables(counts)["raw"]
    crit = IIFLoss(DS(counts))
    net.train()
    hb = {}
    for it in range(3):
        loss, _ = net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
        net.sgd_step(0.05, 0.9, 1e-4)
        rl, _, grads = R.loss_and_grads(ref, x, y, table, arch)
        keys = ["linear.weight", "linear.bias"]
        params, bl = [ref[k] for k in keys], [hb.get(k) for k in keys]
        with torch.no_grad():
            O.sgd_step(params, [grads[k] for k in keys], bl, 0.05, 0.9, 1e-4)
        for k, b in zip(keys, bl):
            hb[k] = b
        assert relerr(loss, rl) <= 1e-4, (it, loss.item(), rl.item())
    assert torch.equal(net.param_arena[:net.block_offsets()["head"]], frozen_before)
    assert relerr(net.linear.weight, ref["linear.weight"]) <= 1e-4
    assert relerr(net.bn1.running_mean, ref["bn1.running_mean"]) <= 1e-5


@pytest.mark.parametrize("arch,C,B,hw", [("resnet50", 1000, 32, 64), ("resnet18", 100, 16, 64)])
def test_bf16_fused_bn_backward_sums_match_the_reduction_pass(arch, C, B, hw, monkeypatch):
    """bf16 mode: the BN-backward sums emitted by the data-gradient epilogues (iif_conv_igemm_dgrad_bnbwd +
    iif_bn_backward_partials) against the standalone reduction pass over (dy, x).  The sums themselves agree to
    1e-6 (tests/test_conv_gpu.py::test_dgrad_emits_upstream_bn_backward_sums); the fp32 summation order differs,
    which flips a few bf16 roundings of dx per layer, and a 50-layer net evaluated on a handful of images amplifies
    that layer by layer (measured at 8 images: 1.5e-4 at layer4.1 growing to ~1e-2 at layer1), so the whole-net bound is loose and
    the per-tensor bound only excludes O(1) errors (a wrong partial row or mask)."""
    from iif_amd.custom import IIFLoss
    monkeypatch.setenv("IIF_NO_NOSTORE", "1")       # both runs on the stored forward (the never-stored one normalises with other statistics)
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=21)
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    xd, yd = x.to(DEV), y.to(DEV)
    net.loss_and_backward(xd, yd, crit)
    plan = net._saved
    assert plan.fuse_bwd
    fused = net._grad_arena.clone()
    plan.fuse_bwd = False
    net.loss_and_backward(xd, yd, crit)
    plain = net._grad_arena.clone()
    plan.fuse_bwd = True
    err = (fused - plain).norm().item() / plain.norm().item()
    assert err <= 3e-2, err
    # per tensor: no gradient tensor drifts (a wrong partial row or mask would show up as O(1))
    o = 0
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        a_, b_ = fused[off:off + rows * pitch], plain[off:off + rows * pitch]
        e = (a_ - b_).norm().item() / max(b_.norm().item(), 1e-12)
        # the stem's BN sits behind all 50 layers of amplification: 5.1e-2 .. 6.2e-2 over three seeds with the round-2 kernels
        # AND with the round-3 ones (profiles/r3_fused_sums_noise.txt, scripts/dbg_fused_sums.py); every other tensor <= 2.3e-2.
        # Round 5 (profiles/r5_fused_sums_noise.txt): 5.1e-2 .. 6.0e-2 for bn1.bias, <= 1.8e-2 elsewhere, 1.1e-2 whole
        assert e <= (8e-2 if m_ is net.bn1 else 3e-2), (type(m_).__name__, attr, e)


@pytest.mark.parametrize("streams", ["three_streams", "no_shortcut_stream", "one_stream"])
@pytest.mark.parametrize("pure", [False, True], ids=["sums_from_producer", "sums_from_P"])
def test_bn3_backward_by_algebra_inside_the_step(monkeypatch, pure, streams):
    """The algebraic conv3 + bn3 backward (csrc/bn3_algebra.hip; both variants: sum g~ xhat from the producing data gradient,
    the default below 1.5e8 elements, and sum g~ y from P = g~^T a2, the default above) against the standard route on the
    same step: they differ by bf16 roundings only, which the 50 layers amplify to ~1e-2 (same size as the fused-sums route's
    distance from the reduction pass, see above)."""
    from iif_amd.custom import IIFLoss
    monkeypatch.setenv("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "0" if pure else "1e30")
    monkeypatch.setenv("IIF_TWOPASS", "1")                  # with sums from P: conv3's output is not even stored in forward
    monkeypatch.setenv("IIF_NO_NOSTORE", "1")               # (the round-6 forward has a test of its own below: its statistics are not bit-identical)
    monkeypatch.setenv("IIF_NO_RX", "1")                    # (and so has the producer that recomputes conv3's tile: these are the round-3 routes)
    if streams == "no_shortcut_stream":                     # the shortcut's BN backward then runs on the compute stream, in front
        monkeypatch.setenv("IIF_NO_BWD_SIDE", "1")          # of a weight-gradient stream that may still read the block's g
    elif streams == "one_stream":
        monkeypatch.setenv("IIF_NO_WGRAD_STREAM", "1")
    arch, C, B, hw = "resnet50", 1000, 32, 64
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=21)
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    xd, yd = x.to(DEV), y.to(DEV)
    net.loss_and_backward(xd, yd, crit)
    plan = net._saved
    assert len(plan.alg3_units) == 13                       # layer1..layer3 (conv3 input channels <= 256)
    alg = net._grad_arena.clone()
    assert len(plan.twopass_units) == (10 if pure else 0)   # the identity blocks of layer1..layer3 (2 + 3 + 5)
    keep, plan.alg3_units, keep2, plan.twopass_units = plan.alg3_units, set(), plan.twopass_units, set()
    loss_std, _ = net.loss_and_backward(xd, yd, crit)
    std = net._grad_arena.clone()
    plan.alg3_units, plan.twopass_units = keep, keep2
    loss_alg, _ = net.loss_and_backward(xd, yd, crit)
    assert loss_alg.item() == loss_std.item()               # the two-pass forward is bit-identical to conv + bn_apply
    assert torch.equal(net._grad_arena, alg)                # fixed summation orders, event-ordered hand-overs
    assert (alg - std).norm().item() / std.norm().item() <= 3e-2
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        a_, b_ = alg[off:off + rows * pitch], std[off:off + rows * pitch]
        e = (a_ - b_).norm().item() / max(b_.norm().item(), 1e-12)
        assert e <= (8e-2 if m_ is net.bn1 else 3e-2), (type(m_).__name__, attr, e)


@pytest.mark.parametrize("streams", ["three_streams", "one_stream"])
def test_never_stored_conv3_forward_inside_the_step(monkeypatch, streams):
    """Round 6: conv3's raw output is neither written nor read (statistics from the accumulators, BN + identity / normalised
    shortcut + ReLU in the convolution's epilogue, "sums from P" backward).  Its statistics are those of the unrounded
    convolution instead of its bf16 rounding, so its forward is NOT bit-identical to the standard route's: a few ReLU decisions
    flip and the 50 layers amplify that (measured, scripts/dbg_nostore.py: the two bf16 routes are 2e-1 apart at this
    initialisation and both 3.1e-1 from the fp32 step).  What is asserted: the loss agrees to bf16-noise level; the route is no
    further from the fp32-compute step of the same network than the standard bf16 route is (whole gradient and per tensor);
    nothing reads conv3's raw output (NaN-filled); repeated steps are bit-identical."""
    from iif_amd.custom import IIFLoss
    monkeypatch.setenv("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "0")
    if streams == "one_stream":
        monkeypatch.setenv("IIF_NO_WGRAD_STREAM", "1")
    arch, C, B, hw = "resnet50", 1000, 32, 128
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    x, y = _data(B, hw, counts, seed=23)
    crit = IIFLoss(DS(counts), variant="raw")
    xd, yd = x.to(DEV), y.to(DEV)
    net32, sd = _build(arch, C, torch.float32)
    net32.load_state_dict(damp_residual_branches(sd, arch))
    net32.train()
    loss32, _ = net32.loss_and_backward(xd, yd, crit)
    ref = net32._grad_arena.clone()
    del net32
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    net.train()
    loss_ns, _ = net.loss_and_backward(xd, yd, crit)
    loss_ns = loss_ns.item()
    plan = net._saved
    assert len(plan.alg3_units) == 13
    # 32 x 32, 16 x 16 and 8 x 8 pixels at batch 32: every algebra unit has >= 1024 rows, downsample blocks included
    assert len(plan.nostore_units) == 13 and not plan.twopass_units
    for u in plan.nostore_units:
        u.x.fill_(float("nan"))                             # nothing may read conv3's raw output
    ns = net._grad_arena.clone()
    loss_again, _ = net.loss_and_backward(xd, yd, crit)
    assert loss_again.item() == loss_ns and torch.equal(net._grad_arena, ns)
    assert not torch.isnan(ns).any()
    keep, plan.alg3_units = plan.alg3_units, set()
    loss_std, _ = net.loss_and_backward(xd, yd, crit)
    std = net._grad_arena.clone()
    plan.alg3_units = keep
    assert abs(loss_ns - loss_std.item()) <= 2e-3 * abs(loss_std.item())
    assert abs(loss_ns - loss32.item()) <= 2e-3 * abs(loss32.item())
    rel = lambda a_, b_: (a_ - b_).norm().item() / max(b_.norm().item(), 1e-12)      # noqa: E731
    assert rel(ns, ref) <= 1.05 * rel(std, ref) + 5e-3, (rel(ns, ref), rel(std, ref))
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        sl = slice(off, off + rows * pitch)
        assert rel(ns[sl], ref[sl]) <= 1.25 * rel(std[sl], ref[sl]) + 2e-2, (type(m_).__name__, attr, rel(ns[sl], ref[sl]), rel(std[sl], ref[sl]))


def test_stride1_shortcut_bn_backward_by_algebra(monkeypatch):
    """The convolutional shortcut of layer1.0 (1x1 / stride 1 + BN, resnet_pytorch.py:152-167 `identity = self.downsample(x)`)
    takes the same algebra as bn3 on the shortcut stream (resnet_engine.py::_ds_algebra): every gradient of the step against the
    step with the shortcut on the standard passes (reduction, normalisation, data gradient, weight gradient) - bf16 roundings
    apart, amplified as in the test above - and bit-identical when repeated."""
    from iif_amd.custom import IIFLoss
    arch, C, B, hw = "resnet50", 1000, 32, 64
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.bfloat16)
    net.load_state_dict(damp_residual_branches(sd, arch))
    x, y = _data(B, hw, counts, seed=22)
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    xd, yd = x.to(DEV), y.to(DEV)
    net.loss_and_backward(xd, yd, crit)
    plan = net._saved
    assert len(plan.ds_alg) == 1                            # layer1.0: the only stride-1 shortcut convolution of a ResNet-50
    alg = net._grad_arena.clone()
    net.loss_and_backward(xd, yd, crit)
    assert torch.equal(net._grad_arena, alg)
    keep, plan.ds_alg = plan.ds_alg, {}
    net.loss_and_backward(xd, yd, crit)
    std = net._grad_arena.clone()
    plan.ds_alg = keep
    assert (alg - std).norm().item() / std.norm().item() <= 2e-2
    ds = net.layer1[0].downsample
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        a_, b_ = alg[off:off + rows * pitch], std[off:off + rows * pitch]
        e = (a_ - b_).norm().item() / max(b_.norm().item(), 1e-12)
        if m_ is ds[0] or m_ is ds[1]:
            assert e <= 2e-2, (type(m_).__name__, attr, e)  # the shortcut's own weight / BN gradients: nothing amplifies them
        else:
            assert e <= (8e-2 if m_ is net.bn1 else 3e-2), (type(m_).__name__, attr, e)


@pytest.mark.parametrize("arch,C,B,hw,alg3", [("resnet50", 1000, 16, 64, False), ("resnet50", 1000, 16, 64, True),
                                              ("resnet50", 1000, 16, 64, "no_shortcut_stream"), ("resnet50", 1000, 16, 64, "one_stream"),
                                              ("resnext50_32x4d", 365, 8, 64, False)])
def test_bf16_step_is_bit_reproducible(arch, C, B, hw, alg3, monkeypatch):
    """Three streams (main, weight gradients, shortcut branch), split-K slabs, fused statistics: every sum has a
    fixed order and every cross-stream hand-over an event, so the same step twice gives bit-identical gradients
    (alg3: with the algebraic BN3 backward forced on for every eligible bottleneck)."""
    from iif_amd.custom import IIFLoss
    if alg3 is True:
        monkeypatch.setenv("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "0")          # sums from P everywhere (the default is mixed)
    elif alg3 == "no_shortcut_stream":                                      # sums from the producer, P on the weight-gradient
        monkeypatch.setenv("IIF_NO_BWD_SIDE", "1")                          # stream, shortcut backward on the compute stream
    elif alg3 == "one_stream":
        monkeypatch.setenv("IIF_NO_WGRAD_STREAM", "1")
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    net, sd = _build(arch, C, torch.bfloat16)
    x, y = _data(B, hw, counts, seed=33)
    crit = IIFLoss(DS(counts), variant="raw")
    net.train()
    xd, yd = x.to(DEV), y.to(DEV)
    ref = None
    for _ in range(4):
        loss, _ = net.loss_and_backward(xd, yd, crit)
        torch.cuda.synchronize()
        cur = (loss.item(), net._grad_arena.clone())
        if ref is None:
            ref = cur
        else:
            assert cur[0] == ref[0]
            assert torch.equal(cur[1], ref[1])


FULL_SIZE = [("resnet50", 1000, 32, 224, torch.float32), ("resnet50", 1000, 32, 224, torch.bfloat16),       # configs 2/3: B=256
             ("resnext101_32x4d", 365, 16, 224, torch.float32), ("resnext101_32x4d", 365, 16, 224, torch.bfloat16),   # config 4: B=128
             ("resnet32", 100, 16, 32, torch.float32), ("resnet32", 100, 16, 32, torch.bfloat16)]          # config 1: B=128


@pytest.mark.parametrize("arch,C,rep,hw,dt", FULL_SIZE)
def test_full_size_step_replication_property(arch, C, rep, hw, dt):
    """BASELINE.json's full sizes (ResNet50 C=1000 B=256 224x224; ResNeXt-101-32x4d C=365 B=128; ResNet32 C=100
    B=128 32x32), checked through a size-independent
    property: a batch made of 8 images repeated ``rep`` times has the batch statistics of the 8 images, so
      (a) every replica's logits are BIT-identical to replica 0's (same arithmetic in every tile, whatever
          tile / halo window / wave the row landed in);
      (b) fp32 mode: logits, loss (mean reduction) and EVERY weight gradient of the 256-image step equal the
          CPU oracle's on the 8 images, given the ReLU decisions the GPU took (see the parity test above).
          Bound per tensor: 2e-4, or twice the distance of the oracle's own fp32 arithmetic from its fp64
          evaluation where that is larger.  The three stem tensors upstream of the max-pool get 1e-2: of the
          1.6 M pooling windows of 8 such images one typically has its two largest candidates within fp32
          rounding (measured 5e-7 apart, scripts/dbg_stem_bwd.py), two fp32 implementations then route
          that window's gradient to different pixels, which shows as ~2e-3 of conv1.weight's gradient;
      (c) bf16 mode: logits and loss against the bf16-storage oracle on the 8 images as in the small-size
          test.  Gradients of a freshly initialised network on 8 images are dominated by bf16 rounding
          whatever the implementation (the oracle's own bf16-storage gradients are ~45 % in L2 from its fp32
          ones: BN backward subtracts nearly equal terms), so the check is relative to that floor: per
          tensor the GPU's distance from the fp32 gradients is at most 1.5x the bf16-storage oracle's
          (+2e-2), and no larger in the median (x1.15)."""
    from iif_amd.custom import IIFLoss
    B = 8
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    x, y = _data(B, hw, counts, seed=21)
    table = O.iif_tables(counts)["raw"]
    crit = IIFLoss(DS(counts), variant="raw")
    fp32 = dt == torch.float32
    out, masks = {}, None
    for name, reps in (("full", rep),):
        net, sd = _build(arch, C, dt)
        if arch not in R.CIFAR_ARCHS:
            net.load_state_dict(damp_residual_branches(sd, arch))      # damps ``sd`` in place
        net.train()
        logits = net(x.repeat(reps, 1, 1, 1).to(DEV))
        loss = crit(logits, y.repeat(reps).to(DEV))
        loss.backward()
        out[name] = (logits.detach().float().cpu(), loss.detach().float().cpu(),
                     {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()})
        if fp32:
            masks = R.ReluMasks(gpu_relu_masks(net, rows=B))
        del net, logits, loss
        torch.cuda.empty_cache()
    lf, loss_f, gf = out["full"]
    assert lf.shape == (B * rep, C)
    reps_view = lf.view(rep, B, C)
    assert all(torch.equal(reps_view[r], reps_view[0]) for r in range(1, rep))                     # (a)
    if fp32:                                                                                         # (b)
        ref_loss, ref_logits, ref_g = R.loss_and_grads({k: v.clone() for k, v in sd.items()}, x, y, table, arch, relu_masks=masks)
        assert masks.disagree <= 1e-4 * masks.total and masks.worst <= 1e-4, (masks.disagree, masks.total, masks.worst)
        masks64 = R.ReluMasks(masks.masks)
        sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        _, _, g64 = R.loss_and_grads(sd64, x.double(), y, table.double(), arch, relu_masks=masks64)
        assert relerr(reps_view[0], ref_logits) <= 1e-4
        assert relerr(loss_f, ref_loss) <= 1e-4
        l2 = lambda a, b: (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-12)   # noqa: E731
        for k in gf:
            own, ref_noise = l2(gf[k], g64[k]), l2(ref_g[k], g64[k])
            bound = 1e-2 if k in ("conv1.weight", "bn1.weight", "bn1.bias") and arch not in R.CIFAR_ARCHS else 2e-4
            assert own <= max(bound, 2.0 * ref_noise), (k, own, ref_noise)
    else:                                                                                            # (c)
        fresh = lambda: {k: v.clone() for k, v in sd.items()}     # noqa: E731
        _, _, ref_g = R.loss_and_grads(fresh(), x, y, table, arch)
        q_loss, q_logits, q_g = R.loss_and_grads(fresh(), x, y, table, arch, q=R.bf16_storage)
        assert relerr(reps_view[0], q_logits) <= 3e-2
        assert relerr(loss_f, q_loss) <= 5e-3, (loss_f.item(), q_loss.item())
        l2 = lambda a, b: (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-12)   # noqa: E731
        own = {k: l2(gf[k], ref_g[k]) for k in gf}
        floor = {k: l2(q_g[k], ref_g[k]) for k in gf}
        worst = max((own[k] / (1.5 * floor[k] + 2e-2), k) for k in gf)
        med = lambda d: sorted(d.values())[len(d) // 2]           # noqa: E731
        print("bf16 full-size gradients vs fp32: median %.3f (bf16-storage oracle %.3f), worst ratio %s" % (med(own), med(floor), worst))
        assert worst[0] <= 1.0, worst
        assert med(own) <= 1.15 * med(floor), (med(own), med(floor))


# ----------------------------------------------------------------------------------------------------------------
# HIP path straight against the REFERENCE's own numbers (tests/golden/g7_nets.npz, g10_se.npz, g16_nets_conditioned.npz:
# the reference's resnet_cifar / resnet_pytorch modules, IIFLoss, torch.optim.SGD and warm-up run by
# tests/golden/make_golden.py on the seed-7 state dict and the seed-99 batch).  No CPU oracle in the loop, no
# ReLU-decision replay.  Every fixture also holds the same reference modules run in float64: the distance between the
# reference's fp32 and fp64 runs is the noise floor of the recipe.
#   * CIFAR networks (random init, 8 images): noise ~1e-7 -> the whole loss curve is held to 1e-4.
#   * ImageNet architectures, conditioned input (G16: last BN gain of every bottleneck x0.1, 8 images): noise <= 1e-5 ->
#     the whole loss curve is held to 1e-4.
#   * ImageNet architectures at plain random init on 2 images (G7/G10): the reference's own fp32 run is 7e-2..3e-1 from its
#     fp64 run after ONE SGD step (profiles/r2_reference_fp32_noise.txt): only step 0 is a 1e-4 statement; the later
#     losses are printed next to the reference's own noise, not asserted.
REF_NET_CASES = [("g7_nets", "resnet32", "resnet32", 100, 8, 32, None), ("g10_se", "se_resnet32", "se_resnet32", 100, 8, 32, None),
                 ("g16_nets_conditioned", "resnet50", "resnet50", 1000, 8, 64, 0.1),
                 ("g16_nets_conditioned", "resnext50", "resnext50_32x4d", 365, 8, 64, 0.1),
                 ("g16_nets_conditioned", "se_resnet50", "se_resnet50", 1000, 8, 64, 0.1),
                 # the headline geometry (224x224: 112/56/28/14/7 feature maps), the reference's own run, no oracle, no replay
                 ("g17_resnet50_224", "resnet50_224", "resnet50", 1000, 8, 224, 0.1),
                 ("g7_nets", "resnet50", "resnet50", 1000, 2, 64, None), ("g7_nets", "resnext50", "resnext50_32x4d", 365, 2, 64, None),
                 ("g10_se", "se_resnet50", "se_resnet50", 1000, 2, 64, None)]
REF_COUNTS = {100: lambda: O.img_num_per_cls(100, 50000, "exp", 0.01),
              1000: lambda: [int(1280 * (5 / 1280) ** (i / 999.0)) for i in range(1000)],
              365: lambda: [int(4980 * (5 / 4980) ** (i / 364.0)) for i in range(365)]}


@pytest.mark.parametrize("fixture,prefix,arch,C,B,hw,damp", REF_NET_CASES,
                         ids=[c[1] + ("_conditioned" if c[6] and c[5] != 224 else "") for c in REF_NET_CASES])
def test_hip_step_against_reference_fixture(golden, fixture, prefix, arch, C, B, hw, damp, stream_mode):
    """fp32 HIP training steps on the reference's inputs vs the reference's outputs.
    Always: logits and loss of step 0 within 1e-4 relative (north_star).  Well-conditioned cases (see above): every
    later loss within 1e-4, the weights after the last step (per tensor: checksum within 2e-3 of the tensor's L1 norm,
    plus three times the reference's own fp32-fp64 distance).  Gradient norms of step 0, per parameter: 1e-3 on the
    well-conditioned cases, 2e-2 at random init on 2 images (a ReLU / max-pool tie that two correct fp32
    implementations decide differently moves every gradient upstream: the reference's own fp32 run shows 6e-3 there)."""
    import numpy as np
    from iif_amd import resnet_cifar, resnet_pytorch
    from iif_amd.custom import IIFLoss
    g = golden(fixture)
    cifar = arch in R.CIFAR_ARCHS
    conditioned = cifar or damp is not None
    sd = R.init_cifar(arch, C, seed=7) if cifar else R.init_imagenet(arch, C, seed=7)
    if damp is not None:
        assert float(g[prefix + "_damp"]) == damp
        for k in sd:
            if k.startswith("layer") and k.endswith("bn3.weight"):
                sd[k] = sd[k] * damp
    chk = np.array([float(v.double().sum()) for k, v in sd.items() if v.is_floating_point()])
    assert np.allclose(chk, g[prefix + "_init_checksum"], rtol=0, atol=1e-9)          # the reference's initial weights
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B, 3, hw, hw, generator=gen)
    assert abs(float(x.double().sum()) - float(g[prefix + "_x_sum"])) < 1e-9          # the reference's batch
    y = torch.from_numpy(g[prefix + "_y"])
    counts = REF_COUNTS[C]()
    if cifar:
        net = getattr(resnet_cifar, arch)(num_classes=C, use_norm="None", compute_dtype=torch.float32)
    else:
        net = getattr(resnet_pytorch, arch)(num_classes=C, use_norm="None", pretrained="None", compute_dtype=torch.float32)
    net.load_state_dict(sd)
    net.train()
    crit = IIFLoss(DS(counts), variant="raw")
    xd, yd = x.to(DEV), y.to(DEV)
    ref_losses, ref_losses64 = g[prefix + "_losses"], g[prefix + "_losses_f64"]
    lr0 = float(g[prefix + "_lr0"])
    report = []
    for it in range(len(ref_losses)):
        net.zero_grad()
        logits = net(xd)
        loss = crit(logits, yd)
        loss.backward()
        dev = abs(loss.item() - ref_losses[it]) / abs(ref_losses[it])
        report.append((dev, abs(ref_losses[it] - ref_losses64[it]) / abs(ref_losses64[it])))
        if it == 0:
            assert relerr(logits, torch.from_numpy(g[prefix + "_logits0"])) <= 1e-4
            gn32 = g[prefix + "_gradnorm0"]
            grads = dict(net.named_parameters())
            tol = 1e-3 if conditioned else 2e-2
            for k, a32 in zip(g[prefix + "_gradnorm_keys"].tolist(), gn32):
                mine = grads[k].grad.double().norm().item()
                assert abs(mine - a32) <= tol * max(a32, 1e-3 * gn32.max()), (k, mine, a32)
        if it == 0 or conditioned:
            assert dev <= 1e-4, (it, loss.item(), ref_losses[it], ref_losses64[it])
        net.sgd_step(lr0 * O.warmup_factor(it, 1000), 0.9, 1e-4)
    print("%s%s: |hip - ref32| / |ref32| per step %s ; reference |fp32 - fp64| / |fp64| %s" % (
        arch, "" if damp is None else " (conditioned)", ["%.1e" % r[0] for r in report], ["%.1e" % r[1] for r in report]))
    if not conditioned:
        return
    final = net.state_dict()
    fc = "linear.weight" if cifar else "fc.weight"
    assert relerr(final[fc][:4].float(), torch.from_numpy(g[prefix + "_final_fc"])) <= 1e-4
    assert relerr(final["bn1.running_mean"].float(), torch.from_numpy(g[prefix + "_final_bn1_rm"])) <= 1e-4
    fl = [(k, v) for k, v in final.items() if v.is_floating_point()]
    chk = np.array([float(v.double().sum()) for _, v in fl])
    l1 = np.array([float(v.double().abs().sum()) for _, v in fl])
    c32, c64 = g[prefix + "_final_checksum"], g[prefix + "_final_checksum_f64"]
    # 2e-3 of the L1 norm: weights move by ~1e-4 of themselves in these steps, so this only bites on the zero-initialised
    # BN biases, which are pure accumulated gradients (per-step gradient agreement 1e-3, a single ReLU tie decided the
    # other way shows up at 2e-4..8e-4 per tensor: scripts/dbg_se_precision.py, profiles/r2_relu_tie_evidence.txt)
    bad = [(fl[i][0], chk[i], c32[i], c64[i]) for i in range(len(fl))
           if abs(chk[i] - c32[i]) > 2e-3 * max(l1[i], 1e-3) + 3.0 * abs(c32[i] - c64[i])]
    assert not bad, bad[:3]


def test_module_surface_does_not_alias_and_refuses_stale_backward():
    """model(x) hands out its own tensor (the plan's logits buffer is reused by the next call); a backward whose
    activations a later forward overwrote raises instead of back-propagating the wrong batch; an eval-mode
    forward under autograd raises (the native backward is the training-mode one)."""
    net, _ = _build("resnet20", 10, torch.float32)
    net.train()
    xa, _ = _data(4, 32, [10] * 10, seed=1)
    xb, _ = _data(4, 32, [10] * 10, seed=2)
    la = net(xa.to(DEV))
    keep = la.detach().clone()
    lb = net(xb.to(DEV))
    assert torch.equal(la.detach(), keep) and not torch.equal(la.detach(), lb.detach())
    with pytest.raises(RuntimeError, match="ONE forward"):
        la.sum().backward()
    lb.sum().backward()                                   # the latest forward is fine
    with torch.no_grad():
        ea, eb = net(xa.to(DEV)), net(xb.to(DEV))
    assert ea.data_ptr() != eb.data_ptr()
    net.eval()
    with pytest.raises(RuntimeError, match="eval-mode forward under autograd"):
        net(xa.to(DEV))
    with torch.no_grad():
        net(xa.to(DEV))


@pytest.mark.parametrize("mix", [False, True])
def test_fused_step_plain_ce_with_class_weights_weighted_mean(mix):
    """``--classif ce --deffered --reduction mean`` through loss_and_backward: the loss and d(loss)/d(logits) of
    nn.CrossEntropyLoss(weight=w) (initialisers.py:43-46), also under mixup (custom.py:116-117: each term divided by
    the weight sum of ITS targets)."""
    import types
    from iif_amd import initialisers
    C, B = 10, 12
    counts = [500, 300, 200, 120, 80, 50, 30, 20, 10, 5]
    net, _ = _build("resnet20", C, torch.float32)
    net.train()
    args = types.SimpleNamespace(classif="ce", deffered=True, reduction="mean", iif="raw", iif_norm=0, device=DEV)
    crit = initialisers.get_criterion(args, DS(counts), net, C)
    x, ya = _data(B, 32, counts, seed=3)
    _, yb = _data(B, 32, counts, seed=4)
    lam = 0.3
    loss, logits = net.loss_and_backward(x.to(DEV), ya.to(DEV), crit, targets_b=yb.to(DEV) if mix else None, lam=lam)
    lg = logits.detach().float().cpu().clone().requires_grad_(True)
    w = torch.tensor(counts); w = (w.sum() / w).float()
    ref_crit = torch.nn.CrossEntropyLoss(weight=w, reduction="mean")
    ref = lam * ref_crit(lg, ya) + (1 - lam) * ref_crit(lg, yb) if mix else ref_crit(lg, ya)
    ref.backward()
    assert relerr(loss, ref) <= 1e-4
    assert relerr(net._saved.dlogits[:, :C], lg.grad) <= 1e-4


def test_bf16_loss_curve_of_the_default_routes_against_the_standard_backward(monkeypatch):
    """The DEFAULT bf16 configuration at the benchmark's image size (224 x 224, batch 64: the 56 x 56 stage takes the algebraic
    BN3 backward with sum g~ y from P = g~^T a2 — sums of the UNROUNDED conv3 output, csrc/bn3_algebra.hip — the other stages
    the producer's sums, the stem its pooled sums) against the same network with every BN backward on the standard
    reduction passes: six SGD steps on the damped initialisation.  The two differ by bf16 roundings only, and the recipe
    amplifies such differences from step to step: measured in round 5 (scripts/dbg_route_curve.py), the standard backward with
    and without the fused BN-backward sums - two orderings of the same sums - are 3.9e-4, 1.7e-3, 3.9e-4, 3.8e-3 and 1.0e-2 apart
    at steps 2 ... 6, the default routes 2.7e-4, 6.8e-4, 2.2e-3, 2.7e-3, 7.5e-3 from the standard backward.  The bound grows with
    the step accordingly (a fixed 5e-3 over all six steps held in round 4 by the luck of one trajectory and broke when the
    stem's sums became MORE exact); a drift of a route shows at steps 2-4, where the bound is still tight."""
    from iif_amd.custom import IIFLoss
    arch, C, B, hw = "resnet50", 1000, 64, 224
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    x, y = _data(B, hw, counts, seed=77)
    xd, yd = x.to(DEV), y.to(DEV)
    crit = IIFLoss(DS(counts), variant="raw")
    curves = {}
    # (batch 64 instead of 256: the threshold between the two algebra variants scaled with it, so the 56 x 56 stage takes
    # "sums from P" and the other stages the producer's sums, as at the benchmark's size)
    monkeypatch.setenv("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "5e7")
    for mode in ("default", "standard"):
        if mode == "standard":
            monkeypatch.setenv("IIF_NO_BN3_ALGEBRA", "1")
            monkeypatch.setenv("IIF_NO_BWD_FUSE", "1")
        net, sd = _build(arch, C, torch.bfloat16)
        net.load_state_dict(damp_residual_branches(sd, arch))
        net.train()
        losses = []
        for it in range(6):
            loss, _ = net.loss_and_backward(xd, yd, crit)
            net.sgd_step(0.002, 0.9, 1e-4)
            losses.append(float(loss.item()))
        if mode == "default":
            plan = net._saved
            # the mixed configuration of the benchmark: the 56 x 56 and 28 x 28 stages never store conv3's output (statistics pass +
            # BN epilogue forward, producer-recomputed sums backward), the 14 x 14 stage takes the stored route
            assert len(plan.alg3_units) == 13 and len(plan.rx_units) == 7 and len(plan.nostore_units) == 7
        curves[mode] = losses
        del net
    # (the first loss is no longer bit-equal: the never-stored forward normalises with the statistics of the unrounded convolution)
    for a, b, tol in zip(curves["default"], curves["standard"], (1e-3, 2e-3, 3e-3, 5e-3, 1e-2, 2e-2)):
        assert abs(a - b) <= tol * abs(b), (curves["default"], curves["standard"])
    assert curves["default"][-1] < curves["default"][0]              # (lr 0.002: the raw IIF recipe descends smoothly)


def test_p_and_gram_from_the_producer_inside_the_step(monkeypatch):
    """Round 6: the 56 x 56 bottlenecks take P = g~^T a2 and Gram = a2^T a2 of the algebraic BN3 backward from the producing data
    gradient's own launch (iif_conv_igemm_dgrad_masksum_rx_pg + iif_slab_sum) instead of a weight-gradient GEMM over g~ and a2.
    Same products of the same stored tensors in another summation order: the step with and without (IIF_NO_PG=1) agrees to fp32
    summation noise in conv3's weight gradients of those blocks and bit for bit everywhere the matrices do not reach."""
    from iif_amd.custom import IIFLoss
    arch, C, B, hw = "resnet50", 100, 32, 224
    counts = [max(int(500 * (5 / 500) ** (i / (C - 1.0))), 1) for i in range(C)]
    x, y = _data(B, hw, counts, seed=5)
    xd, yd = x.to(DEV), y.to(DEV)
    crit = IIFLoss(DS(counts), variant="raw")
    monkeypatch.setenv("IIF_SIDE_STREAMS", "1")
    monkeypatch.setenv("IIF_BN3_ALGEBRA_PURE_MIN_ELEMS", "0")
    out = {}
    for mode in ("pg", "plain"):
        if mode == "plain":
            monkeypatch.setenv("IIF_NO_PG", "1")
        net, sd = _build(arch, C, torch.bfloat16)
        net.load_state_dict(damp_residual_branches(sd, arch))
        net.train()
        loss, _ = net.loss_and_backward(xd, yd, crit)
        torch.cuda.synchronize()
        plan = net._saved
        assert len(plan.pg_units) == (3 if mode == "pg" else 0), len(plan.pg_units)
        out[mode] = (float(loss), net.grad_arena.clone(), {id(u.conv) for u in plan.pg_units}, net)
    assert out["pg"][0] == out["plain"][0]
    gp, gq, net = out["pg"][1], out["plain"][1], out["pg"][3]
    touched = 0
    for (m_, attr, rows, pitch) in net._param_specs():
        off = net._offsets[(id(m_), attr)][0]
        a, b = gp[off:off + rows * pitch], gq[off:off + rows * pitch]
        if torch.equal(a, b):
            continue
        touched += 1
        # conv3 weights of the three blocks (dW = A P + B W Gram + D (x) csum): the summation order of P and Gram only
        assert attr == "weight" and (a - b).norm().item() <= 2e-5 * b.norm().item(), (attr, rows, pitch, (a - b).norm().item(), b.norm().item())
    assert 1 <= touched <= 3


def test_side_streams_follow_the_size_of_the_step(monkeypatch):
    """Without an override the plan of a CIFAR-size step runs on one stream (host-bound: 3.24 -> 2.52 ms at ResNet32 bs 128)
    and an ImageNet-size one gets the weight-gradient and shortcut streams; both produce the same gradients as the forced
    settings (the streams only reorder independent launches)."""
    from iif_amd import resnet_cifar, resnet_pytorch
    from iif_amd.custom import IIFLoss
    dev = torch.device("cuda", 0)

    class DS:
        def get_cls_num_list(self):
            return [100 - i for i in range(10)]

    def run(make, shape, env):
        monkeypatch.delenv("IIF_SIDE_STREAMS", raising=False)
        monkeypatch.delenv("IIF_NO_WGRAD_STREAM", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(3)
        net = make()
        net.train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(*shape, generator=g).to(dev)
        y = torch.randint(0, 10, (shape[0],), generator=g).to(dev)
        crit = IIFLoss(DS(), variant="raw", device=dev)
        loss, _ = net.loss_and_backward(x, y, crit)
        torch.cuda.synchronize()
        return net._saved.wg_stream is not None, float(loss), net.grad_arena.clone()

    small = lambda: resnet_cifar.resnet32(num_classes=10, use_norm="None", device=dev, compute_dtype=torch.bfloat16)  # noqa: E731
    big = lambda: resnet_pytorch.resnet50(num_classes=10, use_norm="None", pretrained="None", device=dev, compute_dtype=torch.bfloat16)  # noqa: E731
    on_s, l_auto, g_auto = run(small, (16, 3, 32, 32), {})
    on_f, l_forced, g_forced = run(small, (16, 3, 32, 32), {"IIF_SIDE_STREAMS": "1"})
    assert not on_s and on_f
    assert l_auto == l_forced and torch.equal(g_auto, g_forced)
    on_b, _, _ = run(big, (48, 3, 224, 224), {})
    off_b, _, _ = run(big, (48, 3, 224, 224), {"IIF_SIDE_STREAMS": "0"})
    assert on_b and not off_b


@pytest.mark.parametrize("arch,C,B,hw", [("resnet50", 1000, 8, 64), ("resnext50_32x4d", 365, 8, 64)])
def test_two_weight_gradient_streams_give_the_same_gradients(arch, C, B, hw, monkeypatch):
    """IIF_WG_STREAMS=2 (weight gradients round robin over two streams, every split-K round sized for half of the device:
    iif_conv_wgrad splits = -2) against the default single stream: the same gradients up to the fp32 summation order of the
    split-K slabs, and bit-identical when repeated (events order every hand-over between the streams)."""
    from iif_amd.custom import IIFLoss
    counts = [max(int(1000 * (5 / 1000) ** (i / (C - 1.0))), 1) for i in range(C)]
    x, y = _data(B, hw, counts, seed=31)
    xd, yd = x.to(DEV), y.to(DEV)
    crit = IIFLoss(DS(counts), variant="raw")
    monkeypatch.setenv("IIF_SIDE_STREAMS", "1")
    grads = []
    for nws in ("1", "2"):
        monkeypatch.setenv("IIF_WG_STREAMS", nws)
        net, sd = _build(arch, C, torch.bfloat16)
        net.load_state_dict(damp_residual_branches(sd, arch))
        net.train()
        net.loss_and_backward(xd, yd, crit)
        assert len(net._saved.wg_streams) == int(nws)
        g1 = net._grad_arena.clone()
        net.loss_and_backward(xd, yd, crit)
        assert torch.equal(net._grad_arena, g1)
        grads.append(g1)
    err = (grads[1] - grads[0]).norm().item() / grads[0].norm().item()
    assert err <= 1e-5, err


def test_bf16_gradients_at_a_settled_point():
    """Round-5 review item 7: bf16 gradients where the network is NOT at its initialisation - ResNet-50, 224 x 224, batch 64,
    damped init, 20 SGD steps taken in fp32 mode (BN statistics and weights settled), then one bf16 step against the fp32 step of
    the same engine on a fresh batch (the fp32 step is what the oracle pins to 1e-4 elsewhere in this file).
    What bf16 STORAGE of activations and gradients costs at such a point was measured (scripts/dbg_bf16_settled.py,
    profiles/r6_bf16_settled.txt): loss 5e-5 away, but every gradient tensor 0.42-0.45 (relative L2) away, the whole gradient
    0.30, cosine 0.956 - synthetic noise images give 64 nearly orthogonal per-image gradients, BN backward subtracts their
    large common part, and the bf16 rounding of the remainder is of its own size; the CPU oracle with bf16 storage shows the
    same level (test_full_size_step_replication_property).  A tolerance of a few bf16 ulps therefore does not exist on this
    workload.  What IS asserted is what a wrong kernel would break and rounding noise does not: the loss, the DIRECTION of the
    whole gradient (cosine, and its length along the fp32 gradient), every tensor's projection on its fp32 counterpart
    (a missing term or a wrong scale moves it far from 1), and that no tensor is further than the measured noise level."""
    from iif_amd.custom import IIFLoss
    arch, C, B, hw = "resnet50", 1000, 64, 224
    counts = [max(int(1280 * (5 / 1280) ** (i / (C - 1.0))), 1) for i in range(C)]
    crit = IIFLoss(DS(counts), variant="raw")
    net, sd = _build(arch, C, torch.float32)
    net.load_state_dict(damp_residual_branches(sd, arch))
    net.train()
    for it in range(20):
        x, y = _data(B, hw, counts, seed=100 + it)
        net.loss_and_backward(x.to(DEV), y.to(DEV), crit)
        net.sgd_step(0.002, 0.9, 1e-4)
    settled = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    del net
    x, y = _data(B, hw, counts, seed=999)
    xd, yd = x.to(DEV), y.to(DEV)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        n2, _ = _build(arch, C, dt)
        n2.load_state_dict(settled)
        n2.train()
        loss, _ = n2.loss_and_backward(xd, yd, crit)
        out[dt] = (loss.item(), [v.double().cpu().clone().flatten() for v in n2._grad_views], [k for k, _ in n2.named_parameters()])
        del n2
    (l32, g32, names), (l16, g16, _) = out[torch.float32], out[torch.bfloat16]
    assert abs(l16 - l32) <= 5e-4 * abs(l32)
    a, b = torch.cat(g16), torch.cat(g32)
    cos = (a @ b / (a.norm() * b.norm())).item()
    proj = (a @ b / (b @ b)).item()
    assert cos >= 0.93 and 0.9 <= proj <= 1.05, (cos, proj)
    for n_, u, v in zip(names, g16, g32):
        p = (u @ v / (v @ v).clamp_min(1e-300)).item()
        e = ((u - v).norm() / v.norm().clamp_min(1e-300)).item()
        assert 0.7 <= p <= 1.15 and e <= 0.75, (n_, p, e)
