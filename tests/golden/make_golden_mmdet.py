#!/usr/bin/env python3
"""Generate the mmdet-half golden vectors (G11-G15) by RUNNING THE REFERENCE'S OWN FILES.

Runs only in the build container (needs /root/reference, read-only).  mmcv / mmdet
are not installed, and the fork's package ``__init__`` files pull in the whole
detector, so the handful of files on the IIF head path are executed one by one
from where they lie, under EMPTY PLACEHOLDER modules for the names they import
but whose arithmetic they do not use:

    mmcv.jit (decorator -> identity), mmcv.utils.Registry / mmcv.cnn.CONV_LAYERS
    (a dict with ``register_module``), mmcv.runner.force_fp32 / get_dist_info,
    mmdet.models.builder.{LOSSES,HEADS}, mmdet.core.utils.reduce_mean (the
    single-process branch: identity), mmdet.utils.get_root_logger, and
    ``ConvFCBBoxHead`` (an nn.Module that only records num_classes / cls_last_dim).

Executed from the reference, unmodified (paths under instance_segmentation/mmdet/):
    models/losses/utils.py, accuracy.py, cross_entropy_loss.py, iif_loss.py,
    fasa_iif_loss.py, models/utils/builder.py, normed_predictor.py,
    models/roi_heads/bbox_heads/lvis_instances.py, fasa_bbox_head.py

The reference hard-codes ``device='cuda'`` / ``.cuda()`` (iif_loss.py:50,
fasa_iif_loss.py:52,64-65, normed_predictor.py:58, fasa_bbox_head.py:51,66-68,
150,161); while its code runs, ``cuda`` is redirected to the CPU (``_cuda_is_cpu``).
Nothing else is patched.  Only inputs and outputs (data) are written to the repo;
the two CSV tables the plugin reads are committed next to the vectors as the data
files they are (tests/golden/lvis_files/idf_1204.csv, coco_files/idf_91.csv).

While generating, every vector is also compared with the CPU oracle
(``oracle/mmdet_iif.py``) so that the restatement is pinned by the reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mmdet.py
"""
import contextlib
import importlib.util
import os
import shutil
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFM = "/root/reference/instance_segmentation"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
torch.set_num_threads(8)

from oracle import mmdet_iif as M          # noqa: E402
from tests import mmdet_cases as K         # noqa: E402
from tests.mmdet_cases import ROW_STEP, head_inputs          # noqa: E402


# ------------------------------------------------------------------ placeholders
class _Registry:
    def __init__(self, name="", **kw):
        self.name, self.module_dict = name, {}

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self.module_dict[name or module.__name__] = module
            return module

        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def __contains__(self, key):
        return key in self.module_dict


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def _identity_decorator(*a, **kw):
    if len(a) == 1 and callable(a[0]) and not kw:
        return a[0]
    return lambda f: f


class _ConvFCBBoxHead(nn.Module):
    """Stands in for the stock head: only the two attributes the FASA subclass reads."""

    def __init__(self, *args, num_classes=1203, cls_last_dim=64, **kwargs):
        super().__init__()
        self.num_classes, self.cls_last_dim = num_classes, cls_last_dim


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFM, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _reference_modules():
    mmcv = _pkg("mmcv"); mmcv.jit = _identity_decorator
    cnn = _pkg("mmcv.cnn"); cnn.CONV_LAYERS = _Registry("conv layer"); mmcv.cnn = cnn
    mu = _pkg("mmcv.utils"); mu.Registry = _Registry; mu.build_from_cfg = None; mmcv.utils = mu
    mr = _pkg("mmcv.runner"); mr.force_fp32 = _identity_decorator; mr.get_dist_info = lambda: (0, 1); mmcv.runner = mr
    _pkg("mmdet"); _pkg("mmdet.models")
    b = _pkg("mmdet.models.builder"); b.LOSSES = _Registry("loss"); b.HEADS = _Registry("head")
    losses = _pkg("mmdet.models.losses")
    _load("mmdet.models.losses.utils", "mmdet/models/losses/utils.py")
    acc = _load("mmdet.models.losses.accuracy", "mmdet/models/losses/accuracy.py")
    ce = _load("mmdet.models.losses.cross_entropy_loss", "mmdet/models/losses/cross_entropy_loss.py")
    losses.accuracy = acc.accuracy
    losses.binary_cross_entropy, losses.mask_cross_entropy = ce.binary_cross_entropy, ce.mask_cross_entropy
    iif = _load("mmdet.models.losses.iif_loss", "mmdet/models/losses/iif_loss.py")
    fiif = _load("mmdet.models.losses.fasa_iif_loss", "mmdet/models/losses/fasa_iif_loss.py")
    _pkg("mmdet.models.utils")
    _load("mmdet.models.utils.builder", "mmdet/models/utils/builder.py")
    npred = _load("mmdet.models.utils.normed_predictor", "mmdet/models/utils/normed_predictor.py")
    _pkg("mmdet.core"); cu = _pkg("mmdet.core.utils"); cu.reduce_mean = lambda t: t        # dist_utils.py:67-69, no process group
    ut = _pkg("mmdet.utils"); ut.get_root_logger = lambda *a, **k: None
    _pkg("mmdet.models.roi_heads")
    bh = _pkg("mmdet.models.roi_heads.bbox_heads", os.path.join(REFM, "mmdet/models/roi_heads/bbox_heads"))
    bh.ConvFCBBoxHead = _ConvFCBBoxHead
    fh = _load("mmdet.models.roi_heads.bbox_heads.fasa_bbox_head", "mmdet/models/roi_heads/bbox_heads/fasa_bbox_head.py")
    return dict(iif=iif, fiif=fiif, ce=ce, acc=acc, npred=npred, fh=fh)


@contextlib.contextmanager
def _cuda_is_cpu():
    """'cuda' means the CPU while the reference's code runs (it hard-codes the device)."""
    real_tensor, real_cuda, real_zeros = torch.tensor, torch.Tensor.cuda, torch.zeros

    def tensor(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return real_tensor(*a, **k)
    torch.tensor = tensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.tensor, torch.Tensor.cuda, torch.zeros = real_tensor, real_cuda, real_zeros


REF = _reference_modules()
LVIS_CSV = os.path.join(REFM, "lvis_files/idf_1204.csv")
COCO_CSV = os.path.join(REFM, "coco_files/idf_91.csv")
VARIANTS14 = ("smooth", "raw", "prob", "normit", "gombit", "base2", "base10", "smooth_obj", "raw_obj", "prob_obj",
              "normit_obj", "gombit_obj", "base2_obj", "base10_obj")


def close(a, b, tol, what):
    a = torch.as_tensor(np.asarray(a.detach() if torch.is_tensor(a) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach() if torch.is_tensor(b) else b), dtype=torch.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = max(b.abs().max().item(), 1.0) if b.numel() else 1.0
    assert err <= tol * ref, "%s: oracle differs from reference by %g" % (what, err)


def ref_iif_loss(**kw):
    with _cuda_is_cpu():
        return REF["iif"].IIFLoss(**kw)


# --------------------------------------------------------------------------- G14
def g14_csv_tables():
    """iif_loss.py:47-50 for every column of the two shipped tables (+ the data files themselves)."""
    out = {}
    for tag, path, nc in (("lvis", LVIS_CSV, 1203), ("coco", COCO_CSV, 80)):
        for v in VARIANTS14:
            t = ref_iif_loss(num_classes=nc, path=path, variant=v).iif_weights
            assert t.dtype == torch.float32 and tuple(t.shape) == (1, nc + 1)
            assert torch.equal(t, M.read_table(path, v)), (tag, v)
            out["%s_%s" % (tag, v)] = t.numpy()
    with _cuda_is_cpu():
        t = REF["npred"].IIFNormedLinear(8, 1204, path=LVIS_CSV).iif_weights             # default variant base2_obj, [C+1, 1]
    assert tuple(t.shape) == (1204, 1)
    out["lvis_normed_default"] = t.numpy()
    for sub, path in (("lvis_files", LVIS_CSV), ("coco_files", COCO_CSV)):
        os.makedirs(os.path.join(HERE, sub), exist_ok=True)
        shutil.copyfile(path, os.path.join(HERE, sub, os.path.basename(path)))
        os.chmod(os.path.join(HERE, sub, os.path.basename(path)), 0o644)
    return out


# --------------------------------------------------------------------------- G11


def _grad_digest(out, key, grad):
    g = grad.double()
    out[key + "_drows"] = grad[::ROW_STEP].numpy()                 # every 16th row in full
    out[key + "_dcolsum"] = g.sum(0).numpy()
    out[key + "_drowabs"] = g.abs().sum(1).numpy()


def g11_cross_entropy():
    out = {}
    for tag, rel, nc, n, seed in (K.LVIS, K.COCO):
        path = os.path.join(REFM, rel)
        c1 = nc + 1
        score, label, weight = head_inputs(n, c1, seed)
        out[tag + "_shape"] = np.array([n, c1, seed])
        out[tag + "_score_sum"] = np.array(float(score.double().sum()))
        out[tag + "_label"] = label.numpy()
        out[tag + "_weight"] = weight.numpy()
        af = max(float((weight > 0).sum().item()), 1.0)                      # bbox_head.py:267
        out[tag + "_avg_factor"] = np.array(af)
        cw = K.class_weight_list(c1, seed)
        out[tag + "_class_weight"] = np.array(cw, dtype=np.float64)
        lab_ign, lab_ign7 = K.ignore_labels(label)
        cases = K.ce_cases(label, weight, af, cw, lab_ign, lab_ign7)
        for variant in K.ce_variants(tag):
            for name, (ckw, fkw, lab) in cases.items():
                if variant != "raw" and name not in ("head", "plain"):
                    continue
                crit = ref_iif_loss(num_classes=nc, path=path, variant=variant, **ckw)
                s = score.clone().requires_grad_(True)
                loss = crit(s, lab, **fkw)
                (loss.sum() if loss.dim() else loss).backward()
                key = "%s_%s_%s" % (tag, variant, name)
                out[key + "_loss"] = loss.detach().numpy()
                _grad_digest(out, key, s.grad)
                # oracle against the reference
                so = score.clone().requires_grad_(True)
                red = fkw.get("reduction_override") or ckw.get("reduction", "mean")
                ign = fkw.get("ignore_index", ckw.get("ignore_index"))
                mine = M.iif_cross_entropy(so, lab, crit.iif_weights, weight=fkw.get("weight"), reduction=red,
                                           avg_factor=fkw.get("avg_factor"),
                                           class_weight=None if "class_weight" not in ckw else torch.tensor(cw),
                                           ignore_index=ign, loss_weight=ckw.get("loss_weight", 1.0))
                (mine.sum() if mine.dim() else mine).backward()
                close(mine, loss, 1e-6, key + " loss")
                close(so.grad, s.grad, 1e-6, key + " grad")
            crit = ref_iif_loss(num_classes=nc, path=path, variant=variant)
            act = crit.get_activation(score)
            close(M.get_activation(score, crit.iif_weights), act, 1e-6, tag + " activation")
            out["%s_%s_act_rows" % (tag, variant)] = act[::ROW_STEP].numpy()
            out["%s_%s_act_rowsum" % (tag, variant)] = act.double().sum(1).numpy()
            out["%s_%s_act_colsum" % (tag, variant)] = act.double().sum(0).numpy()
        crit = ref_iif_loss(num_classes=nc, path=path)
        # error convention (losses/utils.py:53-54) and protocol values
        try:
            crit(score, label, avg_factor=3.0, reduction_override="sum")
            raised = 0
        except ValueError:
            raised = 1
        out[tag + "_sum_avg_factor_raises"] = np.array(raised)
        out[tag + "_cls_channels"] = np.array(crit.get_cls_channels(nc))
        # accuracy on the RAW score (iif_loss.py:92-107 -> accuracy.py:7-51); a score that does hit sometimes
        boosted = K.boosted_score(score, label, seed)
        a = crit.get_accuracy(boosted, label)["acc_classes"]
        assert tuple(a.shape) == (1,)
        assert a.tolist() == M.accuracy_top1(boosted, label).tolist()
        out[tag + "_acc_classes"] = a.numpy()
        a15 = REF["acc"].accuracy(boosted, label, topk=(1, 5))
        out[tag + "_acc_top1_top5"] = np.array([float(a15[0]), float(a15[1])], dtype=np.float32)
        a_thr = REF["acc"].accuracy(boosted, label, topk=1, thresh=6.0)
        out[tag + "_acc_thresh6"] = a_thr.numpy()
        e = crit.get_accuracy(boosted[:0], label[:0])["acc_classes"]
        out[tag + "_acc_empty"] = np.array(float(e)); assert tuple(e.shape) == ()
    # the reference's own CE known answers (tests/test_metrics/test_losses.py:8-32) through IIFLoss with a table of ones
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "ones.csv")
        with open(p, "w") as f:
            f.write("raw\n1\n1.0\n")
        x = torch.tensor([[100.0, -100.0]]); y = torch.tensor([1])
        out["known_ce"] = ref_iif_loss(num_classes=1, path=p)(x, y).numpy()
        out["known_ce_cw"] = ref_iif_loss(num_classes=1, path=p, class_weight=[0.8, 0.2])(x, y).numpy()
        assert float(out["known_ce"]) == 200.0 and abs(float(out["known_ce_cw"]) - 40.0) < 1e-5
    return out


# --------------------------------------------------------------------------- G12
def g12_normed():
    out = {}
    NP = REF["npred"]
    g = torch.Generator().manual_seed(21)
    for name, n, d, c, temp, power, variant in K.NORMED_LINEAR_CASES:
        x = torch.randn(n, d, generator=g) * 1.5
        w = torch.randn(c, d, generator=g) * 0.05
        b = torch.randn(c, generator=g) * 0.1
        gy = torch.randn(n, c, generator=g)
        with _cuda_is_cpu():
            if variant is None:
                m = NP.NormedLinear(d, c, tempearture=temp, power=power)
            else:
                m = NP.IIFNormedLinear(d, c, tempearture=temp, power=power, variant=variant,
                                       path=LVIS_CSV if c == 1204 else COCO_CSV)
        assert abs(m.weight.std().item() - 0.01) < 0.004 and m.bias.abs().max().item() == 0     # init law :29-32
        with torch.no_grad():
            m.weight.copy_(w); m.bias.copy_(b)
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        y.backward(gy)
        for k, v in (("x", x), ("w", w), ("b", b), ("gy", gy), ("out", y.detach()), ("dx", xr.grad), ("dw", m.weight.grad),
                     ("db", m.bias.grad)):
            out["%s_%s" % (name, k)] = v.numpy()
        out[name + "_cfg"] = np.array([temp, power, 1e-6])
        if variant is not None:
            out[name + "_rows"] = m.iif_weights.reshape(-1).numpy()
        xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True); bo = b.clone().requires_grad_(True)
        yo = M.normed_linear(xo, wo, bo, temp, power, 1e-6, None if variant is None else m.iif_weights)
        yo.backward(gy)
        close(yo, y, 1e-6, name + " out"); close(xo.grad, xr.grad, 1e-6, name + " dx")
        close(wo.grad, m.weight.grad, 1e-6, name + " dw"); close(bo.grad, m.bias.grad, 1e-6, name + " db")
    for name, n, cin, cout, hw, nok, ks, stride, pad in K.NORMED_CONV_CASES:
        m = NP.NormedConv2d(cin, cout, ks, stride=stride, padding=pad, tempearture=20, norm_over_kernel=nok)
        x = torch.randn(n, cin, hw, hw, generator=g)
        ho = (hw + 2 * pad - ks) // stride + 1
        gy = torch.randn(n, cout, ho, ho, generator=g)
        with torch.no_grad():
            m.weight.copy_(torch.randn(cout, cin, ks, ks, generator=g) * 0.05); m.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        y.backward(gy)
        for k, v in (("x", x), ("w", m.weight.detach()), ("b", m.bias.detach()), ("gy", gy), ("out", y.detach()), ("dx", xr.grad),
                     ("dw", m.weight.grad), ("db", m.bias.grad)):
            out["%s_%s" % (name, k)] = v.numpy()
        xo = x.clone().requires_grad_(True)
        wo = m.weight.detach().clone().requires_grad_(True); bo = m.bias.detach().clone().requires_grad_(True)
        yo = M.normed_conv2d(xo, wo, bo, 20, 1.0, 1e-6, nok, stride, pad)
        yo.backward(gy)
        close(yo, y, 1e-6, name + " out"); close(xo.grad, xr.grad, 1e-6, name + " dx"); close(wo.grad, m.weight.grad, 1e-6, name + " dw")
    return out


# --------------------------------------------------------------------------- G13
def g13_fasa():
    out = {}
    nc, c1, n = 1203, 1204, K.FASA_N
    with _cuda_is_cpu():
        crit = REF["fiif"].FasaIIFLoss(num_classes=nc, path=LVIS_CSV, variant="raw", loss_weight=1.5, use_cums=True)
    assert crit.reduction == "none" and crit.reduction_old == "mean"
    cl, cn = torch.zeros(c1), torch.zeros(c1)
    losses = []
    for step in range(K.FASA_STEPS):
        score, label, weight = head_inputs(n, c1, K.FASA_SEED0 + step)
        af = max(float((weight > 0).sum().item()), 1.0)
        s = score.clone().requires_grad_(True)
        with _cuda_is_cpu():
            loss = crit(s, label, weight, avg_factor=af)
        loss.backward()
        losses.append(float(loss))
        out["step%d_score_sum" % step] = np.array(float(score.double().sum()))
        out["step%d_label" % step] = label.numpy()
        out["step%d_weight" % step] = weight.numpy()
        out["step%d_drows" % step] = s.grad[::ROW_STEP].numpy()
        rows = 1.5 * M.iif_cross_entropy(score, label, crit.iif_weights, weight=weight, reduction="none")
        close(M.fasa_accumulate(rows, label, cl, cn), loss, 1e-6, "fasa mean")
    close(cl, crit.cum_losses.detach(), 1e-6, "cum_losses"); assert torch.equal(cn, crit.cum_labels)
    out["losses"] = np.array(losses)
    out["cum_losses"] = crit.cum_losses.detach().numpy().copy()     # the reference accumulates with grad attached
    out["cum_labels"] = crit.cum_labels.numpy().copy()
    with _cuda_is_cpu():
        crit.close_cums()
    out["closed_reduction"] = np.array(crit.reduction)
    score, label, weight = head_inputs(8, c1, 399)
    out["closed_loss"] = crit(score, label).detach().numpy()

    # ---- feature bank of ConvFCFASABBoxHead (fasa_bbox_head.py:35-66,118-215), D = 64
    d = 64
    cfg = dict(decay_ratio=0.1, instance_prob_scale=1500.0, instance_prob_power=1)
    with _cuda_is_cpu():
        head = REF["fh"].ConvFCFASABBoxHead(num_classes=nc, cls_last_dim=d, fasa_cfg=cfg)
    out["instance_counts"] = head.instance_count_list.numpy().copy()
    out["prob_list0"] = head.prob_list.data.numpy().copy()
    out["bank_cfg"] = np.array([cfg["decay_ratio"], cfg["instance_prob_scale"], cfg["instance_prob_power"], 1.1, 0.9])
    fm, fv, fu = torch.zeros(nc, d), torch.zeros(nc, d), torch.zeros(nc)
    g = torch.Generator().manual_seed(77)
    for step in range(3):
        k = 200
        emb = torch.randn(k, d, generator=g) * 1.5 + 0.3
        lab = torch.randint(0, 60, (k,), generator=g) * 20              # classes 0, 20, ..., 1180: several rows each
        if step == 2:
            lab[:3] = torch.tensor([7, 9, 11])                          # single-row classes (n = 1: no Bessel factor)
        with _cuda_is_cpu():
            head.fa_update(emb, lab)
        M.fasa_update(emb, lab, fm, fv, fu, cfg["decay_ratio"])
        out["bank_step%d_emb" % step] = emb.numpy()
        out["bank_step%d_lab" % step] = lab.numpy()
    close(fm, head.feature_mean.data, 1e-6, "feature_mean"); close(fv, head.feature_std.data, 1e-6, "feature_var")
    assert torch.equal(fu, head.feature_used.data)
    used = torch.nonzero(head.feature_used.data > 0).reshape(-1)
    out["bank_used_idx"] = used.numpy()
    out["bank_mean_used"] = head.feature_mean.data[used].numpy()
    out["bank_var_used"] = head.feature_std.data[used].numpy()
    # fa_generate: replay its draw order (one torch.rand(C), then one torch.normal per selected+used class) from a seed
    torch.manual_seed(1234)
    with _cuda_is_cpu():
        e, l = head.fa_generate()
    torch.manual_seed(1234)
    rand = torch.rand(nc)
    normal = torch.zeros(nc, d)
    for c in torch.where(rand < head.prob_list.data)[0]:
        if head.feature_used.data[int(c)] == 0:
            continue
        normal[int(c)] = torch.normal(0, 1, size=(d,))
    out["gen_rand"] = rand.numpy(); out["gen_normal_rows"] = normal[l].numpy() if len(l) else np.zeros((0, d), np.float32)
    out["gen_labels"] = l.numpy(); out["gen_emb"] = e.numpy()
    assert len(l) >= 3, "too few generated rows to pin anything (%d)" % len(l)
    me, ml = M.fasa_generate(rand, head.prob_list.data, fu, fm, fv, normal)
    assert ml.tolist() == l.tolist(); close(me, e, 1e-6, "fa_generate")
    # dynamic_sampling (:174-215): two calls in eval mode with accumulators that rise for some classes
    head.eval()
    lc = types.SimpleNamespace(cum_labels=torch.full((c1,), 10.0), cum_losses=torch.linspace(0.5, 3.0, c1))
    head.loss_cls = lc
    with _cuda_is_cpu():
        head.dynamic_sampling()
    out["dyn_prob1"] = head.prob_list.data.numpy().copy()
    out["dyn_groups1"] = np.array([len(grp) for grp in head.group_cluster_list])
    lc.cum_losses = lc.cum_losses * torch.where(torch.arange(c1) % 2 == 0, 1.3, 0.8)
    with _cuda_is_cpu():
        head.dynamic_sampling()
    out["dyn_prob2"] = head.prob_list.data.numpy().copy()
    out["dyn_t0"] = head.cum_loss_perclass_t0.detach().numpy().copy()
    labels_flat = np.full(nc, -1, dtype=np.int64)
    for gi, grp in enumerate(head.group_cluster_list):
        labels_flat[grp] = gi
    out["dyn_group_of_class"] = labels_flat
    return out


# --------------------------------------------------------------------------- G15
def g15_mask():
    """mask_cross_entropy (cross_entropy_loss.py:112-162) and the channel pick of fcn_mask_head.py:289-290."""
    out = {}
    for name, n, c, hw, scale, seed in K.MASK_CASES:
        pred, target, label = K.mask_inputs(n, c, hw, scale, seed)
        p = pred.clone().requires_grad_(True)
        loss = REF["ce"].mask_cross_entropy(p, target, label)
        assert tuple(loss.shape) == (1,)
        (loss * 2.5).sum().backward()
        sel = p.grad[torch.arange(n), label]
        assert float(p.grad.abs().sum()) == float(sel.abs().sum())             # only the picked channels carry gradient
        out[name + "_shape"] = np.array([n, c, hw, seed]); out[name + "_scale"] = np.array(scale)
        out[name + "_pred_sum"] = np.array(float(pred.double().sum()))
        out[name + "_target"] = target.numpy(); out[name + "_label"] = label.numpy()
        out[name + "_loss"] = loss.detach().numpy(); out[name + "_dsel"] = sel.numpy()
        out[name + "_picked"] = pred[range(n), label].numpy()                  # fcn_mask_head.py:289-290
        po = pred.clone().requires_grad_(True)
        mine = M.mask_cross_entropy(po, target, label)
        (mine * 2.5).sum().backward()
        close(mine, loss, 1e-6, name + " mask loss"); close(po.grad, p.grad, 1e-6, name + " mask grad")
        assert torch.equal(M.gather_class_masks(pred, label), pred[range(n), label])
    return out


def main():
    sets = {"g11_mmdet_ce": g11_cross_entropy, "g12_mmdet_normed": g12_normed, "g13_mmdet_fasa": g13_fasa,
            "g14_mmdet_csv": g14_csv_tables, "g15_mmdet_mask": g15_mask}
    only = sys.argv[1:]
    if only:
        sets = {k: v for k, v in sets.items() if k in only}
    for name, fn in sets.items():
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print("%-18s %4d arrays  %8.1f KB" % (name, len(data), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
