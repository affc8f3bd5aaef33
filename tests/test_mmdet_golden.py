"""The mmdet half against vectors produced by RUNNING THE REFERENCE'S OWN FILES
(tests/golden/make_golden_mmdet.py -> g11..g15 + the two CSV tables).

The same assertions run twice: on the CPU oracle (``-m "not gpu"``: pins the restatement) and on the HIP path
(``-m gpu``: IIFLoss / FasaIIFLoss / normed predictors / mask loss of ``iif_amd``).  Tolerance: fp32 loss, gradients
and activations within 1e-4 relative of the reference (BASELINE.json north_star; measured ~1e-6); labels, counts,
channel picks, table values and accuracies bit-exact.
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import mmdet_iif as M
from tests import mmdet_cases as K
from tests.conftest import GOLDEN

REL = 1e-4
DEV = "cuda:0"
VARIANTS14 = ("smooth", "raw", "prob", "normit", "gombit", "base2", "base10", "smooth_obj", "raw_obj", "prob_obj",
              "normit_obj", "gombit_obj", "base2_obj", "base10_obj")
BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")]


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def csv_path(relpath):
    return os.path.join(GOLDEN, relpath)


# ------------------------------------------------------------------ G14: tables
@pytest.mark.parametrize("tag,relpath,nc", [("lvis", K.LVIS[1], 1203), ("coco", K.COCO[1], 80)])
def test_g14_csv_tables_bit_exact(golden, tag, relpath, nc):
    """iif_loss.py:47-50 on the reference's own tables: oracle reader and the product's host reader."""
    from iif_amd.mmdet_iif_loss import read_iif_csv
    g = golden("g14_mmdet_csv")
    for v in VARIANTS14:
        want = torch.from_numpy(g["%s_%s" % (tag, v)])
        assert tuple(want.shape) == (1, nc + 1) and want[0, -1].item() == 1.0
        assert torch.equal(M.read_table(csv_path(relpath), v), want), v
        assert torch.equal(read_iif_csv(csv_path(relpath), v), want), v
    with pytest.raises(KeyError):
        read_iif_csv(csv_path(relpath), "no_such_variant")


@pytest.mark.gpu
def test_g14_plugin_tables_on_device(golden):
    from iif_amd.mmdet_iif_loss import IIFLoss
    from iif_amd.mmdet_normed_predictor import IIFNormedLinear
    g = golden("g14_mmdet_csv")
    for v in ("raw", "base10_obj"):
        crit = IIFLoss(num_classes=1203, path=csv_path(K.LVIS[1]), variant=v)
        assert crit.iif_weights.is_cuda and torch.equal(crit.iif_weights.cpu(), torch.from_numpy(g["lvis_" + v]))
    m = IIFNormedLinear(8, 1204, path=csv_path(K.LVIS[1]))                       # default variant base2_obj
    assert torch.equal(m.iif_weights.cpu(), torch.from_numpy(g["lvis_normed_default"]))


# ----------------------------------------------------------- G11: cross entropy
def _ce_backend(backend, nc, path, variant, ckw):
    if backend == "hip":
        from iif_amd.mmdet_iif_loss import IIFLoss
        crit = IIFLoss(num_classes=nc, path=path, variant=variant, **ckw)
        return crit, DEV
    table = M.read_table(path, variant)

    def crit(s, lab, weight=None, avg_factor=None, reduction_override=None, ignore_index=None):
        red = reduction_override or ckw.get("reduction", "mean")
        ign = ignore_index if ignore_index is not None else ckw.get("ignore_index")
        cw = None if "class_weight" not in ckw else torch.tensor(ckw["class_weight"])
        return M.iif_cross_entropy(s, lab, table, weight=weight, reduction=red, avg_factor=avg_factor, class_weight=cw,
                                   ignore_index=ign, loss_weight=ckw.get("loss_weight", 1.0))
    return crit, "cpu"


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [K.LVIS, K.COCO], ids=["lvis1024x1204", "coco64x81"])
def test_g11_cross_entropy(golden, backend, case):
    """iif_loss.py:109-202 + losses/utils.py:29-55: row weights, avg_factor, ignore_index (default and custom, from
    the constructor and from the call), class_weight, loss_weight, every reduction."""
    g = golden("g11_mmdet_ce")
    tag, relpath, nc, n, seed = case
    c1 = nc + 1
    score, label, weight = K.head_inputs(n, c1, seed)
    assert abs(float(score.double().sum()) - float(g[tag + "_score_sum"])) < 1e-9          # same draw as the generator
    assert torch.equal(label, torch.from_numpy(g[tag + "_label"])) and torch.equal(weight, torch.from_numpy(g[tag + "_weight"]))
    af = float(g[tag + "_avg_factor"])
    cw = K.class_weight_list(c1, seed)
    lab_ign, lab_ign7 = K.ignore_labels(label)
    cases = K.ce_cases(label, weight, af, cw, lab_ign, lab_ign7)
    for variant in K.ce_variants(tag):
        for name, (ckw, fkw, lab) in cases.items():
            key = "%s_%s_%s" % (tag, variant, name)
            if key + "_loss" not in g.files:
                continue
            crit, dev = _ce_backend(backend, nc, csv_path(relpath), variant, ckw)
            s = score.clone().to(dev).requires_grad_(True)
            kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in fkw.items()}
            loss = crit(s, lab.to(dev), **kw)
            (loss.sum() if loss.dim() else loss).backward()
            assert rel(loss, g[key + "_loss"]) <= REL, key
            d = s.grad.cpu()
            assert rel(d[::K.ROW_STEP], g[key + "_drows"]) <= REL, key
            assert rel(d.double().sum(0), g[key + "_dcolsum"]) <= REL, key
            assert rel(d.double().abs().sum(1), g[key + "_drowabs"]) <= REL, key
            if "ign" in name:                               # ignored rows carry neither loss nor gradient
                skip = (lab == (-100 if name == "ign" else 7))
                assert skip.any() and d[skip].abs().max().item() == 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [K.LVIS, K.COCO], ids=["lvis", "coco"])
def test_g11_activation_accuracy_protocol(golden, backend, case):
    """get_activation (iif_loss.py:65-78), get_accuracy on the RAW score (:92-107, accuracy.py:7-51), get_cls_channels,
    and the sum + avg_factor error convention (losses/utils.py:53-54)."""
    g = golden("g11_mmdet_ce")
    tag, relpath, nc, n, seed = case
    score, label, weight = K.head_inputs(n, nc + 1, seed)
    boosted = K.boosted_score(score, label, seed)
    for variant in K.ce_variants(tag):
        if backend == "hip":
            from iif_amd.mmdet_iif_loss import IIFLoss
            crit = IIFLoss(num_classes=nc, path=csv_path(relpath), variant=variant)
            act = crit.get_activation(score.to(DEV)).cpu()
        else:
            act = M.get_activation(score, M.read_table(csv_path(relpath), variant))
        assert rel(act[::K.ROW_STEP], g["%s_%s_act_rows" % (tag, variant)]) <= REL
        assert rel(act.double().sum(1), g["%s_%s_act_rowsum" % (tag, variant)]) <= REL
        assert rel(act.double().sum(0), g["%s_%s_act_colsum" % (tag, variant)]) <= REL
    if backend == "hip":
        acc = crit.get_accuracy(boosted.to(DEV), label.to(DEV))["acc_classes"].cpu()
        empty = crit.get_accuracy(boosted[:0].to(DEV), label[:0].to(DEV))["acc_classes"]
        assert crit.get_cls_channels(nc) == int(g[tag + "_cls_channels"])
        with pytest.raises(AssertionError):
            crit.get_cls_channels(nc + 1)
        assert int(g[tag + "_sum_avg_factor_raises"]) == 1
        with pytest.raises(ValueError):
            crit(score.to(DEV), label.to(DEV), avg_factor=3.0, reduction_override="sum")
        from iif_amd.utils import topk_hit_counts
        hits = topk_hit_counts(boosted.to(DEV), label.to(DEV), (1, 5)).cpu().double() * (100.0 / n)
        assert hits.float().tolist() == g[tag + "_acc_top1_top5"].tolist()                # accuracy.py:46-50, topk=(1,5)
    else:
        acc = M.accuracy_top1(boosted, label)
        empty = M.accuracy_top1(boosted[:0], label[:0])
        with pytest.raises(ValueError):
            M.iif_cross_entropy(score, label, M.read_table(csv_path(relpath), "raw"), avg_factor=3.0, reduction="sum")
    assert tuple(acc.shape) == (1,) and acc.tolist() == g[tag + "_acc_classes"].tolist()     # bit-exact percentage
    assert tuple(empty.shape) == () and float(empty) == float(g[tag + "_acc_empty"]) == 0.0


@pytest.mark.gpu
def test_g11_known_answers(golden, tmp_path):
    """The reference's CE known answers (instance_segmentation/tests/test_metrics/test_losses.py:8-32) as IIFLoss
    with a table of ones produced them."""
    from iif_amd.mmdet_iif_loss import IIFLoss
    g = golden("g11_mmdet_ce")
    p = tmp_path / "ones.csv"
    p.write_text("raw\n1\n1.0\n")
    x = torch.tensor([[100.0, -100.0]], device=DEV); y = torch.tensor([1], device=DEV)
    assert IIFLoss(num_classes=1, path=str(p))(x, y).item() == float(g["known_ce"]) == 200.0
    assert abs(IIFLoss(num_classes=1, path=str(p), class_weight=[0.8, 0.2])(x, y).item() - float(g["known_ce_cw"])) <= 1e-5


# --------------------------------------------------------- G12: normed predictors
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", K.NORMED_LINEAR_CASES, ids=[c[0] for c in K.NORMED_LINEAR_CASES])
def test_g12_normed_linear(golden, backend, case):
    """normed_predictor.py:34-40 (NormedLinear) and :67-73 (IIFNormedLinear, real CSV column): output and all gradients."""
    g = golden("g12_mmdet_normed")
    name, n, d, c, temp, power, variant = case
    x, w, b, gy = (torch.from_numpy(g["%s_%s" % (name, k)]) for k in ("x", "w", "b", "gy"))
    relcsv = K.LVIS[1] if c == 1204 else K.COCO[1]
    if backend == "hip":
        from iif_amd.mmdet_normed_predictor import IIFNormedLinear, NormedLinear
        if variant is None:
            m = NormedLinear(d, c, tempearture=temp, power=power).to(DEV)
        else:
            m = IIFNormedLinear(d, c, tempearture=temp, power=power, variant=variant, path=csv_path(relcsv)).to(DEV)
            assert torch.equal(m.iif_weights.reshape(-1).cpu(), torch.from_numpy(g[name + "_rows"]))
        assert abs(m.weight.std().item() - 0.01) < 0.004 and m.bias.abs().max().item() == 0      # init law :29-32
        with torch.no_grad():
            m.weight.copy_(w); m.bias.copy_(b)
        xr = x.to(DEV).requires_grad_(True)
        y = m(xr)
        y.backward(gy.to(DEV))
        got = dict(out=y, dx=xr.grad, dw=m.weight.grad, db=m.bias.grad)
    else:
        rows = None if variant is None else M.read_table(csv_path(relcsv), variant).reshape(-1)
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = M.normed_linear(xr, wr, br, temp, power, 1e-6, rows)
        y.backward(gy)
        got = dict(out=y, dx=xr.grad, dw=wr.grad, db=br.grad)
    for k, v in got.items():
        assert rel(v, g["%s_%s" % (name, k)]) <= REL, (name, k)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", K.NORMED_CONV_CASES, ids=[c[0] for c in K.NORMED_CONV_CASES])
def test_g12_normed_conv2d(golden, backend, case):
    """normed_predictor.py:104-124: the 1x1 predictor (norm_over_kernel coincides for a 1x1 kernel) and, since round 3,
    k x k kernels with stride / padding and both filter normalisations."""
    g = golden("g12_mmdet_normed")
    name, n, cin, cout, hw, nok, ks, stride, pad = case
    x, w, b, gy = (torch.from_numpy(g["%s_%s" % (name, k)]) for k in ("x", "w", "b", "gy"))
    if backend == "hip":
        from iif_amd.mmdet_normed_predictor import NormedConv2d
        m = NormedConv2d(cin, cout, ks, stride=stride, padding=pad, tempearture=20, norm_over_kernel=nok).to(DEV)
        with torch.no_grad():
            m.weight.copy_(w); m.bias.copy_(b)
        xr = x.to(DEV).requires_grad_(True)
        y = m(xr)
        y.backward(gy.to(DEV))
        got = dict(out=y, dx=xr.grad, dw=m.weight.grad, db=m.bias.grad)
    else:
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = M.normed_conv2d(xr, wr, br, 20, 1.0, 1e-6, nok, stride, pad)
        y.backward(gy)
        got = dict(out=y, dx=xr.grad, dw=wr.grad, db=br.grad)
    for k, v in got.items():
        assert rel(v, g["%s_%s" % (name, k)]) <= REL, (name, k)


# ------------------------------------------------------------------- G13: FASA
@pytest.mark.parametrize("backend", BACKENDS)
def test_g13_fasa_loss_accumulators(golden, backend):
    """fasa_iif_loss.py:60-71,116-162: per-class loss / label accumulators over three calls, then close_cums."""
    g = golden("g13_mmdet_fasa")
    nc, c1, n = 1203, 1204, K.FASA_N
    path = csv_path(K.LVIS[1])
    if backend == "hip":
        from iif_amd.mmdet_fasa import FasaIIFLoss
        crit = FasaIIFLoss(num_classes=nc, path=path, variant="raw", loss_weight=1.5, use_cums=True)
        assert crit.reduction == "none" and crit.reduction_old == "mean"
    else:
        table = M.read_table(path, "raw")
        cl, cn = torch.zeros(c1), torch.zeros(c1)
    losses = []
    for step in range(K.FASA_STEPS):
        score, label, weight = K.head_inputs(n, c1, K.FASA_SEED0 + step)
        assert abs(float(score.double().sum()) - float(g["step%d_score_sum" % step])) < 1e-9
        assert torch.equal(label, torch.from_numpy(g["step%d_label" % step]))
        af = max(float((weight > 0).sum().item()), 1.0)
        if backend == "hip":
            s = score.to(DEV).requires_grad_(True)
            loss = crit(s, label.to(DEV), weight.to(DEV), avg_factor=af)
        else:
            s = score.clone().requires_grad_(True)
            rows = 1.5 * M.iif_cross_entropy(s, label, table, weight=weight, reduction="none")
            loss = M.fasa_accumulate(rows, label, cl, cn)
        loss.backward()
        losses.append(float(loss))
        assert rel(s.grad.cpu()[::K.ROW_STEP], g["step%d_drows" % step]) <= REL
    assert rel(torch.tensor(losses), g["losses"]) <= REL
    cum_l, cum_n = (crit.cum_losses.cpu(), crit.cum_labels.cpu()) if backend == "hip" else (cl.detach(), cn)
    assert torch.equal(cum_n, torch.from_numpy(g["cum_labels"]))                     # counts: exact
    assert rel(cum_l, g["cum_losses"]) <= REL
    if backend == "hip":
        crit.close_cums()
        assert crit.reduction == str(g["closed_reduction"]) == "mean" and crit.cum_labels.abs().sum().item() == 0
        score, label, _ = K.head_inputs(8, c1, 399)
        assert rel(crit(score.to(DEV), label.to(DEV)), g["closed_loss"]) <= REL


def _bank(golden, device):
    from iif_amd.mmdet_fasa import FasaFeatureBank
    g = golden("g13_mmdet_fasa")
    decay, scale, power, up, down = g["bank_cfg"].tolist()
    bank = FasaFeatureBank(1203, 64, g["instance_counts"], dict(decay_ratio=decay, instance_prob_scale=scale,
                                                              instance_prob_power=power), device=device)
    assert bank.dynamic_up == up and bank.dynamic_down == down                        # fasa_bbox_head.py:47-48 defaults
    return g, bank


@pytest.mark.gpu
def test_g13_feature_bank_update_generate(golden):
    """fasa_bbox_head.py:50-59 (sampling probabilities from LVIS_INSTANCES), :118-147 (fa_update), :149-172 (fa_generate)
    against the reference's ConvFCFASABBoxHead."""
    g, bank = _bank(golden, DEV)
    assert rel(bank.prob_list.data, g["prob_list0"]) <= 1e-6
    for step in range(3):
        bank.fa_update(torch.from_numpy(g["bank_step%d_emb" % step]).to(DEV), torch.from_numpy(g["bank_step%d_lab" % step]).to(DEV))
    used = torch.from_numpy(g["bank_used_idx"])
    fu = torch.zeros(1203); fu[used] = 1
    assert torch.equal(bank.feature_used.cpu(), fu)
    assert rel(bank.feature_mean.data.cpu()[used], g["bank_mean_used"]) <= REL
    assert rel(bank.feature_std.data.cpu()[used], g["bank_var_used"]) <= REL
    rest = torch.ones(1203, dtype=torch.bool); rest[used] = False
    assert bank.feature_mean.data.cpu()[rest].abs().max().item() == 0
    labels = torch.from_numpy(g["gen_labels"])
    normal = torch.zeros(1203, 64)
    normal[labels] = torch.from_numpy(g["gen_normal_rows"])
    e, l = bank.fa_generate(torch.from_numpy(g["gen_rand"]).to(DEV), normal.to(DEV))
    assert l.cpu().tolist() == labels.tolist()
    assert rel(e, g["gen_emb"]) <= REL


def test_g13_oracle_feature_bank(golden):
    g = golden("g13_mmdet_fasa")
    fm, fv, fu = torch.zeros(1203, 64), torch.zeros(1203, 64), torch.zeros(1203)
    for step in range(3):
        M.fasa_update(torch.from_numpy(g["bank_step%d_emb" % step]), torch.from_numpy(g["bank_step%d_lab" % step]), fm, fv, fu,
                      float(g["bank_cfg"][0]))
    used = torch.from_numpy(g["bank_used_idx"])
    assert torch.nonzero(fu).reshape(-1).tolist() == used.tolist()
    assert rel(fm[used], g["bank_mean_used"]) <= 1e-6 and rel(fv[used], g["bank_var_used"]) <= 1e-6
    labels = torch.from_numpy(g["gen_labels"])
    normal = torch.zeros(1203, 64); normal[labels] = torch.from_numpy(g["gen_normal_rows"])
    e, l = M.fasa_generate(torch.from_numpy(g["gen_rand"]), torch.from_numpy(g["prob_list0"]), fu, fm, fv, normal)
    assert l.tolist() == labels.tolist() and rel(e, g["gen_emb"]) <= 1e-6


def test_g13_dynamic_sampling_host(golden):
    """fasa_bbox_head.py:174-215 is host logic in the product as well (AffinityPropagation over the class means): runs on
    CPU tensors here, two calls, against the reference's probabilities and clustering."""
    g, bank = _bank(golden, "cpu")
    assert rel(bank.prob_list.data, g["prob_list0"]) <= 1e-6
    used = torch.from_numpy(g["bank_used_idx"])
    bank.feature_mean.data[used] = torch.from_numpy(g["bank_mean_used"])
    c1 = 1204
    lc = types.SimpleNamespace(cum_labels=torch.full((c1,), 10.0), cum_losses=torch.linspace(0.5, 3.0, c1))
    bank.dynamic_sampling(lc, training=True)                                            # training mode: no-op (:175-176)
    assert rel(bank.prob_list.data, g["prob_list0"]) <= 1e-6
    bank.dynamic_sampling(lc)
    assert rel(bank.prob_list.data, g["dyn_prob1"]) <= 1e-6
    assert [len(grp) for grp in bank.group_cluster_list] == g["dyn_groups1"].tolist()
    lc.cum_losses = lc.cum_losses * torch.where(torch.arange(c1) % 2 == 0, 1.3, 0.8)
    bank.dynamic_sampling(lc)
    assert rel(bank.prob_list.data, g["dyn_prob2"]) <= 1e-6
    assert rel(bank.cum_loss_perclass_t0, g["dyn_t0"]) <= 1e-6
    flat = np.full(1203, -1, dtype=np.int64)
    for gi, grp in enumerate(bank.group_cluster_list):
        flat[grp] = gi
    assert flat.tolist() == g["dyn_group_of_class"].tolist()


# ------------------------------------------------------------------- G15: masks
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", K.MASK_CASES, ids=[c[0] for c in K.MASK_CASES])
def test_g15_mask_loss_and_channel_pick(golden, backend, case):
    """mask_cross_entropy (cross_entropy_loss.py:112-162) and mask_pred[range(N), labels] (fcn_mask_head.py:289-290)."""
    g = golden("g15_mmdet_mask")
    name, n, c, hw, scale, seed = case
    pred, target, label = K.mask_inputs(n, c, hw, scale, seed)
    assert abs(float(pred.double().sum()) - float(g[name + "_pred_sum"])) < 1e-6 * max(1.0, scale)
    assert torch.equal(label, torch.from_numpy(g[name + "_label"])) and torch.equal(target, torch.from_numpy(g[name + "_target"]))
    if backend == "hip":
        from iif_amd.mmdet_mask_loss import gather_class_masks, mask_cross_entropy
        p = pred.to(DEV).requires_grad_(True)
        loss = mask_cross_entropy(p, target.to(DEV), label.to(DEV))
        picked = gather_class_masks(pred.to(DEV), label.to(DEV)).cpu()
    else:
        p = pred.clone().requires_grad_(True)
        loss = M.mask_cross_entropy(p, target, label)
        picked = M.gather_class_masks(pred, label)
    assert tuple(loss.shape) == (1,)
    (loss * 2.5).sum().backward()
    assert torch.equal(picked, torch.from_numpy(g[name + "_picked"]))                 # bit-exact channel pick
    assert rel(loss, g[name + "_loss"]) <= REL
    d = p.grad.cpu()
    assert rel(d[torch.arange(n), label], g[name + "_dsel"]) <= REL
    sel = torch.zeros(n, c, dtype=torch.bool); sel[torch.arange(n), label] = True
    assert d[~sel].abs().max().item() == 0
