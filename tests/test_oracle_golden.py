"""The CPU oracle against the golden vectors taken from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import iif_oracle as O
from oracle import mmdet_iif as M
from oracle import resnet_oracle as R

COUNT_NAMES = ("c4", "cifar100_exp100", "places365", "imagenet1000", "lvis1204")


def test_g1_class_counts_bit_exact(golden):
    g = golden("g1_class_counts")
    for key in g.files:
        c, imb_type, imb = key.split("_")
        got = O.img_num_per_cls(int(c[1:]), 50000, imb_type, float(imb))
        assert got == g[key].tolist(), key
    assert sum(g["c100_exp_0.01"].tolist()) == 10847


@pytest.mark.parametrize("case", ["distinct8", "ties12", "lt200", "ties40"])
def test_g2_class_map(golden, case):
    g = golden("g2_class_map")
    C = len(g[case + "_class_map"])
    cmap, tgt, cnl = O.lt_class_map(g[case + "_labels"], C, kind=None)
    assert cmap == g[case + "_class_map"].tolist()
    assert tgt == g[case + "_targets"].tolist()
    assert cnl == g[case + "_cls_num_list"].tolist()
    # documented tie rule (stable): identical count profile, identical map when no ties
    s_cmap, _, s_cnl = O.lt_class_map(g[case + "_labels"], C, kind="stable")
    assert s_cnl == cnl
    assert sorted(s_cmap) == list(range(C))
    if case in ("distinct8", "lt200"):
        assert s_cmap == cmap


@pytest.mark.parametrize("name", COUNT_NAMES)
def test_g3_tables_bit_exact(golden, name):
    g = golden("g3_tables")
    counts = g[name + "_counts"].tolist()
    for norm in (0, 1, 2):
        mine = O.iif_tables(counts, iif_norm=norm)
        for v in O.VARIANTS:
            ref = torch.from_numpy(g["%s_n%d_%s" % (name, norm, v)])
            assert mine[v].dtype == torch.float32 and mine[v].shape == ref.shape
            assert torch.equal(mine[v], ref), (name, norm, v)


@pytest.mark.parametrize("name", COUNT_NAMES)
def test_g4_loss_and_grad(golden, name):
    g, t = golden("g4_loss"), golden("g3_tables")
    counts = t[name + "_counts"].tolist()
    pred = torch.from_numpy(g[name + "_pred"])
    tgt = torch.from_numpy(g[name + "_targets"])
    cw = torch.from_numpy(g[name + "_class_weight"])
    assert torch.equal(cw, O.deferred_class_weight(counts))
    tables = O.iif_tables(counts)
    variants = O.VARIANTS if len(counts) <= 1000 else ("raw", "smooth", "base10")
    for v in variants:
        for red in ("mean", "sum"):
            for wname, w in (("nw", None), ("cw", cw)):
                key = "%s_%s_%s_%s" % (name, v, red, wname)
                p = pred.clone().requires_grad_(True)
                loss = O.iif_ce(p, tgt, tables[v], w, red)
                loss.backward()
                ref_l, ref_d = float(g[key + "_loss"]), torch.from_numpy(g[key + "_dpred"])
                assert abs(loss.item() - ref_l) <= 1e-6 * max(1.0, abs(ref_l)), key
                assert (p.grad - ref_d).abs().max().item() <= 1e-6 * max(1.0, ref_d.abs().max().item()), key
                l64, d64, _ = O.iif_ce_closed_form(pred, tgt, tables[v], w, red)
                assert abs(l64.item() - ref_l) <= 2e-6 * max(1.0, abs(ref_l)), key
                assert (d64 - ref_d.double()).abs().max().item() <= 2e-6 * max(1.0, ref_d.abs().max().item()), key
        scaled = O.iif_infer(pred, tables[v])
        a1, a5 = O.accuracy(scaled, tgt, topk=(1, min(5, len(counts))))
        assert [a1.item(), a5.item()] == g["%s_%s_infer_acc" % (name, v)].tolist()
    r1, r5 = O.accuracy(pred, tgt, topk=(1, min(5, len(counts))))
    assert [r1.item(), r5.item()] == g[name + "_rawacc"].tolist()
    perm = torch.from_numpy(g[name + "_perm"])
    p = pred.clone().requires_grad_(True)
    ml = O.mixup_criterion(p, tgt, tgt[perm], 0.3, tables["raw"])
    ml.backward()
    assert abs(ml.item() - float(g[name + "_mixup_loss"])) <= 1e-6 * max(1.0, abs(ml.item()))
    ref_d = torch.from_numpy(g[name + "_mixup_dpred"])
    assert (p.grad - ref_d).abs().max().item() <= 1e-6 * max(1.0, ref_d.abs().max().item())


def _checksum(sd):
    return np.array([float(v.double().sum()) for v in sd.values() if v.is_floating_point()])


@pytest.mark.parametrize("fixture,prefix,arch,C,cname,B,hw", [
    ("g7_nets", "resnet32", "resnet32", 100, "cifar100_exp100", 8, 32),
    ("g7_nets", "resnet50", "resnet50", 1000, "imagenet1000", 2, 64),
    ("g7_nets", "resnext50", "resnext50_32x4d", 365, "places365", 2, 64),
    ("g10_se", "se_resnet32", "se_resnet32", 100, "cifar100_exp100", 8, 32),
    ("g10_se", "se_resnet50", "se_resnet50", 1000, "imagenet1000", 2, 64),
])
def test_g7_train_steps(golden, fixture, prefix, arch, C, cname, B, hw):
    """Forward, IIF loss, backward and SGD(+warm-up) of the oracle reproduce the
    reference model's logits, gradient norms, loss sequence and final weights
    (g10: the squeeze-and-excitation variants)."""
    g, t = golden(fixture), golden("g3_tables")
    counts = t[cname + "_counts"].tolist()
    sd = (R.init_cifar if arch in R.CIFAR_ARCHS else R.init_imagenet)(arch, C, seed=7)
    if not np.allclose(_checksum(sd), g[prefix + "_init_checksum"], rtol=0, atol=1e-9):
        pytest.skip("torch CPU RNG stream differs from the build container; seeded init not reproducible")
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B, 3, hw, hw, generator=gen)
    prior = torch.tensor(counts, dtype=torch.float64)
    y = torch.multinomial(prior / prior.sum(), B, replacement=True, generator=gen)
    assert abs(float(x.double().sum()) - float(g[prefix + "_x_sum"])) < 1e-9
    assert y.tolist() == g[prefix + "_y"].tolist()
    table = O.iif_tables(counts)["raw"]
    ref_losses = g[prefix + "_losses"]
    lr0 = float(g[prefix + "_lr0"])
    bufs = {}
    for it in range(len(ref_losses)):
        lr = lr0 * O.warmup_factor(it, 1000)
        if it == 0:
            probe = {k: v.clone() for k, v in sd.items()}
            _, logits, grads = R.loss_and_grads(probe, x, y, table, arch)
            ref_logits = torch.from_numpy(g[prefix + "_logits0"])
            assert (logits - ref_logits).abs().max().item() <= 1e-5 * max(1.0, ref_logits.abs().max().item())
            keys = g[prefix + "_gradnorm_keys"].tolist()
            for k, n in zip(keys, g[prefix + "_gradnorm0"]):
                assert abs(float(grads[k].double().norm()) - n) <= 1e-4 * max(n, 1e-3), k
        loss, _ = R.train_step(sd, bufs, x, y, table, arch, lr)
        assert abs(loss.item() - ref_losses[it]) <= 1e-5 * max(1.0, abs(ref_losses[it])), it
    np.testing.assert_allclose(_checksum(sd), g[prefix + "_final_checksum"], rtol=1e-4, atol=2e-3)
    head = "linear.weight" if arch in R.CIFAR_ARCHS else "fc.weight"
    np.testing.assert_allclose(sd[head][:4].numpy(), g[prefix + "_final_fc"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sd["bn1.running_mean"].numpy(), g[prefix + "_final_bn1_rm"], rtol=0, atol=2e-6)


def test_g8_warmup(golden):
    g = golden("g8_warmup")
    for iters in (5, 84, 1000):
        mine = np.array([O.warmup_factor(i, iters) for i in range(iters + 3)])
        assert np.array_equal(mine, g["warmup_%d" % iters])


def test_sgd_matches_torch_optim():
    torch.manual_seed(0)
    for nesterov in (False, True):
        p = [torch.randn(5, 3), torch.randn(7)]
        q = [t.clone().requires_grad_(True) for t in p]
        opt = torch.optim.SGD(q, lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=nesterov)
        bufs = [None, None]
        for _ in range(4):
            grads = [torch.randn_like(t) for t in p]
            for t, gr in zip(q, grads):
                t.grad = gr.clone()
            opt.step()
            O.sgd_step(p, grads, bufs, 0.05, 0.9, 1e-4, nesterov)
        for a, b in zip(p, q):
            assert torch.allclose(a, b.detach(), rtol=0, atol=1e-6)


# ------------------------------------------------------------------ mmdet half
def test_mmdet_ce_known_answers():
    """instance_segmentation/tests/test_metrics/test_losses.py:8-32 with an
    all-ones table (IIF scaling off): 200, 40 (class_weight [0.8,0.2])."""
    ones = torch.ones(1, 2)
    cls = torch.tensor([[100.0, -100.0]])
    lbl = torch.tensor([1])
    assert torch.allclose(M.iif_cross_entropy(cls, lbl, ones), torch.tensor(200.0))
    cw = torch.tensor([0.8, 0.2])
    assert torch.allclose(M.iif_cross_entropy(cls, lbl, ones, class_weight=cw), torch.tensor(40.0))


def test_mmdet_reduces_to_classification_loss(golden):
    g, t = golden("g4_loss"), golden("g3_tables")
    pred = torch.from_numpy(g["lvis1204_pred"])
    tgt = torch.from_numpy(g["lvis1204_targets"])
    table = torch.from_numpy(t["lvis1204_n0_raw"])
    got = M.iif_cross_entropy(pred, tgt, table)
    assert abs(got.item() - float(g["lvis1204_raw_mean_nw_loss"])) <= 1e-6 * max(1.0, abs(got.item()))
    w = torch.linspace(0, 2, pred.shape[0])
    ref = (torch.nn.functional.cross_entropy(pred * table, tgt, reduction="none") * w).sum() / 5.0
    assert torch.allclose(M.iif_cross_entropy(pred, tgt, table, weight=w, avg_factor=5.0), ref)
    with pytest.raises(ValueError):
        M.iif_cross_entropy(pred, tgt, table, reduction="sum", avg_factor=3.0)
    sm = M.get_activation(pred, table)
    assert torch.allclose(sm.sum(-1), torch.ones(pred.shape[0]), atol=1e-5)


def test_mmdet_csv_tables_match_formulas(tmp_path):
    """The CSV data contract (iif_loss.py:47-50): column -> [1, C+1] with a
    trailing 1.0; a synthetic CSV built from the classification formulas."""
    counts = [64, 9, 300, 12]
    tabs = O.iif_tables(counts)
    p = tmp_path / "idf.csv"
    with open(p, "w") as f:
        f.write("raw,smooth\n1,1\n")
        for i in range(len(counts)):
            f.write("%r,%r\n" % (float(np.log(sum(counts) / counts[i])),
                                 float(np.log((sum(counts) + 1) / (counts[i] + 1)) + 1)))
    tb = M.read_table(str(p), "raw")
    assert tb.shape == (1, 5) and tb[0, -1].item() == 1.0
    assert torch.equal(tb[:, :4], tabs["raw"])
    assert torch.equal(M.read_table(str(p), "smooth")[:, :4], tabs["smooth"])


@pytest.mark.parametrize("head", ["cosine", "lr_cosine", "norm"])
def test_g9_classifier_heads(golden, head):
    """Cosine / normed heads of the oracle against vectors from the reference's own forward code."""
    g = golden("g9_heads")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    gy = torch.from_numpy(g["gy"])
    sd = {"linear.weight": torch.from_numpy(g[head + "_weight"]).clone().requires_grad_(True)}
    if head == "lr_cosine":
        sd["linear.scale"] = (5.0 * torch.ones(1)).requires_grad_(True)
    y = R.head_forward(sd, "linear", x, head)
    y.backward(gy)
    for got, key in ((y, "_logits"), (x.grad, "_dx"), (sd["linear.weight"].grad, "_dw")):
        ref = torch.from_numpy(g[head + key])
        assert (got.detach() - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item()), (head, key)
    if head == "lr_cosine":
        assert abs(sd["linear.scale"].grad.item() - float(g[head + "_dscale"])) <= 1e-5 * abs(float(g[head + "_dscale"]))
