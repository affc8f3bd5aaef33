"""Inputs and case tables shared by tests/golden/make_golden_mmdet.py (which runs the reference on them) and the
tests that replay them through the oracle and the HIP path.  Inputs are regenerated from seeds (torch CPU
generator); the fixtures hold checksums of the draws next to the reference's outputs."""
import torch

ROW_STEP = 64          # gradient / activation rows stored in full: every 64th; the rest is pinned by row / column sums
LVIS = ("lvis", "lvis_files/idf_1204.csv", 1203, 1024, 11)
COCO = ("coco", "coco_files/idf_91.csv", 80, 64, 12)


def head_inputs(n, c1, seed):
    """cls_score ~ 2*N(0,1) [n, c1], labels 75 % background (= c1-1) / 25 % foreground, label weights with
    zeros (SURVEY 8d synthetic head inputs)."""
    g = torch.Generator().manual_seed(seed)
    score = torch.randn(n, c1, generator=g) * 2.0
    fg = torch.randint(0, c1 - 1, (n,), generator=g)
    label = torch.where(torch.rand(n, generator=g) < 0.75, torch.full((n,), c1 - 1), fg)
    weight = (torch.rand(n, generator=g) > 0.1).float() * (0.5 + torch.rand(n, generator=g))
    return score, label, weight


def class_weight_list(c1, seed):
    return (0.5 + torch.rand(c1, generator=torch.Generator().manual_seed(seed + 100))).tolist()


def ignore_labels(label):
    a = label.clone(); a[::7] = -100
    b = label.clone(); b[::5] = 7
    return a, b


def ce_cases(label, weight, af, cw, lab_ign, lab_ign7):
    """name -> (constructor kwargs, forward kwargs, labels) for IIFLoss (iif_loss.py:15-23,109-152)."""
    return {
        "plain": (dict(), dict(), label),
        "head": (dict(), dict(weight=weight, avg_factor=af), label),                              # the bbox head's call
        "none_w": (dict(), dict(weight=weight, reduction_override="none"), label),
        "sum_w": (dict(reduction="sum"), dict(weight=weight), label),
        "none_af": (dict(), dict(weight=weight, avg_factor=af, reduction_override="none"), label),
        "ign": (dict(), dict(weight=weight, avg_factor=af), lab_ign),                             # default ignore_index -100
        "ign7": (dict(ignore_index=7), dict(weight=weight), lab_ign7),
        "ign7_call": (dict(), dict(ignore_index=7), lab_ign7),
        "cw_lw": (dict(class_weight=cw, loss_weight=0.5), dict(weight=weight, avg_factor=af), label),
        "cw_mean": (dict(class_weight=cw), dict(), label),                                        # mean over rows, not over weights
    }


def ce_variants(tag):
    return ("raw", "smooth_obj") if tag == "lvis" else ("raw",)


def boosted_score(score, label, seed):
    """A score whose top-1 does hit sometimes (accuracy fixtures)."""
    n = score.shape[0]
    boosted = score.clone()
    hit = torch.rand(n, generator=torch.Generator().manual_seed(seed + 7)) < 0.4
    boosted[hit, label[hit]] += 12.0
    return boosted


NORMED_LINEAR_CASES = (("lin81", 16, 64, 81, 20, 1.0, None), ("lin_pow2", 12, 48, 30, 8, 2.0, None),
                       ("iif1204", 32, 64, 1204, 8, 1.0, "base2_obj"), ("iif81", 9, 32, 81, 20, 1.0, "raw"))
# name, n, cin, cout, hw, norm_over_kernel, kernel, stride, padding (round 3: kernels beyond the 1x1 predictor)
NORMED_CONV_CASES = (("conv80", 2, 256, 80, 7, False, 1, 1, 0), ("conv_nok", 3, 32, 12, 7, True, 1, 1, 0),
                     ("conv3x3", 2, 32, 20, 9, False, 3, 1, 1), ("conv3x3_nok_s2", 2, 16, 12, 8, True, 3, 2, 1))
MASK_CASES = (("lvis", 6, 1203, 14, 3.0, 31), ("coco", 16, 80, 28, 1.0, 32), ("doc", 3, 11, 2, 1000.0, 33))


def mask_inputs(n, c, hw, scale, seed):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(n, c, hw, hw, generator=g) * scale
    target = (torch.rand(n, hw, hw, generator=g) > 0.5).float()
    label = torch.randint(0, c, (n,), generator=g)
    return pred, target, label


FASA_N, FASA_STEPS, FASA_SEED0 = 256, 3, 300
