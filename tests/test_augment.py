"""Host-side colour / auto-augment transforms (iif_amd/augment.py; reference: imbalanced_dataset.py:196-225 on torchvision and
the un-vendored ``randaugment`` package).  Parity is unpinned (no PIL / torchvision to take vectors from): these tests pin the
algebra of every operation and the shape of the policy tables."""
import math

import torch

from iif_amd import augment as A


def _img(seed=0, h=24, w=20):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, h, w, generator=g)


def test_identity_factors_leave_the_image_alone():
    x = _img()
    for f in (A.adjust_brightness, A.adjust_contrast, A.adjust_saturation):
        assert torch.allclose(f(x, 1.0), x, atol=1e-6)
    assert torch.equal(A.adjust_hue(x, 0.0), x)
    assert torch.allclose(A._hsv_to_rgb(A._rgb_to_hsv(x)), x, atol=1e-5)
    assert torch.allclose(A.adjust_hue(x, 1.0), x, atol=1e-5)                 # a whole turn of the hue circle


def test_colour_adjustments_against_their_definitions():
    x = _img(1)
    grey = 0.2989 * x[0] + 0.587 * x[1] + 0.114 * x[2]
    assert torch.allclose(A.adjust_brightness(x, 0.5), 0.5 * x)
    assert torch.allclose(A.adjust_brightness(x, 1.7), (1.7 * x).clamp(0, 1))
    assert torch.allclose(A.adjust_saturation(x, 0.0), grey.expand(3, -1, -1), atol=1e-6)       # factor 0: the grey image
    assert torch.allclose(A.adjust_contrast(x, 0.0), torch.full_like(x, grey.mean().item()), atol=1e-6)
    # hue: a third of a turn maps R -> G -> B on pure colours
    red = torch.zeros(3, 2, 2); red[0] = 1.0
    out = A.adjust_hue(red, 1.0 / 3.0)
    assert torch.allclose(out[1], torch.ones(2, 2), atol=1e-5) and out[0].abs().max() < 1e-5 and out[2].abs().max() < 1e-5
    g = A._grey(x)
    assert torch.allclose(A.adjust_hue(g.expand(3, -1, -1).clone(), 0.2), g.expand(3, -1, -1), atol=1e-5)   # grey has no hue


def test_color_jitter_draws_in_range_and_is_reproducible():
    x = _img(2)
    cj = A.ColorJitter(0.4, 0.4, 0.4, 0.25)
    a = cj(x, torch.Generator().manual_seed(7))
    b = cj(x, torch.Generator().manual_seed(7))
    c = cj(x, torch.Generator().manual_seed(8))
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.min() >= 0 and a.max() <= 1 and a.shape == x.shape
    assert torch.equal(A.ColorJitter()(x, torch.Generator().manual_seed(1)), x)              # all amounts 0: nothing happens


def test_pointwise_operations():
    x = _img(3)
    u = (x * 255).round().to(torch.uint8)
    assert torch.equal((A._posterize(x, 4) * 255).round().to(torch.uint8), u & 0xF0)
    assert torch.equal((A._posterize(x, 8) * 255).round().to(torch.uint8), u)
    s = (A._solarize(x, 128) * 255).round().to(torch.int64)
    assert torch.equal(s, torch.where(u < 128, u.to(torch.int64), 255 - u.to(torch.int64)))
    assert torch.equal((A._solarize(x, 256) * 255).round().to(torch.uint8), u)               # threshold 256: untouched
    ac = A._autocontrast(0.25 + 0.5 * x)
    assert torch.allclose(ac.amin((1, 2)), torch.zeros(3), atol=1e-6) and torch.allclose(ac.amax((1, 2)), torch.ones(3), atol=1e-6)
    flat = torch.full((3, 4, 4), 0.3)
    assert torch.equal(A._autocontrast(flat), flat) and torch.allclose(A._equalize(flat), (flat * 255).round() / 255)
    assert torch.allclose(A.apply_op(x, "Invert", 0), 1.0 - x)
    # equalisation flattens the cumulative histogram: a ramp of 256 levels, each once, maps to itself
    ramp = (torch.arange(256, dtype=torch.float32) / 255.0).view(1, 16, 16).expand(3, -1, -1).clone()
    assert torch.allclose(A._equalize(ramp), ramp, atol=1.0 / 255)
    sk = torch.rand(1, 64, 64, generator=torch.Generator().manual_seed(4)).pow(3).expand(3, -1, -1).clone()      # skewed to dark
    eq = A._equalize(sk)
    assert abs(eq.mean().item() - 0.5) < 0.05 < abs(sk.mean().item() - 0.5)
    assert torch.allclose(A._sharpness(x, 1.0), x, atol=1e-6)


def test_geometric_operations():
    x = _img(5, 16, 16)
    assert torch.equal(A._affine(x, 1, 0, 0, 0, 1, 0), x)                                     # the identity map
    t = A._affine(x, 1, 0, 3, 0, 1, 0)                                                        # output (x, y) <- input (x + 3, y)
    assert torch.equal(t[:, :, :13], x[:, :, 3:]) and torch.allclose(t[:, :, 13:], torch.full((3, 16, 3), 128 / 255.0))
    assert torch.equal(A._rotate(x, 0.0), x)
    r90 = A._rotate(x, 90.0)                                                                  # counter-clockwise, as PIL
    assert torch.equal(r90, torch.rot90(x, 1, (1, 2)))
    g = torch.Generator().manual_seed(0)
    for name in ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate"):
        assert torch.equal(A.apply_op(x, name, 0, g), x)                                      # magnitude 0 of the ten: no movement
        assert A.apply_op(x, name, 9, g).shape == x.shape


def test_policy_tables_and_sampling():
    for name, n in (("imagenet", 25), ("cifar10", 25)):
        pol = A.AutoAugmentPolicy(name)
        assert len(pol.policies) == n
        for p in pol.policies:
            assert p[0] in A.RandAugment.OPS and p[3] in A.RandAugment.OPS and 0 <= p[1] <= 1 and 0 <= p[4] <= 1
            assert 0 <= p[2] <= 9 and 0 <= p[5] <= 9
    assert A.AutoAugmentPolicy("cifar").policies is A.AutoAugmentPolicy("cifar10").policies
    r = A._ranges()
    assert r["Posterize"] == [8, 8, 7, 7, 6, 6, 5, 5, 4, 4] and r["Solarize"][0] == 256 and abs(r["Rotate"][9] - 30) < 1e-9
    assert all(len(v) == 10 for v in r.values()) and abs(r["TranslateX"][9] - 150 / 331) < 1e-12
    x = _img(6, 32, 32)
    for aug in (A.AutoAugmentPolicy("imagenet"), A.AutoAugmentPolicy("cifar10"), A.RandAugment()):
        outs = [aug(x, torch.Generator().manual_seed(s)) for s in range(12)]
        assert all(o.shape == x.shape and o.min() >= 0 and o.max() <= 1 and not torch.isnan(o).any() for o in outs)
        assert any(not torch.equal(o, x) for o in outs)
        assert torch.equal(aug(x, torch.Generator().manual_seed(3)), outs[3])


def test_training_transform_applies_the_colour_stage():
    import numpy as np
    from iif_amd.imbalanced_dataset import TensorTransform
    img = (np.random.RandomState(0).rand(40, 48, 3) * 255).astype(np.uint8)
    torch.manual_seed(11)
    plain = TensorTransform("imagenet_lt", True, 32, seed=5, color_jitter=False)(img)
    torch.manual_seed(11)
    jit = TensorTransform("imagenet_lt", True, 32, seed=5)(img)
    torch.manual_seed(11)
    aa = TensorTransform("imagenet_lt", True, 32, seed=5, auto_augment="imagenet")
    assert isinstance(aa.colour, A.AutoAugmentPolicy) and isinstance(TensorTransform("inat18", True).colour, A.ColorJitter)
    assert TensorTransform("inat18", True).colour.hue == 0.25 and TensorTransform("places_lt", True).colour.hue == 0.0
    assert TensorTransform("imagenet_lt", False).colour is None
    assert plain.shape == jit.shape == (3, 32, 32) and not torch.equal(plain, jit)            # same crop / flip draws, then the jitter
    assert math.isfinite(aa(img).sum().item())
