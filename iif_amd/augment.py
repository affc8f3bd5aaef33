"""Host-side colour / auto-augment transforms of the reference's training pipelines, on CHW float tensors in [0, 1].

The reference composes them from torchvision and from the un-vendored ``randaugment`` package, both on PIL images
(classification/imbalanced_dataset.py:10,196-225; initialisers.py:57,118-126): neither torchvision nor PIL is in this
image, so this module restates the PUBLISHED algorithms on tensors -

* ``ColorJitter(brightness, contrast, saturation, hue)``: torchvision's transform - one factor per property drawn uniformly
  from [max(0, 1 - v), 1 + v] (hue: [-h, h]), the four adjustments applied in a random order; each adjustment is a blend
  ``clamp(f * img + (1 - f) * other, 0, 1)`` with ``other`` = 0 (brightness), the mean of the grey image (contrast), the grey
  image (saturation; grey = 0.2989 R + 0.587 G + 0.114 B), hue a shift of H in HSV space.
* ``AutoAugmentPolicy("imagenet" | "cifar10")``: the 25 two-operation sub-policies of Cubuk et al., "AutoAugment" (2019),
  tables 9 / 7 of the appendix as the ``randaugment`` package's ``ImageNetPolicy`` / ``CIFAR10Policy`` carry them: one sub-policy
  per image, each operation applied with its probability at one of ten magnitudes; geometric operations fill with grey 128.
* ``RandAugment(n, m)``: n operations drawn uniformly from the same operation set, all at magnitude m.

PARITY: unpinned.  There is no reference run to take vectors from (PIL / torchvision absent here and on the GPU box); the tests
pin the algebra (identity factors, grey-image invariants, histogram / bit properties, the policy tables' shape) - not the
resampling details of PIL's affine transforms, which use nearest-neighbour sampling as PIL does by default.  This is data
augmentation on the host, outside the measured path.
"""
import math

import torch

_GREY = (0.2989, 0.587, 0.114)


def _grey(img):
    return (_GREY[0] * img[0] + _GREY[1] * img[1] + _GREY[2] * img[2]).unsqueeze(0)


def _blend(img, other, f):
    return (f * img + (1.0 - f) * other).clamp_(0.0, 1.0)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    return _blend(img, _grey(img).mean(), f)


def adjust_saturation(img, f):
    return _blend(img, _grey(img), f)


def _rgb_to_hsv(img):
    r, g, b = img[0], img[1], img[2]
    maxc, minc = img.max(0).values, img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc))


def _hsv_to_rgb(hsv):
    h, s, v = hsv[0], hsv[1], hsv[2]
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int64) % 6
    p = (v * (1.0 - s)).clamp(0.0, 1.0)
    q = (v * (1.0 - f * s)).clamp(0.0, 1.0)
    t = (v * (1.0 - (1.0 - f) * s)).clamp(0.0, 1.0)
    r = torch.stack((v, q, p, p, t, v)).gather(0, i.unsqueeze(0))[0]
    g = torch.stack((t, v, v, q, p, p)).gather(0, i.unsqueeze(0))[0]
    b = torch.stack((p, p, t, v, v, q)).gather(0, i.unsqueeze(0))[0]
    return torch.stack((r, g, b))


def adjust_hue(img, shift):
    if shift == 0:
        return img
    hsv = _rgb_to_hsv(img)
    hsv[0] = torch.remainder(hsv[0] + shift, 1.0)
    return _hsv_to_rgb(hsv)


class ColorJitter(object):
    """torchvision.transforms.ColorJitter on a CHW float tensor in [0, 1] (imbalanced_dataset.py:197,205: 0.4, 0.4, 0.4 and
    hue 0.25 for iNaturalist, 0 otherwise)."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

    def __call__(self, img, gen=None):
        r = lambda: torch.rand((), generator=gen).item()      # noqa: E731
        order = torch.randperm(4, generator=gen).tolist()
        fb = max(0.0, 1.0 - self.brightness) + (1.0 + self.brightness - max(0.0, 1.0 - self.brightness)) * r() if self.brightness else None
        fc = max(0.0, 1.0 - self.contrast) + (1.0 + self.contrast - max(0.0, 1.0 - self.contrast)) * r() if self.contrast else None
        fs = max(0.0, 1.0 - self.saturation) + (1.0 + self.saturation - max(0.0, 1.0 - self.saturation)) * r() if self.saturation else None
        fh = -self.hue + 2.0 * self.hue * r() if self.hue else None
        for k in order:
            if k == 0 and fb is not None:
                img = adjust_brightness(img, fb)
            elif k == 1 and fc is not None:
                img = adjust_contrast(img, fc)
            elif k == 2 and fs is not None:
                img = adjust_saturation(img, fs)
            elif k == 3 and fh is not None:
                img = adjust_hue(img, fh)
        return img


# ------------------------------------------------------------------------------------------------ auto-augment operations
def _affine(img, a, b, c, d, e, f, fill=128.0 / 255.0):
    """PIL's Image.transform(size, AFFINE, (a, b, c, d, e, f)): output pixel (x, y) takes input pixel (a x + b y + c,
    d x + e y + f), nearest neighbour, ``fill`` outside."""
    _, h, w = img.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
    sx = torch.floor(a * xs + b * ys + c).to(torch.int64)
    sy = torch.floor(d * xs + e * ys + f).to(torch.int64)
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    out = img[:, sy.clamp(0, h - 1), sx.clamp(0, w - 1)]
    return torch.where(ok.unsqueeze(0), out, torch.full_like(out, fill))


def _rotate(img, deg):
    _, h, w = img.shape
    t = -math.radians(deg)
    cx, cy = w / 2.0, h / 2.0
    a, b, d, e = math.cos(t), math.sin(t), -math.sin(t), math.cos(t)
    return _affine(img, a, b, cx - a * cx - b * cy, d, e, cy - d * cx - e * cy)


def _u8(img):
    return (img * 255.0).round().clamp(0, 255).to(torch.uint8)


def _posterize(img, bits):
    mask = (0xFF << (8 - int(bits))) & 0xFF
    return (_u8(img) & mask).float() / 255.0


def _solarize(img, threshold):
    u = _u8(img).to(torch.int64)                     # (a uint8 comparison would wrap the threshold 256 to 0)
    return torch.where(u < int(math.ceil(threshold)), u, 255 - u).float() / 255.0


def _autocontrast(img):
    lo = img.amin((1, 2), keepdim=True)
    hi = img.amax((1, 2), keepdim=True)
    scale = torch.where(hi > lo, 1.0 / (hi - lo).clamp_min(1e-12), torch.ones_like(hi))
    return torch.where(hi > lo, (img - lo) * scale, img)


def _equalize(img):
    """PIL.ImageOps.equalize per channel: lut[i] = (cumulative histogram below i + step // 2) // step, step = (pixels - last
    non-empty bin) // 255; a channel with step 0 is left alone."""
    u = _u8(img)
    out = torch.empty_like(u)
    for c in range(u.shape[0]):
        hist = torch.bincount(u[c].reshape(-1).to(torch.int64), minlength=256)
        nz = hist[hist > 0]
        step = int((hist.sum() - nz[-1]) // 255)
        if step == 0:
            out[c] = u[c]
            continue
        lut = ((torch.cumsum(hist, 0) - hist) + step // 2) // step
        out[c] = lut.clamp(0, 255).to(torch.uint8)[u[c].to(torch.int64)]
    return out.float() / 255.0


def _sharpness(img, f):
    """PIL.ImageEnhance.Sharpness: blend with the image smoothed by the 3x3 kernel (1 1 1; 1 5 1; 1 1 1) / 13 (borders kept)."""
    k = torch.tensor([[1.0, 1.0, 1.0], [1.0, 5.0, 1.0], [1.0, 1.0, 1.0]]) / 13.0
    sm = torch.nn.functional.conv2d(img.unsqueeze(1), k.view(1, 1, 3, 3))[:, 0]
    blur = img.clone()
    blur[:, 1:-1, 1:-1] = sm
    return _blend(img, blur, f)


def _ranges():
    lin = lambda a, b: [a + (b - a) * i / 9.0 for i in range(10)]      # noqa: E731
    return {
        "ShearX": lin(0.0, 0.3), "ShearY": lin(0.0, 0.3), "TranslateX": lin(0.0, 150.0 / 331.0), "TranslateY": lin(0.0, 150.0 / 331.0),
        "Rotate": lin(0.0, 30.0), "Color": lin(0.0, 0.9), "Posterize": [int(round(v)) for v in lin(8.0, 4.0)], "Solarize": lin(256.0, 0.0),
        "Contrast": lin(0.0, 0.9), "Sharpness": lin(0.0, 0.9), "Brightness": lin(0.0, 0.9), "AutoContrast": [0] * 10, "Equalize": [0] * 10,
        "Invert": [0] * 10,
    }


def apply_op(img, name, magnitude_idx, gen=None):
    """One auto-augment operation at magnitude index 0..9 (random sign for the signed ones, as the package draws it)."""
    m = _ranges()[name][magnitude_idx]
    sign = 1.0 if torch.rand((), generator=gen).item() < 0.5 else -1.0
    _, h, w = img.shape
    if name == "ShearX":
        return _affine(img, 1.0, m * sign, 0.0, 0.0, 1.0, 0.0)
    if name == "ShearY":
        return _affine(img, 1.0, 0.0, 0.0, m * sign, 1.0, 0.0)
    if name == "TranslateX":
        return _affine(img, 1.0, 0.0, m * w * sign, 0.0, 1.0, 0.0)
    if name == "TranslateY":
        return _affine(img, 1.0, 0.0, 0.0, 0.0, 1.0, m * h * sign)
    if name == "Rotate":
        return _rotate(img, m * sign)
    if name == "Color":
        return adjust_saturation(img, 1.0 + m * sign)
    if name == "Contrast":
        return adjust_contrast(img, 1.0 + m * sign)
    if name == "Brightness":
        return adjust_brightness(img, 1.0 + m * sign)
    if name == "Sharpness":
        return _sharpness(img, 1.0 + m * sign)
    if name == "Posterize":
        return _posterize(img, m)
    if name == "Solarize":
        return _solarize(img, m)
    if name == "AutoContrast":
        return _autocontrast(img)
    if name == "Equalize":
        return _equalize(img)
    if name == "Invert":
        return 1.0 - img
    raise ValueError("unknown auto-augment operation %r" % (name,))


_P = {
    "imagenet": [
        ("Posterize", 0.4, 8, "Rotate", 0.6, 9), ("Solarize", 0.6, 5, "AutoContrast", 0.6, 5), ("Equalize", 0.8, 8, "Equalize", 0.6, 3),
        ("Posterize", 0.6, 7, "Posterize", 0.6, 6), ("Equalize", 0.4, 7, "Solarize", 0.2, 4), ("Equalize", 0.4, 4, "Rotate", 0.8, 8),
        ("Solarize", 0.6, 3, "Equalize", 0.6, 7), ("Posterize", 0.8, 5, "Equalize", 1.0, 2), ("Rotate", 0.2, 3, "Solarize", 0.6, 8),
        ("Equalize", 0.6, 8, "Posterize", 0.4, 6), ("Rotate", 0.8, 8, "Color", 0.4, 0), ("Rotate", 0.4, 9, "Equalize", 0.6, 2),
        ("Equalize", 0.0, 7, "Equalize", 0.8, 8), ("Invert", 0.6, 4, "Equalize", 1.0, 8), ("Color", 0.6, 4, "Contrast", 1.0, 8),
        ("Rotate", 0.8, 8, "Color", 1.0, 2), ("Color", 0.8, 8, "Solarize", 0.8, 7), ("Sharpness", 0.4, 7, "Invert", 0.6, 8),
        ("ShearX", 0.6, 5, "Equalize", 1.0, 9), ("Color", 0.4, 0, "Equalize", 0.6, 3), ("Equalize", 0.4, 7, "Solarize", 0.2, 4),
        ("Solarize", 0.6, 5, "AutoContrast", 0.6, 5), ("Invert", 0.6, 4, "Equalize", 1.0, 8), ("Color", 0.6, 4, "Contrast", 1.0, 8),
        ("Equalize", 0.8, 8, "Equalize", 0.6, 3),
    ],
    "cifar10": [
        ("Invert", 0.1, 7, "Contrast", 0.2, 6), ("Rotate", 0.7, 2, "TranslateX", 0.3, 9), ("Sharpness", 0.8, 1, "Sharpness", 0.9, 3),
        ("ShearY", 0.5, 8, "TranslateY", 0.7, 9), ("AutoContrast", 0.5, 8, "Equalize", 0.9, 2), ("ShearY", 0.2, 7, "Posterize", 0.3, 7),
        ("Color", 0.4, 3, "Brightness", 0.6, 7), ("Sharpness", 0.3, 9, "Brightness", 0.7, 9), ("Equalize", 0.6, 5, "Equalize", 0.5, 1),
        ("Contrast", 0.6, 7, "Sharpness", 0.6, 5), ("Color", 0.7, 7, "TranslateX", 0.5, 8), ("Equalize", 0.3, 7, "AutoContrast", 0.4, 8),
        ("TranslateY", 0.4, 3, "Sharpness", 0.2, 6), ("Brightness", 0.9, 6, "Color", 0.2, 8), ("Solarize", 0.5, 2, "Invert", 0.0, 3),
        ("Equalize", 0.2, 0, "AutoContrast", 0.6, 0), ("Equalize", 0.2, 8, "Equalize", 0.6, 4), ("Color", 0.9, 9, "Equalize", 0.6, 6),
        ("AutoContrast", 0.8, 4, "Solarize", 0.2, 8), ("Brightness", 0.1, 3, "Color", 0.7, 0), ("Solarize", 0.4, 5, "AutoContrast", 0.9, 3),
        ("TranslateY", 0.9, 9, "TranslateY", 0.7, 9), ("AutoContrast", 0.9, 2, "Solarize", 0.8, 3), ("Equalize", 0.8, 8, "Invert", 0.1, 3),
        ("TranslateY", 0.7, 9, "AutoContrast", 0.9, 1),
    ],
}
_P["cifar"] = _P["cifar10"]


class AutoAugmentPolicy(object):
    """``ImageNetPolicy()`` / ``CIFAR10Policy()`` of the ``randaugment`` package (imbalanced_dataset.py:210-217,
    initialisers.py:120-126): one of 25 sub-policies per image, its two operations each with its own probability."""

    def __init__(self, name):
        if name not in _P:
            raise ValueError("auto-augment policy %r (known: imagenet, cifar10 / cifar)" % (name,))
        self.policies = _P[name]

    def __call__(self, img, gen=None):
        p = self.policies[int(torch.randint(0, len(self.policies), (), generator=gen).item())]
        for (name, prob, mag) in ((p[0], p[1], p[2]), (p[3], p[4], p[5])):
            if torch.rand((), generator=gen).item() < prob:
                img = apply_op(img, name, mag, gen)
        return img


class RandAugment(object):
    """``RandAugment()`` of the same package (imbalanced_dataset.py:218-225): n operations drawn uniformly from the
    operation set, every one at magnitude index m (of 0..9)."""
    OPS = ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Color", "Posterize", "Solarize", "Contrast", "Sharpness",
           "Brightness", "AutoContrast", "Equalize", "Invert")

    def __init__(self, n=2, m=9):
        self.n, self.m = int(n), min(max(int(m), 0), 9)

    def __call__(self, img, gen=None):
        for _ in range(self.n):
            img = apply_op(img, self.OPS[int(torch.randint(0, len(self.OPS), (), generator=gen).item())], self.m, gen)
        return img
