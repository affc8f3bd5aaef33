"""mmdet normed predictors over the gfx950 kernels.

Mirror of instance_segmentation/mmdet/models/utils/normed_predictor.py:
``NormedLinear`` (:11-40), ``IIFNormedLinear`` (:43-73) and ``NormedConv2d``
(:75-124: the 1x1 kernel ``FCNMaskHead.conv_logits`` builds, fcn_mask_head.py, and k x k kernels with
``norm_over_kernel``) with the same constructor keywords (``tempearture`` is the
reference's spelling), parameter names and init law.  Used through
``cls_predictor_cfg=dict(type='NormedLinear', tempearture=8)`` in every cos-norm
LVIS recipe (configs/fasa/*cos_norm*.py:56,76).

    weight_ = (iif * W) / (|iif * W|_row^power + eps)
    x_      = tempearture * x / (|x|_row^power + eps)
    out     = x_ @ weight_^T + bias

Row normalisations and their backward are `iif_rownorm_*`; the products run on the
exact-fp32 MFMA kernels (`iif_conv_igemm`, `iif_conv_wgrad`).  When mmdet is
importable the classes register themselves under the reference's names.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .mmdet_iif_loss import read_iif_csv


def _round_up(v, m):
    return (v + m - 1) // m * m


class _NormedLinearFn(torch.autograd.Function):
    """x [N, D] fp32, weight [C, D] fp32, bias [C] or None, row_scale [C] or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, row_scale, temperature, power, eps):
        _lib.require_gpu(x, weight)
        x = x.float().contiguous()
        w = weight.float().contiguous()
        n, d = x.shape
        c = w.shape[0]
        cp = _round_up(c, 8)
        dev = x.device
        xn = torch.empty((n, d), dtype=torch.float32, device=dev)
        xnorm = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        wn = torch.zeros((cp, d), dtype=torch.float32, device=dev)          # pad rows stay zero
        wnorm = torch.empty(c, dtype=torch.float32, device=dev)
        ops.rownorm_forward(x, power, temperature, eps, xn, xnorm)
        ops.rownorm_forward(w, power, 1.0, eps, wn, wnorm, row_scale=row_scale)
        out = torch.zeros((n, cp), dtype=torch.float32, device=dev)
        if n > 0:
            b = None
            if bias is not None:
                b = torch.zeros(cp, dtype=torch.float32, device=dev)
                b[:c] = bias.float()
            ops.conv_forward(xn.view(n, 1, 1, d), wn, 1, 1, 1, 0, out=out.view(n, 1, 1, cp), bias=b)
        ctx.save_for_backward(x, w, xn, xnorm, wn, wnorm, row_scale if row_scale is not None else torch.empty(0, device=dev))
        ctx.cfg = (temperature, power, eps, row_scale is not None, bias is not None, c, cp)
        return out[:, :c]

    @staticmethod
    def backward(ctx, g):
        x, w, xn, xnorm, wn, wnorm, rs = ctx.saved_tensors
        temperature, power, eps, has_rs, has_bias, c, cp = ctx.cfg
        n, d = x.shape
        dev = x.device
        gp = torch.zeros((n, cp), dtype=torch.float32, device=dev)
        gp[:, :c] = g
        dx = torch.zeros_like(x)
        dw = torch.zeros_like(w)
        db = None
        if n > 0:
            if has_bias:
                db = torch.empty(cp, dtype=torch.float32, device=dev)
                ops.colsum_f32(gp, n, cp, cp, db)
                db = db[:c]
            dwn = ops.conv_wgrad(xn.view(n, 1, 1, d), gp.view(n, 1, 1, cp), 1, 1, 1, 0, ldw=d)        # [cp, d]
            wt = torch.zeros((d, _round_up(cp, 16)), dtype=torch.float32, device=dev)
            ops.weight_transpose(wn, cp, d, 1, wt)
            dxn = ops.conv_dgrad(gp.view(n, 1, 1, cp), wt, 1, 1, 1, 0, (1, 1)).view(n, d)
            ops.rownorm_backward(x, xnorm, dxn, power, temperature, eps, dx)
            ops.rownorm_backward(w, wnorm, dwn[:c], power, 1.0, eps, dw, row_scale=rs if has_rs else None)
        elif has_bias:
            db = torch.zeros(c, dtype=torch.float32, device=dev)
        return dx, dw, db, None, None, None, None


class NormedLinear(nn.Linear):
    """normed_predictor.py:11-40."""

    def __init__(self, *args, tempearture=20, power=1.0, eps=1e-6, **kwargs):
        super().__init__(*args, **kwargs)
        self.tempearture = tempearture
        self.power = power
        self.eps = eps
        self.init_weights()

    def init_weights(self):
        nn.init.normal_(self.weight, mean=0, std=0.01)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def _row_scale(self):
        return None

    def forward(self, x):
        return _NormedLinearFn.apply(x, self.weight, self.bias, self._row_scale(), float(self.tempearture), float(self.power),
                                     float(self.eps))


class IIFNormedLinear(NormedLinear):
    """normed_predictor.py:43-73: rows of W are multiplied by the class's IIF weight (CSV column
    ``variant``, first row dropped, 1.0 appended for the background) before the normalisation."""

    def __init__(self, *args, tempearture=20, power=1.0, eps=1e-6, path="./lvis_files/idf_1204.csv", variant="base2_obj",
                 device="cuda", **kwargs):
        super().__init__(*args, tempearture=tempearture, power=power, eps=eps, **kwargs)
        self.iif_weights = read_iif_csv(path, variant).to(device).reshape(-1, 1)       # [C+1, 1] as in the reference
        if self.iif_weights.shape[0] != self.out_features:
            raise ValueError("IIF table has %d rows, the layer %d outputs" % (self.iif_weights.shape[0], self.out_features))

    def _row_scale(self):
        return self.iif_weights.reshape(-1).to(self.weight.device)


class _NormedConvFn(torch.autograd.Function):
    """x [N, C, H, W] fp32, weight [O, C, kh, kw] fp32, bias [O] or None.  Pixels are NHWC rows (normalised over their
    channels), filters KRSC rows: one row per (filter, tap) — or one row per filter with norm_over_kernel — through the same
    row-normalisation kernels as the linear predictors; the convolution, its data gradient and its weight gradient run on the
    exact-fp32 MFMA kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, temperature, power, eps, over_kernel, stride, pad):
        _lib.require_gpu(x, weight)
        n, c, h, w_ = x.shape
        o, _, kh, kw = weight.shape
        dev = x.device
        op = _round_up(o, 8)
        xr = x.float().permute(0, 2, 3, 1).contiguous().view(n * h * w_, c)                  # NHWC pixel rows
        wr = weight.float().permute(0, 2, 3, 1).contiguous()                                   # KRSC
        wrows = wr.view(o, kh * kw * c) if over_kernel else wr.view(o * kh * kw, c)
        xn = torch.empty_like(xr)
        xnorm = torch.empty(max(xr.shape[0], 1), dtype=torch.float32, device=dev)
        wn = torch.zeros((op, kh * kw * c), dtype=torch.float32, device=dev)                   # pad filters stay zero
        wnorm = torch.empty(wrows.shape[0], dtype=torch.float32, device=dev)
        ops.rownorm_forward(xr, power, temperature, eps, xn, xnorm)
        ops.rownorm_forward(wrows, power, 1.0, eps, wn[:o].view(wrows.shape), wnorm)
        ho, wo = ops.conv_out_hw(h, w_, kh, kw, stride, pad)
        out = torch.zeros((n, ho, wo, op), dtype=torch.float32, device=dev)
        b = None
        if bias is not None:
            b = torch.zeros(op, dtype=torch.float32, device=dev)
            b[:o] = bias.float()
        if n > 0:
            ops.conv_forward(xn.view(n, h, w_, c), wn, kh, kw, stride, pad, out=out, bias=b)
        ctx.save_for_backward(xr, wrows, xn, xnorm, wn, wnorm)
        ctx.cfg = (temperature, power, eps, over_kernel, stride, pad, bias is not None, (n, c, h, w_), (o, kh, kw), op)
        return out[..., :o].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        xr, wrows, xn, xnorm, wn, wnorm = ctx.saved_tensors
        temperature, power, eps, over_kernel, stride, pad, has_bias, (n, c, h, w_), (o, kh, kw), op = ctx.cfg
        dev = xr.device
        ho, wo = ops.conv_out_hw(h, w_, kh, kw, stride, pad)
        gp = torch.zeros((n, ho, wo, op), dtype=torch.float32, device=dev)
        gp[..., :o] = g.permute(0, 2, 3, 1)
        db = None
        if has_bias:
            db = torch.empty(op, dtype=torch.float32, device=dev)
            ops.colsum_f32(gp.view(-1, op), n * ho * wo, op, op, db)
            db = db[:o]
        k = kh * kw * c
        dwn = ops.conv_wgrad(xn.view(n, h, w_, c), gp, kh, kw, stride, pad, ldw=k)                          # [op, kh*kw*c]
        wt = torch.zeros((c, _round_up(kh * kw * op, 16)), dtype=torch.float32, device=dev)
        ops.weight_transpose(wn, op, c, kh * kw, wt)
        dxn = ops.conv_dgrad(gp, wt, kh, kw, stride, pad, (h, w_)).view(n * h * w_, c)
        dx = torch.zeros_like(xr)
        ops.rownorm_backward(xr, xnorm, dxn, power, temperature, eps, dx)
        dw = torch.zeros_like(wrows)
        ops.rownorm_backward(wrows, wnorm, dwn[:o].contiguous().view(wrows.shape), power, 1.0, eps, dw)
        return (dx.view(n, h, w_, c).permute(0, 3, 1, 2), dw.view(o, kh, kw, c).permute(0, 3, 1, 2), db,
                None, None, None, None, None, None)


class NormedConv2d(nn.Conv2d):
    """normed_predictor.py:75-124.  1x1 kernels (what ``FCNMaskHead.conv_logits`` builds) take the row-matrix path of the
    linear predictors; k x k kernels with stride 1 or 2 and symmetric padding run as a normalised convolution
    (``norm_over_kernel`` selects the filter normalisation, :105-113).  Dense, undilated, channels a multiple of 4."""

    def __init__(self, *args, tempearture=20, power=1.0, eps=1e-6, norm_over_kernel=False, **kwargs):
        super().__init__(*args, **kwargs)
        kh, kw = self.kernel_size
        if (self.groups != 1 or self.dilation != (1, 1) or self.stride[0] != self.stride[1] or self.stride[0] not in (1, 2)
                or self.padding[0] != self.padding[1] or isinstance(self.padding, str) or self.in_channels % 4 or kh * kw > 16):
            raise NotImplementedError("the native NormedConv2d covers dense, undilated kernels of up to 16 taps, stride 1 or 2, "
                                      "symmetric padding, input channels in fours")
        self.tempearture = tempearture
        self.power = power
        self.norm_over_kernel = norm_over_kernel
        self.eps = eps

    def forward(self, x):
        n, c, h, w = x.shape
        if self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0):
            rows = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
            out = _NormedLinearFn.apply(rows, self.weight.view(self.out_channels, c), self.bias, None, float(self.tempearture),
                                        float(self.power), float(self.eps))
            return out.view(n, h, w, self.out_channels).permute(0, 3, 1, 2)
        return _NormedConvFn.apply(x, self.weight, self.bias, float(self.tempearture), float(self.power), float(self.eps),
                                   bool(self.norm_over_kernel), int(self.stride[0]), int(self.padding[0]))


def register_into_mmdet():
    """Register under the reference's names if mmdet / mmcv are importable."""
    try:
        from mmcv.cnn import CONV_LAYERS
        from mmdet.models.utils.builder import LINEAR_LAYERS
    except Exception:
        return False
    LINEAR_LAYERS.register_module(name="NormedLinear", force=True, module=NormedLinear)
    LINEAR_LAYERS.register_module(name="IIFNormedLinear", force=True, module=IIFNormedLinear)
    CONV_LAYERS.register_module(name="NormedConv2d", force=True, module=NormedConv2d)
    return True


register_into_mmdet()
