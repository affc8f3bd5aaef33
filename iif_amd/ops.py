"""Thin tensor-level wrappers over the C ABI (one call = one kernel launch on
torch's current HIP stream).  Layout everywhere: activations NHWC, weights
[Cout][R][S][Cin] rows with pitch ``ldw``.  No fallbacks: GPU tensors only."""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check, dtype_code, lib, ptr, require_gpu, stream_ptr


def _desc(n, hs, ws, cs, hd, wd, cd, r, s, stride, pad, transposed, ldw, dtype, dst_dtype, groups=1, w_frag=None):
    """``w_frag``: the weights as MFMA fragments - a tensor (iif_conv_pack_fragments' format) or ``(tensor, 1)`` for the grouped
    16-channel format of iif_conv_pack_fragments_g16."""
    kind = 0
    if isinstance(w_frag, tuple):
        w_frag, kind = w_frag
    return ConvDesc(n, hs, ws, cs, hd, wd, cd, r, s, stride, pad, transposed, ldw, dtype, dst_dtype, groups,
                    ptr(w_frag) if w_frag is not None else None, kind)


def conv_out_hw(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


def conv_forward(x, w, r, s, stride, pad, out=None, out_dtype=None, bias=None, res=None, groups=1, out_hw=None, w_frag=None):
    """x: [N,H,W,Cin] NHWC; w: [Cout, ldw] (rows = r*s*Cin K-contiguous, KRSC).  groups > 1: the
    channels split into `groups` chunks, w holds the chunk matrices one after the other.  w_frag: the same weights as
    MFMA fragments (pack_fragments), used by the 3x3 / stride-1 kernel where it applies."""
    require_gpu(x, w, bias, res)
    n, h, wd_, cin = x.shape
    cout, ldw = w.shape
    ho, wo = out_hw or conv_out_hw(h, wd_, r, s, stride, pad)     # out_hw: explicit grid (asymmetric padding)
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty((n, ho, wo, cout), dtype=odt, device=x.device)
    d = _desc(n, h, wd_, cin // groups, ho, wo, cout // groups, r, s, stride, pad, 0, ldw, dtype_code(x), dtype_code(out),
              groups, w_frag)
    check(lib().iif_conv_igemm(ctypes.byref(d), ptr(x), ptr(w), ptr(out), ptr(res), ptr(bias), stream_ptr()), "iif_conv_igemm")
    return out


def conv_forward_bnstats(x, w, r, s, stride, pad, out, partial, groups=1, w_frag=None):
    """conv_forward (bf16) that also writes per-tile BN partial sums; returns the tile count."""
    n, h, wd_, cin = x.shape
    cout, ldw = w.shape
    ho, wo = out.shape[1], out.shape[2]
    d = _desc(n, h, wd_, cin // groups, ho, wo, cout // groups, r, s, stride, pad, 0, ldw, dtype_code(x), dtype_code(out),
              groups, w_frag)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_bnstats(ctypes.byref(d), ptr(x), ptr(w), ptr(out), 0, 0, ptr(partial), partial.numel(),
                                       ctypes.byref(nt), stream_ptr()), "iif_conv_igemm_bnstats")
    return nt.value


def bn_finalize_stats(partial, n_partials, m, c, gamma, beta, running_mean, running_var, stats, eps=1e-5, momentum=0.1,
                      scratch=None, tickets=None, extra_sums=None):
    """``tickets`` (zeroed int32[64], one per stream): both reduction stages in one launch (iif_bn_finalize_stats_fused).
    ``extra_sums`` = (rows [n, 2, c2], n, c2, out [2, c2]): a second set of partial rows column-summed by the same launch."""
    if extra_sums is not None:
        rows2, n2, c2, out2 = extra_sums
        check(lib().iif_bn_finalize_stats_sums(ptr(partial), n_partials, m, c, ptr(gamma), ptr(beta), eps, momentum,
                                               ptr(running_mean), ptr(running_var), ptr(stats), ptr(scratch),
                                               0 if scratch is None else scratch.numel(), ptr(tickets), ptr(rows2), n2, c2, ptr(out2),
                                               stream_ptr()), "iif_bn_finalize_stats_sums", tickets)
        return stats
    if tickets is not None:
        check(lib().iif_bn_finalize_stats_fused(ptr(partial), n_partials, m, c, ptr(gamma), ptr(beta), eps, momentum,
                                                ptr(running_mean), ptr(running_var), ptr(stats), ptr(scratch),
                                                0 if scratch is None else scratch.numel(), ptr(tickets), stream_ptr()),
              "iif_bn_finalize_stats_fused", tickets)
        return stats
    check(lib().iif_bn_finalize_stats(ptr(partial), n_partials, m, c, ptr(gamma), ptr(beta), eps, momentum,
                                      ptr(running_mean), ptr(running_var), ptr(stats), ptr(scratch),
                                      0 if scratch is None else scratch.numel(), stream_ptr()),
          "iif_bn_finalize_stats")
    return stats


def conv_dgrad(dy, wt, r, s, stride, pad, in_hw, out=None, res=None, groups=1, res_bits=None, w_frag=None):
    """dy: [N,Ho,Wo,Cout]; wt: [Cin, ldw] rows of r*s*Cout (the CRSK transpose);
    returns dx [N,H,W,Cin] (+ res, gated element-wise by the ReLU bits ``res_bits`` if given)."""
    require_gpu(dy, wt, res)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    if out is None:
        out = torch.empty((n, h, w_, cin), dtype=dy.dtype, device=dy.device)
    d = _desc(n, ho, wo, cout // groups, h, w_, cin // groups, r, s, stride, pad, 1, ldw, dtype_code(dy), dtype_code(out),
              groups, w_frag)
    if res_bits is not None:
        check(lib().iif_conv_igemm_masked_res(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), ptr(res_bits),
                                              stream_ptr()), "iif_conv_igemm_masked_res")
        return out
    check(lib().iif_conv_igemm(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), 0, stream_ptr()), "iif_conv_igemm(dgrad)")
    return out


def conv_wgrad(x, dy, r, s, stride, pad, ldw=None, out=None, workspace=None, splits=0, groups=1):
    """x: [N,H,W,Cin], dy: [N,Ho,Wo,Cout] -> dw float32 [Cout, ldw] (KRSC rows)."""
    require_gpu(x, dy, out, workspace)
    n, h, w_, cin = x.shape
    _, ho, wo, cout = dy.shape
    k = r * s * (cin // groups)
    ldw = ldw or k
    if out is None:
        out = torch.zeros((cout, ldw), dtype=torch.float32, device=x.device)
    elif not out.is_contiguous() or out.numel() < cout * ldw:
        raise ValueError("conv_wgrad: out must be a contiguous float32 [Cout, ldw] tensor (rows are written with pitch ldw)")
    d = _desc(n, h, w_, cin // groups, ho, wo, cout // groups, r, s, stride, pad, 0, ldw, dtype_code(x), _lib.IIF_F32, groups)
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(lib().iif_conv_wgrad(ctypes.byref(d), ptr(x), ptr(dy), ptr(out), ptr(workspace), wsb, splits, stream_ptr()),
          "iif_conv_wgrad")
    return out


def wgrad1x1_stacked(x2d, dy2d, dy2_2d, out, workspace, splits=0):
    """out[:cd1] = dy^T x, out[cd1:] = dy2^T x in one pass over x (bf16 [m, c] views; out float32 [cd1 + cd2, ldw])."""
    require_gpu(x2d, dy2d, dy2_2d, out, workspace)
    m, cs = x2d.shape
    cd1, cd2 = dy2d.shape[1], dy2_2d.shape[1]
    if not out.is_contiguous() or out.shape[0] < cd1 + cd2:
        raise ValueError("wgrad1x1_stacked: out must be a contiguous float32 [cd1 + cd2, ldw] tensor")
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(lib().iif_wgrad1x1_stacked(ptr(x2d), ptr(dy2d), ptr(dy2_2d), m, cs, cd1, cd2, out.shape[1], ptr(out), ptr(workspace), wsb,
                                     splits, stream_ptr()), "iif_wgrad1x1_stacked")
    return out


# ------------------------------------------------------------------ batch norm
def bn_workspace(m, c, device):
    nbytes = lib().iif_bn_workspace_bytes(m, c)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def bn_forward_stats(x2d, gamma, beta, running_mean, running_var, stats, ws, eps=1e-5, momentum=0.1):
    """x2d: [M, C] view of an NHWC activation; fills stats [4, C]."""
    m, c = x2d.shape
    check(lib().iif_bn_forward_stats(ptr(x2d), dtype_code(x2d), m, c, ptr(gamma), ptr(beta), eps, momentum,
                                     ptr(running_mean), ptr(running_var), ptr(stats), ptr(ws), ws.numel(),
                                     stream_ptr()), "iif_bn_forward_stats")
    return stats


def bn_apply(x2d, stats, y2d, relu=True, residual=None, residual_stats=None, relu_bits=None):
    m, c = x2d.shape
    check(lib().iif_bn_apply(ptr(x2d), dtype_code(x2d), m, c, ptr(stats), ptr(residual), ptr(residual_stats),
                             1 if relu else 0, ptr(y2d), ptr(relu_bits), stream_ptr()), "iif_bn_apply")
    return y2d


def bn_backward(gy, y_mask, x2d, stats, gamma, dgamma, dbeta, dx, ws, gmasked=None, relu_bits=None):
    m, c = x2d.shape
    check(lib().iif_bn_backward(ptr(gy), ptr(y_mask), ptr(relu_bits), ptr(x2d), dtype_code(x2d), m, c, ptr(stats),
                                ptr(gamma), ptr(dgamma), ptr(dbeta), ptr(dx), ptr(gmasked), ptr(ws), ws.numel(),
                                stream_ptr()), "iif_bn_backward")
    return dx


# ----------------------------------------------------- cross-replica BN pieces
def bn_partial_sums(partial, n_partials, c, sums):
    """sums [2, c] = column sums of the partial rows [n, 2, c] (per-rank half of a SyncBatchNorm reduction)."""
    check(lib().iif_bn_partial_sums(ptr(partial), n_partials, c, ptr(sums), stream_ptr()), "iif_bn_partial_sums")
    return sums


def bn_stats_sums(x2d, sums, ws):
    m, c = x2d.shape
    check(lib().iif_bn_stats_sums(ptr(x2d), dtype_code(x2d), m, c, ptr(sums), ptr(ws), ws.numel(), stream_ptr()), "iif_bn_stats_sums")
    return sums


def bn_backward_sums(gy, y_mask, x2d, stats, sums, ws, relu_bits=None):
    m, c = x2d.shape
    check(lib().iif_bn_backward_sums(ptr(gy), ptr(y_mask), ptr(relu_bits), ptr(x2d), dtype_code(x2d), m, c, ptr(stats), ptr(sums),
                                     ptr(ws), ws.numel(), stream_ptr()), "iif_bn_backward_sums")
    return sums


def bn_backward_apply_sums(gy, y_mask, x2d, stats, gamma, local_sums, total_sums, total_count, dgamma, dbeta, dx, coef, gmasked=None,
                           relu_bits=None):
    m, c = x2d.shape
    check(lib().iif_bn_backward_apply_sums(ptr(gy), ptr(y_mask), ptr(relu_bits), ptr(x2d), dtype_code(x2d), m, c, ptr(stats), ptr(gamma),
                                           ptr(local_sums), ptr(total_sums), float(total_count), ptr(dgamma), ptr(dbeta), ptr(dx),
                                           ptr(gmasked), ptr(coef), stream_ptr()), "iif_bn_backward_apply_sums")
    return dx


# --------------------------------------------------------------------- pooling
def maxpool_forward(x, k, stride, pad):
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    check(lib().iif_maxpool_forward(ptr(x), dtype_code(x), n, h, w, c, k, stride, pad, ptr(y), ptr(idx),
                                    stream_ptr()), "iif_maxpool_forward")
    return y, idx


def maxpool_backward(gy, idx, in_shape, k, stride, pad, out=None):
    n, h, w, c = in_shape
    dx = out if out is not None else torch.empty(in_shape, dtype=gy.dtype, device=gy.device)
    check(lib().iif_maxpool_backward(ptr(gy), ptr(idx), dtype_code(gy), n, h, w, c, k, stride, pad, ptr(dx),
                                     stream_ptr()), "iif_maxpool_backward")
    return dx


def avgpool_forward(x, out=None):
    n, h, w, c = x.shape
    y = out if out is not None else torch.empty((n, c), dtype=x.dtype, device=x.device)
    check(lib().iif_avgpool_forward(ptr(x), dtype_code(x), n, h * w, c, ptr(y), stream_ptr()), "iif_avgpool_forward")
    return y


def avgpool_backward(gy, hw, out=None):
    n, c = gy.shape
    dx = out if out is not None else torch.empty((n, hw, c), dtype=gy.dtype, device=gy.device)
    check(lib().iif_avgpool_backward(ptr(gy), dtype_code(gy), n, hw, c, ptr(dx), stream_ptr()), "iif_avgpool_backward")
    return dx


# ------------------------------------------------------------------------ misc
def im2col_nchw(img, r, s, stride, pad, kp, dtype, out=None):
    require_gpu(img)
    if img.dtype != torch.float32 or not img.is_contiguous():
        img = img.float().contiguous()
    n, cin, h, w = img.shape
    ho, wo = conv_out_hw(h, w, r, s, stride, pad)
    if out is None:
        out = torch.empty((n, ho, wo, kp), dtype=dtype, device=img.device)
    check(lib().iif_im2col_nchw(ptr(img), n, cin, h, w, r, s, stride, pad, kp, dtype_code(out), ptr(out),
                                stream_ptr()), "iif_im2col_nchw")
    return out


def cast(src, dst):
    check(lib().iif_cast(ptr(src), dtype_code(src), ptr(dst), dtype_code(dst), src.numel(), stream_ptr()), "iif_cast")
    return dst


def weight_transpose(w2d, cout, cin, rs, wt2d):
    check(lib().iif_weight_transpose(ptr(w2d), cout, cin, rs, w2d.shape[1], wt2d.shape[1], dtype_code(wt2d),
                                     ptr(wt2d), stream_ptr()), "iif_weight_transpose")
    return wt2d


def shortcut_a_forward(x, cout, out=None):
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout), dtype=x.dtype, device=x.device)
    check(lib().iif_shortcut_a_forward(ptr(x), dtype_code(x), n, h, w, cin, cout, ptr(y), stream_ptr()),
          "iif_shortcut_a_forward")
    return y


def shortcut_a_backward_acc(g, dx):
    n, h, w, cin = dx.shape
    cout = g.shape[-1]
    check(lib().iif_shortcut_a_backward_acc(ptr(g), dtype_code(g), n, h, w, cin, cout, ptr(dx), stream_ptr()),
          "iif_shortcut_a_backward_acc")
    return dx


def colsum_f32(a, rows, cols, ld, out):
    check(lib().iif_colsum_f32(ptr(a), rows, cols, ld, ptr(out), stream_ptr()), "iif_colsum_f32")
    return out


def sgd_step(params, grads, bufs, lr, momentum, weight_decay, nesterov=False, grad_scale=1.0, d_lr=None):
    check(lib().iif_sgd_step(ptr(params), ptr(grads), ptr(bufs), params.numel(), float(lr), ptr(d_lr), float(momentum),
                             float(weight_decay), 1 if nesterov else 0, float(grad_scale), stream_ptr()), "iif_sgd_step")


# ------------------------------------------------------- cosine / normed heads
def rowmap_forward(x, mode, scale, out, norms=None, eps=1e-12):
    rows, cols = x.shape
    check(lib().iif_rowmap_forward(ptr(x), dtype_code(x), rows, cols, x.stride(0), mode, float(scale), eps, ptr(out),
                                   dtype_code(out), out.stride(0), ptr(norms), stream_ptr()), "iif_rowmap_forward")
    return out


def rowmap_backward(x, norms, g, mode, scale, dx, eps=1e-12):
    rows, cols = x.shape
    check(lib().iif_rowmap_backward(ptr(x), dtype_code(x), ptr(norms), ptr(g), dtype_code(g), rows, cols, x.stride(0),
                                    g.stride(0), mode, float(scale), eps, ptr(dx), dtype_code(dx), dx.stride(0),
                                    stream_ptr()), "iif_rowmap_backward")
    return dx


def transpose_f32(src, dst):
    rows, cols = src.shape
    check(lib().iif_transpose_f32(ptr(src), rows, cols, src.stride(0), ptr(dst), dst.stride(0), stream_ptr()),
          "iif_transpose_f32")
    return dst


def dot_window_f32(a, b, rows, cols, alpha, out, alpha_div=None):
    check(lib().iif_dot_window_f32(ptr(a), ptr(b), rows, cols, a.stride(0), b.stride(0), float(alpha), ptr(alpha_div),
                                   ptr(out), stream_ptr()), "iif_dot_window_f32")
    return out


# ------------------------------------------------------------ grouped convs
def group_pack(master, channels, cg, chunk, rs, out, transposed=False):
    check(lib().iif_group_pack(ptr(master), channels, cg, chunk, rs, master.shape[1], out.shape[1], 1 if transposed else 0,
                               dtype_code(out), ptr(out), stream_ptr()), "iif_group_pack")
    return out


def group_pack_table(entries, device):
    """Device table for group_pack_batched.  entries: (master fp32 [C, ldm], C, cg, chunk, rs, out [C, ldp], transposed);
    every ``out`` of one dtype.  The tensors must stay alive and in place as long as the table is used."""
    import struct
    raw = b"".join(struct.pack("<QQ8i", ptr(m) or 0, ptr(o) or 0, c, cg, ch, rs, m.shape[1], o.shape[1], 1 if tr else 0, 0)
                   for (m, c, cg, ch, rs, o, tr) in entries)
    tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    biggest = max(c * o.shape[1] for (m, c, cg, ch, rs, o, tr) in entries)
    return tab, len(entries), max(1, min(64, (biggest + 2047) // 2048)), entries[0][5]


def group_pack_batched(table):
    tab, n, blocks, like = table
    check(lib().iif_group_pack_batched(ptr(tab), n, blocks, dtype_code(like), stream_ptr()), "iif_group_pack_batched")


def group_unpack_grad(packed, channels, cg, chunk, rs, master_grad):
    check(lib().iif_group_unpack_grad(ptr(packed), channels, cg, chunk, rs, packed.shape[1], master_grad.shape[1],
                                      ptr(master_grad), stream_ptr()), "iif_group_unpack_grad")
    return master_grad


# ------------------------------------------------------------ space-to-depth stem
def space_to_depth_nchw(img, cpad, out):
    require_gpu(img)
    if img.dtype != torch.float32 or not img.is_contiguous():
        img = img.float().contiguous()
    n, c, h, w = img.shape
    check(lib().iif_space_to_depth_nchw(ptr(img), n, c, h, w, cpad, dtype_code(out), ptr(out), stream_ptr()),
          "iif_space_to_depth_nchw")
    return out


def stem_s2d_pack(master, k, c, r, cpad, out):
    check(lib().iif_stem_s2d_pack(ptr(master), k, c, r, master.shape[1], cpad, dtype_code(out), ptr(out), stream_ptr()),
          "iif_stem_s2d_pack")
    return out


def stem_s2d_unpack_grad(packed, k, c, r, cpad, master_grad):
    check(lib().iif_stem_s2d_unpack_grad(ptr(packed), k, c, r, cpad, master_grad.shape[1], ptr(master_grad), stream_ptr()),
          "iif_stem_s2d_unpack_grad")
    return master_grad


# ------------------------------------------------------------ squeeze-and-excitation
def se_squeeze(x, sums):
    """x [N,H,W,C] raw conv output -> sums [N,C] fp32 over H*W."""
    n, h, w, c = x.shape
    check(lib().iif_se_squeeze(ptr(x), dtype_code(x), n, h * w, c, ptr(sums), stream_ptr()), "iif_se_squeeze")
    return sums


def se_apply(x, stats, excite, y, relu_bits, residual=None, residual_stats=None):
    n, h, w, c = x.shape
    check(lib().iif_se_apply(ptr(x), dtype_code(x), n, h * w, c, ptr(stats), ptr(excite), ptr(residual),
                             ptr(residual_stats), ptr(y), ptr(relu_bits), stream_ptr()), "iif_se_apply")
    return y


def se_excite_forward(sums, stats, hw, w1, w2t, q, h, e):
    """q = mean_hw(bn(x)), h = relu(W1 q), e = sigmoid(W2 h) in one launch; w2t = W2^T [hid, C]."""
    n, c = sums.shape
    hid = w1.shape[0]
    check(lib().iif_se_excite_forward(ptr(sums), ptr(stats), n, hw, c, hid, ptr(w1), w1.stride(0), ptr(w2t), w2t.stride(0),
                                      ptr(q), ptr(h), ptr(e), stream_ptr()), "iif_se_excite_forward")
    return e


def se_excite_backward(s1, s2, stats, hw, w1, w2t, e, h, q, dz2, dz1, offset, dw1, dw2):
    n, c = s1.shape
    hid = w1.shape[0]
    check(lib().iif_se_excite_backward(ptr(s1), ptr(s2), ptr(stats), n, hw, c, hid, ptr(w1), w1.stride(0), ptr(w2t),
                                       w2t.stride(0), ptr(e), ptr(h), ptr(q), ptr(dz2), ptr(dz1), ptr(offset), ptr(dw1),
                                       dw1.stride(0), ptr(dw2), dw2.stride(0), stream_ptr()), "iif_se_excite_backward")
    return offset


def se_backward_sums(g, relu_bits, x, s1, s2):
    n, h, w, c = x.shape
    check(lib().iif_se_backward_sums(ptr(g), ptr(relu_bits), ptr(x), dtype_code(x), n, h * w, c, ptr(s1), ptr(s2),
                                     stream_ptr()), "iif_se_backward_sums")


def se_backward_form(g, excite, offset, out):
    n, h, w, c = g.shape
    check(lib().iif_se_backward_form(ptr(g), dtype_code(g), n, h * w, c, ptr(excite), ptr(offset), ptr(out), stream_ptr()),
          "iif_se_backward_form")
    return out


# ------------------------------------------------------------ mmdet normed predictors
def rownorm_forward(x, power, scale, eps, out, norms, row_scale=None):
    rows, cols = x.shape
    check(lib().iif_rownorm_forward(ptr(x), ptr(row_scale), rows, cols, x.stride(0), float(power), float(scale), float(eps),
                                    ptr(out), out.stride(0), ptr(norms), stream_ptr()), "iif_rownorm_forward")
    return out


def rownorm_backward(x, norms, g, power, scale, eps, dx, row_scale=None):
    rows, cols = x.shape
    check(lib().iif_rownorm_backward(ptr(x), ptr(row_scale), ptr(norms), ptr(g), rows, cols, x.stride(0), g.stride(0),
                                     float(power), float(scale), float(eps), ptr(dx), dx.stride(0), stream_ptr()),
          "iif_rownorm_backward")
    return dx


# ------------------------------------------------------------ batched weight preparation
def wt_table(entries, device):
    """entries: [(src_off, dst_off, cout, cin, rs, ldw, ldwt)] -> (device int32 table, total blocks).  Layout =
    struct iif_wt_desc {int64 src_off, dst_off; int32 cout, cin, rs, ldw, ldwt, block_start;} (40 bytes)."""
    import struct
    blob, start = b"", 0
    for (so, do, cout, cin, rs, ldw, ldwt) in entries:
        blob += struct.pack("<qqiiiiii", so, do, cout, cin, rs, ldw, ldwt, start)
        start += rs * ((cin + 31) // 32) * ((cout + 31) // 32)           # one block per 32x32 tile of one tap
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    return t, start


def conv3x3_frag_ok(n, h, w, cin, cout, dtype, groups=1):
    """True when a 3x3 / stride-1 / pad-1 convolution of this geometry (cin / cout per group) runs on the fragment-weights
    kernel."""
    d = _desc(n, h, w, cin, h, w, cout, 3, 3, 1, 1, 0, 9 * cin, _lib.IIF_BF16 if dtype == torch.bfloat16 else _lib.IIF_F32,
              _lib.IIF_BF16 if dtype == torch.bfloat16 else _lib.IIF_F32, groups)
    return bool(lib().iif_conv3x3_frag_ok(ctypes.byref(d)))


def pack_table(entries, device):
    """entries: [(src_off, dst_off, rows, taps, k, ld)] in elements -> (device table of iif_pack_desc, total blocks)."""
    import struct
    blob, start = b"", 0
    for (so, do, rows, taps, k, ld) in entries:
        blob += struct.pack("<qqiiiiii", so, do, rows, taps, k, ld, start, 0)
        start += (rows * taps * k // 8 + 255) // 256
    return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device), start


def pack_table_g16(entries, device):
    """entries: [(src_off, dst_off, rows, taps, k, ld)] of grouped layers (k = 64, taps = 9) for iif_conv_pack_fragments_g16:
    20 fragments of 1 KB per 64-channel chunk, four fragments per block."""
    import struct
    blob, start = b"", 0
    for (so, do, rows, taps, k, ld) in entries:
        blob += struct.pack("<qqiiiiii", so, do, rows, taps, k, ld, start, 0)
        start += (rows // 64) * 5
    return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device), start


def pack_fragments_g16(src, table, n_desc, total_blocks, out):
    require_gpu(src, out, table)
    check(lib().iif_conv_pack_fragments_g16(ptr(src), ptr(table), n_desc, total_blocks, ptr(out), stream_ptr()),
          "iif_conv_pack_fragments_g16")
    return out


def pack_fragments(src, table, n_desc, total_blocks, out):
    """src / out: bf16 arenas the table's element offsets refer to (iif_conv_pack_fragments)."""
    require_gpu(src, out, table)
    check(lib().iif_conv_pack_fragments(ptr(src), ptr(table), n_desc, total_blocks, ptr(out), stream_ptr()),
          "iif_conv_pack_fragments")
    return out


def weight_transpose_batched(arena, table, n_desc, total_blocks, out):
    check(lib().iif_weight_transpose_batched(ptr(arena), ptr(table), n_desc, total_blocks, dtype_code(out), ptr(out),
                                             stream_ptr()), "iif_weight_transpose_batched")
    return out


# ------------------------------------------------------------ BN backward sums fused into the producing dgrad
def conv_dgrad_bnbwd(dy, wt, r, s, stride, pad, in_hw, out, up_x, up_bits, up_stats, partial, res=None, res_bits=None,
                     w_frag=None, groups=1):
    """conv_dgrad that also writes the upstream unit's BN-backward partial sums; returns the partial row count."""
    require_gpu(dy, wt, res, up_x)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    d = _desc(n, ho, wo, cout // groups, h, w_, cin // groups, r, s, stride, pad, 1, ldw, dtype_code(dy), dtype_code(out), groups,
              w_frag)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_dgrad_bnbwd(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), ptr(res_bits), ptr(up_x),
                                           ptr(up_bits), ptr(up_stats), ptr(partial), partial.numel(), ctypes.byref(nt),
                                           stream_ptr()), "iif_conv_igemm_dgrad_bnbwd")
    return nt.value


# ------------------------------------------------------------ BN backward through the expanding 1x1 layer, by algebra
def conv_forward_stats_only(x, w, partial):
    """Pass 1 of the two-pass conv + BN forward (1x1 / stride 1, bf16): the partial rows of conv_forward_bnstats, no output."""
    require_gpu(x, w, partial)
    n, h, w_, cin = x.shape
    cout, ldw = w.shape
    d = _desc(n, h, w_, cin, h, w_, cout, 1, 1, 1, 0, 0, ldw, dtype_code(x), dtype_code(x), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_stats_only(ctypes.byref(d), ptr(x), ptr(w), ptr(partial), partial.numel(), ctypes.byref(nt),
                                          stream_ptr()), "iif_conv_igemm_stats_only")
    return nt.value


def conv_forward_bn_relu(x, w, out, stats, res=None, relu_bits=None):
    """Pass 2: out = relu(stats[2] * bf16(conv) + stats[3] + res), ReLU decisions into relu_bits (1x1 / stride 1, bf16)."""
    require_gpu(x, w, out, res, stats)
    n, h, w_, cin = x.shape
    cout, ldw = w.shape
    d = _desc(n, h, w_, cin, h, w_, cout, 1, 1, 1, 0, 0, ldw, dtype_code(x), dtype_code(out), 1)
    check(lib().iif_conv_igemm_bn_relu(ctypes.byref(d), ptr(x), ptr(w), ptr(out), ptr(res), ptr(stats), ptr(relu_bits),
                                       stream_ptr()), "iif_conv_igemm_bn_relu")
    return out


def conv_fwdbn_ok(n, h, w, cin, cout, dtype):
    """Whether the never-stored forward (statistics from the accumulators + BN epilogue, register-weight kernel) takes a
    1x1 / stride-1 layer of this shape."""
    if dtype != torch.bfloat16:
        return False
    d = _desc(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 0, cin, _lib.IIF_BF16, _lib.IIF_BF16, 1)
    return bool(lib().iif_conv_fwdbn_ok(ctypes.byref(d)))


def conv_pro_ok(n, h, w, cin, cout, dtype, stats_only):
    """Whether the register-weight kernel takes this 1x1 layer with the previous unit's BN + ReLU in its operand path."""
    if dtype != torch.bfloat16:
        return False
    d = _desc(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 0, cin, _lib.IIF_BF16, _lib.IIF_BF16, 1)
    return bool(lib().iif_conv_pro_ok(ctypes.byref(d), 1 if stats_only else 0))


def conv_forward_bnstats_pro(x_raw, x_stats, act_out, act_bits, w, out, partial, act_csum=None):
    """1x1 forward (out None: statistics only, from the accumulators) of relu(bn(x_raw)); the activation and its ReLU bits are
    written to act_out / act_bits on the way (and its column-sum rows to act_csum); returns the partial row count."""
    require_gpu(x_raw, x_stats, act_out, act_bits, w, out, partial, act_csum)
    n, h, w_, cin = x_raw.shape
    cout, ldw = w.shape
    d = _desc(n, h, w_, cin, h, w_, cout, 1, 1, 1, 0, 0, ldw, dtype_code(x_raw), dtype_code(x_raw), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_bnstats_pro(ctypes.byref(d), ptr(x_raw), ptr(x_stats), ptr(act_out), ptr(act_bits), ptr(act_csum), ptr(w),
                                           ptr(out), ptr(partial), partial.numel(), ctypes.byref(nt), stream_ptr()),
          "iif_conv_igemm_bnstats_pro")
    return nt.value


def conv_forward_stats_acc(x, w, partial):
    """Pass 1 on the register-weight kernel: partial rows of (sum, sum of squares) of the unrounded accumulators; no output."""
    require_gpu(x, w, partial)
    n, h, w_, cin = x.shape
    cout, ldw = w.shape
    d = _desc(n, h, w_, cin, h, w_, cout, 1, 1, 1, 0, 0, ldw, dtype_code(x), dtype_code(x), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_stats_acc(ctypes.byref(d), ptr(x), ptr(w), ptr(partial), partial.numel(), ctypes.byref(nt),
                                         stream_ptr()), "iif_conv_igemm_stats_acc")
    return nt.value


def conv_forward_bn_relu2(x, w, out, stats, relu_bits, res=None, res_stats=None):
    """Pass 2: out = relu(stats[2] * bf16(conv) + stats[3] + r), r = res or res_stats[2] * res + res_stats[3]."""
    require_gpu(x, w, out, res, stats, relu_bits)
    n, h, w_, cin = x.shape
    cout, ldw = w.shape
    d = _desc(n, h, w_, cin, h, w_, cout, 1, 1, 1, 0, 0, ldw, dtype_code(x), dtype_code(out), 1)
    check(lib().iif_conv_igemm_bn_relu2(ctypes.byref(d), ptr(x), ptr(w), ptr(out), ptr(res), ptr(res_stats), ptr(stats),
                                        ptr(relu_bits), stream_ptr()), "iif_conv_igemm_bn_relu2")
    return out


def conv_dgrad_masksum(dy, wt, in_hw, out, up_bits, partial, res=None, res_bits=None, up_x=None, up_stats=None):
    """1x1 / stride-1 conv_dgrad whose result is stored gated by ``up_bits`` (the ReLU decisions of the block output it is the
    gradient of) and whose per-tile column sums go to ``partial`` (second half of every row: zero, or with ``up_x`` /
    ``up_stats`` the sums of gradient * xhat); returns the row count."""
    require_gpu(dy, wt, res, out, up_x)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    d = _desc(n, ho, wo, cout, h, w_, cin, 1, 1, 1, 0, 1, ldw, dtype_code(dy), dtype_code(out), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_dgrad_masksum(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), ptr(res_bits), ptr(up_x),
                                             ptr(up_bits), ptr(up_stats), ptr(partial), partial.numel(), ctypes.byref(nt),
                                             stream_ptr()),
          "iif_conv_igemm_dgrad_masksum")
    return nt.value


def conv_dgrad_rx_ok(n, h, w, cs, cd, c2, dtype):
    """Whether iif_conv_igemm_dgrad_masksum_rx takes a 1x1 data gradient cs -> cd with the upstream x recomputed over c2 channels."""
    if dtype != torch.bfloat16:
        return False
    d = _desc(n, h, w, cs, h, w, cd, 1, 1, 1, 0, 1, cs, _lib.IIF_BF16, _lib.IIF_BF16, 1)
    return bool(lib().iif_conv_dgrad_rx_ok(ctypes.byref(d), c2))


def conv_dgrad_masksum_rx(dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats, res=None, res_bits=None):
    """conv_dgrad_masksum with (sum g~, sum g~ xhat) rows whose xhat comes from the upstream conv3 RECOMPUTED per tile
    (up_a2 [n, h, w, c2], up_w3 [cd, ldw3] bf16) instead of read from memory; returns the row count."""
    require_gpu(dy, wt, res, out, up_a2, up_w3)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    d = _desc(n, ho, wo, cout, h, w_, cin, 1, 1, 1, 0, 1, ldw, dtype_code(dy), dtype_code(out), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_dgrad_masksum_rx(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), ptr(res_bits), ptr(up_a2),
                                                up_a2.shape[3], ptr(up_w3), up_w3.stride(0), ptr(up_bits), ptr(up_stats), ptr(partial),
                                                partial.numel(), ctypes.byref(nt), stream_ptr()), "iif_conv_igemm_dgrad_masksum_rx")
    return nt.value


def conv_dgrad_rx_pg_ok(n, h, w, cs, cd, c2, dtype):
    """Whether the recomputing producer also has the instance that leaves P = g~^T a2 and Gram = a2^T a2 behind."""
    if dtype != torch.bfloat16:
        return False
    d = _desc(n, h, w, cs, h, w, cd, 1, 1, 1, 0, 1, cs, _lib.IIF_BF16, _lib.IIF_BF16, 1)
    return bool(lib().iif_conv_dgrad_rx_pg_ok(ctypes.byref(d), c2))


def conv_dgrad_masksum_rx_pg(dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats, pg_slabs, pg_ld, res=None, res_bits=None):
    """conv_dgrad_masksum_rx that also writes one fp32 slab [(cd + c2), pg_ld] per tile sequence into ``pg_slabs`` (P = out^T a2
    rows first, then Gram = a2^T a2); returns (partial row count, slab count).  ``slab_sum`` adds the slabs up."""
    require_gpu(dy, wt, res, out, up_a2, up_w3, pg_slabs)
    n, ho, wo, cout = dy.shape
    cin, ldw = wt.shape
    h, w_ = in_hw
    d = _desc(n, ho, wo, cout, h, w_, cin, 1, 1, 1, 0, 1, ldw, dtype_code(dy), dtype_code(out), 1)
    nt, ns = ctypes.c_int32(0), ctypes.c_int32(0)
    check(lib().iif_conv_igemm_dgrad_masksum_rx_pg(ctypes.byref(d), ptr(dy), ptr(wt), ptr(out), ptr(res), ptr(res_bits), ptr(up_a2),
                                                   up_a2.shape[3], ptr(up_w3), up_w3.stride(0), ptr(up_bits), ptr(up_stats), ptr(partial),
                                                   partial.numel(), ctypes.byref(nt), ptr(pg_slabs), pg_slabs.numel(), pg_ld,
                                                   ctypes.byref(ns), stream_ptr()), "iif_conv_igemm_dgrad_masksum_rx_pg")
    return nt.value, ns.value


def slab_sum(slabs, n, rows, ld, cols, out):
    """out[r, c] = sum of the first ``n`` fp32 slabs [rows, ld] of ``slabs`` (columns < cols), in slab order."""
    require_gpu(slabs, out)
    check(lib().iif_slab_sum(ptr(slabs), slabs.numel(), n, rows, ld, cols, ptr(out), stream_ptr()), "iif_slab_sum")
    return out


def conv_dgrad2_bnbwd(src, src2, wt, bias, out, up_x=None, up_bits=None, up_stats=None, partial=None):
    """out[m, j] = sum_k [src | src2][m, k] * wt[j, k] + bias[j]  (1x1, bf16); with up_x: the upstream BN-backward sums too."""
    require_gpu(src, src2, wt, out, up_x)
    n, h, w_, c1 = src.shape
    c2 = src2.shape[3]
    cout, ldw = wt.shape
    d = _desc(n, h, w_, c1, h, w_, cout, 1, 1, 1, 0, 1, ldw, dtype_code(src), dtype_code(out), 1)
    nt = ctypes.c_int32(0)
    check(lib().iif_conv_igemm_dgrad2_bnbwd(ctypes.byref(d), ptr(src), ptr(src2), c2, ptr(wt), ptr(bias), ptr(out), ptr(up_x),
                                            ptr(up_bits), ptr(up_stats), ptr(partial), 0 if partial is None else partial.numel(),
                                            ctypes.byref(nt), stream_ptr()), "iif_conv_igemm_dgrad2_bnbwd")
    return nt.value


def bn3_algebra_prep_scratch(C, c, device):
    return torch.empty(lib().iif_bn3_algebra_prep_scratch_floats(C, c), dtype=torch.float32, device=device)


def bn3_algebra_prep(P, w_bf16, c, partial, n_partials, stats, gamma, m, coef, dgamma, dbeta, wt, bias, scratch, tickets, colsum2=None):
    """P [C, ldp] fp32 (or None: sum g~ xhat comes from the second half of the partial rows), w_bf16 [C, ldw] (c valid
    columns), the producer's partial rows -> coef [3, C], dgamma, dbeta, the stacked bf16 weights wt [c, ldwt >= C + c]
    (g~ half and a2 half) and bias [c].  tickets: int32[64], zero before the first call (self-resetting).  colsum2 [c]:
    column sums of the data gradient's second source (a2): the bias then keeps the data gradient's column sums at zero."""
    C = w_bf16.shape[0]
    check(lib().iif_bn3_algebra_prep(ptr(P), 0 if P is None else P.stride(0), ptr(w_bf16), w_bf16.stride(0), ptr(partial), n_partials, ptr(stats),
                                     ptr(gamma), C, c, int(m), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(wt), wt.stride(0), ptr(bias),
                                     ptr(scratch), scratch.numel(), ptr(tickets), ptr(colsum2), stream_ptr()), "iif_bn3_algebra_prep", tickets)


def bn3_algebra_dw(P, w_bf16, c, gram, csum, coef, dW):
    C = P.shape[0]
    check(lib().iif_bn3_algebra_dw(ptr(P), P.stride(0), ptr(w_bf16), w_bf16.stride(0), ptr(gram), gram.stride(0), ptr(csum), ptr(coef),
                                   C, c, ptr(dW), dW.stride(0), stream_ptr()), "iif_bn3_algebra_dw")


def bn_backward_partials(gy, relu_bits, x2d, stats, gamma, partial, n_partials, dgamma, dbeta, dx, ws, tickets=None):
    m, c = x2d.shape
    check(lib().iif_bn_backward_partials_fused(ptr(gy), ptr(relu_bits), ptr(x2d), dtype_code(x2d), m, c, ptr(stats), ptr(gamma),
                                               ptr(partial), n_partials, ptr(dgamma), ptr(dbeta), ptr(dx), ptr(ws), ws.numel(),
                                               ptr(tickets), stream_ptr()), "iif_bn_backward_partials_fused", tickets)
    return dx
