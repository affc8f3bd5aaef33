/*
 * iif_amd.h — C ABI of libiif_amd.so: the MI355X (gfx950) kernels behind the IIF
 * training hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions
 *   - every `d_*` / device pointer is BORROWED: owned by the caller, must stay
 *     alive until the work enqueued on `stream` has completed; nothing is
 *     allocated or freed inside the library and no call synchronises.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     All work is enqueued on it; calls are re-entrant across streams and keep
 *     no global or thread-local state, so they may be captured into a hipGraph.
 *   - return value: IIF_OK (0) or a negative IIF_E* code; nothing throws.
 *   - dtype codes: IIF_F32 / IIF_BF16.  Activations are NHWC ("channels last"),
 *     weights KRSC ([Cout][R][S][Cin/groups]); this is the layout the kernels
 *     are tiled for (16-byte channel vectors feed the MFMA K dimension).
 *
 * Each entry point names the reference interface it replaces (file:line under
 * kostas1515/iif).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 */
#ifndef IIF_AMD_H
#define IIF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IIF_OK 0
#define IIF_EINVAL (-1)       /* bad argument (null pointer, size <= 0, bad enum) */
#define IIF_EUNSUPPORTED (-2) /* shape / alignment the kernels are not built for */
#define IIF_ELAUNCH (-3)      /* hipLaunch / hip runtime error */

#define IIF_F32 0
#define IIF_BF16 1

/* table variants: classification/custom.py:16-23 */
#define IIF_RAW 0
#define IIF_SMOOTH 1
#define IIF_REL 2
#define IIF_NORMIT 3
#define IIF_GOMBIT 4
#define IIF_BASE2 5
#define IIF_BASE10 6

/* library / build info: "iif_amd <version> gfx950" */
const char* iif_version(void);

/* ------------------------------------------------------------------ IIF head */

/* HOST.  Per-class IIF weights from class counts.
 * Replaces the table construction of classification/custom.py:14-26
 * (float64 arithmetic, one cast to float32, optional division by the p-norm
 * of the float32 vector when norm_p > 0).  out_host: float[C]. */
int iif_build_table(const int64_t* counts_host, int C, int variant, int norm_p, float* out_host);

/* Fused IIF softmax cross-entropy, forward + gradient in ONE pass over logits.
 * Replaces classification/custom.py:28-36 (pred*iif -> CrossEntropyLoss('none',
 * weight) -> mean/sum), custom.py:116-117 (mixup criterion: pass targets_b and
 * lam) and mmdet/models/losses/iif_loss.py:184-202 + losses/utils.py:42-55
 * (row weights, ignore_index, avg_factor folded into `scale`).
 *
 *   z_i      = logits_i * table                               (row i, C classes)
 *   nll(i,t) = logsumexp(z_i) - z_i[t]   (0 if t == ignore_index)
 *   r_i      = row_weight_i * ( lam*cw[ta_i]*nll(i,ta_i) + (1-lam)*cw[tb_i]*nll(i,tb_i) )
 *   loss     = scale * sum_i r_i            (scale = 1/B for 'mean', 1 for 'sum',
 *                                            loss_weight/avg_factor for mmdet)
 *   dlogits[i,c] = scale * d r_i / d logits[i,c]
 *
 * logits/dlogits: [B, C] row-major with leading dimensions ld_logits / ld_dlogits
 * (elements), dtype IIF_F32 or IIF_BF16 (math is fp32 either way).
 * targets_b == NULL means no mixup (lam ignored).  row_weight / class_weight may
 * be NULL (= ones).  loss_per_row: float[B] (required; receives r_i, the
 * reduction='none' value).  loss_out: float[1] or NULL.  dlogits may be NULL
 * (loss only).  d_status: int32[1] or NULL; set to 1 if any target is outside
 * [0,C) and != ignore_index (such rows contribute 0).
 * d_workspace: IIF_CE_WORKSPACE_BYTES of device memory or NULL.  Its first int32 is a
 * ticket that must be ZERO on entry (zero it once; the kernel leaves it zero again),
 * the rest holds one partial sum per block.  With a workspace the scalar loss comes
 * out of the SAME launch (the last block to finish sums the per-block partials);
 * without it a second, one-block launch sums loss_per_row.  A workspace must not be
 * shared by calls that can run concurrently (one per stream).
 * Deterministic either way (fixed-order trees, no float atomics); the two paths
 * associate the sum differently and may differ in the last fp32 bits.
 * The softmax runs in base 2 on v_exp_f32 / v_log_f32 (~1 ulp each). */
int iif_ce_fwd_bwd(const void* logits, int dtype, int64_t ld_logits,
                   const float* table, const int64_t* targets_a, const int64_t* targets_b,
                   float lam, const float* row_weight, const float* class_weight,
                   int64_t ignore_index, float scale, int B, int C,
                   float* loss_per_row, float* loss_out,
                   void* dlogits, int64_t ld_dlogits, int32_t* d_status, void* d_workspace, void* stream);
#define IIF_CE_WORKSPACE_BYTES (4 * (1 + 2048))

/* out = logits * table.  Replaces classification/custom.py:37-39 (infer=True). */
int iif_scale_logits(const void* logits, int dtype, int64_t ld_logits, const float* table,
                     int B, int C, void* out, int64_t ld_out, void* stream);

/* out = softmax(logits * table, dim=-1), fp32 out.
 * Replaces mmdet/models/losses/iif_loss.py:65-78 (get_activation). */
int iif_softmax(const void* logits, int dtype, int64_t ld_logits, const float* table,
                int B, int C, float* out, int64_t ld_out, void* stream);

/* Top-k hit counts.  hits[j] += #rows whose target ranks < k[j] in
 * logits*table (table NULL = raw logits); rank = #greater + #equal-with-lower-index.
 * Replaces classification/utils.py:165-179 and mmdet/models/losses/accuracy.py:7-51
 * (callers turn counts into percent).  hits: int32[nk], zeroed by the caller;
 * integer atomics => exact and order independent.  nk <= 4. */
int iif_topk_hits(const void* logits, int dtype, int64_t ld_logits, const float* table,
                  const int64_t* targets, int B, int C, const int32_t* k_host, int nk,
                  int32_t* hits, void* stream);

/* out[i] = x[i] * *d_scalar (device scalar; out may alias x).  Used by the autograd
 * bridge to apply the upstream gradient of the scalar loss without a host sync. */
int iif_scale_by_device_scalar(const void* x, int dtype, int64_t n, const float* d_scalar, void* out, void* stream);

/* out[b,:] = lam*x[b,:] + (1-lam)*x[perm[b],:], rows of n elements.
 * Replaces the image blend of classification/custom.py:112 (Mixup.__call__). */
int iif_mix_rows(const void* x, int dtype, const int64_t* perm, float lam, int B, int64_t n,
                 void* out, void* stream);

/* ------------------------------------------------------- ResNet forward/backward */

/* Geometry of one convolution-shaped contraction (all tensors NHWC).
 * "source" is what the taps gather from, "destination" is the output grid. */
typedef struct iif_conv_desc {
    int32_t n, hs, ws, cs;   /* source      [n, hs, ws, cs]                       */
    int32_t hd, wd, cd;      /* destination [n, hd, wd, cd]                       */
    int32_t r, s;            /* taps                                               */
    int32_t stride, pad;     /* of the FORWARD convolution (stride 1 or 2)         */
    int32_t transposed;      /* 0: ys = y*stride - pad + r   (forward)             */
                             /* 1: ys = (y + pad - r)/stride (data gradient)       */
    int32_t ldw;             /* weight row pitch in elements (>= r*s*cs, 16-B mult) */
    int32_t dtype;           /* IIF_F32 / IIF_BF16 of src and wgt                  */
    int32_t dst_dtype;       /* dtype of dst / res: == dtype, or IIF_F32           */
    int32_t groups;          /* 0/1: dense.  G > 1: cs / cd are PER-GROUP widths, the   */
                             /* tensors hold G*cs / G*cd channels per pixel, wgt is G   */
                             /* consecutive [cd][ldw] matrices (grouped convolution,    */
                             /* resnet_pytorch.py:137,141 ResNeXt)                      */
    const void* wgt_frag;    /* nullable: the SAME weights as MFMA fragments            */
                             /* (iif_conv_pack_fragments); used by the 3x3 / stride-1   */
                             /* kernel where iif_conv3x3_frag_ok(d) says so, ignored    */
                             /* elsewhere.  wgt must be valid either way.               */
    int32_t wgt_frag_kind;   /* 0: iif_conv_pack_fragments' format.  1 (round 6): the  */
                             /* grouped 16-channel format of iif_conv_pack_fragments_g16 */
                             /* (groups > 1, cs = cd = 64, every group <= 16 channels). */
} iif_conv_desc;

/* Implicit-GEMM convolution on the matrix cores:
 *   dst[m, k] = sum_{r,s,c} gather(src)[m; r,s,c] * wgt[k][r][s][c]  (+ bias[k]) (+ res[m, k])
 * wgt: [cd][ldw] rows of r*s*cs K-contiguous elements.  With transposed=0 and
 * KRSC weights this is conv2d forward (resnet_pytorch.py:46-62 conv3x3/conv1x1,
 * resnet_cifar.py:112-115); with transposed=1 and [cin][R][S][cout] weights it is
 * the data gradient; r=s=1 on a [B,1,1,C] tensor is the fully connected layer
 * (resnet_pytorch.py:219, resnet_cifar.py:192) with `bias`.  res (nullable, same
 * shape/dtype as dst, may alias dst) is added in the epilogue: gradient
 * accumulation at residual joins.  bf16 inputs multiply on v_mfma_f32_16x16x32_bf16
 * with fp32 accumulation; f32 inputs on v_mfma_f32_16x16x4_f32 (exact fp32). */
int iif_conv_igemm(const iif_conv_desc* d, const void* src, const void* wgt, void* dst,
                   const void* res, const float* bias, void* stream);

/* iif_conv_igemm that also emits batch-norm statistics of its (bf16) output from the
 * epilogue: bn_partial[t][0][k] / [t][1][k] = sum / sum of squares of the stored output
 * over pixel tile t (128 consecutive pixels) for channel k; *n_partials (host int,
 * nullable) receives the tile count.  Feed them to iif_bn_finalize_stats: no separate
 * pass over the activation.  bf16 in/out, cd % 8 == 0, no bias/res; otherwise
 * IIF_EUNSUPPORTED.  bn_partial must hold >= ceil(m/128)*2*cd floats. */
int iif_conv_igemm_bnstats(const iif_conv_desc* d, const void* src, const void* wgt, void* dst,
                           const void* res, const float* bias, float* bn_partial,
                           int64_t bn_partial_floats, int32_t* n_partials, void* stream);

/* Weight gradient: dw[k][r][s][c] = sum_m dy[m, k] * gather(x)[m; r,s,c], fp32 out.
 * d describes the FORWARD convolution (source = x, destination = dy grid,
 * transposed must be 0).  dw: float [cd][ldw] (columns >= r*s*cs are not
 * written).  The pixel reduction is split over `splits` workgroup rows
 * (0 = choose: one co-resident round of workgroups on the whole device; -n =
 * choose for 1/n of the device, for a caller that runs n weight gradients side
 * by side on n streams - every workgroup writes its accumulator tile once, so
 * n half-size rounds write 1/n of the slab bytes each) into fp32 slabs in
 * `workspace` (>= splits*cd*ldw*4 bytes; fewer
 * splits are used if it is smaller, NULL = no split) that are then summed in
 * fixed order: deterministic, no float atomics.  bf16 fragments are formed by
 * ds_read_b64_tr_b16.  Replaces the wgrad half of autograd's conv backward for
 * resnet_pytorch.py:46-62 / resnet_cifar.py:112-115 and the fc layer. */
int iif_conv_wgrad(const iif_conv_desc* d, const void* x, const void* dy, float* dw,
                   void* workspace, int64_t workspace_bytes, int splits, void* stream);

/* 1x1 weight gradient against TWO gradient tensors stacked along the channels, one pass over x (bf16):
 *   out[k][:] = sum_m dy[m][k] x[m][:] for k < cd1,   out[cd1 + j][:] = sum_m dy2[m][j] x[m][:] for j < cd2;
 * out float [cd1 + cd2][ldw], split-K slabs and `splits` as iif_conv_wgrad.  The BN-by-algebra backward
 * (iif_bn3_algebra_*) takes P = g~^T a2 and the Gram matrix a2^T a2 from one launch this way (dy = g~, dy2 = x = a2).
 * cd1 % 128 == 0, cs % 8 == 0, cd2 % 8 == 0; otherwise IIF_EUNSUPPORTED. */
int iif_wgrad1x1_stacked(const void* x, const void* dy, const void* dy2, int64_t m, int cs, int cd1, int cd2,
                         int ldw, float* out, void* workspace, int64_t workspace_bytes, int splits,
                         void* stream);

/* Training-mode batch norm, NHWC activation viewed as x[m, c] (m = N*H*W).
 * Replaces F.batch_norm(training=True) + ReLU + residual add and their autograd
 * backward under resnet_pytorch.py:152-167 / resnet_cifar.py:133-138.
 *
 * forward_stats: batch mean / biased variance per channel (fixed-order partial
 *   sums, fp64 finalisation), running stats updated in place with `momentum`
 *   (unbiased variance, as torch), and `stats` = float[4*c]:
 *   mean | invstd | a = gamma*invstd | b = beta - mean*a.
 * apply: y = act(a*x + b + R) with R = 0, `residual`, or a2*residual + b2 when
 *   residual_stats (another BN's stats block) is given; act = ReLU if relu.
 * backward: given gy = dL/dy (y = the activated output), y_mask (the stored
 *   activated output; NULL = no ReLU), the BN input x and `stats`:
 *   dy = gy*[y_mask>0]; dgamma = sum dy*xhat; dbeta = sum dy;
 *   dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)).
 *   gmasked (nullable, may alias gy) receives dy — the gradient that also flows
 *   into the residual branch.  dx may alias gy when gmasked is NULL.
 * relu_bits (apply: nullable output, backward: nullable input that replaces y_mask): one byte per
 *   16-byte channel vector (8 bf16 / 4 f32 channels), bit k = [pre-activation k > 0] — the backward
 *   passes then read 1/16 of the mask bytes.  m*c/8 (bf16) or m*c/4 (f32) bytes.
 * workspace: iif_bn_workspace_bytes(m, c) bytes. */
int64_t iif_bn_workspace_bytes(int64_t m, int c);
int iif_bn_forward_stats(const void* x, int dtype, int64_t m, int c, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* stats, void* workspace,
                         int64_t workspace_bytes, void* stream);
int iif_bn_apply(const void* x, int dtype, int64_t m, int c, const float* stats,
                 const void* residual, const float* residual_stats, int relu, void* y,
                 uint8_t* relu_bits, void* stream);
/* second half of iif_bn_forward_stats on externally produced partial sums
 * (partial[t][0][k] = sum, [t][1][k] = sum of squares; fixed-order fp64 reduction).
 * scratch (nullable, >= 128*c floats) lets > 512 partial rows be reduced in two
 * parallel stages instead of one serial walk. */
int iif_bn_finalize_stats(const float* partial, int n_partials, int64_t m, int c, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, float* stats, float* scratch, int64_t scratch_floats,
                          void* stream);
/* iif_bn_finalize_stats with both reduction stages in ONE launch when there are > 512 partial rows: the 64 slice sums
 * are published with agent-scope atomics and the last block of every 32-channel group (a ticket per group) finishes
 * the statistics.  tickets: int32[64], ZERO on entry (zero it once; the kernel leaves it zero), not shared by calls
 * that can run concurrently; NULL = the two-launch path.  Same arithmetic (fixed-order fp64 sums). */
int iif_bn_finalize_stats_fused(const float* partial, int n_partials, int64_t m, int c, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean,
                                float* running_var, float* stats, float* scratch, int64_t scratch_floats,
                                int32_t* tickets, void* stream);
/* Round 6: iif_bn_finalize_stats_fused plus a second, independent job in the same launch: sums2[2][c2] = column sums of the partial
 * rows partial2 [n_partials2][2][c2] (iif_bn_partial_sums' result) - the column sums of a2 that conv3's prologue emits
 * (iif_conv_igemm_bnstats_pro's act_csum) are reduced by the finalisation that follows that launch anyway. */
int iif_bn_finalize_stats_sums(const float* partial, int n_partials, int64_t m, int c, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* stats,
                               float* scratch, int64_t scratch_floats, int32_t* tickets, const float* partial2, int n_partials2,
                               int c2, float* sums2, void* stream);
int iif_bn_backward(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x,
                    int dtype, int64_t m, int c, const float* stats, const float* gamma, float* dgamma,
                    float* dbeta, void* dx, void* gmasked, void* workspace, int64_t workspace_bytes,
                    void* stream);

/* Max pooling k x k / stride / pad on NHWC (resnet_pytorch.py:206 MaxPool2d(3,2,1)).
 * argmax: uint8 per output element = kh*k + kw of the first maximum in scan
 * order (torch's tie rule); backward routes gy to that input position. */
int iif_maxpool_forward(const void* x, int dtype, int n, int h, int w, int c, int k, int stride,
                        int pad, void* y, uint8_t* argmax, void* stream);
int iif_maxpool_backward(const void* gy, const uint8_t* argmax, int dtype, int n, int h, int w, int c,
                         int k, int stride, int pad, void* dx, void* stream);

/* Global average pooling over hw pixels (resnet_pytorch.py:211 AdaptiveAvgPool2d(1),
 * resnet_cifar.py:209 avg_pool2d). */
int iif_avgpool_forward(const void* x, int dtype, int n, int hw, int c, void* y, void* stream);
int iif_avgpool_backward(const void* gy, int dtype, int n, int hw, int c, void* dx, void* stream);

/* Stem patches: NCHW fp32 image -> [n*ho*wo][kp] matrix, column (r*S+s)*cin + c,
 * zero padded to kp, so the few-channel first convolution (resnet_pytorch.py:203
 * 7x7/2, resnet_cifar.py:179 3x3) runs as a K-contiguous MFMA GEMM. */
int iif_im2col_nchw(const float* img, int n, int cin, int h, int w, int r, int s, int stride, int pad,
                    int kp, int out_dtype, void* out, void* stream);

/* Element-wise precision cast (f32 -> bf16 | f32, bf16 -> f32). */
int iif_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* fp32 master weights [cout][ldw] (r,s,cin order) -> [cin][ldwt] (r,s,cout order) in
 * out_dtype: the operand layout of the data-gradient contraction. */
int iif_weight_transpose(const float* w, int cout, int cin, int rs, int ldw, int ldwt, int out_dtype,
                         void* wt, void* stream);

/* Option-A shortcut of the CIFAR ResNet (resnet_cifar.py:125-126):
 * y[n,y,x,c] = x[n,2y,2x,c-(cout-cin)/2] inside the channel band, else 0;
 * backward_acc adds g back into dx at the even pixels. */
int iif_shortcut_a_forward(const void* x, int dtype, int n, int h, int w, int cin, int cout, void* y,
                           void* stream);
int iif_shortcut_a_backward_acc(const void* g, int dtype, int n, int h, int w, int cin, int cout,
                                void* dx, void* stream);

/* out[c] = sum_r a[r*ld + c] (bias gradient of the fc layer). */
int iif_colsum_f32(const float* a, int rows, int cols, int64_t ld, float* out, void* stream);

/* One fused SGD update over a flat fp32 arena (all parameters of the model in
 * one launch).  torch.optim.SGD semantics as used at classification/train.py:199-204
 * (dampening 0; a zero-initialised momentum buffer reproduces buf = grad on the
 * first step): d = grad_scale*g + wd*p; buf = m*buf + d;
 * p -= lr*(nesterov ? d + m*buf : buf).  d_lr (nullable) overrides lr with a
 * device scalar so a captured graph can follow the schedule. */
int iif_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr,
                 const float* d_lr, float momentum, float weight_decay, int nesterov,
                 float grad_scale, void* stream);

/* Cosine / normed classifier heads (resnet_cifar.py:38-78 CosNorm_Classifier,
 * NormedLinear; mmdet normed_predictor.py).  Row maps and their backward; the
 * products run on iif_conv_igemm / iif_conv_wgrad.
 *   mode 0: out = x * scale/(1+|x|)           (cosine feature squashing)
 *   mode 1: out = x / max(|x|, eps)           (F.normalize; zero rows stay zero)
 * norms (nullable in forward) receives |x| per row, fp32.
 * backward: dx = d(out)/dx^T g given the forward input x and its stored norms. */
int iif_rowmap_forward(const void* x, int x_dtype, int rows, int cols, int64_t ldx, int mode,
                       float scale, float eps, void* out, int out_dtype, int64_t ldo, float* norms,
                       void* stream);
int iif_rowmap_backward(const void* x, int x_dtype, const float* norms, const void* g, int g_dtype,
                        int rows, int cols, int64_t ldx, int64_t ldg, int mode, float scale, float eps,
                        void* dx, int dx_dtype, int64_t lddx, void* stream);
/* out[c][r] = in[r][c] (fp32); NormedLinear keeps its weight [in][out]. */
int iif_transpose_f32(const float* in, int rows, int cols, int64_t ldi, float* out, int64_t ldo,
                      void* stream);
/* out[0] = alpha / (*d_alpha_div or 1) * sum_{r<rows,c<cols} a[r][c]*b[r][c], fixed order
 * (gradient of the learnable cosine scale). */
int iif_dot_window_f32(const float* a, const float* b, int rows, int cols, int64_t lda, int64_t ldb,
                       float alpha, const float* d_alpha_div, float* out, void* stream);

/* Grouped convolution (ResNeXt, resnet_pytorch.py:137,141) on the MFMA kernels: channels are
 * cut into chunks of `chunk` (64) and each chunk runs as a dense convolution whose weights are
 * block-diagonal over the groups inside it (iif_conv_desc.groups = channels/chunk, cs = cd = chunk).
 * iif_group_pack: fp32 master weights [channels][ldm] (rows of (tap, cin_local < cg)) -> packed
 * [channels][ldp] rows of (tap, chunk-local channel); transposed=1 gives the data-gradient operand
 * (rows = input channels, columns = (tap, chunk-local output channel)).
 * iif_group_unpack_grad: dense-in-chunk weight gradient -> master layout (in-group entries). */
int iif_group_pack(const float* master, int channels, int cg, int chunk, int rs, int ldm, int ldp,
                   int transposed, int out_dtype, void* out, void* stream);
int iif_group_unpack_grad(const float* packed, int channels, int cg, int chunk, int rs, int ldp,
                          int ldm, float* master, void* stream);
/* iif_group_pack for every grouped layer in ONE launch: `table` = `entries` device-resident
 * iif_group_pack_entry records (all outputs of dtype out_dtype), `blocks_per_entry` 256-thread
 * blocks walk each entry.  Same arithmetic as iif_group_pack, entry by entry. */
typedef struct iif_group_pack_entry {
    const float* master;   /* fp32 [channels][ldm] */
    void* out;             /* [channels][ldp] */
    int32_t channels, cg, chunk, rs, ldm, ldp, transposed, reserved;
} iif_group_pack_entry;
int iif_group_pack_batched(const void* table, int entries, int blocks_per_entry, int out_dtype,
                           void* stream);

/* Stem as a space-to-depth convolution: the RxR / stride-2 / pad-(R-1)/2 convolution on a c-channel
 * NCHW fp32 image (resnet_pytorch.py:203: 7x7/2 on 3 channels) equals an AxA / stride-1 / pad-A/2
 * convolution, A = (R+1)/2, on the 2x2 space-to-depth image [n, h/2, w/2, cpad] (channel (di*2+dj)*c + ch,
 * zero padded to cpad), with output grid h/2 x w/2.  No patch matrix is written.
 * iif_space_to_depth_nchw builds that image, iif_stem_s2d_pack the [k][A*A*cpad] weight rows from the
 * master [k][ldm] (r, s, c) rows, iif_stem_s2d_unpack_grad maps the weight gradient back. */
int iif_space_to_depth_nchw(const float* img, int n, int c, int h, int w, int cpad, int out_dtype,
                            void* out, void* stream);
int iif_stem_s2d_pack(const float* master, int k, int c, int r, int ldm, int cpad, int out_dtype,
                      void* out, void* stream);
int iif_stem_s2d_unpack_grad(const float* packed, int k, int c, int r, int cpad, int ldm, float* master,
                             void* stream);

/* Squeeze-and-excitation around the last BN of a residual block (SE_Block.forward,
 * classification/resnet_pytorch.py:313-317 / resnet_cifar.py:102-106, inside SEBottleneck.forward :358-381
 * and Se_Block.forward resnet_cifar.py:163-169).  x is the raw convolution output [n, hw, c] (NHWC), stats
 * the BN stats block (scale at [2c], shift at [3c]), excite / offset fp32 [n, c].
 *   iif_se_squeeze        sums[n,c] = sum_hw x
 *   iif_se_apply          y = relu((a*x + b) * excite[n,c] + identity), identity = residual or
 *                         a2*residual + b2 (residual_stats), relu_bits = 1 bit per element
 *   iif_se_backward_sums  g <- g * [y > 0] in place; s1[n,c] = sum_hw g; s2[n,c] = sum_hw g*x
 *   iif_se_backward_form  out = g * excite[n,c] + offset[n,c]  (gradient w.r.t. the BN output)
 * The [n, c]-sized excitation (two bias-free linears, ReLU, sigmoid; resnet_pytorch.py:306-311, 315), fp32:
 *   iif_se_excite_forward   q = a * sums / hw + b (mean of the BN output);  h = relu(W1 q);  e = sigmoid(W2 h)
 *   iif_se_excite_backward  from s1 / s2 of iif_se_backward_sums: dz2, dz1 (scratch [n,c] / [n,hid]), offset = (dz1 W1) / hw
 *                           (the `offset` of iif_se_backward_form) and the weight gradients dw1 [hid][ldg1], dw2 [c][ldg2]
 * w1 is W1 [hid][ld1]; w2t is the TRANSPOSE of W2, [hid][ldt] (iif_transpose_f32), so both matrices are read along c. */
int iif_se_squeeze(const void* x, int dtype, int n, int hw, int c, float* sums, void* stream);
int iif_se_apply(const void* x, int dtype, int n, int hw, int c, const float* stats, const float* excite,
                 const void* residual, const float* residual_stats, void* y, unsigned char* relu_bits,
                 void* stream);
int iif_se_backward_sums(void* g, const unsigned char* relu_bits, const void* x, int dtype, int n, int hw,
                         int c, float* s1, float* s2, void* stream);
int iif_se_backward_form(const void* g, int dtype, int n, int hw, int c, const float* excite,
                         const float* offset, void* out, void* stream);
int iif_se_excite_forward(const float* sums, const float* stats, int n, int hw, int c, int hid, const float* w1, int ld1,
                          const float* w2t, int ldt, float* q, float* h, float* e, void* stream);
int iif_se_excite_backward(const float* s1, const float* s2, const float* stats, int n, int hw, int c, int hid, const float* w1,
                           int ld1, const float* w2t, int ldt, const float* e, const float* h, const float* q, float* dz2,
                           float* dz1, float* offset, float* dw1, int ldg1, float* dw2, int ldg2, void* stream);

/* mmdet normed predictors (instance_segmentation/mmdet/models/utils/normed_predictor.py: NormedLinear :34-40,
 * IIFNormedLinear :67-73, NormedConv2d with a 1x1 kernel :104-112), fp32:
 *   v = row_scale[row] * x[row] (row_scale nullable = 1);  out = v * scale / (|v|^power + eps);  norms[row] = |v|
 * backward: dx = d(out)/dx^T g.  The products run on iif_conv_igemm / iif_conv_wgrad. */
int iif_rownorm_forward(const float* x, const float* row_scale, int rows, int cols, int64_t ldx, float power,
                        float scale, float eps, float* out, int64_t ldo, float* norms, void* stream);
int iif_rownorm_backward(const float* x, const float* row_scale, const float* norms, const float* g, int rows,
                         int cols, int64_t ldx, int64_t ldg, float power, float scale, float eps, float* dx,
                         int64_t lddx, void* stream);

/* All dense convolutions' transposed copies ([cin][rs*cout], for the data gradient) in ONE launch.  `table` is a
 * DEVICE array of n_desc descriptors sorted by block_start; descriptor i owns the blocks
 * [block_start_i, block_start_{i+1}): rs * ceil(cin/32) * ceil(cout/32) of them, one 32x32 tile of one tap each
 * (tap-major, then cin tiles, then cout tiles).
 * src_off / dst_off are element offsets into the fp32 parameter arena / the output arena. */
typedef struct iif_wt_desc {
    int64_t src_off, dst_off;
    int32_t cout, cin, rs, ldw, ldwt, block_start;
} iif_wt_desc;
int iif_weight_transpose_batched(const float* arena, const iif_wt_desc* table, int n_desc, int total_blocks,
                                 int out_dtype, void* out, void* stream);

/* Weights as ready-made MFMA fragments for the 3x3 / stride-1 / pad-1 bf16 kernel (conv3x3 of resnet_pytorch.py:46-57,
 * forward and data gradient).  Source: bf16 rows [rows][ld] of `taps` x `k` channels — the forward weights
 * [cout][9 * cin] or the transposed copy [cin][9 * cout] of iif_weight_transpose.  Destination, rows * taps * k elements:
 * fragment (row / 16, tap, channel / 32) = 1 KB, lane l's 16 bytes (row l & 15, channels (l >> 4) * 8 .. + 8) at l * 16,
 * so that a wavefront fetches one operand of v_mfma_f32_16x16x32_bf16 with one coalesced load and the weights never pass
 * through LDS.  rows % 16 == 0, k % 32 == 0.  `table` is a DEVICE array of descriptors (element offsets into src_base /
 * dst_base, block_start = first 256-thread block of the descriptor, ascending); total_blocks = sum of
 * ceil(rows * taps * k / 8 / 256).  iif_conv3x3_frag_ok: 1 when a convolution with this descriptor (wgt_frag set) runs on
 * the fragment kernel, 0 when it would take the row-weights path. */
typedef struct iif_pack_desc {
    int64_t src_off, dst_off;
    int32_t rows, taps, k, ld, block_start, reserved;
} iif_pack_desc;
int iif_conv_pack_fragments(const void* src_base, const iif_pack_desc* table, int n_desc, int total_blocks, void* dst_base,
                            void* stream);
int iif_conv3x3_frag_ok(const iif_conv_desc* d);
/* Round 6, ResNeXt's narrow groups (resnet_pytorch.py:137,141 with groups = 32, base width 4: 4 / 8 / 16 channels per group): the
 * block-diagonal 64-channel chunk matrices of iif_group_pack re-packed as 20 fragments per chunk (4 output tiles x 5 tap PAIRS:
 * K of an MFMA = two taps x the tile's own 16 input channels), for iif_conv_desc.wgt_frag with wgt_frag_kind = 1.  Table entries
 * as for iif_conv_pack_fragments (rows = channels of the layer, taps = 9, k = 64, ld = row pitch of the chunk matrix); one
 * 256-thread block per 4 fragments: blocks of an entry = ceil(rows / 64 * 20 / 4).  Valid where every group is <= 16 channels wide. */
int iif_conv_pack_fragments_g16(const void* src_base, const iif_pack_desc* table, int n_desc, int total_blocks, void* dst_base,
                                void* stream);

/* Batch-norm backward through the expanding 1x1 convolution of a bottleneck (conv3 -> bn3 -> += identity -> relu,
 * classification/resnet_pytorch.py:160-167) WITHOUT re-reading the convolution's output y = a2 W^T.  BN backward is affine per
 * channel in (g~, y), dy = A o g~ + B o y + D, with g~ the block-output gradient already gated by the block's ReLU bits, so
 * with P = g~^T a2 (iif_conv_wgrad), Gram = a2^T a2 (iif_conv_wgrad of a2 with itself) and csum = colsum(a2) (iif_bn_stats_sums):
 *   sum g~ y = rowdot(P, W);  dW = diag(A) P + diag(B) W Gram + D (x) csum;  da2 = [g~ | a2] [A o W ; W^T diag(B) W]^T + D W.
 *   iif_conv_igemm_dgrad_masksum  the data gradient that PRODUCES the block-output gradient stores it gated by up_bits and
 *                                 emits per tile into `partial` either (sum dst, 0) (up_x NULL: no read of y) or, with up_x /
 *                                 up_stats (y and its batch statistics), (sum dst, sum dst * xhat) as iif_conv_igemm_dgrad_bnbwd;
 *   iif_bn3_algebra_prep          those partial rows, W (the bf16 copy the forward used) and EITHER P (sum g~ y = rowdot(P, W):
 *                                 y is never read, but P is needed before the data gradient) OR P = NULL (sum g~ xhat from the
 *                                 second half of the rows: y is read once by the producer and P is only needed for dW, off the
 *                                 critical path) -> coef [3][C] = (A, B, D), dgamma, dbeta and the stacked bf16 weights
 *                                 wt [c][ldwt >= C + c]: wt[j][ch] = A[ch] W[ch][j], wt[j][C + i] = sum_ch bf16(B[ch] W[ch][j]) W[ch][i],
 *                                 and bias[j] = sum_ch D[ch] W[ch][j].  Two launches (one ticketed kernel over channel groups x row
 *                                 slices, one slab sum); scratch: iif_bn3_algebra_prep_scratch_floats(C, c) floats; tickets:
 *                                 int32[64], zero on entry, zero again on exit.  c in {64, 128, 256}, C <= 4096.  With colsum2
 *                                 (sum over the pixels of the second K source of the data gradient, i.e. of a2) the bias also absorbs
 *                                 what the bf16 rounding of wt adds to the COLUMN SUMS of the data gradient (exactly zero in exact
 *                                 arithmetic): bias[j] -= (sum_i colsum2[i] d2[j][i] + sum_ch sum(g~)[ch] d1[j][ch]) / m;
 *   iif_conv_igemm_dgrad2_bnbwd   dst = [src | src2] wgt^T + bias (1x1, stride 1, bf16; K runs over src's cs then src2's cs2
 *                                 channels), optionally with the upstream BN-backward sums of iif_conv_igemm_dgrad_bnbwd;
 *   iif_bn3_algebra_dw            dW [C][lddw] from P, W, Gram, csum, coef.
 * c in {64, 128, 256} (the 56x56 ... 14x14 stages of the ImageNet networks).  Everything sums in a fixed order. */
/* Two-pass forward of conv + BN (+ identity) + ReLU for the expanding 1x1 layer of a bottleneck (resnet_pytorch.py:160-167),
 * bf16, 1x1 / stride 1: the raw convolution output is never stored.
 *   iif_conv_igemm_stats_only   pass 1: the per-tile (sum, sum of squares) partial rows of iif_conv_igemm_bnstats — same
 *                               bf16 rounding of the tile, same sums — without the store;
 *   iif_conv_igemm_bn_relu      pass 2 (after iif_bn_finalize_stats): dst = relu(fma(a, bf16(conv), b) + res), a / b at
 *                               stats[2 Cd + c] / stats[3 Cd + c], one byte of ReLU decisions per 8 channels into relu_bits
 *                               (nullable): bit-identical to iif_conv_igemm followed by iif_bn_apply with that residual. */
int iif_conv_igemm_stats_only(const iif_conv_desc* d, const void* src, const void* wgt, float* bn_partial,
                              int64_t bn_partial_floats, int32_t* n_partials, void* stream);
int iif_conv_igemm_bn_relu(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                           const float* stats, unsigned char* relu_bits, void* stream);
/* Round 6: the same two passes on the register-weight kernel (cs in {64, 128, 256}, cd a multiple of 256, n*hd*wd a multiple
 * of 64; iif_conv_fwdbn_ok says whether a descriptor qualifies), which is what makes the never-stored forward pay
 * (resnet_pytorch.py:160-167: conv3 -> bn3 -> += identity -> relu):
 *   iif_conv_igemm_stats_acc    pass 1: (sum, sum of squares) of the UNROUNDED fp32 accumulators, one partial row per tile
 *                               sequence, nothing staged or stored (the statistics of the convolution itself rather than of
 *                               its bf16 rounding; feed iif_bn_finalize_stats as usual);
 *   iif_conv_igemm_bn_relu2     pass 2: as iif_conv_igemm_bn_relu; with res_stats the residual is the RAW output of the
 *                               block's convolutional shortcut and is normalised on the way in, fma(a2, res, b2) with a2 / b2 at
 *                               res_stats[2 Cd + c] / [3 Cd + c] (iif_bn_apply's residual_stats arithmetic).  relu_bits required. */
int iif_conv_fwdbn_ok(const iif_conv_desc* d);
int iif_conv_igemm_stats_acc(const iif_conv_desc* d, const void* src, const void* wgt, float* bn_partial,
                             int64_t bn_partial_floats, int32_t* n_partials, void* stream);
int iif_conv_igemm_bn_relu2(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                            const float* res_stats, const float* stats, unsigned char* relu_bits, void* stream);
int iif_conv_igemm_dgrad_masksum(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                 const unsigned char* res_bits, const void* up_x, const unsigned char* up_bits,
                                 const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                                 void* stream);
/* Round 6: the batch norm + ReLU of the PREVIOUS unit applied in the convolution's operand path (resnet_pytorch.py:157-160:
 * bn2 -> relu -> conv3).  src_raw is that unit's raw convolution output [n, h, w, cs] and src_stats its statistics as
 * iif_bn_finalize_stats wrote them; each tile is normalised in LDS (iif_bn_apply's arithmetic) before the matrix pipe reads
 * it, and the activated tensor is written out as a by-product - act_out [n, h, w, cs] bf16 and act_bits (one byte per 8
 * channels), bit-identical to iif_bn_apply's - because backward needs it; act_csum (nullable) receives per partial row one
 * row [2][cs] = (its column sums, zeros), the layout iif_bn_partial_sums reduces.  dst NULL: the statistics-only pass of iif_conv_igemm_stats_acc; otherwise the stored forward
 * with statistics of iif_conv_igemm_bnstats.  One launch and one pass over the activation less per bottleneck.
 * iif_conv_pro_ok(d, stats_only) says whether the register-weight kernel has the instance. */
int iif_conv_pro_ok(const iif_conv_desc* d, int stats_only);
int iif_conv_igemm_bnstats_pro(const iif_conv_desc* d, const void* src_raw, const float* src_stats, void* act_out,
                               unsigned char* act_bits, float* act_csum, const void* wgt, void* dst, float* bn_partial,
                               int64_t bn_partial_floats, int32_t* n_partials, void* stream);
/* Round 6: iif_conv_igemm_dgrad_masksum whose (sum dst, sum dst * xhat) rows are formed WITHOUT the upstream block's conv3 output:
 * each tile of it is recomputed on the matrix pipe from that block's a2 (up_a2, [n*h*w, up_c2] bf16) and conv3 weights as the
 * forward multiplied them (up_w3, [cd, up_ldw3] bf16), rounded to bf16 as the stored tensor would have been.  Together with
 * iif_conv_igemm_stats_acc / iif_conv_igemm_bn_relu2 this lets the forward pass never store that output, with "sums from the
 * producer" (P and Gram off the critical path).  Register-weight kernel only: iif_conv_dgrad_rx_ok(d, up_c2) says whether the
 * (cs, up_c2) pair has an instance ((64, 64), (128, 64), (128, 128), (256, 128)). */
int iif_conv_dgrad_rx_ok(const iif_conv_desc* d, int up_c2);
int iif_conv_igemm_dgrad_masksum_rx(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                    const unsigned char* res_bits, const void* up_a2, int up_c2, const void* up_w3, int up_ldw3,
                                    const unsigned char* up_bits, const float* up_stats, float* partial, int64_t partial_floats,
                                    int32_t* n_partials, void* stream);
/* ... and with the two small matrices the algebraic BN3 backward of that upstream block needs as BY-PRODUCTS of the same launch
 * (classification/resnet_pytorch.py:162-163, conv3 + bn3 backward as autograd derives them): P = dst^T a2 ([cd, up_c2]: what the
 * conv3 weight-gradient GEMM computes) and Gram = a2^T a2 ([up_c2, up_c2]).  The block that forms a tile of dst holds it in its
 * staging buffers and the a2 tile in LDS; it accumulates both products over its tiles and writes ONE fp32 slab
 * [(cd + up_c2), pg_ld] (P rows first) per tile sequence into pg_slabs (pg_floats floats available; *n_slabs receives the slab count).
 * iif_slab_sum adds the slabs in sequence order (deterministic).  The stacked weight-gradient launch that re-read dst and a2 from
 * memory (0.5 GB per bottleneck at 56 x 56) is not needed.  iif_conv_dgrad_rx_pg_ok: up_c2 = 64, cd = 256, cs in {64, 128}. */
int iif_conv_dgrad_rx_pg_ok(const iif_conv_desc* d, int up_c2);
int iif_conv_igemm_dgrad_masksum_rx_pg(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                       const unsigned char* res_bits, const void* up_a2, int up_c2, const void* up_w3, int up_ldw3,
                                       const unsigned char* up_bits, const float* up_stats, float* partial, int64_t partial_floats,
                                       int32_t* n_partials, float* pg_slabs, int64_t pg_floats, int pg_ld, int32_t* n_slabs,
                                       void* stream);
/* out[r][c] = sum over the n slabs [rows, ld] of slabs[s][r][c], c < cols, in slab order (the split-K reduction of
 * iif_conv_wgrad for slabs another kernel wrote).  slab_floats: floats available behind `slabs`; with room for ceil(n / 16)
 * more slabs behind the n the reduction runs in two parallel stages. */
int iif_slab_sum(float* slabs, int64_t slab_floats, int n, int rows, int ld, int cols, float* out, void* stream);
int iif_conv_igemm_dgrad2_bnbwd(const iif_conv_desc* d, const void* src, const void* src2, int cs2, const void* wgt,
                                const float* bias, void* dst, const void* up_x, const unsigned char* up_bits,
                                const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                                void* stream);
int64_t iif_bn3_algebra_prep_scratch_floats(int C, int c);
int iif_bn3_algebra_prep(const float* P, int ldp, const void* w_bf16, int ldw, const float* partial, int n_partials,
                         const float* stats, const float* gamma, int C, int c, int64_t m, float* coef, float* dgamma, float* dbeta,
                         void* wt, int ldwt, float* bias, float* scratch, int64_t scratch_floats, int32_t* tickets,
                         const float* colsum2 /* nullable: colsum of the data gradient's second source, [c] */, void* stream);
int iif_bn3_algebra_dw(const float* P, int ldp, const void* w_bf16, int ldw, const float* gram, int ldg, const float* csum,
                       const float* coef, int C, int c, float* dW, int lddw, void* stream);

/* The convolution library reads its experiment / test switches (IIF_CONV_NO_STREAM1X1, IIF_CONV_STREAM1X1_FORCE,
 * IIF_CONV_NO_SHORTK, IIF_CONV_TWOSTAGE_K, IIF_CONV_FORCE_BN64, IIF_CONV_REGSTAGE, IIF_CONV_NO_V2) from the environment once, when it is
 * loaded: nothing on the launch path calls getenv.  A harness that changes one of them afterwards calls this to have them
 * read again.  Not needed (and not used) by the product path. */
int iif_conv_reload_env(void);

/* Data gradient with a MASKED residual: dst = dgrad(src, wgt) + res * [bit], where res_bits holds one ReLU
 * decision bit per element of res (the relu_bits of iif_bn_apply: one byte per 16-byte vector).  This is the
 * identity path of a residual block in backward (resnet_pytorch.py:163-167: out += identity; relu): the
 * gradient of the block output is gated by the block's ReLU while it is added, so no masked copy is ever
 * written.  transposed=1, stride 1 only. */
int iif_conv_igemm_masked_res(const iif_conv_desc* d, const void* src, const void* wgt, void* dst,
                              const void* res, const unsigned char* res_bits, void* stream);

/* Mask-side class-channel selection.  pred is [n, c, hw] (NCHW mask logits / probabilities,
 * fp32 or bf16), labels int64 [n] the IIF-derived class of every RoI.
 *   iif_mask_gather       out[n, hw] = pred[i, labels[i], :]        (FCNMaskHead.get_seg_masks,
 *                         mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:289-290)
 *   iif_mask_bce_fwd_bwd  loss = mean_i,p BCEWithLogits(pred[i, labels[i], p], target[i, p])   (mask_cross_entropy,
 *                         mmdet/models/losses/cross_entropy_loss.py:158-162); dpred (nullable, fp32 [n, c, hw],
 *                         ZERO-FILLED by the caller) receives grad_scale * d loss / d pred in the selected
 *                         channels only.  row_loss: [n] scratch.  An out-of-range label sets bit 0 of *status
 *                         (device int, caller-zeroed) and contributes nothing. */
int iif_mask_gather(const void* pred, int dtype, const int64_t* labels, int n, int c, int hw, float* out,
                    int* status, void* stream);
int iif_mask_bce_fwd_bwd(const void* pred, int dtype, const float* target, const int64_t* labels, int n, int c,
                         int hw, float grad_scale, float* row_loss, float* loss, float* dpred, int* status,
                         void* stream);

/* FASA side outputs of the IIF classifier loss (instance_segmentation/mmdet).
 *   iif_class_accumulate  FasaIIFLoss.forward with use_cums (losses/fasa_iif_loss.py:154-160): for every class
 *                         c in [0, C): cum_labels[c] += #{labels == c}, cum_losses[c] += sum of rows over them
 *                         (row order; labels outside [0, C) are skipped).
 *   iif_fasa_update       fa_update / fa_update_push (roi_heads/bbox_heads/fasa_bbox_head.py:118-147): per-class mean
 *                         and unbiased variance of embedding [n, d] rows, first sight -> copy, afterwards
 *                         exponential moving average with `decay`; feature_used[c] becomes 1.
 *   iif_fasa_generate     fa_generate (:149-172) with the random numbers supplied by the caller: classes with
 *                         rnd[c] < prob[c] and feature_used[c] > 0, ascending, get out[k] = mean + sqrt(var) *
 *                         normal[c], out_labels[k] = c; *count (device) = number of rows written (<= c).
 *                         slot_class: int32 [c] scratch. */
int iif_class_accumulate(const float* rows, const int64_t* labels, int n, int c, float* cum_losses,
                         float* cum_labels, void* stream);
int iif_fasa_update(const float* embedding, const int64_t* labels, int n, int d, int64_t ld, int c, float decay,
                    float* feature_mean, float* feature_var, float* feature_used, void* stream);
int iif_fasa_generate(const float* rnd, const float* prob, const float* feature_used, const float* feature_mean,
                      const float* feature_var, const float* normal, int c, int d, int* slot_class, int* count,
                      float* out, int64_t* out_labels, void* stream);

/* Data gradient that also emits the batch-norm BACKWARD partial sums of the upstream unit.  dst (= dL/dy of the
 * unit that produced this convolution's input) is gated by that unit's ReLU bits `up_bits` (nullable: no ReLU) and
 * reduced per pixel tile against its pre-normalisation output `up_x` (same shape as dst) and statistics
 * `up_stats` (mean at [c], invstd at [cd + c]): partial[t] = (sum g, sum g * xhat) per channel, *n_partials rows.
 * iif_bn_backward_partials then finishes that unit's BN backward without the reduction pass over dst and up_x
 * (what F.batch_norm's backward re-reads in resnet_pytorch.py:152-167).  bf16, cd % 8 == 0, dense;
 * res / res_bits as in iif_conv_igemm / iif_conv_igemm_masked_res; stride-2 launches visit every pixel once. */
int iif_conv_igemm_dgrad_bnbwd(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                               const unsigned char* res_bits, const void* up_x, const unsigned char* up_bits,
                               const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                               void* stream);
int iif_bn_backward_partials(const void* gy, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                             const float* stats, const float* gamma, const float* partial, int n_partials,
                             float* dgamma, float* dbeta, void* dx, void* workspace, int64_t workspace_bytes,
                             void* stream);
/* Cross-replica batch statistics (nn.SyncBatchNorm, classification/train.py:190-191): reduction and normalisation as
 * separate calls so that the host can all-reduce the per-channel sums of all ranks in between.
 *   iif_bn_partial_sums         sums[0][c] / sums[1][c] = column sums of partial rows [n][2][c] (fixed order, fp64)
 *   iif_bn_stats_sums           the same straight from x [m, c]: (sum x, sum x^2); workspace as iif_bn_forward_stats
 *   forward: all-reduce sums, then iif_bn_finalize_stats(sums, 1, m * world, ...) (one "partial row", global count)
 *   iif_bn_backward_sums        (sum g, sum g*xhat) of this rank, g gated by y_mask > 0 or relu_bits
 *   iif_bn_backward_apply_sums  dgamma / dbeta from the LOCAL sums (the gradient all-reduce averages them as it does every
 *                               parameter gradient), dx with mean(g), mean(g*xhat) from the all-reduced total_sums and
 *                               total_count = m * world; gmasked nullable; coef_scratch: 3*c floats */
int iif_bn_partial_sums(const float* partial, int n_partials, int c, float* sums, void* stream);
int iif_bn_stats_sums(const void* x, int dtype, int64_t m, int c, float* sums, void* workspace, int64_t workspace_bytes,
                      void* stream);
int iif_bn_backward_sums(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x, int dtype, int64_t m,
                         int c, const float* stats, float* sums, void* workspace, int64_t workspace_bytes, void* stream);
int iif_bn_backward_apply_sums(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x, int dtype,
                               int64_t m, int c, const float* stats, const float* gamma, const float* local_sums,
                               const float* total_sums, double total_count, float* dgamma, float* dbeta, void* dx,
                               void* gmasked, float* coef_scratch, void* stream);
/* the same with the slice reduction and the finalisation of > 512 partial rows in one launch (tickets as above) */
int iif_bn_backward_partials_fused(const void* gy, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                                   const float* stats, const float* gamma, const float* partial, int n_partials,
                                   float* dgamma, float* dbeta, void* dx, void* workspace, int64_t workspace_bytes,
                                   int32_t* tickets, void* stream);

/* Stem without a stored activation (resnet_pytorch.py:284-287: bn1 -> relu -> maxpool):
 *   iif_maxpool_bn_forward           max pool over relu(a*x + b) of the RAW convolution output x (stats: scale at [2c],
 *                                    shift at [3c]); candidates are rounded to the storage type first, so value and
 *                                    argmax equal iif_bn_apply followed by iif_maxpool_forward
 *   iif_bn_backward_relu_recompute   iif_bn_backward with the ReLU mask recomputed as a*x + b > 0 (same fmaf)
 *   iif_bn_backward_relu_recompute_pooled   the same, its two column sums taken from the POOLED gradient g_pool and the RAW stem
 *                                    output at each window's arg max, pool_x ([pool_pixels][c] each; written by
 *                                    iif_maxpool_bn_forward when its pool_x argument is given) instead of a reduction pass over
 *                                    gy and x: sum g = sum g_pool [a x* + b > 0], sum g xhat = sum g_pool [a x* + b > 0] (x* - mean) invstd
 *                                    - term for term what the reduction pass adds, in another order.  gy is the pooled gradient
 *                                    already scattered to the stem's resolution (iif_maxpool_backward) */
int iif_maxpool_bn_forward(const void* x, int dtype, const float* stats, int n, int h, int w, int c, int k, int stride,
                           int pad, void* y, uint8_t* argmax, void* pool_x /* nullable */, void* stream);
int iif_bn_backward_relu_recompute(const void* gy, const void* x, int dtype, int64_t m, int c, const float* stats,
                                   const float* gamma, float* dgamma, float* dbeta, void* dx, void* workspace,
                                   int64_t workspace_bytes, void* stream);
int iif_bn_backward_relu_recompute_pooled(const void* gy, const void* x, int dtype, int64_t m, int c, const float* stats,
                                          const float* gamma, float* dgamma, float* dbeta, void* dx, void* workspace,
                                          int64_t workspace_bytes, const void* g_pool, const void* pool_x, int64_t pool_pixels,
                                          void* stream);
/* iif_maxpool_backward (3x3 / stride 2 / pad 1) + iif_bn_backward_relu_recompute_pooled in one: the pooled gradient g_pool
 * [n][ho][wo][c] is gathered through `argmax` inside the normalisation pass, the scattered gradient at the stem's resolution
 * [n][h][w][c] is never formed.  dx and dgamma / dbeta bit-identical to the two calls (the gathered value is rounded to the
 * storage type as the stored tensor was).  workspace >= (min(512, ceil(n*ho*wo/64)) * 2c + 3c) * 4 bytes. */
int iif_bn_backward_pool_fused(const void* g_pool, const uint8_t* argmax, const void* pool_x, const void* x, int dtype, int n,
                               int h, int w, int c, int ho, int wo, const float* stats, const float* gamma, float* dgamma,
                               float* dbeta, void* dx, void* workspace, int64_t workspace_bytes, void* stream);

/* Compute-unit budget of the persistent grids (the weights-in-registers kernels, the stem, the streaming 1x1 kernel size their
 * grids to one or two resident blocks per CU).  Process-wide, default 0 = every CU of the device; a rank whose gradient
 * all-reduce (RCCL kernels, classification/train.py:230-234 DDP) overlaps backward sets e.g. 240 so that the reduction's
 * channels find free CUs instead of queueing behind a resident grid.  Rounded down to a multiple of 8, at least 64.
 * iif_get_cu_budget: the count the next persistent launch will use. */
int iif_set_cu_budget(int cus);
int iif_get_cu_budget(void);

#ifdef __cplusplus
}
#endif
#endif /* IIF_AMD_H */
