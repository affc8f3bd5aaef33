// Batch-norm backward THROUGH the expanding 1x1 convolution of a bottleneck without re-reading the convolution's output.
//
// Reference: Bottleneck.forward, classification/resnet_pytorch.py:149-169 (conv3 -> bn3 -> += identity -> relu), backward
// as autograd derives it.  With y = a2 W^T the output of conv3 ([M, C], C = 4c), a2 its input ([M, c]) and g~ the gradient of
// the block output already gated by the block's ReLU, training-mode BN backward is affine per channel in (g~, y):
//     dy = A o g~ + B o y + D,   A = gamma invstd,  B = -A mean(g~ xhat) invstd,  D = -A mean(g~) - B mu
// and y is a product of the NARROW tensor, so everything the backward pass needs from y follows from three small matrices:
//     P    = g~^T a2        [C, c]   (the weight-gradient GEMM, computed anyway)
//     Gram = a2^T a2        [c, c],  csum = colsum(a2) [c]
//     sum_m g~ y  = rowdot(P, W)                         -> mean(g~ xhat) without a pass over y
//     dW   = diag(A) P + diag(B) W Gram + D (x) csum      -> the weight gradient
//     da2  = g~ (A o W) + a2 (W^T diag(B) W) + D W        -> the data gradient: ONE GEMM over [g~ | a2] with stacked weights
// y is never read in backward: 27 -> 12 channel-widths of HBM traffic for this unit of a bottleneck.
// The kernels below are the small fp32 pieces; the GEMMs are iif_conv_wgrad and iif_conv_igemm_dgrad2_bnbwd.
// W is the bf16 copy the forward pass multiplied with (so rowdot(P, W) is the sum over the y the statistics saw, up to the
// bf16 rounding of the stored y).  Fixed summation orders: deterministic.
#include "common.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block = 32 channels x 32 columns (grid C/32 x c/32).  Phase 1 (8 threads per channel, every block of a channel group
// repeats it: a lone block walking all c / 32 column tiles was a chain of dependent round trips, 26-40 us inside the step):
// sum_g from the `slices` rows of column sums, sum g~ xhat from their second half or, with P, from
// sgy = sum_j P[ch][j] W[ch][j]; the coefficients; the blocks of column tile 0 write coef / dgamma / dbeta.
// Phase 2: this block's tile of the g~ half of the stacked data-gradient weights, wt[j][ch] = bf16(A[ch] W[ch][j]),
// transposed through LDS, and of BW[ch][j] = bf16(B[ch] W[ch][j]) (the scaled operand of the Gm product).
__global__ void __launch_bounds__(256) bn3_coef_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* sg_slices,
                                                       int slices, int sg_pitch, const float* stats, const float* gamma, int C, int c,
                                                       double count, float* coef, float* dgamma, float* dbeta, unsigned short* wt,
                                                       int ldwt, unsigned short* BW) {
    __shared__ float cA[32], cB[32];
    __shared__ float tile[32][33];
    const int ch0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    // phase 2's operand first: its loads fly while phase 1 reduces (one dependent round trip instead of two)
    float wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = ch0 + ty + 8 * i, j = j0 + tx;
        wv[i] = (ch < C && j < c) ? bf16_bits_to_f32(W[(int64_t)ch * ldw + j]) : 0.f;
    }
    {
        const int lc = threadIdx.x >> 3, sub = threadIdx.x & 7;
        const int ch = ch0 + lc;
        float acc = 0.f, s1 = 0.f, sq = 0.f;
        if (ch < C) {
            if (P)
                for (int j = sub; j < c; j += 8) acc += P[(int64_t)ch * ldp + j] * bf16_bits_to_f32(W[(int64_t)ch * ldw + j]);
            for (int r = sub; r < slices; r += 8) {
                s1 += sg_slices[(int64_t)r * sg_pitch + ch];
                sq += sg_slices[(int64_t)r * sg_pitch + C + ch];
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); s1 += __shfl_xor(s1, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (sub == 0 && ch < C) {
            const float mu = stats[ch], invstd = stats[C + ch];
            const float s2 = P ? invstd * (acc - mu * s1) : sq;        // sum g~ xhat
            const float A = gamma[ch] * invstd;
            const float B = -A * (float)((double)s2 / count) * invstd;
            const float D = -A * (float)((double)s1 / count) - B * mu;
            if (blockIdx.y == 0) {
                dbeta[ch] = s1;
                dgamma[ch] = s2;
                coef[ch] = A; coef[C + ch] = B; coef[2 * C + ch] = D;
            }
            cA[lc] = A; cB[lc] = B;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        const int ch = ch0 + r, j = j0 + tx;
        if (ch < C && j < c) BW[(int64_t)ch * c + j] = f32_to_bf16_bits(cB[r] * wv[i]);
        tile[r][tx] = (ch < C) ? cA[r] * wv[i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, ch = ch0 + tx;
        if (j < c && ch < C) wt[(int64_t)j * ldwt + ch] = f32_to_bf16_bits(tile[tx][r]);
    }
}

// 64 x 64 output tile (4 x 4 per thread) of  Out[r][q] = sum_k L(r, k) R(k, q),  k staged through LDS 32 at a time.
// LF(row, k) / RF(k, col) return 0 outside the problem.  After every staged chunk `each(k0)` runs with the chunk still in LDS
// (rs[k][col] = R(k0 + k, q0 + col)).
constexpr int kTP = 68;                                  // LDS row pitch in floats: 16-byte aligned rows, conflict-free float4 reads
template <typename LF, typename RF, typename EF>
__device__ __forceinline__ void tile_gemm64(int kbeg, int K, LF lf, RF rf, EF each, float (&acc)[4][4], float (*ls)[kTP],
                                            float (*rs)[kTP]) {
    const int tj = (threadIdx.x >> 4) * 4, ti = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) acc[a_][b_] = 0.f;
    for (int k0 = kbeg; k0 < K; k0 += 64) {
        // TWO chunks' loads in flight before the first LDS write: on a loaded memory system a round trip costs 3-5 us and the
        // arithmetic of a chunk 0.5 us, so the depth of the dependent chain is what a launch takes
        float lv[2][8], rv[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = threadIdx.x + t * 256;
                lv[h][t] = lf(q & 63, k0 + 32 * h + (q >> 6));      // (lf / rf return 0 past K)
                rv[h][t] = rf(k0 + 32 * h + (q >> 6), q & 63);
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (k0 + 32 * h >= K) break;
            __syncthreads();                             // the previous chunk has been consumed
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = threadIdx.x + t * 256;
                ls[q >> 6][q & 63] = lv[h][t];
                rs[q >> 6][q & 63] = rv[h][t];
            }
            __syncthreads();
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const float4 l4 = *reinterpret_cast<const float4*>(&ls[r][tj]);
                const float4 r4 = *reinterpret_cast<const float4*>(&rs[r][ti]);
                const float l_[4] = {l4.x, l4.y, l4.z, l4.w}, r_[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
                    for (int b_ = 0; b_ < 4; ++b_) acc[a_][b_] += l_[a_] * r_[b_];
            }
            each(k0 + 32 * h);
        }
    }
}

// Gm[jo][i] = sum_ch BW[ch][jo] W[ch][i]  ->  wt[jo][C + i] (bf16);  bias[jo] = sum_ch D[ch] W[ch][jo].
// Two launches, because a lone block walking all C channels is a chain of C / 32 dependent memory round trips on a loaded
// memory system (160 us measured in the step):  (1) grid ((c/64)^2, C/kGmK): every block one kGmK-channel slice of one 64 x 64
// tile -> fp32 slab [slice][c][c], the diagonal tiles also their slice of the bias -> [slice][c] behind the slabs;
// (2) sum of the slices in a fixed order (deterministic), bf16 conversion into the stacked weights.
constexpr int kGmK = 64;
__global__ void __launch_bounds__(256) bn3_gm_slab_kernel(const unsigned short* W, int ldw, const unsigned short* BW, const float* coef,
                                                          int C, int c, float* slab, float* bias_slab) {
    __shared__ __attribute__((aligned(16))) float ls[32][kTP], rs[32][kTP];
    __shared__ float dsl[kGmK];
    const int nt = (c + 63) / 64;
    const int jo0 = (blockIdx.x / nt) * 64, i0 = (blockIdx.x % nt) * 64;
    const int kb = blockIdx.y * kGmK;
    const int ke = kb + kGmK < C ? kb + kGmK : C;
    const bool diag = jo0 == i0;
    // (D of this channel slice goes through LDS with the first batch of loads: read inside the chunk loop it was one more
    // dependent round trip per chunk; the first barrier of tile_gemm64 publishes it)
    if (threadIdx.x < kGmK) dsl[threadIdx.x] = kb + (int)threadIdx.x < C ? coef[2 * C + kb + threadIdx.x] : 0.f;
    float bacc = 0.f;
    float acc[4][4];
    tile_gemm64(
        kb, ke,
        [&](int row, int ch) { return (ch < ke && jo0 + row < c) ? bf16_bits_to_f32(BW[(int64_t)ch * c + jo0 + row]) : 0.f; },
        [&](int ch, int col) { return (ch < ke && i0 + col < c) ? bf16_bits_to_f32(W[(int64_t)ch * ldw + i0 + col]) : 0.f; },
        [&](int ch0) {
            if (diag && threadIdx.x < 64) {
                const int lim = ke - ch0 < 32 ? ke - ch0 : 32;
                for (int r = 0; r < lim; ++r) bacc += dsl[ch0 - kb + r] * rs[r][threadIdx.x];
            }
        },
        acc, ls, rs);
    const int tj = (threadIdx.x >> 4) * 4, ti = (threadIdx.x & 15) * 4;
    float* out = slab + (int64_t)blockIdx.y * c * c;
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_) {
        const int jo = jo0 + tj + a_, i = i0 + ti;
        if (jo < c && i < c) *reinterpret_cast<float4*>(out + (int64_t)jo * c + i) = make_float4(acc[a_][0], acc[a_][1], acc[a_][2], acc[a_][3]);
    }
    if (diag && threadIdx.x < 64 && jo0 + (int)threadIdx.x < c) bias_slab[(int64_t)blockIdx.y * c + jo0 + threadIdx.x] = bacc;
}

__global__ void __launch_bounds__(256) bn3_gm_finish_kernel(const float* slab, const float* bias_slab, int slices, int C, int c,
                                                            unsigned short* wt, int ldwt, float* bias) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < c * c) {
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += slab[(int64_t)k * c * c + q];
        wt[(int64_t)(q / c) * ldwt + C + q % c] = f32_to_bf16_bits(s);
    }
    if (q < c) {
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += bias_slab[(int64_t)k * c + q];
        bias[q] = s;
    }
}

// dW[ch][j] = A P[ch][j] + B sum_i W[ch][i] Gram[i][j] + D csum[j].  Grid (C / 64, c / 64).
__global__ void __launch_bounds__(256) bn3_dw_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* gram,
                                                     int ldg, const float* csum, const float* coef, int C, int c, float* dW,
                                                     int lddw) {
    __shared__ __attribute__((aligned(16))) float ls[32][kTP], rs[32][kTP];
    const int ch0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    float acc[4][4];
    tile_gemm64(
        0, c,
        [&](int row, int i) { return (ch0 + row < C && i < c) ? bf16_bits_to_f32(W[(int64_t)(ch0 + row) * ldw + i]) : 0.f; },
        [&](int i, int col) { return (i < c && j0 + col < c) ? gram[(int64_t)i * ldg + j0 + col] : 0.f; },
        [](int) {}, acc, ls, rs);
    const int tj = (threadIdx.x >> 4) * 4, ti = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_) {
        const int ch = ch0 + tj + a_;
        if (ch >= C) continue;
        const float A = coef[ch], B = coef[C + ch], D = coef[2 * C + ch];
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) {
            const int j = j0 + ti + b_;
            if (j < c) dW[(int64_t)ch * lddw + j] = A * P[(int64_t)ch * ldp + j] + B * acc[a_][b_] + D * csum[j];
        }
    }
}

// stage 1 of the column sums of many partial rows ([2][C] each): slice blockIdx.y of the rows, 32 of the 2 C columns per
// block -> out[slice][2 C]
__global__ void __launch_bounds__(256) bn3_slice_sums_kernel(const float* partial, int nrows, int C2, int rows_per_slice, float* out) {
    __shared__ float sh[256];
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5;
    const int ch = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per_slice;
    int r1 = r0 + rows_per_slice; if (r1 > nrows) r1 = nrows;
    float a = 0.f;
    if (ch < C2)
        for (int r = r0 + ln; r < r1; r += 8) a += partial[(int64_t)r * C2 + ch];
    sh[ln * 32 + cl] = a;
    __syncthreads();
    if (ln == 0 && ch < C2) {
        float s = 0.f;
        for (int q = 0; q < 8; ++q) s += sh[q * 32 + cl];
        out[(int64_t)blockIdx.y * C2 + ch] = s;
    }
}

}  // namespace

extern "C" {

int iif_bn3_algebra_coef(const float* P, int ldp, const void* w_bf16, int ldw, const float* partial, int n_partials,
                         const float* stats, const float* gamma, int C, int c, int64_t m, float* coef, float* dgamma, float* dbeta,
                         void* wt, int ldwt, void* bw, float* scratch, int64_t scratch_floats, void* stream) {
    if (!w_bf16 || !partial || !stats || !gamma || !coef || !dgamma || !dbeta || !wt || !bw || !scratch || C <= 0 || c <= 0 ||
        m <= 0 || n_partials <= 0)
        return IIF_EINVAL;
    if (c > 256 || (c % 32) || (P && ldp < c) || ldw < c || ldwt < C + c) return IIF_EUNSUPPORTED;
    const int slices = n_partials < 64 ? n_partials : 64;
    if ((int64_t)slices * 2 * C > scratch_floats) return IIF_EINVAL;
    const int rps = (n_partials + slices - 1) / slices;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(bn3_slice_sums_kernel, dim3((2 * C + 31) / 32, slices), dim3(256), 0, st, partial, n_partials, 2 * C, rps, scratch);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn3_coef_kernel, dim3((C + 31) / 32, (c + 31) / 32), dim3(256), 0, st, P, ldp, (const unsigned short*)w_bf16, ldw, scratch, slices, 2 * C,
                       stats, gamma, C, c, (double)m, coef, dgamma, dbeta, (unsigned short*)wt, ldwt, (unsigned short*)bw);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int64_t iif_bn3_algebra_gm_scratch_floats(int C, int c) { return (int64_t)((C + kGmK - 1) / kGmK) * ((int64_t)c * c + c); }

int iif_bn3_algebra_gm(const void* w_bf16, int ldw, const void* bw, const float* coef, int C, int c, void* wt, int ldwt, float* bias,
                       float* scratch, int64_t scratch_floats, void* stream) {
    if (!w_bf16 || !bw || !coef || !wt || !bias || !scratch || C <= 0 || c <= 0) return IIF_EINVAL;
    if (c > 256 || (c % 4) || ldw < c || ldwt < C + c) return IIF_EUNSUPPORTED;
    if (scratch_floats < iif_bn3_algebra_gm_scratch_floats(C, c)) return IIF_EINVAL;
    const int slices = (C + kGmK - 1) / kGmK, nt = (c + 63) / 64;
    float* bias_slab = scratch + (int64_t)slices * c * c;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(bn3_gm_slab_kernel, dim3(nt * nt, slices), dim3(256), 0, st, (const unsigned short*)w_bf16, ldw,
                       (const unsigned short*)bw, coef, C, c, scratch, bias_slab);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn3_gm_finish_kernel, dim3((c * c + 255) / 256), dim3(256), 0, st, scratch, bias_slab, slices, C, c,
                       (unsigned short*)wt, ldwt, bias);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn3_algebra_dw(const float* P, int ldp, const void* w_bf16, int ldw, const float* gram, int ldg, const float* csum,
                       const float* coef, int C, int c, float* dW, int lddw, void* stream) {
    if (!P || !w_bf16 || !gram || !csum || !coef || !dW || C <= 0 || c <= 0) return IIF_EINVAL;
    if (c > 256 || ldp < c || ldw < c || ldg < c || lddw < c) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(bn3_dw_kernel, dim3((C + 63) / 64, (c + 63) / 64), dim3(256), 0, as_stream(stream), P, ldp, (const unsigned short*)w_bf16,
                       ldw, gram, ldg, csum, coef, C, c, dW, lddw);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
