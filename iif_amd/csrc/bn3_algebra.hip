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

// one wave per channel: sgy = sum_j P[ch][j] W[ch][j]; coefficients; dgamma / dbeta
__global__ void __launch_bounds__(256) bn3_coef_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* sg,
                                                       const float* stats, const float* gamma, int C, int c, double count,
                                                       float* coef, float* dgamma, float* dbeta) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= C) return;                              // wave-uniform
    float acc = 0.f;
    for (int j = lane; j < c; j += 64) acc += P[(int64_t)ch * ldp + j] * bf16_bits_to_f32(W[(int64_t)ch * ldw + j]);
    acc = wsum(acc);
    if (lane != 0) return;
    const float mu = stats[ch], invstd = stats[C + ch];
    const float s1 = sg[ch];
    const float s2 = invstd * (acc - mu * s1);        // sum g~ xhat
    dbeta[ch] = s1;
    dgamma[ch] = s2;
    const float A = gamma[ch] * invstd;
    const float B = -A * (float)((double)s2 / count) * invstd;
    const float D = -A * (float)((double)s1 / count) - B * mu;
    coef[ch] = A; coef[C + ch] = B; coef[2 * C + ch] = D;
}

// wt[j][ch] = bf16(A[ch] W[ch][j]): the g~ half of the stacked data-gradient weights, transposed through a 32 x 32 LDS tile
__global__ void __launch_bounds__(256) bn3_scaled_transpose_kernel(const unsigned short* W, int ldw, const float* coef, int C, int c,
                                                                   unsigned short* wt, int ldwt) {
    __shared__ float tile[32][33];
    const int ch0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ch = ch0 + r, j = j0 + tx;
        tile[r][tx] = (ch < C && j < c) ? coef[ch] * bf16_bits_to_f32(W[(int64_t)ch * ldw + j]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, ch = ch0 + tx;
        if (j < c && ch < C) wt[(int64_t)j * ldwt + ch] = f32_to_bf16_bits(tile[tx][r]);
    }
}

// block = output row jo of the data gradient: Gm[jo][i] = sum_ch W[ch][jo] B[ch] W[ch][i] -> wt[jo][C + i];
// bias[jo] = sum_ch D[ch] W[ch][jo].  Threads over i (c <= 256).
__global__ void __launch_bounds__(256) bn3_gm_kernel(const unsigned short* W, int ldw, const float* coef, int C, int c,
                                                     unsigned short* wt, int ldwt, float* bias) {
    __shared__ float red[256];
    const int jo = blockIdx.x, i = threadIdx.x;
    float acc = 0.f, bpart = 0.f;
    const float* B = coef + C;
    const float* D = coef + 2 * C;
    for (int ch = 0; ch < C; ++ch) {
        const float wj = bf16_bits_to_f32(W[(int64_t)ch * ldw + jo]);        // block-uniform address: one scalar-like load
        if (i < c) acc += (wj * B[ch]) * bf16_bits_to_f32(W[(int64_t)ch * ldw + i]);
    }
    for (int ch = i; ch < C; ch += 256) bpart += D[ch] * bf16_bits_to_f32(W[(int64_t)ch * ldw + jo]);
    if (i < c) wt[(int64_t)jo * ldwt + C + i] = f32_to_bf16_bits(acc);
    red[i] = bpart;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (i < o) red[i] += red[i + o];
        __syncthreads();
    }
    if (i == 0) bias[jo] = red[0];
}

// dW[ch][j] = A P[ch][j] + B sum_i W[ch][i] Gram[i][j] + D csum[j].  Block = 256 / c channels... one channel per c threads.
__global__ void __launch_bounds__(256) bn3_dw_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* gram,
                                                     int ldg, const float* csum, const float* coef, int C, int c, float* dW,
                                                     int lddw) {
    __shared__ float wrow[256];
    const int per = 256 / c > 0 ? 256 / c : 1;        // channels per block (c <= 256)
    const int lc = threadIdx.x / c, j = threadIdx.x % c;
    const int ch = blockIdx.x * per + lc;
    const bool on = lc < per && ch < C;
    if (on) wrow[threadIdx.x] = bf16_bits_to_f32(W[(int64_t)ch * ldw + j]);
    __syncthreads();
    if (!on) return;
    float t = 0.f;
    const float* wr = wrow + lc * c;
    for (int i = 0; i < c; ++i) t += wr[i] * gram[(int64_t)i * ldg + j];
    dW[(int64_t)ch * lddw + j] = coef[ch] * P[(int64_t)ch * ldp + j] + coef[C + ch] * t + coef[2 * C + ch] * csum[j];
}

}  // namespace

extern "C" {

int iif_bn3_algebra_coef(const float* P, int ldp, const void* w_bf16, int ldw, const float* sum_g, const float* stats,
                         const float* gamma, int C, int c, int64_t m, float* coef, float* dgamma, float* dbeta, void* wt,
                         int ldwt, void* stream) {
    if (!P || !w_bf16 || !sum_g || !stats || !gamma || !coef || !dgamma || !dbeta || !wt || C <= 0 || c <= 0 || m <= 0) return IIF_EINVAL;
    if (c > 256 || ldp < c || ldw < c || ldwt < C + c) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(bn3_coef_kernel, dim3((C + 3) / 4), dim3(256), 0, st, P, ldp, (const unsigned short*)w_bf16, ldw, sum_g, stats,
                       gamma, C, c, (double)m, coef, dgamma, dbeta);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn3_scaled_transpose_kernel, dim3((C + 31) / 32, (c + 31) / 32), dim3(256), 0, st, (const unsigned short*)w_bf16,
                       ldw, coef, C, c, (unsigned short*)wt, ldwt);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn3_algebra_gm(const void* w_bf16, int ldw, const float* coef, int C, int c, void* wt, int ldwt, float* bias, void* stream) {
    if (!w_bf16 || !coef || !wt || !bias || C <= 0 || c <= 0) return IIF_EINVAL;
    if (c > 256 || ldw < c || ldwt < C + c) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(bn3_gm_kernel, dim3(c), dim3(256), 0, as_stream(stream), (const unsigned short*)w_bf16, ldw, coef, C, c,
                       (unsigned short*)wt, ldwt, bias);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn3_algebra_dw(const float* P, int ldp, const void* w_bf16, int ldw, const float* gram, int ldg, const float* csum,
                       const float* coef, int C, int c, float* dW, int lddw, void* stream) {
    if (!P || !w_bf16 || !gram || !csum || !coef || !dW || C <= 0 || c <= 0) return IIF_EINVAL;
    if (c > 256 || ldp < c || ldw < c || ldg < c || lddw < c) return IIF_EUNSUPPORTED;
    const int per = 256 / c > 0 ? 256 / c : 1;
    hipLaunchKernelGGL(bn3_dw_kernel, dim3((C + per - 1) / per), dim3(256), 0, as_stream(stream), P, ldp, (const unsigned short*)w_bf16,
                       ldw, gram, ldg, csum, coef, C, c, dW, lddw);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
