// Squeeze-and-excitation around the last BN of a residual block, NHWC, gfx950.
//
// Reference: SE_Block.forward (classification/resnet_pytorch.py:313-317, resnet_cifar.py:102-106) inside
// SEBottleneck.forward (:358-381) / Se_Block.forward (resnet_cifar.py:163-169):
//     o = bn(conv(x));  e = sigmoid(W2 relu(W1 mean_hw(o)));  y = relu(o * e + identity)
// The per-pixel work is four HBM-bound streaming passes over the [N, HW, C] tensor; the [N, C]-sized
// excitation (two tiny GEMMs) stays on the host side of the C-ABI.
//   squeeze  : S[n,c]  = sum_hw x[n,hw,c]                      (the BN affine is applied to the sums)   [1 read]
//   apply    : y = relu((a*x + b) * e[n,c] + identity) + 1-bit ReLU decisions                     [1-2 reads, 1 write]
//   bwd_sums : g <- g * [y > 0] in place;  S1[n,c] = sum_hw g,  S2[n,c] = sum_hw g*x              [2 reads, 1 write]
//   bwd_form : G = g * e[n,c] + o[n,c]   (gradient w.r.t. the BN output, o = d(squeeze)/HW)       [1 read, 1 write]
// Sums are accumulated in a fixed order per (sample, channel): deterministic.
#include "common.h"
#include "vec16.h"

namespace {

// block = one sample x up to CVB channel vectors; thread = (channel vector, row lane)
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) se_sums_kernel(T* g_, const unsigned char* bits, const T* x, int hw, int C, int cvb,
                                                      float* s1o, float* s2o) {
    constexpr int V = VT<T>::V;
    __shared__ float sh[2][256 * V];
    const int cv = C / V;
    const int tid = threadIdx.x;
    const int rpb = 256 / cvb;
    const int cvl = tid % cvb, rl = tid / cvb;
    const int cvec = blockIdx.y * cvb + cvl;
    const int n = blockIdx.x;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    if (rl < rpb && cvec < cv) {
        for (int r = rl; r < hw; r += rpb) {
            const int64_t row = (int64_t)n * hw + r;
            const int64_t o = row * C + cvec * V;
            float xv[V];
            VT<T>::load(x + o, xv);
            if (BWD) {
                float gv[V];
                VT<T>::load(g_ + o, gv);
                const unsigned mb = bits[row * cv + cvec];
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float d = ((mb >> i) & 1u) ? gv[i] : 0.f;
                    gv[i] = d;
                    s1[i] += d;
                    s2[i] += d * xv[i];
                }
                VT<T>::store(g_ + o, gv);
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) s1[i] += xv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { sh[0][tid * V + i] = s1[i]; sh[1][tid * V + i] = s2[i]; }
    __syncthreads();
    if (rl == 0 && cvec < cv) {
        for (int j = 1; j < rpb; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) { s1[i] += sh[0][(j * cvb + cvl) * V + i]; s2[i] += sh[1][(j * cvb + cvl) * V + i]; }
        float* p1 = s1o + (int64_t)n * C + cvec * V;
#pragma unroll
        for (int i = 0; i < V; ++i) p1[i] = s1[i];
        if (BWD) {
            float* p2 = s2o + (int64_t)n * C + cvec * V;
#pragma unroll
            for (int i = 0; i < V; ++i) p2[i] = s2[i];
        }
    }
}

template <typename T, int RES>   // RES 0: none, 1: + r, 2: + a2*r + b2
__global__ void __launch_bounds__(256) se_apply_kernel(const T* x, const float* stats, const float* e, const T* r,
                                                       const float* stats2, T* y, int64_t total_vec, int cv, int C, int hw,
                                                       unsigned char* relu_bits) {
    constexpr int V = VT<T>::V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * V;
        const int64_t n = (i / cv) / hw;
        const float* en = e + n * C + c0;
        float v[V], w[V];
        VT<T>::load(x + i * V, v);
        if (RES) VT<T>::load(r + i * V, w);
        unsigned bitsv = 0;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = fmaf(stats[2 * C + c0 + k], v[k], stats[3 * C + c0 + k]) * en[k];
            if (RES == 1) t += w[k];
            if (RES == 2) t += fmaf(stats2[2 * C + c0 + k], w[k], stats2[3 * C + c0 + k]);
            bitsv |= (t > 0.f ? 1u : 0u) << k;
            v[k] = fmaxf(t, 0.f);
        }
        VT<T>::store(y + i * V, v);
        relu_bits[i] = (unsigned char)bitsv;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) se_form_kernel(const T* g, const float* e, const float* o, T* out, int64_t total_vec,
                                                      int cv, int C, int hw) {
    constexpr int V = VT<T>::V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * V;
        const int64_t n = (i / cv) / hw;
        const float* en = e + n * C + c0;
        const float* on = o + n * C + c0;
        float v[V];
        VT<T>::load(g + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = fmaf(v[k], en[k], on[k]);
        VT<T>::store(out + i * V, v);
    }
}

inline unsigned stream_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}
inline bool bad(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
inline int vec_of(int dtype) { return dtype == IIF_F32 ? 4 : 8; }

template <typename T, bool BWD>
int launch_sums(T* g, const unsigned char* bits, const T* x, int n, int hw, int c, float* s1, float* s2, hipStream_t st) {
    const int cv = c / VT<T>::V;
    const int cvb = cv < 32 ? cv : 32;              // >= 8 row lanes per block
    const dim3 grid((unsigned)n, (unsigned)((cv + cvb - 1) / cvb));
    hipLaunchKernelGGL((se_sums_kernel<T, BWD>), grid, dim3(256), 0, st, g, bits, x, hw, c, cvb, s1, s2);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // namespace

extern "C" {

int iif_se_squeeze(const void* x, int dtype, int n, int hw, int c, float* sums, void* stream) {
    if (!x || !sums || n <= 0 || hw <= 0 || c <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16)) return IIF_EINVAL;
    if (c % vec_of(dtype) || bad(x)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) return launch_sums<float, false>(nullptr, nullptr, (const float*)x, n, hw, c, sums, nullptr, as_stream(stream));
    return launch_sums<unsigned short, false>(nullptr, nullptr, (const unsigned short*)x, n, hw, c, sums, nullptr, as_stream(stream));
}

int iif_se_apply(const void* x, int dtype, int n, int hw, int c, const float* stats, const float* excite, const void* residual,
                 const float* residual_stats, void* y, unsigned char* relu_bits, void* stream) {
    if (!x || !stats || !excite || !y || !relu_bits || n <= 0 || hw <= 0 || c <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16))
        return IIF_EINVAL;
    if (residual_stats && !residual) return IIF_EINVAL;
    const int V = vec_of(dtype);
    if (c % V || bad(x) || bad(y) || (residual && bad(residual))) return IIF_EUNSUPPORTED;
    const int cv = c / V;
    const int64_t tv = (int64_t)n * hw * cv;
    const dim3 grid(stream_blocks(tv)), blk(256);
    hipStream_t st = as_stream(stream);
    const int res = residual ? (residual_stats ? 2 : 1) : 0;
#define IIF_SE_APPLY(T, RS) hipLaunchKernelGGL((se_apply_kernel<T, RS>), grid, blk, 0, st, (const T*)x, stats, excite, (const T*)residual, residual_stats, (T*)y, tv, cv, c, hw, relu_bits)
    if (dtype == IIF_F32) { if (res == 0) IIF_SE_APPLY(float, 0); else if (res == 1) IIF_SE_APPLY(float, 1); else IIF_SE_APPLY(float, 2); }
    else { if (res == 0) IIF_SE_APPLY(unsigned short, 0); else if (res == 1) IIF_SE_APPLY(unsigned short, 1); else IIF_SE_APPLY(unsigned short, 2); }
#undef IIF_SE_APPLY
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_se_backward_sums(void* g, const unsigned char* relu_bits, const void* x, int dtype, int n, int hw, int c, float* s1,
                         float* s2, void* stream) {
    if (!g || !relu_bits || !x || !s1 || !s2 || n <= 0 || hw <= 0 || c <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16))
        return IIF_EINVAL;
    if (c % vec_of(dtype) || bad(x) || bad(g)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) return launch_sums<float, true>((float*)g, relu_bits, (const float*)x, n, hw, c, s1, s2, as_stream(stream));
    return launch_sums<unsigned short, true>((unsigned short*)g, relu_bits, (const unsigned short*)x, n, hw, c, s1, s2, as_stream(stream));
}

int iif_se_backward_form(const void* g, int dtype, int n, int hw, int c, const float* excite, const float* offset, void* out,
                         void* stream) {
    if (!g || !excite || !offset || !out || n <= 0 || hw <= 0 || c <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16))
        return IIF_EINVAL;
    const int V = vec_of(dtype);
    if (c % V || bad(g) || bad(out)) return IIF_EUNSUPPORTED;
    const int cv = c / V;
    const int64_t tv = (int64_t)n * hw * cv;
    const dim3 grid(stream_blocks(tv)), blk(256);
    if (dtype == IIF_F32) hipLaunchKernelGGL(se_form_kernel<float>, grid, blk, 0, as_stream(stream), (const float*)g, excite, offset, (float*)out, tv, cv, c, hw);
    else hipLaunchKernelGGL(se_form_kernel<unsigned short>, grid, blk, 0, as_stream(stream), (const unsigned short*)g, excite, offset, (unsigned short*)out, tv, cv, c, hw);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
