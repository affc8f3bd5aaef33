// Squeeze-and-excitation around the last BN of a residual block, NHWC, gfx950.
//
// Reference: SE_Block.forward (classification/resnet_pytorch.py:313-317, resnet_cifar.py:102-106) inside
// SEBottleneck.forward (:358-381) / Se_Block.forward (resnet_cifar.py:163-169):
//     o = bn(conv(x));  e = sigmoid(W2 relu(W1 mean_hw(o)));  y = relu(o * e + identity)
// The per-pixel work is four HBM-bound streaming passes over the [N, HW, C] tensor; the [N, C]-sized excitation
// (two bias-free linears, ReLU, sigmoid: 2 x C x C/r MACs per sample, not a GEMM-shaped problem at 256 samples) is three
// fp32 launches of its own: forward, backward, weight gradients (se_excite_* below).
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

// ------------------------------------------------------------------------------------------- excitation MLP
// e = sigmoid(W2 relu(W1 q)),  q[n,c] = a[c] * sums[n,c] / HW + b[c]  (= mean_hw of the BN output: the affine commutes with
// the mean).  W1 [hid][C] and W2T [hid][C] (the transpose of W2 [C][hid], made by iif_transpose_f32) are both read row-wise
// along C, so every access is coalesced.  A block owns SE_G samples: it pulls each weight matrix once from L2 and keeps the
// samples' vectors in LDS; two access patterns cover all four products:
//   rows_dot : out[s][j] = sum_c v[s][c] * W[j][c]   a wave per row j, lanes split C (float4), wave shuffle reduce
//   cols_mac : out[s][c] = sum_j u[s][j] * W[j][c]   a thread per 4 consecutive c, u broadcast from LDS
// Fixed summation order per output: deterministic.  (Round 2's native attempt re-read the weights per sample: 0.29 ms.)
constexpr int SE_G = 4, SE_T = 512;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out_s[s][j] (LDS, stride hid) = f(sum_c v_s[s][c] * W[j][c]);  C % 4 == 0
template <typename F>
__device__ __forceinline__ void se_rows_dot(const float* v_s, int C, const float* W, int ldw, int hid, float* out_s, F post) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = SE_T / 64;
    for (int j = wave; j < hid; j += nw) {
        float acc[SE_G];
#pragma unroll
        for (int s = 0; s < SE_G; ++s) acc[s] = 0.f;
        const float* wr = W + (int64_t)j * ldw;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
            for (int s = 0; s < SE_G; ++s) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(v_s + s * C + c);
                acc[s] += (w.x * v.x + w.y * v.y) + (w.z * v.z + w.w * v.w);
            }
        }
#pragma unroll
        for (int s = 0; s < SE_G; ++s) acc[s] = wave_sum_f(acc[s]);
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < SE_G; ++s) out_s[s * hid + j] = post(acc[s], s, j);
        }
    }
}

// for every 4 consecutive c: acc[s][0..3] = sum_j u_s[s][j] * W[j][c..c+3], handed to emit(c, acc)
template <typename F>
__device__ __forceinline__ void se_cols_mac(const float* u_s, int hid, const float* W, int ldw, int C, F emit) {
    for (int c = threadIdx.x * 4; c < C; c += SE_T * 4) {
        float acc[SE_G][4];
#pragma unroll
        for (int s = 0; s < SE_G; ++s) { acc[s][0] = 0.f; acc[s][1] = 0.f; acc[s][2] = 0.f; acc[s][3] = 0.f; }
        for (int j = 0; j < hid; ++j) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W + (int64_t)j * ldw + c);
#pragma unroll
            for (int s = 0; s < SE_G; ++s) {
                const float u = u_s[s * hid + j];
                acc[s][0] += u * w.x; acc[s][1] += u * w.y; acc[s][2] += u * w.z; acc[s][3] += u * w.w;
            }
        }
        emit(c, acc);
    }
}

__global__ void __launch_bounds__(SE_T) se_excite_fwd_kernel(const float* sums, const float* stats, float inv_hw, int N, int C, int hid,
                                                              const float* W1, int ld1, const float* W2T, int ldt,
                                                              float* q, float* h, float* e) {
    extern __shared__ __attribute__((aligned(16))) float se_sh[];
    float* q_s = se_sh;                    // [SE_G][C]
    float* h_s = se_sh + SE_G * C;         // [SE_G][hid]
    const int n0 = blockIdx.x * SE_G;
    const float* a = stats + 2 * C;
    const float* b = stats + 3 * C;
    for (int i = threadIdx.x; i < SE_G * C; i += SE_T) {
        const int s = i / C, c = i - s * C, n = n0 + s;
        const float v = n < N ? __builtin_fmaf(a[c] * sums[(int64_t)n * C + c], inv_hw, b[c]) : 0.f;
        q_s[i] = v;
        if (n < N) q[(int64_t)n * C + c] = v;
    }
    __syncthreads();
    se_rows_dot(q_s, C, W1, ld1, hid, h_s, [&](float v, int s, int j) {
        const float r = v > 0.f ? v : 0.f;
        if (n0 + s < N) h[(int64_t)(n0 + s) * hid + j] = r;
        return r;
    });
    __syncthreads();
    se_cols_mac(h_s, hid, W2T, ldt, C, [&](int c, float (&acc)[SE_G][4]) {
#pragma unroll
        for (int s = 0; s < SE_G; ++s) {
            if (n0 + s >= N) continue;
            f32x4 o;
            o.x = 1.f / (1.f + __expf(-acc[s][0])); o.y = 1.f / (1.f + __expf(-acc[s][1]));
            o.z = 1.f / (1.f + __expf(-acc[s][2])); o.w = 1.f / (1.f + __expf(-acc[s][3]));
            *reinterpret_cast<f32x4*>(e + (int64_t)(n0 + s) * C + c) = o;
        }
    });
}

// de = a * S2 + b * S1 (d/d e of sum_hw g * (a x + b) * e); dz2 = de * e (1 - e); dh = dz2 W2; dz1 = dh * [h > 0];
// o = (dz1 W1) / HW  (gradient reaching the BN output through the squeeze)
__global__ void __launch_bounds__(SE_T) se_excite_bwd_kernel(const float* s1, const float* s2, const float* stats, float inv_hw, int N,
                                                              int C, int hid, const float* W1, int ld1, const float* W2T, int ldt,
                                                              const float* e, const float* h, float* dz2, float* dz1, float* o) {
    extern __shared__ __attribute__((aligned(16))) float se_sh[];
    float* z2_s = se_sh;                   // [SE_G][C]
    float* z1_s = se_sh + SE_G * C;        // [SE_G][hid]
    const int n0 = blockIdx.x * SE_G;
    const float* a = stats + 2 * C;
    const float* b = stats + 3 * C;
    for (int i = threadIdx.x; i < SE_G * C; i += SE_T) {
        const int s = i / C, c = i - s * C, n = n0 + s;
        float v = 0.f;
        if (n < N) {
            const int64_t k = (int64_t)n * C + c;
            const float ev = e[k];
            v = (a[c] * s2[k] + b[c] * s1[k]) * ev * (1.f - ev);
            dz2[k] = v;
        }
        z2_s[i] = v;
    }
    __syncthreads();
    se_rows_dot(z2_s, C, W2T, ldt, hid, z1_s, [&](float v, int s, int j) {
        float r = 0.f;
        if (n0 + s < N) {
            r = h[(int64_t)(n0 + s) * hid + j] > 0.f ? v : 0.f;
            dz1[(int64_t)(n0 + s) * hid + j] = r;
        }
        return r;
    });
    __syncthreads();
    se_cols_mac(z1_s, hid, W1, ld1, C, [&](int c, float (&acc)[SE_G][4]) {
#pragma unroll
        for (int s = 0; s < SE_G; ++s) {
            if (n0 + s >= N) continue;
            *reinterpret_cast<f32x4*>(o + (int64_t)(n0 + s) * C + c) = f32x4{acc[s][0] * inv_hw, acc[s][1] * inv_hw, acc[s][2] * inv_hw, acc[s][3] * inv_hw};
        }
    });
}

// dW[r][c] = sum_n L[n][r] * Rm[n][c]  (L: [N][R], Rm: [N][Cc]): dW2 = dz2^T h (R = C, Cc = hid) and dW1 = dz1^T q (R = hid,
// Cc = C).  Block = 32 rows r x 64 columns c, samples in LDS chunks of 32, 8 accumulators per thread, fixed order.
__global__ void __launch_bounds__(256) se_outer_sum_kernel(const float* L, int ldl, const float* Rm, int ldr, int N, int R, int Cc,
                                                            float* dW, int ldw) {
    __shared__ float l_s[32][33], r_s[32][65];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 64;
    const int tr = threadIdx.x >> 6, tc = threadIdx.x & 63;          // rows tr, tr + 4, ..., tr + 28; column tc
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int nb = 0; nb < N; nb += 32) {
        for (int i = threadIdx.x; i < 32 * 32; i += 256) {
            const int s = i >> 5, r = i & 31;
            l_s[s][r] = (nb + s < N && r0 + r < R) ? L[(int64_t)(nb + s) * ldl + r0 + r] : 0.f;
        }
        for (int i = threadIdx.x; i < 32 * 64; i += 256) {
            const int s = i >> 6, c = i & 63;
            r_s[s][c] = (nb + s < N && c0 + c < Cc) ? Rm[(int64_t)(nb + s) * ldr + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
            const float rv = r_s[s][tc];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += l_s[s][tr + 4 * i] * rv;
        }
        __syncthreads();
    }
    if (c0 + tc < Cc) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (r0 + tr + 4 * i < R) dW[(int64_t)(r0 + tr + 4 * i) * ldw + c0 + tc] = acc[i];
    }
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


int iif_se_excite_forward(const float* sums, const float* stats, int n, int hw, int c, int hid, const float* w1, int ld1,
                          const float* w2t, int ldt, float* q, float* h, float* e, void* stream) {
    if (!sums || !stats || !w1 || !w2t || !q || !h || !e || n <= 0 || hw <= 0 || c <= 0 || hid <= 0) return IIF_EINVAL;
    if ((c & 3) || (ld1 & 3) || (ldt & 3) || ld1 < c || ldt < c) return IIF_EUNSUPPORTED;
    const size_t lds = (size_t)SE_G * (c + hid) * sizeof(float);
    if (lds > 64 * 1024) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(se_excite_fwd_kernel, dim3((n + SE_G - 1) / SE_G), dim3(SE_T), lds, as_stream(stream), sums, stats,
                       1.0f / (float)hw, n, c, hid, w1, ld1, w2t, ldt, q, h, e);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_se_excite_backward(const float* s1, const float* s2, const float* stats, int n, int hw, int c, int hid, const float* w1,
                           int ld1, const float* w2t, int ldt, const float* e, const float* h, const float* q, float* dz2,
                           float* dz1, float* offset, float* dw1, int ldg1, float* dw2, int ldg2, void* stream) {
    if (!s1 || !s2 || !stats || !w1 || !w2t || !e || !h || !q || !dz2 || !dz1 || !offset || !dw1 || !dw2) return IIF_EINVAL;
    if (n <= 0 || hw <= 0 || c <= 0 || hid <= 0) return IIF_EINVAL;
    if ((c & 3) || (ld1 & 3) || (ldt & 3) || ld1 < c || ldt < c || ldg1 < c || ldg2 < hid) return IIF_EUNSUPPORTED;
    const size_t lds = (size_t)SE_G * (c + hid) * sizeof(float);
    if (lds > 64 * 1024) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(se_excite_bwd_kernel, dim3((n + SE_G - 1) / SE_G), dim3(SE_T), lds, st, s1, s2, stats, 1.0f / (float)hw, n, c,
                       hid, w1, ld1, w2t, ldt, e, h, dz2, dz1, offset);
    IIF_LAUNCH_CHECK();
    // dW2 [c][hid] = dz2^T h,  dW1 [hid][c] = dz1^T q
    hipLaunchKernelGGL(se_outer_sum_kernel, dim3((c + 31) / 32, (hid + 63) / 64), dim3(256), 0, st, dz2, c, h, hid, n, c, hid, dw2, ldg2);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(se_outer_sum_kernel, dim3((hid + 31) / 32, (c + 63) / 64), dim3(256), 0, st, dz1, hid, q, c, n, hid, c, dw1, ldg1);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
