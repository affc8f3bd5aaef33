// Pooling, stem patch gather, layout/precision casts, option-A shortcut and the
// fused SGD update — the HBM-bound glue of the ResNet step on gfx950.  NHWC, 16-byte
// channel vectors per lane wherever alignment allows.
#include "common.h"
#include "pool_gather.h"

namespace {

template <typename T> struct PT;
template <> struct PT<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct PT<unsigned short> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<u32x4*>(p) = t;
    }
    static __device__ __forceinline__ float load1(const unsigned short* p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void store1(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};

inline int sblocks(int64_t n) { const int64_t b = (n + 255) / 256; return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

// ---------------------------------------------------------------- max pool 3x3/s2/p1 (general k,s,p)
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* x, int N, int H, int W, int C, int k, int s, int p,
                                                          int Ho, int Wo, T* y, unsigned char* idx) {
    constexpr int V = PT<T>::V;
    const int cv = C / V;
    const int64_t total = (int64_t)N * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        int64_t pix = i / cv;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float best[V]; unsigned char bi[V];
#pragma unroll
        for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bi[q] = 0; }
        bool first = true;
        for (int kh = 0; kh < k; ++kh) {
            const int h = ho * s - p + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int w = wo * s - p + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[V];
                PT<T>::load(x + (((int64_t)n * H + h) * W + w) * C + c, v);
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if (first || v[q] > best[q] || v[q] != v[q]) { best[q] = v[q]; bi[q] = (unsigned char)(kh * k + kw); }
                first = false;
            }
        }
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
        PT<T>::store(y + o, best);
#pragma unroll
        for (int q = 0; q < V; ++q) idx[o + q] = bi[q];
    }
}

// max pool over relu(a*x + b) of the RAW convolution output: the stem's activation (resnet_pytorch.py:284-287: bn1, relu,
// maxpool) is never written.  Every candidate is rounded to the storage type before the comparison, so value and
// argmax are exactly those of bn_apply followed by maxpool_fwd_kernel.
template <typename T>
__global__ void __launch_bounds__(256) maxpool_bn_fwd_kernel(const T* x, const float* stats, int N, int H, int W, int C, int k,
                                                             int s, int p, int Ho, int Wo, T* y, unsigned char* idx, T* px) {
    constexpr int V = PT<T>::V;
    const int cv = C / V;
    const int64_t total = (int64_t)N * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        int64_t pix = i / cv;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float best[V], bx[V], ca[V], cb[V]; unsigned char bi[V];
#pragma unroll
        for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bx[q] = 0.f; bi[q] = 0; ca[q] = stats[2 * C + c + q]; cb[q] = stats[3 * C + c + q]; }
        bool first = true;
        for (int kh = 0; kh < k; ++kh) {
            const int h = ho * s - p + kh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int w = wo * s - p + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[V];
                PT<T>::load(x + (((int64_t)n * H + h) * W + w) * C + c, v);
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    float t = fmaxf(fmaf(ca[q], v[q], cb[q]), 0.f);
                    if constexpr (sizeof(T) == 2) t = bf16_bits_to_f32(f32_to_bf16_bits(t));
                    if (first || t > best[q] || t != t) { best[q] = t; bx[q] = v[q]; bi[q] = (unsigned char)(kh * k + kw); }
                }
                first = false;
            }
        }
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
        PT<T>::store(y + o, best);
        if (px) PT<T>::store(px + o, bx);
#pragma unroll
        for (int q = 0; q < V; ++q) idx[o + q] = bi[q];
    }
}

// The stem's shape (3 x 3 / stride 2 / pad 1, 256 % (C / V) == 0): one output row per block pass, a thread keeps ONE channel
// vector (coefficients loaded once), the nine candidate vectors are all in flight before the first comparison (clamped
// addresses, predicated use), 32-bit index arithmetic, the eight arg-max codes leave as one store.  Candidates are visited
// in the (kh, kw) order of the general kernel with its first-valid / strictly-greater / NaN rule: identical results.
template <typename T>
__global__ void __launch_bounds__(256) maxpool321_bn_fwd_kernel(const T* x, const float* stats, int N, int H, int W, int C, int Ho,
                                                                int Wo, T* y, unsigned char* idx, T* px) {
    constexpr int V = PT<T>::V;
    const unsigned cv = C / V, rowv = (unsigned)Wo * cv;
    const int c = (int)(threadIdx.x % cv) * V;
    float ca[V], cb[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { ca[q] = stats[2 * C + c + q]; cb[q] = stats[3 * C + c + q]; }
    for (unsigned row = blockIdx.x; row < (unsigned)(N * Ho); row += gridDim.x) {
        const unsigned n = row / (unsigned)Ho, ho = row - n * Ho;
        const int h0 = 2 * (int)ho - 1;
        for (unsigned j = threadIdx.x; j < rowv; j += 256) {
            const unsigned wo = j / cv;
            const int w0 = 2 * (int)wo - 1;
            float v[9][V];
            bool ok[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int h = h0 + kh, w = w0 + kw;
                    ok[kh * 3 + kw] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
                    const int hc = h < 0 ? 0 : (h >= H ? H - 1 : h), wc = w < 0 ? 0 : (w >= W ? W - 1 : w);
                    PT<T>::load(x + (((int64_t)n * H + hc) * W + wc) * C + c, v[kh * 3 + kw]);
                }
            float best[V], bx[V];
            unsigned bi[V];
#pragma unroll
            for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bx[q] = 0.f; bi[q] = 0; }
            bool first = true;
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                if (!ok[t9]) continue;
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    float t = fmaxf(fmaf(ca[q], v[t9][q], cb[q]), 0.f);
                    if constexpr (sizeof(T) == 2) t = bf16_bits_to_f32(f32_to_bf16_bits(t));
                    if (first || t > best[q] || t != t) { best[q] = t; bx[q] = v[t9][q]; bi[q] = (unsigned)t9; }
                }
                first = false;
            }
            const int64_t o = ((int64_t)row * Wo + wo) * C + c;
            PT<T>::store(y + o, best);
            if (px) PT<T>::store(px + o, bx);          // the RAW value at the arg max: what the backward's column sums start from
            if constexpr (V == 8) {
                const unsigned lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
                *reinterpret_cast<uint2*>(idx + o) = make_uint2(lo, hi);
            } else {
                *reinterpret_cast<unsigned*>(idx + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* gy, const unsigned char* idx, int N, int H, int W,
                                                          int C, int k, int s, int p, int Ho, int Wo, T* dx) {
    constexpr int V = PT<T>::V;
    const int cv = C / V;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        int64_t pix = i / cv;
        const int w = (int)(pix % W); pix /= W;
        const int h = (int)(pix % H);
        const int n = (int)(pix / H);
        float acc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = 0.f;
        // windows (ho, wo) with ho*s - p <= h <= ho*s - p + k - 1
        int ho_lo = (h + p - (k - 1) + s - 1) / s; if (h + p - (k - 1) < 0) ho_lo = 0;
        int ho_hi = (h + p) / s; if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        int wo_lo = (w + p - (k - 1) + s - 1) / s; if (w + p - (k - 1) < 0) wo_lo = 0;
        int wo_hi = (w + p) / s; if (wo_hi > Wo - 1) wo_hi = Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int code = (h - (ho * s - p)) * k + (w - (wo * s - p));
                const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
                float g[V];
                PT<T>::load(gy + o, g);
                // the V argmax codes of this vector in one load (o is a multiple of V)
                unsigned long long codes;
                if constexpr (V == 8) codes = *reinterpret_cast<const unsigned long long*>(idx + o);
                else codes = *reinterpret_cast<const unsigned int*>(idx + o);
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if ((int)((codes >> (8 * q)) & 0xffu) == code) acc[q] += g[q];
            }
        PT<T>::store(dx + (((int64_t)n * H + h) * W + w) * C + c, acc);
    }
}

// 3x3 / stride 2 / pad 1 (the ResNet stem pool): unrolled, all window loads in flight at once
template <typename T>
__global__ void __launch_bounds__(256) maxpool321_bwd_kernel(const T* gy, const unsigned char* idx, int N, int H, int W,
                                                             int C, int Ho, int Wo, T* dx) {
    constexpr int V = PT<T>::V;
    const unsigned cv = C / V, rowv = (unsigned)W * cv;
    // one image row per block pass: h is uniform, only 32-bit index arithmetic
    for (unsigned row = blockIdx.x; row < (unsigned)(N * H); row += gridDim.x) {
        const unsigned n = row / (unsigned)H, h = row - n * H;
        for (unsigned j = threadIdx.x; j < rowv; j += 256) {
            const unsigned w = j / cv, c = (j - w * cv) * V;
            float acc[V];
            pool321_gather<T>(gy, idx, (int)n, (int)h, (int)w, (int)c, C, Ho, Wo, acc);
            PT<T>::store(dx + ((int64_t)row * W + w) * C + c, acc);
        }
    }
}

// ---------------------------------------------------------------- global average pool
template <typename T>
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const T* x, int N, int HW, int C, T* y) {
    constexpr int V = PT<T>::V;
    const int cv = C / V;
    const int64_t total = (int64_t)N * cv;
    const float inv = 1.0f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const int n = (int)(i / cv);
        float acc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = 0.f;
        for (int j = 0; j < HW; ++j) {
            float v[V];
            PT<T>::load(x + ((int64_t)n * HW + j) * C + c, v);
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += v[q];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] *= inv;
        PT<T>::store(y + (int64_t)n * C + c, acc);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const T* gy, int N, int HW, int C, T* dx) {
    constexpr int V = PT<T>::V;
    const int cv = C / V;
    const int64_t total = (int64_t)N * HW * cv;
    const float inv = 1.0f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * V;
        const int n = (int)(i / ((int64_t)HW * cv));
        float g[V];
        PT<T>::load(gy + (int64_t)n * C + c, g);
#pragma unroll
        for (int q = 0; q < V; ++q) g[q] *= inv;
        PT<T>::store(dx + i * V, g);
    }
}

// ---------------------------------------------------------------- stem patches (im2col of a few-channel image)
// img: NCHW fp32 [N, Cin, H, W] -> patches [N*Ho*Wo][kp], column (r*S + s)*Cin + c, zero pad to kp
template <typename T>
__global__ void __launch_bounds__(256) im2col_kernel(const float* img, int N, int Cin, int H, int W, int R, int S,
                                                     int stride, int pad, int Ho, int Wo, int kp, T* out) {
    constexpr int V = PT<T>::V;
    const int gv = kp / V;
    const int K = R * S * Cin;
    const int64_t total = (int64_t)N * Ho * Wo * gv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k0 = (int)(i % gv) * V;
        int64_t pix = i / gv;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float v[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int kk = k0 + q;
            float t = 0.f;
            if (kk < K) {
                const int c = kk % Cin, tap = kk / Cin;
                const int r = tap / S, s = tap - r * S;
                const int h = ho * stride - pad + r, w = wo * stride - pad + s;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
                    t = img[(((int64_t)n * Cin + c) * H + h) * W + w];
            }
            v[q] = t;
        }
        PT<T>::store(out + i * V, v);
    }
}

// ---------------------------------------------------------------- casts
template <typename T>
__global__ void __launch_bounds__(256) cast_from_f32_kernel(const float* src, T* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        PT<T>::store1(dst + i, src[i]);
}
template <typename T>
__global__ void __launch_bounds__(256) cast_to_f32_kernel(const T* src, float* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = PT<T>::load1(src + i);
}
// w: fp32 [Cout][ldw] (r,s,c) -> wt: T [Cin][ldwt] (r,s,k); pad columns zeroed
template <typename T>
__global__ void __launch_bounds__(256) weight_transpose_kernel(const float* w, int Cout, int Cin, int RS, int ldw,
                                                               int ldwt, T* wt) {
    const int64_t total = (int64_t)Cin * ldwt;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldwt), c = (int)(i / ldwt);
        float v = 0.f;
        if (col < RS * Cout) {
            const int tap = col / Cout, k = col - tap * Cout;
            v = w[(int64_t)k * ldw + tap * Cin + c];
        }
        PT<T>::store1(wt + i, v);
    }
}

// every dense convolution's transposed copy in ONE launch: a device table of per-tensor descriptors, each block
// finds its tensor by binary search over the table's block prefix and transposes one 32 (cout) x 32 (cin) tile of
// one tap through LDS, so both the fp32 reads and the T writes are coalesced 128-byte runs (the element-wise
// gather this replaces pulled a whole line per element: 1.4 GB of fetch for 0.15 GB of data).
// Tiles of a tensor: tap-major, then cin tiles, then cout tiles; pad columns of wt are zeroed by the last cout tile.
template <typename T>
__global__ void __launch_bounds__(256) weight_transpose_batched_kernel(const float* arena, const iif_wt_desc* tab, int n, T* out) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= b) lo = mid; else hi = mid - 1;
    }
    const iif_wt_desc d = tab[lo];
    const float* w = arena + d.src_off;
    T* wt = out + d.dst_off;
    const int kt_n = (d.cout + 31) / 32, ct_n = (d.cin + 31) / 32;
    int t = b - d.block_start;
    const int kt = t % kt_n; t /= kt_n;
    const int ct = t % ct_n;
    const int tap = t / ct_n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    const int k0 = kt * 32, c0 = ct * 32;
    for (int j = ty; j < 32; j += 8) {                                 // rows k0+j of w, columns tap*cin + c0 + tx
        const int k = k0 + j, c = c0 + tx;
        tile[j][tx] = (k < d.cout && c < d.cin) ? w[(int64_t)k * d.ldw + tap * d.cin + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                                 // rows c0+j of wt, columns tap*cout + k0 + tx
        const int c = c0 + j, k = k0 + tx;
        if (c < d.cin && k < d.cout) PT<T>::store1(wt + (int64_t)c * d.ldwt + tap * d.cout + k, tile[tx][j]);
    }
    if (kt == kt_n - 1 && tap == d.rs - 1) {                            // zero the row padding [rs*cout, ldwt)
        const int padn = d.ldwt - d.rs * d.cout;
        for (int i = threadIdx.x; i < 32 * padn; i += 256) {
            const int c = c0 + i / padn, col = d.rs * d.cout + i % padn;
            if (c < d.cin) PT<T>::store1(wt + (int64_t)c * d.ldwt + col, 0.f);
        }
    }
}

// ---------------------------------------------------------------- option-A shortcut (resnet_cifar.py:125-126)
template <typename T>
__global__ void __launch_bounds__(256) shortcut_a_fwd_kernel(const T* x, int N, int H, int W, int Cin, int Ho, int Wo,
                                                             int Cout, int cpad, T* y) {
    const int64_t total = (int64_t)N * Ho * Wo * Cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cout);
        int64_t pix = i / Cout;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float v = 0.f;
        const int ci = c - cpad;
        if (ci >= 0 && ci < Cin) v = PT<T>::load1(x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * Cin + ci);
        PT<T>::store1(y + i, v);
    }
}
// dx[n,h,w,ci] += g[n,h/2,w/2,ci+cpad] at even (h,w)
template <typename T>
__global__ void __launch_bounds__(256) shortcut_a_bwd_kernel(const T* g, int N, int H, int W, int Cin, int Ho, int Wo,
                                                             int Cout, int cpad, T* dx) {
    const int64_t total = (int64_t)N * Ho * Wo * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        int64_t pix = i / Cin;
        const int wo = (int)(pix % Wo); pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        const int64_t o = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * Cin + ci;
        const float gv = PT<T>::load1(g + (((int64_t)n * Ho + ho) * Wo + wo) * Cout + ci + cpad);
        PT<T>::store1(dx + o, PT<T>::load1(dx + o) + gv);
    }
}

// ---------------------------------------------------------------- column sums (bias gradient)
// 32 columns x 8 row lanes per block, 4 independent loads in flight per thread, row lanes combined in fixed order
__global__ void __launch_bounds__(256) colsum_kernel(const float* a, int rows, int cols, int64_t ld, float* out) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = rl;
        for (; r + 24 < rows; r += 32) {
            s0 += a[(int64_t)r * ld + c]; s1 += a[(int64_t)(r + 8) * ld + c];
            s2 += a[(int64_t)(r + 16) * ld + c]; s3 += a[(int64_t)(r + 24) * ld + c];
        }
        for (; r < rows; r += 8) s0 += a[(int64_t)r * ld + c];
    }
    sh[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += sh[j][cl];
        out[c] = s;
    }
}

// ---------------------------------------------------------------- fused SGD over a flat arena
// torch.optim.SGD semantics (dampening 0): d = g + wd*p; buf = m*buf + d (zero-initialised buf gives
// buf = d on the first step); p -= lr * (nesterov ? d + m*buf : buf).  grad_scale folds 1/world_size.
__global__ void __launch_bounds__(256) sgd_kernel(float* p, const float* g, float* buf, int64_t n, float lr,
                                                  const float* d_lr, float momentum, float wd, int nesterov,
                                                  float grad_scale) {
    const float step = d_lr ? *d_lr : lr;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = gv[q] * grad_scale + wd * pv[q];
            bv[q] = momentum * bv[q] + d;
            pv[q] -= step * (nesterov ? d + momentum * bv[q] : bv[q]);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(buf)[i] = bv;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = g[i] * grad_scale + wd * p[i];
        const float b = momentum * buf[i] + d;
        buf[i] = b;
        p[i] -= step * (nesterov ? d + momentum * b : b);
    }
}

inline bool mis(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

#define IIF_BY_DTYPE(dtype, CALL_F32, CALL_BF16) \
    if ((dtype) == IIF_F32) { CALL_F32; } else if ((dtype) == IIF_BF16) { CALL_BF16; } else return IIF_EINVAL;

extern "C" {

int iif_maxpool_forward(const void* x, int dtype, int n, int h, int w, int c, int k, int stride, int pad, void* y,
                        uint8_t* argmax, void* stream) {
    if (!x || !y || !argmax || n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0)
        return IIF_EINVAL;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return IIF_EINVAL;
    if (mis(x) || mis(y) || c % (dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * ho * wo * (c / (dtype == IIF_F32 ? 4 : 8));
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)x, n, h, w, c, k, stride, pad, ho, wo, (float*)y, argmax),
        hipLaunchKernelGGL(maxpool_fwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)x, n, h, w, c, k, stride, pad, ho, wo, (unsigned short*)y, argmax))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_maxpool_bn_forward(const void* x, int dtype, const float* stats, int n, int h, int w, int c, int k, int stride, int pad,
                           void* y, uint8_t* argmax, void* pool_x, void* stream) {
    if (!x || !stats || !y || !argmax || n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0)
        return IIF_EINVAL;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return IIF_EINVAL;
    if (mis(x) || mis(y) || (pool_x && mis(pool_x)) || c % (dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * ho * wo * (c / (dtype == IIF_F32 ? 4 : 8));
    const int cvs = c / (dtype == IIF_F32 ? 4 : 8);
    if (k == 3 && stride == 2 && pad == 1 && cvs <= 256 && 256 % cvs == 0 && (int64_t)n * ho < 0x7fffffffLL &&
        (int64_t)n * h * w * c < 0x7fffffffLL) {
        const int rowblocks = (int)((int64_t)n * ho < 16384 ? (int64_t)n * ho : 16384);
        IIF_BY_DTYPE(dtype,
            hipLaunchKernelGGL(maxpool321_bn_fwd_kernel<float>, dim3(rowblocks), dim3(256), 0, st, (const float*)x, stats, n, h, w, c, ho, wo, (float*)y, argmax, (float*)pool_x),
            hipLaunchKernelGGL(maxpool321_bn_fwd_kernel<unsigned short>, dim3(rowblocks), dim3(256), 0, st, (const unsigned short*)x, stats, n, h, w, c, ho, wo, (unsigned short*)y, argmax, (unsigned short*)pool_x))
        IIF_LAUNCH_CHECK();
        return IIF_OK;
    }
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(maxpool_bn_fwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)x, stats, n, h, w, c, k, stride, pad, ho, wo, (float*)y, argmax, (float*)pool_x),
        hipLaunchKernelGGL(maxpool_bn_fwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)x, stats, n, h, w, c, k, stride, pad, ho, wo, (unsigned short*)y, argmax, (unsigned short*)pool_x))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_maxpool_backward(const void* gy, const uint8_t* argmax, int dtype, int n, int h, int w, int c, int k, int stride,
                         int pad, void* dx, void* stream) {
    if (!gy || !dx || !argmax || n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0)
        return IIF_EINVAL;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return IIF_EINVAL;
    if (mis(gy) || mis(dx) || c % (dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * h * w * (c / (dtype == IIF_F32 ? 4 : 8));
    if (k == 3 && stride == 2 && pad == 1 && (int64_t)n * h < 0x7fffffffLL) {
        // one image row per block, every row its own block (round 5: a grid capped at 16 384 blocks walked 28 672 rows in two
        // uneven passes, and stores of a looping grid spread over DRAM pages - bn.hip, stream_grid)
        const int rowblocks = (int)((int64_t)n * h);
        IIF_BY_DTYPE(dtype,
            hipLaunchKernelGGL(maxpool321_bwd_kernel<float>, dim3(rowblocks), dim3(256), 0, st, (const float*)gy, argmax, n, h, w, c, ho, wo, (float*)dx),
            hipLaunchKernelGGL(maxpool321_bwd_kernel<unsigned short>, dim3(rowblocks), dim3(256), 0, st, (const unsigned short*)gy, argmax, n, h, w, c, ho, wo, (unsigned short*)dx))
        IIF_LAUNCH_CHECK();
        return IIF_OK;
    }
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)gy, argmax, n, h, w, c, k, stride, pad, ho, wo, (float*)dx),
        hipLaunchKernelGGL(maxpool_bwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)gy, argmax, n, h, w, c, k, stride, pad, ho, wo, (unsigned short*)dx))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_avgpool_forward(const void* x, int dtype, int n, int hw, int c, void* y, void* stream) {
    if (!x || !y || n <= 0 || hw <= 0 || c <= 0) return IIF_EINVAL;
    if (mis(x) || mis(y) || c % (dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * (c / (dtype == IIF_F32 ? 4 : 8));
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)x, n, hw, c, (float*)y),
        hipLaunchKernelGGL(avgpool_fwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)x, n, hw, c, (unsigned short*)y))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_avgpool_backward(const void* gy, int dtype, int n, int hw, int c, void* dx, void* stream) {
    if (!gy || !dx || n <= 0 || hw <= 0 || c <= 0) return IIF_EINVAL;
    if (mis(gy) || mis(dx) || c % (dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * hw * (c / (dtype == IIF_F32 ? 4 : 8));
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)gy, n, hw, c, (float*)dx),
        hipLaunchKernelGGL(avgpool_bwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)gy, n, hw, c, (unsigned short*)dx))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_im2col_nchw(const float* img, int n, int cin, int h, int w, int r, int s, int stride, int pad, int kp,
                    int out_dtype, void* out, void* stream) {
    if (!img || !out || n <= 0 || cin <= 0 || h <= 0 || w <= 0 || r <= 0 || s <= 0 || stride <= 0 || pad < 0)
        return IIF_EINVAL;
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
    if (ho <= 0 || wo <= 0 || kp < r * s * cin) return IIF_EINVAL;
    if (mis(out) || kp % (out_dtype == IIF_F32 ? 4 : 8)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * ho * wo * (kp / (out_dtype == IIF_F32 ? 4 : 8));
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, img, n, cin, h, w, r, s, stride, pad, ho, wo, kp, (float*)out),
        hipLaunchKernelGGL(im2col_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, img, n, cin, h, w, r, s, stride, pad, ho, wo, kp, (unsigned short*)out))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    if (n < 0) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!src || !dst) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const dim3 grid(sblocks(n)), blk(256);
    if (src_dtype == IIF_F32 && dst_dtype == IIF_BF16)
        hipLaunchKernelGGL(cast_from_f32_kernel<unsigned short>, grid, blk, 0, st, (const float*)src, (unsigned short*)dst, n);
    else if (src_dtype == IIF_BF16 && dst_dtype == IIF_F32)
        hipLaunchKernelGGL(cast_to_f32_kernel<unsigned short>, grid, blk, 0, st, (const unsigned short*)src, (float*)dst, n);
    else if (src_dtype == IIF_F32 && dst_dtype == IIF_F32)
        hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid, blk, 0, st, (const float*)src, (float*)dst, n);
    else
        return IIF_EINVAL;
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_weight_transpose(const float* w, int cout, int cin, int rs, int ldw, int ldwt, int out_dtype, void* wt,
                         void* stream) {
    if (!w || !wt || cout <= 0 || cin <= 0 || rs <= 0 || ldw < rs * cin || ldwt < rs * cout) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)cin * ldwt;
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(weight_transpose_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, w, cout, cin, rs, ldw, ldwt, (float*)wt),
        hipLaunchKernelGGL(weight_transpose_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, w, cout, cin, rs, ldw, ldwt, (unsigned short*)wt))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_weight_transpose_batched(const float* arena, const iif_wt_desc* table, int n_desc, int total_blocks, int out_dtype,
                                 void* out, void* stream) {
    if (!arena || !table || !out || n_desc <= 0 || total_blocks <= 0) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(weight_transpose_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, st, arena, table, n_desc, (float*)out),
        hipLaunchKernelGGL(weight_transpose_batched_kernel<unsigned short>, dim3(total_blocks), dim3(256), 0, st, arena, table, n_desc, (unsigned short*)out))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_shortcut_a_forward(const void* x, int dtype, int n, int h, int w, int cin, int cout, void* y, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout < cin || (cout - cin) % 2) return IIF_EINVAL;
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, cpad = (cout - cin) / 2;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * ho * wo * cout;
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(shortcut_a_fwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)x, n, h, w, cin, ho, wo, cout, cpad, (float*)y),
        hipLaunchKernelGGL(shortcut_a_fwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)x, n, h, w, cin, ho, wo, cout, cpad, (unsigned short*)y))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_shortcut_a_backward_acc(const void* g, int dtype, int n, int h, int w, int cin, int cout, void* dx, void* stream) {
    if (!g || !dx || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout < cin || (cout - cin) % 2) return IIF_EINVAL;
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, cpad = (cout - cin) / 2;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * ho * wo * cin;
    IIF_BY_DTYPE(dtype,
        hipLaunchKernelGGL(shortcut_a_bwd_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, (const float*)g, n, h, w, cin, ho, wo, cout, cpad, (float*)dx),
        hipLaunchKernelGGL(shortcut_a_bwd_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, (const unsigned short*)g, n, h, w, cin, ho, wo, cout, cpad, (unsigned short*)dx))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_colsum_f32(const float* a, int rows, int cols, int64_t ld, float* out, void* stream) {
    if (!a || !out || rows <= 0 || cols <= 0 || ld < cols) return IIF_EINVAL;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 31) / 32), dim3(256), 0, as_stream(stream), a, rows, cols, ld, out);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, const float* d_lr,
                 float momentum, float weight_decay, int nesterov, float grad_scale, void* stream) {
    if (n < 0) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!params || !grads || !momentum_buf) return IIF_EINVAL;
    if (mis(params) || mis(grads) || mis(momentum_buf)) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(sgd_kernel, dim3(sblocks(n / 4 + 1)), dim3(256), 0, as_stream(stream), params, grads, momentum_buf,
                       n, lr, d_lr, momentum, weight_decay, nesterov, grad_scale);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- grouped-convolution weight packing
// A grouped KxK convolution (ResNeXt, resnet_pytorch.py:137,141) runs on the MFMA kernels as a dense
// convolution inside channel CHUNKS of `ch` (= 64) channels: chunk weights are block-diagonal over the
// groups they contain.  master: fp32 [cout][ldm] rows of (tap, cin_local<cg);  packed: [cout][ldp] rows of
// (tap, chunk-local input channel < ch).  transposed = rows are INPUT channels, columns (tap, chunk-local
// output channel): the data-gradient operand.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) group_pack_kernel(const float* m, int C, int cg, int ch, int rs, int ldm, int ldp,
                                                         int transposed, T* out) {
    const int64_t total = (int64_t)C * ldp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldp), row = (int)(i / ldp);
        float v = 0.f;
        if (col < rs * ch) {
            const int tap = col / ch, loc = col - tap * ch;
            const int other = (row / ch) * ch + loc;          // the channel on the other side of the weight
            if (other / cg == row / cg) {
                const int k = transposed ? other : row, c = transposed ? row : other;
                v = m[(int64_t)k * ldm + tap * cg + (c % cg)];
            }
        }
        PT<T>::store1(out + i, v);
    }
}
// every grouped layer's packed copies in ONE launch (ResNeXt-101: 66 launches per step, ~0.9 ms of host time at the start
// of the step during which the compute stream had nothing queued): blockIdx.y = table entry
struct GroupPackEntry { const float* m; void* out; int C, cg, ch, rs, ldm, ldp, transposed, pad; };
template <typename T>
__global__ void __launch_bounds__(256) group_pack_batched_kernel(const GroupPackEntry* tab) {
    const GroupPackEntry e = tab[blockIdx.y];
    const int64_t total = (int64_t)e.C * e.ldp;
    T* out = (T*)e.out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % e.ldp), row = (int)(i / e.ldp);
        float v = 0.f;
        if (col < e.rs * e.ch) {                              // (the arithmetic of group_pack_kernel)
            const int tap = col / e.ch, loc = col - tap * e.ch;
            const int other = (row / e.ch) * e.ch + loc;
            if (other / e.cg == row / e.cg) {
                const int k = e.transposed ? other : row, c = e.transposed ? row : other;
                v = e.m[(int64_t)k * e.ldm + tap * e.cg + (c % e.cg)];
            }
        }
        PT<T>::store1(out + i, v);
    }
}
// dense-in-chunk weight gradient [cout][ldp] -> master layout [cout][ldm] (only the in-group entries exist)
__global__ void __launch_bounds__(256) group_unpack_kernel(const float* p, int C, int cg, int ch, int rs, int ldp, int ldm,
                                                           float* m) {
    const int64_t total = (int64_t)C * rs * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cl = (int)(i % cg);
        const int tap = (int)((i / cg) % rs);
        const int k = (int)(i / ((int64_t)cg * rs));
        const int c = (k / cg) * cg + cl;                     // global input channel
        m[(int64_t)k * ldm + tap * cg + cl] = p[(int64_t)k * ldp + tap * ch + (c % ch)];
    }
}
}  // namespace

extern "C" int iif_group_pack(const float* master, int channels, int cg, int chunk, int rs, int ldm, int ldp, int transposed,
                              int out_dtype, void* out, void* stream) {
    if (!master || !out || channels <= 0 || cg <= 0 || chunk <= 0 || rs <= 0) return IIF_EINVAL;
    if (channels % chunk || chunk % cg || ldm < rs * cg || ldp < rs * chunk) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)channels * ldp;
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(group_pack_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, master, channels, cg, chunk, rs, ldm, ldp, transposed, (float*)out),
        hipLaunchKernelGGL(group_pack_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, master, channels, cg, chunk, rs, ldm, ldp, transposed, (unsigned short*)out))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_group_pack_batched(const void* table, int entries, int blocks_per_entry, int out_dtype, void* stream) {
    if (!table || entries <= 0 || entries > 65535 || blocks_per_entry <= 0) return IIF_EINVAL;
    static_assert(sizeof(GroupPackEntry) == 48, "table layout (iif_group_pack_entry in include/iif_amd.h)");
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)blocks_per_entry, (unsigned)entries);
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(group_pack_batched_kernel<float>, grid, dim3(256), 0, st, (const GroupPackEntry*)table),
        hipLaunchKernelGGL(group_pack_batched_kernel<unsigned short>, grid, dim3(256), 0, st, (const GroupPackEntry*)table))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_group_unpack_grad(const float* packed, int channels, int cg, int chunk, int rs, int ldp, int ldm,
                                     float* master, void* stream) {
    if (!packed || !master || channels <= 0 || cg <= 0 || chunk <= 0 || rs <= 0) return IIF_EINVAL;
    if (channels % chunk || chunk % cg || ldm < rs * cg || ldp < rs * chunk) return IIF_EINVAL;
    const int64_t tot = (int64_t)channels * rs * cg;
    hipLaunchKernelGGL(group_unpack_kernel, dim3(sblocks(tot)), dim3(256), 0, as_stream(stream), packed, channels, cg, chunk,
                       rs, ldp, ldm, master);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// ---------------------------------------------------------------- stem as a space-to-depth convolution
// The 7x7 / stride-2 / pad-3 stem on a 3-channel image (resnet_pytorch.py:203) equals a 4x4 / stride-1
// convolution on the 2x2 space-to-depth image (12 channels, padded to `cpad` = 32 so that a K step is one
// tap): input row i = 2I + di, original tap r = 2a + di - 1 for s2d tap a = 0..3 (r outside 0..6: zero
// weight).  No patch matrix is materialised: the MFMA kernels gather the 16 taps themselves.
namespace {
// one thread per output pixel: 2 rows x float2 per image plane (consecutive threads read consecutive float2:
// coalesced), then the pixel's cpad channels in one run of stores
template <typename T>
__global__ void __launch_bounds__(256) s2d_kernel(const float* img, int N, int C, int H, int W, int cpad, T* out) {
    const int H2 = H / 2, W2 = W / 2;
    const int64_t total = (int64_t)N * H2 * W2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t pix = i;
        const int x = (int)(pix % W2); pix /= W2;
        const int y = (int)(pix % H2);
        const int n = (int)(pix / H2);
        T* o = out + i * cpad;
        float v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = 0.f;
        if (C == 3) {                                                   // RGB: fully unrolled, values stay in registers
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* p = img + (((int64_t)n * 3 + c) * H + 2 * y) * W + 2 * x;
                const float2 a = *reinterpret_cast<const float2*>(p), b2 = *reinterpret_cast<const float2*>(p + W);
                v[c] = a.x; v[3 + c] = a.y; v[6 + c] = b2.x; v[9 + c] = b2.y;      // (di, dj) = (0,0) (0,1) (1,0) (1,1)
            }
            if constexpr (sizeof(T) == 2) {
                for (int q0 = 0; q0 < cpad; q0 += 8) {
                    u32x4 w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = q0 < 16 ? pack_bf16x2(v[(q0 & 8) + 2 * k], v[(q0 & 8) + 2 * k + 1]) : 0u;
                    *reinterpret_cast<u32x4*>(o + q0) = w;
                }
            } else {
                for (int q0 = 0; q0 < cpad; q0 += 4)
                    *reinterpret_cast<f32x4*>(o + q0) = q0 < 12 ? f32x4{v[q0 & 15], v[(q0 & 15) + 1], v[(q0 & 15) + 2], v[(q0 & 15) + 3]}
                                                               : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            continue;
        }
        for (int c = 0; c < C; ++c) {
            const float* p = img + (((int64_t)n * C + c) * H + 2 * y) * W + 2 * x;
            PT<T>::store1(o + c, p[0]);
            PT<T>::store1(o + C + c, p[1]);
            PT<T>::store1(o + 2 * C + c, p[W]);
            PT<T>::store1(o + 3 * C + c, p[W + 1]);
        }
        for (int q = 4 * C; q < cpad; ++q) PT<T>::store1(o + q, 0.f);
    }
}
// master [K][ldm] rows of (r, s, c) over R x R taps -> packed [K][A*A*cpad] rows of (a, b, q), A = (R+1)/2
template <typename T>
__global__ void __launch_bounds__(256) stem_pack_kernel(const float* m, int K, int C, int R, int ldm, int cpad, T* out) {
    const int A = (R + 1) / 2, ldp = A * A * cpad;
    const int64_t total = (int64_t)K * ldp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldp), k = (int)(i / ldp);
        const int q = col % cpad, ab = col / cpad, a = ab / A, b = ab - a * A;
        float v = 0.f;
        if (q < 4 * C) {
            const int sub = q / C, c = q - sub * C;
            const int r = 2 * a + (sub >> 1) - 1, s = 2 * b + (sub & 1) - 1;
            if (r >= 0 && r < R && s >= 0 && s < R) v = m[(int64_t)k * ldm + (r * R + s) * C + c];
        }
        PT<T>::store1(out + i, v);
    }
}
__global__ void __launch_bounds__(256) stem_unpack_kernel(const float* p, int K, int C, int R, int cpad, int ldm, float* m) {
    const int A = (R + 1) / 2, ldp = A * A * cpad;
    const int64_t total = (int64_t)K * R * R * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int s = (int)((i / C) % R);
        const int r = (int)((i / ((int64_t)C * R)) % R);
        const int k = (int)(i / ((int64_t)C * R * R));
        const int a = (r + 1) / 2, di = (r + 1) & 1, b = (s + 1) / 2, dj = (s + 1) & 1;
        m[(int64_t)k * ldm + (r * R + s) * C + c] = p[(int64_t)k * ldp + (a * A + b) * cpad + (di * 2 + dj) * C + c];
    }
}
}  // namespace

extern "C" int iif_space_to_depth_nchw(const float* img, int n, int c, int h, int w, int cpad, int out_dtype, void* out,
                                       void* stream) {
    if (!img || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || cpad < 4 * c) return IIF_EINVAL;
    // vector stores: whole 16-byte groups per pixel, 8-byte aligned image rows
    if (cpad % 8 || cpad > 32 || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(img) & 7)) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t tot = (int64_t)n * (h / 2) * (w / 2);
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(s2d_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, img, n, c, h, w, cpad, (float*)out),
        hipLaunchKernelGGL(s2d_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, img, n, c, h, w, cpad, (unsigned short*)out))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_stem_s2d_pack(const float* master, int k, int c, int r, int ldm, int cpad, int out_dtype, void* out,
                                 void* stream) {
    if (!master || !out || k <= 0 || c <= 0 || r <= 0 || !(r & 1) || ldm < r * r * c || cpad < 4 * c) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const int a = (r + 1) / 2;
    const int64_t tot = (int64_t)k * a * a * cpad;
    IIF_BY_DTYPE(out_dtype,
        hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(sblocks(tot)), dim3(256), 0, st, master, k, c, r, ldm, cpad, (float*)out),
        hipLaunchKernelGGL(stem_pack_kernel<unsigned short>, dim3(sblocks(tot)), dim3(256), 0, st, master, k, c, r, ldm, cpad, (unsigned short*)out))
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_stem_s2d_unpack_grad(const float* packed, int k, int c, int r, int cpad, int ldm, float* master,
                                        void* stream) {
    if (!packed || !master || k <= 0 || c <= 0 || r <= 0 || !(r & 1) || ldm < r * r * c || cpad < 4 * c) return IIF_EINVAL;
    const int64_t tot = (int64_t)k * r * r * c;
    hipLaunchKernelGGL(stem_unpack_kernel, dim3(sblocks(tot)), dim3(256), 0, as_stream(stream), packed, k, c, r, cpad, ldm,
                       master);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}
