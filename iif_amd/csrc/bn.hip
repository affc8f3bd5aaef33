// Training-mode batch normalisation on NHWC activations, gfx950.
//
// The activation is a row-major [M, C] matrix (M = N*H*W pixels, C channels
// fastest), so per-channel statistics are column reductions.  Every thread owns
// one 16-byte channel vector (8 bf16 / 4 f32) and walks rows; a wave therefore
// reads whole 128-byte lines.  All kernels are HBM-bound streaming passes:
//   stats    : x -> per-block partial (sum, sumsq)            [1 read]
//   finalize : partials -> mean, invstd, scale a, shift b, running stats (fp64 sums)
//   apply    : y = relu?(a*x + b (+ a2*r + b2 | + r))          [1-2 reads, 1 write]
//   bwd_red  : partial (sum dy, sum dy*xhat), dy = g * [y > 0] [2-3 reads]
//   bwd_fin  : dgamma, dbeta, per-channel coefficients
//   bwd_apply: dx = k1*(dy - k2 - (x-mean)*k3) (+ masked g out) [2-3 reads, 1-2 writes]
// Partials are reduced in a fixed order (no float atomics): deterministic.
// Replaces torch's batch_norm / relu / add forward+backward under
// classification/resnet_pytorch.py:152-167 and resnet_cifar.py:133-138.
#include <stdlib.h>

#include "common.h"
#include "vec16.h"
#include "pool_gather.h"

namespace {

// thread -> (channel vector, row lane) for a [M, C] matrix
struct Map {
    int cvb;      // channel vectors handled per block row (<= 256)
    int rpb;      // rows per block pass = 256 / cvb
};

struct Geo {
    int M, C, cv, cvb, rpb, colblocks, rows_per_block, nblk;
};

inline Geo make_geo(int64_t M, int C, int V, int target_blocks) {
    Geo g;
    g.M = (int)M; g.C = C; g.cv = C / V;
    g.cvb = g.cv < 256 ? g.cv : 256;
    g.rpb = 256 / g.cvb;
    g.colblocks = (g.cv + g.cvb - 1) / g.cvb;
    int nblk = target_blocks / g.colblocks;
    if (nblk < 1) nblk = 1;
    int64_t rows = (M + nblk - 1) / nblk;
    rows = (rows + g.rpb - 1) / g.rpb * g.rpb;
    if (rows < g.rpb) rows = g.rpb;
    g.rows_per_block = (int)rows;
    g.nblk = (int)((M + rows - 1) / rows);
    return g;
}

// ------------------------------------------------------------------ forward
template <typename T>
__global__ void __launch_bounds__(256) bn_stats_kernel(const T* x, Geo g, float* partial) {
    constexpr int V = VT<T>::V;
    __shared__ float sh[2][256 * V];
    const int tid = threadIdx.x;
    const int cvl = tid % g.cvb, rl = tid / g.cvb;
    const int cvec = blockIdx.y * g.cvb + cvl;
    float s[V], q[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s[i] = 0.f; q[i] = 0.f; }
    const int r0 = blockIdx.x * g.rows_per_block;
    int r1 = r0 + g.rows_per_block; if (r1 > g.M) r1 = g.M;
    if (rl < g.rpb && cvec < g.cv) {
        for (int r = r0 + rl; r < r1; r += g.rpb) {
            float v[V];
            VT<T>::load(x + (int64_t)r * g.C + cvec * V, v);
#pragma unroll
            for (int i = 0; i < V; ++i) { s[i] += v[i]; q[i] += v[i] * v[i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { sh[0][tid * V + i] = s[i]; sh[1][tid * V + i] = q[i]; }
    __syncthreads();
    // thread (rl == 0) sums its column over the row lanes in fixed order
    if (rl == 0 && cvec < g.cv) {
        for (int j = 1; j < g.rpb; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) { s[i] += sh[0][(j * g.cvb + cvl) * V + i]; q[i] += sh[1][(j * g.cvb + cvl) * V + i]; }
        float* p = partial + (int64_t)blockIdx.x * 2 * g.C + cvec * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { p[i] = s[i]; p[g.C + i] = q[i]; }
    }
}

// Column sums of the [nblk][2][C] partials: a 256-thread block owns CB channels with LN = 256/CB lanes
// each; a lane walks the partial rows with stride LN in double, lanes are combined in fixed order via LDS.
template <int CB>
__device__ __forceinline__ void reduce_partials(const float* partial, int nblk, int C, int c, int ln,
                                                double* sh /* [2][256] */, int cl, double& s, double& q) {
    constexpr int LN = 256 / CB;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 16 rows in flight per lane, requested unconditionally from a clamped row and masked afterwards (a conditional or
        // run-time-counted load is waited for on its own); the sums keep the order of the 4-deep loop: rows ln, ln + LN, ...
        // Round 4: the register-weight convolutions hand over <= 256 rows, i.e. two round trips here instead of eight.
        for (int r0 = ln; r0 < nblk; r0 += 16 * LN) {
            float av[16], bv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = r0 + i * LN;
                const int64_t o = (int64_t)(rr < nblk ? rr : r0) * 2 * C + c;
                av[i] = partial[o]; bv[i] = partial[o + C];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (r0 + i * LN < nblk) { a += av[i]; b += bv[i]; }
            }
        }
    }
    sh[ln * CB + cl] = a; sh[256 + ln * CB + cl] = b;
    __syncthreads();
    s = 0.0; q = 0.0;
    if (ln == 0)
        for (int j = 0; j < LN; ++j) { s += sh[j * CB + cl]; q += sh[256 + j * CB + cl]; }
}

// stats layout: [4][C] = mean, invstd, a (=gamma*invstd), b (=beta-mean*a)
// Blocks >= nb_main (round 6; partial2 non-null): a SECOND, independent job riding in the same launch - the plain column sums of
// another set of partial rows [nblk2][2][C2] into sums2[2][C2] (the column sums of a2 that conv3's prologue left behind: one launch
// and one cross-stream event less per bottleneck than a reduction of its own).
template <int CB>
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* partial, int nblk, int C, double count,
                                                          const float* gamma, const float* beta, float eps,
                                                          float momentum, float* running_mean, float* running_var,
                                                          float* stats, int nb_main = 0x7fffffff, const float* partial2 = nullptr,
                                                          int nblk2 = 0, int C2 = 0, float* sums2 = nullptr) {
    __shared__ double sh[512];
    const int cl = threadIdx.x % CB, ln = threadIdx.x / CB;
    if ((int)blockIdx.x >= nb_main) {
        const int c2 = ((int)blockIdx.x - nb_main) * CB + cl;
        double s2, q2;
        reduce_partials<CB>(partial2, nblk2, C2, c2, ln, sh, cl, s2, q2);
        if (ln == 0 && c2 < C2) { sums2[c2] = (float)s2; sums2[C2 + c2] = (float)q2; }
        return;
    }
    const int c = blockIdx.x * CB + cl;
    // per-channel parameters are fetched together with the partial rows, not in a second round trip after them
    float ga = 0.f, be = 0.f, rm = 0.f, rv = 0.f;
    if (ln == 0 && c < C) {
        ga = gamma[c]; be = beta[c];
        if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    }
    double s, q;
    reduce_partials<CB>(partial, nblk, C, c, ln, sh, cl, s, q);
    if (ln != 0 || c >= C) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float a = ga * invstd;
    stats[c] = (float)mean;
    stats[C + c] = invstd;
    stats[2 * C + c] = a;
    stats[3 * C + c] = be - (float)mean * a;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
}

// relu_bits (nullable): one byte per channel vector, bit k = [pre-activation k > 0].  The backward passes read
// this byte instead of the 16-byte activated vector: 1/16 of the mask traffic.
// Loop form (round 5, scripts/micro/stream_rw.hip: this pass's shape on 411 MB tensors): a resident grid striding over the
// tensor one vector per trip ran at 5.4 TB/s - the stores of blocks that have drifted apart land in DRAM pages 16 MB apart
// (a write-only sweep of the same form: 4.2 TB/s at 4 096 blocks, 6.1 at 16 384); ONE trip per thread over U vectors 256 apart
// (the same channel vector, so the coefficients are still loaded once), every load requested before the first use,
// non-temporal loads: 6.2 TB/s (6.6 with non-temporal stores too).  The grid is as large as the tensor (stream_grid below).
template <typename T, bool RELU, int RES, int U, bool NTS>   // RES 0: none, 1: + r, 2: + a2*r + b2
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* x, const float* stats, const T* r, const float* stats2,
                                                       T* y, int64_t total_vec, int cv, int C, unsigned char* relu_bits) {
    constexpr int V = VT<T>::V;
    // cv divides 256: a thread keeps ONE channel vector for all its vectors (a block's trips start 256 U apart), so the
    // 2 x V (4 x V with a normalised residual) coefficients are loaded once, not per 16-byte vector
    const bool fixed = (256 % cv) == 0;
    float ca[V], cb[V], ra[V], rb[V];
    int c0 = (int)(threadIdx.x % cv) * V;
    if (fixed) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            ca[k] = stats[2 * C + c0 + k]; cb[k] = stats[3 * C + c0 + k];
            if (RES == 2) { ra[k] = stats2[2 * C + c0 + k]; rb[k] = stats2[3 * C + c0 + k]; }
        }
    }
    for (int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x; base < total_vec; base += (int64_t)gridDim.x * (256 * U)) {
        u32x4 xr[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                     // (a vector past the end reads the trip's first one again)
            const int64_t i = base + 256 * u < total_vec ? base + 256 * u : base;
            xr[u] = VT<T>::raw_nt(x + i * V);
            if (RES) rr[u] = VT<T>::raw_nt(r + i * V);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + 256 * u;
            if (!fixed) {
                c0 = (int)(i % cv) * V;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    ca[k] = stats[2 * C + c0 + k]; cb[k] = stats[3 * C + c0 + k];
                    if (RES == 2) { ra[k] = stats2[2 * C + c0 + k]; rb[k] = stats2[3 * C + c0 + k]; }
                }
            }
            float v[V], w[V];
            VT<T>::unpack(xr[u], v);
            if (RES) VT<T>::unpack(rr[u], w);
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float t = fmaf(ca[k], v[k], cb[k]);   // one rounding, as vec fmadd
                if (RES == 1) t += w[k];
                if (RES == 2) t += fmaf(ra[k], w[k], rb[k]);
                bits |= (t > 0.f ? 1u : 0u) << k;
                v[k] = RELU ? fmaxf(t, 0.f) : t;
            }
            if (i < total_vec) {
                VT<T>::template store_as<NTS>(y + i * V, v);
                if (RELU && relu_bits) relu_bits[i] = (unsigned char)bits;
            }
        }
    }
}

// ----------------------------------------------------------------- backward
// MASK 0: no ReLU, 1: mask from the activated tensor, 2: mask from the bit bytes, 3: recomputed as a*x + b > 0 (the
// same fmaf as bn_apply: identical decisions) for units whose activation is never stored (stem fused with the max pool)
template <typename T, int MASK>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* g_, const T* ymask, const unsigned char* bits,
                                                            const T* x, const float* stats, Geo g, float* partial) {
    constexpr int V = VT<T>::V;
    __shared__ float sh[2][256 * V];
    const int tid = threadIdx.x;
    const int cvl = tid % g.cvb, rl = tid / g.cvb;
    const int cvec = blockIdx.y * g.cvb + cvl;
    float s1[V], s2[V], mean[V], istd[V], aa[V], bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s1[i] = 0.f; s2[i] = 0.f; mean[i] = 0.f; istd[i] = 0.f; aa[i] = 0.f; bb[i] = 0.f; }
    const int r0 = blockIdx.x * g.rows_per_block;
    int r1 = r0 + g.rows_per_block; if (r1 > g.M) r1 = g.M;
    if (rl < g.rpb && cvec < g.cv) {
#pragma unroll
        for (int i = 0; i < V; ++i) { mean[i] = stats[cvec * V + i]; istd[i] = stats[g.C + cvec * V + i]; }
        if (MASK == 3) {
#pragma unroll
            for (int i = 0; i < V; ++i) { aa[i] = stats[2 * g.C + cvec * V + i]; bb[i] = stats[3 * g.C + cvec * V + i]; }
        }
        for (int r = r0 + rl; r < r1; r += g.rpb) {
            const int64_t o = (int64_t)r * g.C + cvec * V;
            float dy[V], xv[V], yv[V];
            VT<T>::load(g_ + o, dy);
            VT<T>::load(x + o, xv);
            if (MASK == 1) VT<T>::load(ymask + o, yv);
            unsigned mb = 0;
            if (MASK == 2) mb = bits[(int64_t)r * g.cv + cvec];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const bool on = MASK == 0 ? true : (MASK == 1 ? yv[i] > 0.f : (MASK == 2 ? ((mb >> i) & 1u) != 0
                                                                                                : fmaf(aa[i], xv[i], bb[i]) > 0.f));
                const float d = on ? dy[i] : 0.f;
                s1[i] += d;
                s2[i] += d * ((xv[i] - mean[i]) * istd[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { sh[0][tid * V + i] = s1[i]; sh[1][tid * V + i] = s2[i]; }
    __syncthreads();
    if (rl == 0 && cvec < g.cv) {
        for (int j = 1; j < g.rpb; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) { s1[i] += sh[0][(j * g.cvb + cvl) * V + i]; s2[i] += sh[1][(j * g.cvb + cvl) * V + i]; }
        float* p = partial + (int64_t)blockIdx.x * 2 * g.C + cvec * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { p[i] = s1[i]; p[g.C + i] = s2[i]; }
    }
}

// coef layout [3][C]: k1 = gamma*invstd, k2 = mean(dy), k3 = mean(dy*xhat)*invstd
template <int CB>
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* partial, int nblk, int C, double count,
                                                              const float* gamma, const float* stats, float* dgamma,
                                                              float* dbeta, float* coef) {
    __shared__ double sh[512];
    const int cl = threadIdx.x % CB, ln = threadIdx.x / CB;
    const int c = blockIdx.x * CB + cl;
    float ga = 0.f, invstd = 0.f;
    if (ln == 0 && c < C) { ga = gamma[c]; invstd = stats[C + c]; }
    double s1, s2;
    reduce_partials<CB>(partial, nblk, C, c, ln, sh, cl, s1, s2);
    if (ln != 0 || c >= C) return;
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    coef[c] = ga * invstd;
    coef[C + c] = (float)(s1 / count);
    coef[2 * C + c] = (float)(s2 / count) * invstd;
}

// Waves per SIMD asked of the compiler: 8 (<= 64 VGPRs, no spills) for the one-vector form - the small tensors (the two- and
// four-vector forms spill under such a bound and keep the compiler's 113-138).  At 98 VGPRs
// a wave of this kernel did not fit beside the nine-tap weight-gradient blocks of the other stream (two per CU, 448 of a SIMD's
// 512 registers): the launch waited for that kernel to END, 80-90 us for a 15 us pass at 7 x 7 (profiles/r5_g_step_listing.txt).
template <typename T, int MASK, bool GMOUT, int U, bool NTS>      // (loop form: see bn_apply_kernel)
__global__ void __launch_bounds__(256, (U == 1 ? 8 : 1)) bn_bwd_apply_kernel(const T* g_, const T* ymask, const unsigned char* bits, const T* x,
                                                           const float* stats, const float* coef, T* dx, T* gm,
                                                           int64_t total_vec, int cv, int C) {
    constexpr int V = VT<T>::V;
    const bool fixed = (256 % cv) == 0;          // see bn_apply_kernel: one channel vector per thread
    float k1[V], k2[V], k3[V], mu[V], aa[V], bb[V];
    int c0 = (int)(threadIdx.x % cv) * V;
    if (fixed) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k]; k3[k] = coef[2 * C + c0 + k]; mu[k] = stats[c0 + k];
            if (MASK == 3) { aa[k] = stats[2 * C + c0 + k]; bb[k] = stats[3 * C + c0 + k]; }
        }
    }
    for (int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x; base < total_vec; base += (int64_t)gridDim.x * (256 * U)) {
        u32x4 gr[U], xr[U], yr[U];
        unsigned mbv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                     // in place (dx == g_) is fine: a thread reads all its vectors before it writes any
            const int64_t i = base + 256 * u < total_vec ? base + 256 * u : base;
            gr[u] = VT<T>::raw_nt(g_ + i * V);
            xr[u] = VT<T>::raw_nt(x + i * V);
            if (MASK == 1) yr[u] = VT<T>::raw(ymask + i * V);
            mbv[u] = MASK == 2 ? bits[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + 256 * u;
            if (!fixed) {
                c0 = (int)(i % cv) * V;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k]; k3[k] = coef[2 * C + c0 + k]; mu[k] = stats[c0 + k];
                    if (MASK == 3) { aa[k] = stats[2 * C + c0 + k]; bb[k] = stats[3 * C + c0 + k]; }
                }
            }
            float dy[V], xv[V], yv[V];
            VT<T>::unpack(gr[u], dy);
            VT<T>::unpack(xr[u], xv);
            if (MASK == 1) VT<T>::unpack(yr[u], yv);
            const unsigned mb = mbv[u];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const bool on = MASK == 0 ? true : (MASK == 1 ? yv[k] > 0.f : (MASK == 2 ? ((mb >> k) & 1u) != 0
                                                                                                : fmaf(aa[k], xv[k], bb[k]) > 0.f));
                const float d = on ? dy[k] : 0.f;
                dy[k] = d;
                xv[k] = k1[k] * (d - k2[k] - (xv[k] - mu[k]) * k3[k]);
            }
            if (i < total_vec) {
                VT<T>::template store_as<NTS>(dx + i * V, xv);
                if (GMOUT) VT<T>::template store_as<false>(gm + i * V, dy);
            }
        }
    }
}

// Sum of rows r, r + 8, r + 16, ... < r1 of column c (both halves of the [2][C] rows) in ROW ORDER, 8 loads in flight: under
// load a dependent round trip costs 3-5 us, so the depth of this chain is what the small reduction launches take
// (4 in flight: 17 us per launch inside the step).  The order of the additions is the one the 4-deep loop had.
__device__ __forceinline__ void lane_sums(const float* partial, int C, int c, int r, int r1, double& a, double& b) {
    // 16 rows (stride 8) per batch, requested unconditionally from a clamped row and masked afterwards: a slice of <= 128 rows is
    // ONE round trip (round 3: 8 + 4 + 1 in flight in three loops; a slice of 98 rows took three trips and a serial tail)
    for (int r0 = r; r0 < r1; r0 += 128) {
        float av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = r0 + 8 * i;
            const int64_t o = (int64_t)(rr < r1 ? rr : r0) * 2 * C + c;
            av[i] = partial[o]; bv[i] = partial[o + C];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (r0 + 8 * i < r1) { a += av[i]; b += bv[i]; }
        }
    }
}

// Stage 1 for many partial rows: slice `blockIdx.y` of the rows is summed (double, fixed order) by 32 channels
// x 8 lanes per block into out[slice][2][C]; the finalize kernels then see only `slices` rows.
__global__ void __launch_bounds__(256) bn_partial_reduce_kernel(const float* partial, int nblk, int C, int rows_per_slice,
                                                                float* out) {
    __shared__ double sh[512];
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per_slice;
    int r1 = r0 + rows_per_slice; if (r1 > nblk) r1 = nblk;
    double a = 0.0, b = 0.0;
    if (c < C) {
        lane_sums(partial, C, c, r0 + ln, r1, a, b);
    }
    sh[ln * 32 + cl] = a; sh[256 + ln * 32 + cl] = b;
    __syncthreads();
    if (ln == 0 && c < C) {
        double s = 0.0, q = 0.0;
        for (int j = 0; j < 8; ++j) { s += sh[j * 32 + cl]; q += sh[256 + j * 32 + cl]; }
        out[(int64_t)blockIdx.y * 2 * C + c] = (float)s;
        out[(int64_t)blockIdx.y * 2 * C + C + c] = (float)q;
    }
}

// Both stages in ONE launch.  Stage 1 as bn_partial_reduce_kernel (slice blockIdx.y, 32 channels blockIdx.x); the slice
// sums are published with agent-scope atomic exchanges (performed at the coherence point; a returning atomic has
// completed when its value is back), then the block takes a ticket of its channel group; the block that draws the last
// ticket reads all slices back with agent-scope atomic loads, sums them in slice order (double) and finalises its 32
// channels.  No fence: an agent-scope release would write back the whole L2 (measured on the IIF loss kernel: 40 % of
// its time), the ordinary stores of this kernel need no ordering at all.  tickets: int32[>= gridDim.x], zero on entry,
// zero again on exit.  MODE 0: forward statistics (+ running statistics), MODE 1: backward sums -> dgamma, dbeta, coef.
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_finalize_kernel(const float* partial, int nblk, int C, int rows_per_slice,
                                                                 float* slices_out, int* tickets, double count,
                                                                 const float* gamma, const float* beta_or_stats, float eps,
                                                                 float momentum, float* o0, float* o1, float* o2) {
    __shared__ double sh[512];
    __shared__ int last;
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per_slice;
    int r1 = r0 + rows_per_slice; if (r1 > nblk) r1 = nblk;
    double a = 0.0, b = 0.0;
    // what the finishing block needs besides the sums travels with the first batch of loads, not in a round trip of its own
    float pre_ga = 0.f, pre_b = 0.f, pre_rm = 0.f, pre_rv = 0.f;
    if (ln == 0 && c < C) {
        pre_ga = gamma[c];
        pre_b = MODE == 0 ? beta_or_stats[c] : beta_or_stats[C + c];
        if (MODE == 0 && o1) { pre_rm = o1[c]; pre_rv = o2[c]; }
    }
    if (c < C) {
        lane_sums(partial, C, c, r0 + ln, r1, a, b);
    }
    sh[ln * 32 + cl] = a; sh[256 + ln * 32 + cl] = b;
    __syncthreads();
    if (ln == 0) {                                      // threads 0..31: one wave
        float keep = 0.f;
        if (c < C) {
            double s = 0.0, q = 0.0;
            for (int j = 0; j < 8; ++j) { s += sh[j * 32 + cl]; q += sh[256 + j * 32 + cl]; }
            float* o = slices_out + (int64_t)blockIdx.y * 2 * C + c;
            keep = __hip_atomic_exchange(o, (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            keep += __hip_atomic_exchange(o + C, (float)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(keep) : "memory");       // every lane's exchanges have returned
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == (int)gridDim.y - 1);
        }
    }
    __syncthreads();
    if (!last) return;
    // ---- the last block of this channel group: slices in order, 8 lanes x 32 channels
    const int ns = (int)gridDim.y;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int j = ln;
        for (; j + 56 < ns; j += 64) {                   // 8 slices in flight per lane, added in slice order
            float sv[8], qv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sv[i] = __hip_atomic_load(slices_out + (int64_t)(j + 8 * i) * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qv[i] = __hip_atomic_load(slices_out + (int64_t)(j + 8 * i) * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s += (double)sv[i]; q += (double)qv[i]; }
        }
        for (; j < ns; j += 8) {
            s += (double)__hip_atomic_load(slices_out + (int64_t)j * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q += (double)__hip_atomic_load(slices_out + (int64_t)j * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    sh[ln * 32 + cl] = s; sh[256 + ln * 32 + cl] = q;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ln != 0 || c >= C) return;
    s = 0.0; q = 0.0;
    for (int j = 0; j < 8; ++j) { s += sh[j * 32 + cl]; q += sh[256 + j * 32 + cl]; }
    if (MODE == 0) {                                    // o0 = stats[4][C], o1 / o2 = running mean / var (nullable)
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float ga = pre_ga, be = pre_b;
        const float k = ga * invstd;
        o0[c] = (float)mean; o0[C + c] = invstd; o0[2 * C + c] = k; o0[3 * C + c] = be - (float)mean * k;
        if (o1) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            o1[c] = (1.f - momentum) * pre_rm + momentum * (float)mean;
            o2[c] = (1.f - momentum) * pre_rv + momentum * (float)unbiased;
        }
    } else {                                            // o0 = dgamma, o1 = dbeta, o2 = coef[3][C]
        const float ga = pre_ga, invstd = pre_b;
        o0[c] = (float)q;
        o1[c] = (float)s;
        o2[c] = ga * invstd;
        o2[C + c] = (float)(s / count);
        o2[2 * C + c] = (float)(q / count) * invstd;
    }
}

// slices of the first reduction stage: 64, or 256 for very many partial rows (the stem of the 224-pixel networks: 25 088 rows
// -> 392 rows per slice took 105 us in two blocks' worth of lanes) when the scratch rows are there.  Both the one-launch
// and the two-launch path use this, so they stay bit-identical to each other.
inline int stage1_slices(int n_partials, int c, int64_t scratch_floats) {
    return (n_partials > 8192 && scratch_floats >= (int64_t)256 * 2 * c) ? 256 : 64;
}

// partial rows above which the sums are pre-reduced into 64 slices by a separate launch
inline int two_stage_rows() {
    // round 5, same-call A/B of the step (ms): 512 rows 17.56, 1024 17.52, 2048 17.61, 4096 17.58 (784-row sums of the 28 x 28
    // stage are quicker in one stage of 8 channels x 32 lanes than through slices + ticket)
    static const int v = [] { const char* e = getenv("IIF_BN_TWO_STAGE_ROWS"); return e ? atoi(e) : 1024; }();
    return v;
}

// channels per finalize block: few channels x many lanes when there are many partial rows
inline int finalize_cb(int nblk) {
    return nblk > 2048 ? 4 : (nblk > 256 ? 8 : 32);
}

inline int launch_bn_finalize(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta,
                              float eps, float momentum, float* rm, float* rv, float* stats, hipStream_t st,
                              const float* partial2 = nullptr, int nblk2 = 0, int C2 = 0, float* sums2 = nullptr) {
    const int cb = finalize_cb(nblk > nblk2 ? nblk : nblk2);
    const int nb = (C + cb - 1) / cb, nb2 = partial2 ? (C2 + cb - 1) / cb : 0;
    const dim3 grid(nb + nb2), blk(256);
    if (cb == 4) hipLaunchKernelGGL(bn_finalize_kernel<4>, grid, blk, 0, st, partial, nblk, C, count, gamma, beta, eps, momentum, rm, rv, stats, nb, partial2, nblk2, C2, sums2);
    else if (cb == 8) hipLaunchKernelGGL(bn_finalize_kernel<8>, grid, blk, 0, st, partial, nblk, C, count, gamma, beta, eps, momentum, rm, rv, stats, nb, partial2, nblk2, C2, sums2);
    else hipLaunchKernelGGL(bn_finalize_kernel<32>, grid, blk, 0, st, partial, nblk, C, count, gamma, beta, eps, momentum, rm, rv, stats, nb, partial2, nblk2, C2, sums2);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// grid of the two normalisation passes: one trip per thread over U vectors (U = 4 where that still leaves >= 2 048 blocks).
// In the ResNet-50 step (same-call A/B, ms per step): round 4's form (4 096 blocks looping, U = 1) 18.59 / 18.62; one trip,
// U <= 4 18.32 / 18.33; the same with non-temporal stores 18.34 / 18.41 (alone they are the faster form, 6.6 against 6.2 TB/s;
// in the step the consumer reads the tensor next); U = 1 one trip 19.09 / 19.04 (100 k blocks of one vector per thread: alone
// 6.1 TB/s, in the step the other streams' blocks wait behind them); a grid capped at 4 096 / 2 048 / 1 024 blocks with U <= 4
// 18.42 / 18.47 / 18.51.  IIF_BN_GRID_CAP=<blocks>, IIF_BN_UNROLL=<1|2|4>, IIF_BN_NT_STORES=1 select the other forms.
struct StreamGrid { int blocks, u; bool nts; };
inline StreamGrid stream_grid(int64_t total_vec) {
    static const int64_t cap = [] { const char* e = getenv("IIF_BN_GRID_CAP"); return e ? atoll(e) : (1LL << 30); }();
    static const bool nts = getenv("IIF_BN_NT_STORES") != nullptr;
    static const int umax = [] { const char* e = getenv("IIF_BN_UNROLL"); return e ? atoi(e) : 4; }();
    int u = total_vec >= (int64_t)4 * 256 * 2048 ? 4 : (total_vec >= (int64_t)2 * 256 * 2048 ? 2 : 1);
    if (u > umax) u = umax;
    const int64_t b = (total_vec + 256 * u - 1) / (256 * u);
    return StreamGrid{(int)(b < cap ? b : cap), u, nts};
}

template <typename T>
int bn_forward_t(const T* x, int64_t M, int C, const float* gamma, const float* beta, float eps, float momentum,
                 float* rm, float* rv, float* stats, float* ws, int64_t ws_bytes, hipStream_t st) {
    constexpr int V = VT<T>::V;
    Geo g = make_geo(M, C, V, 512);
    if ((int64_t)g.nblk * 2 * C * 4 > ws_bytes) return IIF_EINVAL;
    hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(g.nblk, g.colblocks), dim3(256), 0, st, x, g, ws);
    IIF_LAUNCH_CHECK();
    return launch_bn_finalize(ws, g.nblk, C, (double)M, gamma, beta, eps, momentum, rm, rv, stats, st);
}

template <typename T>
int bn_apply_t(const T* x, const float* stats, const T* r, const float* stats2, T* y, int64_t M, int C, int relu,
               unsigned char* relu_bits, hipStream_t st) {
    constexpr int V = VT<T>::V;
    const int cv = C / V;
    const int64_t tv = M * cv;
    const StreamGrid sg = stream_grid(tv);
    const dim3 grid(sg.blocks), blk(256);
    const int res = r ? (stats2 ? 2 : 1) : 0;
#define IIF_APPLY3(RL, RS, UU, NT) hipLaunchKernelGGL((bn_apply_kernel<T, RL, RS, UU, NT>), grid, blk, 0, st, x, stats, r, stats2, y, tv, cv, C, relu_bits)
#define IIF_APPLY2(RL, RS, UU) do { if (sg.nts) IIF_APPLY3(RL, RS, UU, true); else IIF_APPLY3(RL, RS, UU, false); } while (0)
#define IIF_APPLY(RL, RS) do { if (sg.u == 4) IIF_APPLY2(RL, RS, 4); else if (sg.u == 2) IIF_APPLY2(RL, RS, 2); else IIF_APPLY2(RL, RS, 1); } while (0)
    if (relu) { if (res == 0) IIF_APPLY(true, 0); else if (res == 1) IIF_APPLY(true, 1); else IIF_APPLY(true, 2); }
    else { if (res == 0) IIF_APPLY(false, 0); else if (res == 1) IIF_APPLY(false, 1); else IIF_APPLY(false, 2); }
#undef IIF_APPLY
#undef IIF_APPLY2
#undef IIF_APPLY3
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// ext_partial (nullable): [n_ext][2][C] rows of (sum dy, sum dy*xhat) already produced elsewhere (the epilogue of the
// data-gradient convolution that wrote gy): the reduction pass over gy and x is skipped.
template <typename T>
int bn_backward_t(const T* gy, const T* ymask, const unsigned char* bits, const T* x, const float* stats, const float* gamma,
                  int64_t M, int C, float* dgamma, float* dbeta, T* dx, T* gm, float* ws, int64_t ws_bytes, hipStream_t st,
                  const float* ext_partial = nullptr, int n_ext = 0, bool recompute = false, int* tickets = nullptr) {
    constexpr int V = VT<T>::V;
    Geo g = make_geo(M, C, V, 512);
    const int64_t need = ((int64_t)g.nblk * 2 * C + 3 * C) * 4;
    if (need > ws_bytes) return IIF_EINVAL;
    float* coef = ws + (int64_t)g.nblk * 2 * C;
    const dim3 rgrid(g.nblk, g.colblocks), blk(256);
    const float* fin_src = ws;
    bool finalized = false;
    if (ext_partial) {
        fin_src = ext_partial;
        g.nblk = n_ext;
        if (n_ext > two_stage_rows() && tickets) {   // both stages in one launch (see bn_reduce_finalize_kernel)
            const int slices = 64, rps = (n_ext + slices - 1) / slices;
            if ((int64_t)(slices * 2 * C + 3 * C) * 4 > ws_bytes || (C + 31) / 32 > 64) return IIF_EINVAL;
            coef = ws + (int64_t)slices * 2 * C;
            hipLaunchKernelGGL(bn_reduce_finalize_kernel<1>, dim3((C + 31) / 32, slices), blk, 0, st, ext_partial, n_ext, C, rps, ws,
                               tickets, (double)M, gamma, stats, 0.f, 0.f, dgamma, dbeta, coef);
            IIF_LAUNCH_CHECK();
            finalized = true;
        } else if (n_ext > two_stage_rows()) {   // many tile rows: 64 slices first (fixed order), as in the forward path
            const int slices = 64, rps = (n_ext + slices - 1) / slices;
            if ((int64_t)(slices * 2 * C + 3 * C) * 4 > ws_bytes) return IIF_EINVAL;
            hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((C + 31) / 32, slices), blk, 0, st, ext_partial, n_ext, C, rps, ws);
            IIF_LAUNCH_CHECK();
            fin_src = ws; g.nblk = slices;
            coef = ws + (int64_t)slices * 2 * C;
        }
    } else {
        if (recompute) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 3>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
        else if (bits) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
        else if (ymask) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
        else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
        IIF_LAUNCH_CHECK();
    }
    if (!finalized) {
        const int cb = finalize_cb(g.nblk);
        const dim3 fgrid((C + cb - 1) / cb);
        if (cb == 4) hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, fgrid, blk, 0, st, fin_src, g.nblk, C, (double)M, gamma, stats, dgamma, dbeta, coef);
        else if (cb == 8) hipLaunchKernelGGL(bn_bwd_finalize_kernel<8>, fgrid, blk, 0, st, fin_src, g.nblk, C, (double)M, gamma, stats, dgamma, dbeta, coef);
        else hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, fgrid, blk, 0, st, fin_src, g.nblk, C, (double)M, gamma, stats, dgamma, dbeta, coef);
    }
    IIF_LAUNCH_CHECK();
    const int cv = C / V;
    const int64_t tv = M * cv;
    const StreamGrid sg = stream_grid(tv);
    const dim3 agrid(sg.blocks);
#define IIF_BAPPLY3(MK, GO, UU, NT) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MK, GO, UU, NT>), agrid, blk, 0, st, gy, ymask, bits, x, stats, coef, dx, gm, tv, cv, C)
#define IIF_BAPPLY2(MK, GO, UU) do { if (sg.nts) IIF_BAPPLY3(MK, GO, UU, true); else IIF_BAPPLY3(MK, GO, UU, false); } while (0)
#define IIF_BAPPLY(MK, GO) do { if (sg.u == 4) IIF_BAPPLY2(MK, GO, 4); else if (sg.u == 2) IIF_BAPPLY2(MK, GO, 2); else IIF_BAPPLY2(MK, GO, 1); } while (0)
    if (recompute) { if (gm) IIF_BAPPLY(3, true); else IIF_BAPPLY(3, false); }
    else if (bits) { if (gm) IIF_BAPPLY(2, true); else IIF_BAPPLY(2, false); }
    else if (ymask) { if (gm) IIF_BAPPLY(1, true); else IIF_BAPPLY(1, false); }
    else { if (gm) IIF_BAPPLY(0, true); else IIF_BAPPLY(0, false); }
#undef IIF_BAPPLY
#undef IIF_BAPPLY2
#undef IIF_BAPPLY3
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// ---------------------------------------------------------------- cross-replica statistics (SyncBatchNorm)
// The reduction and the normalisation are separate entry points, so the host can all-reduce the per-channel sums of all
// ranks in between (classification/train.py:190-191 converts to nn.SyncBatchNorm: statistics over the global batch).
//   forward : sums[0][c] = sum x, sums[1][c] = sum x^2 of THIS rank -> all-reduce -> iif_bn_finalize_stats(sums, 1 row,
//             m * world): the same finalisation as the single-rank path, on a single "partial row".
//   backward: local (sum g, sum g*xhat) -> dgamma / dbeta stay LOCAL sums (the gradient all-reduce averages them like
//             every other parameter gradient), the subtraction terms mean(g), mean(g*xhat) use the all-reduced sums and
//             the global count.
inline int launch_column_sums(const float* partial, int n, int C, float* sums, hipStream_t st) {
    hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, st, partial, n, C, n, sums);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

__global__ void __launch_bounds__(256) bn_bwd_coef_kernel(const float* local, const float* total, int C, double count, const float* gamma,
                                                          const float* stats, float* dgamma, float* dbeta, float* coef) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float invstd = stats[C + c];
    dbeta[c] = local[c];
    dgamma[c] = local[C + c];
    coef[c] = gamma[c] * invstd;
    coef[C + c] = (float)((double)total[c] / count);
    coef[2 * C + c] = (float)((double)total[C + c] / count) * invstd;
}

template <typename T>
int bn_stats_sums_t(const T* x, int64_t M, int C, float* sums, float* ws, int64_t ws_bytes, hipStream_t st) {
    constexpr int V = VT<T>::V;
    Geo g = make_geo(M, C, V, 512);
    if ((int64_t)g.nblk * 2 * C * 4 > ws_bytes) return IIF_EINVAL;
    hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(g.nblk, g.colblocks), dim3(256), 0, st, x, g, ws);
    IIF_LAUNCH_CHECK();
    return launch_column_sums(ws, g.nblk, C, sums, st);
}

template <typename T>
int bn_backward_sums_t(const T* gy, const T* ymask, const unsigned char* bits, const T* x, const float* stats, int64_t M, int C,
                       float* sums, float* ws, int64_t ws_bytes, hipStream_t st) {
    constexpr int V = VT<T>::V;
    Geo g = make_geo(M, C, V, 512);
    if ((int64_t)g.nblk * 2 * C * 4 > ws_bytes) return IIF_EINVAL;
    const dim3 rgrid(g.nblk, g.colblocks), blk(256);
    if (bits) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
    else if (ymask) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0>), rgrid, blk, 0, st, gy, ymask, bits, x, stats, g, ws);
    IIF_LAUNCH_CHECK();
    return launch_column_sums(ws, g.nblk, C, sums, st);
}

template <typename T>
int bn_backward_apply_sums_t(const T* gy, const T* ymask, const unsigned char* bits, const T* x, const float* stats, const float* gamma,
                             const float* local, const float* total, double count, int64_t M, int C, float* dgamma, float* dbeta,
                             T* dx, T* gm, float* coef, hipStream_t st) {
    constexpr int V = VT<T>::V;
    const dim3 blk(256);
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), blk, 0, st, local, total, C, count, gamma, stats, dgamma, dbeta, coef);
    IIF_LAUNCH_CHECK();
    const int cv = C / V;
    const int64_t tv = M * cv;
    const StreamGrid sg = stream_grid(tv);
    const dim3 agrid(sg.blocks);
#define IIF_BAPPLY3(MK, GO, UU, NT) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MK, GO, UU, NT>), agrid, blk, 0, st, gy, ymask, bits, x, stats, coef, dx, gm, tv, cv, C)
#define IIF_BAPPLY2(MK, GO, UU) do { if (sg.nts) IIF_BAPPLY3(MK, GO, UU, true); else IIF_BAPPLY3(MK, GO, UU, false); } while (0)
#define IIF_BAPPLY(MK, GO) do { if (sg.u == 4) IIF_BAPPLY2(MK, GO, 4); else if (sg.u == 2) IIF_BAPPLY2(MK, GO, 2); else IIF_BAPPLY2(MK, GO, 1); } while (0)
    if (bits) { if (gm) IIF_BAPPLY(2, true); else IIF_BAPPLY(2, false); }
    else if (ymask) { if (gm) IIF_BAPPLY(1, true); else IIF_BAPPLY(1, false); }
    else { if (gm) IIF_BAPPLY(0, true); else IIF_BAPPLY(0, false); }
#undef IIF_BAPPLY
#undef IIF_BAPPLY2
#undef IIF_BAPPLY3
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BN-backward sums of the stem (bn1 -> relu -> 3x3/2 max pool, resnet_pytorch.py:284-287) from POOLED tensors.
// The gradient of the stem's BN output is the pooled gradient scattered to the arg-max positions, gated by the ReLU; its
// column sums do not care where an element lands:
//     sum g      = sum over pooled elements of gp [a x* + b > 0]
//     sum g xhat = sum over pooled elements of gp [a x* + b > 0] (x* - mean) invstd,      x* = the RAW stem output at the arg max
// so the reduction pass over the 4x larger scattered gradient AND the stem output (822 MB, 190 us at batch 256) becomes a pass
// over two pooled tensors (206 MB).  x* is stored by the fused forward pool (iif_maxpool_bn_forward, pool_x: +103 MB written
// per step) - round 4 recovered xhat from the pooled ACTIVATION instead, xhat = ((out - b) / a - mean) invstd, whose error is
// 2^-9 |xhat + beta / gamma|: noise for a channel with a small gain against its shift (round-4 advice).  With x* every term
// is the term of the standard pass (same mask test, same (x - mean) * invstd), only the order of the additions differs.
// One partial row [2][C] per block, fixed order: deterministic.  Grid <= 512 blocks (bn_backward_t's one-stage finalisation).
template <typename T>
__global__ void __launch_bounds__(256) pool_bwd_sums_kernel(const T* gp, const T* px, const float* stats, int64_t npix, int C,
                                                            int64_t pix_per_block, float* partial) {
    constexpr int V = VT<T>::V;
    __shared__ float sh[2][256][V + 1];
    const int cv = C / V;                                 // channel vectors per pixel (host: 256 % cv == 0)
    const int vec = threadIdx.x % cv, pl = threadIdx.x / cv, ppb = 256 / cv;
    const int c0 = vec * V;
    float aa[V], bb[V], mean[V], istd[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        aa[q] = stats[2 * C + c0 + q]; bb[q] = stats[3 * C + c0 + q]; mean[q] = stats[c0 + q]; istd[q] = stats[C + c0 + q];
    }
    float s1[V], s2[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    int64_t p1 = p0 + pix_per_block; if (p1 > npix) p1 = npix;
    for (int64_t p = p0 + pl; p < p1; p += 2 * ppb) {     // two pixels in flight per thread
        float g0[V], x0[V], g1[V], x1[V];
        const bool two = p + ppb < p1;
        const int64_t pb = two ? p + ppb : p;
        VT<T>::load_nt(gp + p * C + c0, g0); VT<T>::load_nt(px + p * C + c0, x0);
        VT<T>::load_nt(gp + pb * C + c0, g1); VT<T>::load_nt(px + pb * C + c0, x1);
#pragma unroll
        for (int q = 0; q < V; ++q) {                     // mask and xhat exactly as bn_bwd_reduce_kernel<T, 3> forms them
            const float ga = fmaf(aa[q], x0[q], bb[q]) > 0.f ? g0[q] : 0.f;
            s1[q] += ga; s2[q] += ga * ((x0[q] - mean[q]) * istd[q]);
            const float gb = (two && fmaf(aa[q], x1[q], bb[q]) > 0.f) ? g1[q] : 0.f;
            s1[q] += gb; s2[q] += gb * ((x1[q] - mean[q]) * istd[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) { sh[0][threadIdx.x][q] = s1[q]; sh[1][threadIdx.x][q] = s2[q]; }
    __syncthreads();
    if ((int)threadIdx.x < 2 * C) {                       // (C <= 128 here; one thread per (sum, channel), pixel lanes in order)
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        float a = 0.f;
        for (int j = 0; j < ppb; ++j) a += sh[which][j * cv + c / V][c % V];
        partial[(int64_t)blockIdx.x * 2 * C + which * C + c] = a;
    }
}

inline bool bad_align(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

// Stem, round 5: the max-pool backward gathered INSIDE the normalisation pass.  The scattered gradient (the stem's resolution,
// three quarters zeros: 411 MB written by iif_maxpool_backward and read again here) is never formed: a thread gathers its vector
// from the (at most four) pooled windows whose arg max points at its pixel (pool_gather.h), rounds it to the storage type as the
// stored tensor was, and goes on as bn_bwd_apply_kernel<T, 3>: 1.64 -> 0.98 GB, one launch less at the end of backward where little
// else runs.  One image row per block; 256 % (C / V) == 0, so a thread keeps one channel vector's coefficients.
// the four windows of pool321_gather (pool_gather.h) as one batch of loads: (row B, col B), (row B, col A), (row A, col B),
// (row A, col A) - clamped addresses, the predicates kept for the sum, which adds in that same order
template <typename T> struct PoolTaps {
    u32x4 g[4];
    unsigned code[4][2];
    int want[4];
    bool use[4];
};
template <typename T>
__device__ __forceinline__ void pool321_issue(const T* gy, const unsigned char* idx, int n, int h, int w, int c, int C, int Ho, int Wo,
                                              PoolTaps<T>& t) {
    constexpr int V = VT<T>::V;
    const int th = h + 1, tw = w + 1;
    const int hoA = th >> 1, woA = tw >> 1;
    const int rA = th & 1, sA = tw & 1;
    const bool vhA = hoA < Ho, vwA = woA < Wo;
    const bool vhB = rA == 0, vwB = sA == 0;
    const int hA = vhA ? hoA : Ho - 1, wA = vwA ? woA : Wo - 1;
    const int hB = vhB ? hoA - 1 : hA, wB = vwB ? woA - 1 : wA;
    const int64_t base = (int64_t)n * Ho;
    const int hh[4] = {hB, hB, hA, hA}, ww[4] = {wB, wA, wB, wA};
    t.want[0] = 2 * 3 + 2; t.want[1] = 2 * 3 + sA; t.want[2] = rA * 3 + 2; t.want[3] = rA * 3 + sA;
    t.use[0] = vhB && vwB; t.use[1] = vhB && vwA; t.use[2] = vhA && vwB; t.use[3] = vhA && vwA;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t o = ((base + hh[k]) * Wo + ww[k]) * C + c;
        t.g[k] = VT<T>::raw(gy + o);
        if constexpr (V == 8) {
            const uint2 cc = *reinterpret_cast<const uint2*>(idx + o);
            t.code[k][0] = cc.x; t.code[k][1] = cc.y;
        } else {
            t.code[k][0] = *reinterpret_cast<const unsigned int*>(idx + o); t.code[k][1] = 0;
        }
    }
}
template <typename T>
__device__ __forceinline__ void pool321_sum(const PoolTaps<T>& t, float (&acc)[VT<T>::V]) {
    constexpr int V = VT<T>::V;
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g[V];
        VT<T>::unpack(t.g[k], g);
#pragma unroll
        for (int q = 0; q < V; ++q)
            if (t.use[k] && (int)((t.code[k][q >> 2] >> (8 * (q & 3))) & 0xffu) == t.want[k]) acc[q] += g[q];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) pool_bn_bwd_apply_kernel(const T* gp, const unsigned char* idx, const T* x, const float* stats,
                                                                const float* coef, T* dx, int H, int W, int C, int Ho, int Wo,
                                                                int total_rows, int rows_per_block) {
    constexpr int V = VT<T>::V;
    const unsigned cv = (unsigned)C / V, rowv = (unsigned)W * cv;
    // Blocks go round the 8 XCDs; a pooled row is gathered by three image rows.  XCD k walks the k-th eighth of the row groups in
    // order, so neighbouring rows meet in one L2.
    const unsigned ngroups = gridDim.x, per = ngroups >> 3;
    const unsigned grp = (ngroups & 7u) == 0 ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned r0 = grp * (unsigned)rows_per_block;
    unsigned nr = (unsigned)rows_per_block;
    if (r0 + nr > (unsigned)total_rows) nr = (unsigned)total_rows - r0;
    const unsigned nvec = nr * rowv;                                     // this block's vectors: thread t takes t, t + 256, ...
    const int c0 = (int)(threadIdx.x % cv) * V;                          // (rowv % cv == 0 and 256 % cv == 0: one channel vector per thread)
    float k1[V], k2[V], k3[V], mu[V], aa[V], bb[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k]; k3[k] = coef[2 * C + c0 + k]; mu[k] = stats[c0 + k];
        aa[k] = stats[2 * C + c0 + k]; bb[k] = stats[3 * C + c0 + k];
    }
    const int64_t base = (int64_t)r0 * W * C;
    // one vector per step, the NEXT step's seven loads requested before this step's arithmetic: a block is one stream of
    // round trips instead of load / compute / store phases (3 blocks per CU at 136 VGPRs did not hide those: 297 us)
    u32x4 xr[2];
    PoolTaps<T> taps[2];
    auto issue = [&](unsigned t, int slot) {
        const unsigned tt = t < nvec ? t : threadIdx.x;
        const unsigned rr = tt / rowv, j = tt - rr * rowv;
        const unsigned row = r0 + rr, n = row / (unsigned)H, h = row - n * (unsigned)H;
        xr[slot] = VT<T>::raw_nt(x + base + (int64_t)tt * V);
        pool321_issue<T>(gp, idx, (int)n, (int)h, (int)(j / cv), c0, C, Ho, Wo, taps[slot]);
    };
    auto finish = [&](unsigned t, int slot) {
        float g[V], xv[V];
        pool321_sum<T>(taps[slot], g);
        VT<T>::unpack(xr[slot], xv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = fmaf(aa[k], xv[k], bb[k]) > 0.f ? VT<T>::round_trip(g[k]) : 0.f;
            xv[k] = k1[k] * (d - k2[k] - (xv[k] - mu[k]) * k3[k]);
        }
        VT<T>::template store_as<false>(dx + base + (int64_t)t * V, xv);
    };
    if (threadIdx.x >= nvec) return;
    issue(threadIdx.x, 0);
    for (unsigned t = threadIdx.x; t < nvec; t += 512) {
        issue(t + 256, 1);
        finish(t, 0);
        if (t + 256 < nvec) {
            issue(t + 512, 0);
            finish(t + 256, 1);
        }
    }
}

extern "C" {

int64_t iif_bn_workspace_bytes(int64_t m, int c) { return ((int64_t)1100 * 2 * c + 3 * (int64_t)c) * 4 + 256; }

int iif_bn_forward_stats(const void* x, int dtype, int64_t m, int c, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* stats, void* workspace,
                         int64_t workspace_bytes, void* stream) {
    if (!x || !gamma || !beta || !stats || !workspace || m <= 0 || c <= 0) return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(x)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) {
        if (c % 4) return IIF_EUNSUPPORTED;
        return bn_forward_t<float>((const float*)x, m, c, gamma, beta, eps, momentum, running_mean, running_var, stats,
                                   (float*)workspace, workspace_bytes, as_stream(stream));
    }
    if (dtype == IIF_BF16) {
        if (c % 8) return IIF_EUNSUPPORTED;
        return bn_forward_t<unsigned short>((const unsigned short*)x, m, c, gamma, beta, eps, momentum, running_mean,
                                            running_var, stats, (float*)workspace, workspace_bytes, as_stream(stream));
    }
    return IIF_EINVAL;
}

int iif_bn_finalize_stats(const float* partial, int n_partials, int64_t m, int c, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* stats,
                          float* scratch, int64_t scratch_floats, void* stream) {
    if (!partial || !gamma || !beta || !stats || n_partials <= 0 || m <= 0 || c <= 0) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (n_partials > two_stage_rows() && scratch && scratch_floats >= (int64_t)64 * 2 * c) {
        const int slices = stage1_slices(n_partials, c, scratch_floats), rps = (n_partials + slices - 1) / slices;
        hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((c + 31) / 32, slices), dim3(256), 0, st, partial, n_partials, c,
                           rps, scratch);
        IIF_LAUNCH_CHECK();
        return launch_bn_finalize(scratch, slices, c, (double)m, gamma, beta, eps, momentum, running_mean, running_var,
                                  stats, st);
    }
    return launch_bn_finalize(partial, n_partials, c, (double)m, gamma, beta, eps, momentum, running_mean, running_var,
                              stats, st);
}

int iif_bn_finalize_stats_fused(const float* partial, int n_partials, int64_t m, int c, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* stats,
                                float* scratch, int64_t scratch_floats, int32_t* tickets, void* stream) {
    if (!partial || !gamma || !beta || !stats || n_partials <= 0 || m <= 0 || c <= 0) return IIF_EINVAL;
    if (!tickets || n_partials <= two_stage_rows() || !scratch || scratch_floats < (int64_t)64 * 2 * c || (c + 31) / 32 > 64)
        return iif_bn_finalize_stats(partial, n_partials, m, c, gamma, beta, eps, momentum, running_mean, running_var, stats,
                                     scratch, scratch_floats, stream);
    const int slices = stage1_slices(n_partials, c, scratch_floats), rps = (n_partials + slices - 1) / slices;
    hipLaunchKernelGGL(bn_reduce_finalize_kernel<0>, dim3((c + 31) / 32, slices), dim3(256), 0, as_stream(stream), partial,
                       n_partials, c, rps, scratch, tickets, (double)m, gamma, beta, eps, momentum, stats, running_mean, running_var);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn_finalize_stats_sums(const float* partial, int n_partials, int64_t m, int c, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* stats,
                               float* scratch, int64_t scratch_floats, int32_t* tickets, const float* partial2, int n_partials2, int c2,
                               float* sums2, void* stream) {
    if (!partial2 || !sums2 || n_partials2 <= 0 || c2 <= 0) return IIF_EINVAL;
    if (!partial || !gamma || !beta || !stats || n_partials <= 0 || m <= 0 || c <= 0) return IIF_EINVAL;
    if (n_partials > two_stage_rows() || n_partials2 > two_stage_rows()) {          // (many rows: the two reductions as they were)
        const int rc = iif_bn_finalize_stats_fused(partial, n_partials, m, c, gamma, beta, eps, momentum, running_mean, running_var, stats,
                                                   scratch, scratch_floats, tickets, stream);
        return rc != IIF_OK ? rc : launch_column_sums(partial2, n_partials2, c2, sums2, as_stream(stream));
    }
    return launch_bn_finalize(partial, n_partials, c, (double)m, gamma, beta, eps, momentum, running_mean, running_var, stats,
                              as_stream(stream), partial2, n_partials2, c2, sums2);
}

int iif_bn_apply(const void* x, int dtype, int64_t m, int c, const float* stats, const void* residual,
                 const float* residual_stats, int relu, void* y, uint8_t* relu_bits, void* stream) {
    if (!x || !stats || !y || m <= 0 || c <= 0) return IIF_EINVAL;
    if (bad_align(x) || bad_align(y) || bad_align(residual)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) {
        if (c % 4) return IIF_EUNSUPPORTED;
        return bn_apply_t<float>((const float*)x, stats, (const float*)residual, residual_stats, (float*)y, m, c, relu,
                                 relu_bits, as_stream(stream));
    }
    if (dtype == IIF_BF16) {
        if (c % 8) return IIF_EUNSUPPORTED;
        return bn_apply_t<unsigned short>((const unsigned short*)x, stats, (const unsigned short*)residual,
                                          residual_stats, (unsigned short*)y, m, c, relu, relu_bits, as_stream(stream));
    }
    return IIF_EINVAL;
}

int iif_bn_backward(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                    const float* stats, const float* gamma, float* dgamma, float* dbeta, void* dx, void* gmasked,
                    void* workspace, int64_t workspace_bytes, void* stream) {
    if (!gy || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || m <= 0 || c <= 0) return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(dx) || bad_align(y_mask) || bad_align(gmasked))
        return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) {
        if (c % 4) return IIF_EUNSUPPORTED;
        return bn_backward_t<float>((const float*)gy, (const float*)y_mask, relu_bits, (const float*)x, stats, gamma, m, c, dgamma,
                                    dbeta, (float*)dx, (float*)gmasked, (float*)workspace, workspace_bytes,
                                    as_stream(stream));
    }
    if (dtype == IIF_BF16) {
        if (c % 8) return IIF_EUNSUPPORTED;
        return bn_backward_t<unsigned short>((const unsigned short*)gy, (const unsigned short*)y_mask, relu_bits,
                                             (const unsigned short*)x, stats, gamma, m, c, dgamma, dbeta,
                                             (unsigned short*)dx, (unsigned short*)gmasked, (float*)workspace,
                                             workspace_bytes, as_stream(stream));
    }
    return IIF_EINVAL;
}

int iif_bn_backward_relu_recompute(const void* gy, const void* x, int dtype, int64_t m, int c, const float* stats,
                                   const float* gamma, float* dgamma, float* dbeta, void* dx, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    if (!gy || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || m <= 0 || c <= 0) return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(dx)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) {
        if (c % 4) return IIF_EUNSUPPORTED;
        return bn_backward_t<float>((const float*)gy, nullptr, nullptr, (const float*)x, stats, gamma, m, c, dgamma, dbeta, (float*)dx,
                                    nullptr, (float*)workspace, workspace_bytes, as_stream(stream), nullptr, 0, true);
    }
    if (dtype == IIF_BF16) {
        if (c % 8) return IIF_EUNSUPPORTED;
        return bn_backward_t<unsigned short>((const unsigned short*)gy, nullptr, nullptr, (const unsigned short*)x, stats, gamma, m, c,
                                             dgamma, dbeta, (unsigned short*)dx, nullptr, (float*)workspace, workspace_bytes,
                                             as_stream(stream), nullptr, 0, true);
    }
    return IIF_EINVAL;
}

int iif_bn_backward_relu_recompute_pooled(const void* gy, const void* x, int dtype, int64_t m, int c, const float* stats,
                                          const float* gamma, float* dgamma, float* dbeta, void* dx, void* workspace,
                                          int64_t workspace_bytes, const void* g_pool, const void* pool_x, int64_t pool_pixels,
                                          void* stream) {
    if (!gy || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || !g_pool || !pool_x || m <= 0 || c <= 0 ||
        pool_pixels <= 0)
        return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(dx) || bad_align(g_pool) || bad_align(pool_x)) return IIF_EUNSUPPORTED;
    const int v = dtype == IIF_F32 ? 4 : 8;
    if ((dtype != IIF_F32 && dtype != IIF_BF16) || c % v || 256 % (c / v) || 2 * c > 256) return IIF_EUNSUPPORTED;
    int nblk = (int)(pool_pixels / 64 < 512 ? (pool_pixels + 63) / 64 : 512);
    const int64_t ppb = (pool_pixels + nblk - 1) / nblk;
    nblk = (int)((pool_pixels + ppb - 1) / ppb);
    // the partial rows take the END of the workspace; bn_backward_t gets (and checks) what is in front of them
    const int64_t row_bytes = (int64_t)nblk * 2 * c * 4;
    if (row_bytes + 4096 > workspace_bytes) return IIF_EINVAL;
    float* ws = (float*)workspace;
    float* rows = ws + (workspace_bytes - row_bytes) / 16 * 4;
    workspace_bytes = (workspace_bytes - row_bytes) / 16 * 16;
    hipStream_t st = as_stream(stream);
    if (dtype == IIF_F32) {
        hipLaunchKernelGGL(pool_bwd_sums_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)g_pool, (const float*)pool_x, stats,
                           pool_pixels, c, ppb, rows);
        IIF_LAUNCH_CHECK();
        return bn_backward_t<float>((const float*)gy, nullptr, nullptr, (const float*)x, stats, gamma, m, c, dgamma, dbeta, (float*)dx, nullptr,
                                    ws, workspace_bytes, st, rows, nblk, true);
    }
    hipLaunchKernelGGL(pool_bwd_sums_kernel<unsigned short>, dim3(nblk), dim3(256), 0, st, (const unsigned short*)g_pool,
                       (const unsigned short*)pool_x, stats, pool_pixels, c, ppb, rows);
    IIF_LAUNCH_CHECK();
    return bn_backward_t<unsigned short>((const unsigned short*)gy, nullptr, nullptr, (const unsigned short*)x, stats, gamma, m, c, dgamma,
                                         dbeta, (unsigned short*)dx, nullptr, ws, workspace_bytes, st, rows, nblk, true);
}

int iif_bn_backward_pool_fused(const void* g_pool, const uint8_t* argmax, const void* pool_x, const void* x, int dtype, int n, int h,
                               int w, int c, int ho, int wo, const float* stats, const float* gamma, float* dgamma, float* dbeta,
                               void* dx, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!g_pool || !argmax || !pool_x || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || n <= 0 || h <= 0 ||
        w <= 0 || c <= 0)
        return IIF_EINVAL;
    if (ho != (h + 2 - 3) / 2 + 1 || wo != (w + 2 - 3) / 2 + 1) return IIF_EINVAL;            // 3x3 / stride 2 / pad 1 only
    const int64_t m = (int64_t)n * h * w, pool_pixels = (int64_t)n * ho * wo;
    if (m > 0x7fffff00LL || bad_align(g_pool) || bad_align(pool_x) || bad_align(x) || bad_align(dx)) return IIF_EUNSUPPORTED;
    const int v = dtype == IIF_F32 ? 4 : 8;
    if ((dtype != IIF_F32 && dtype != IIF_BF16) || c % v || 256 % (c / v) || 2 * c > 256) return IIF_EUNSUPPORTED;
    int nblk = (int)(pool_pixels / 64 < 512 ? (pool_pixels + 63) / 64 : 512);
    const int64_t ppb = (pool_pixels + nblk - 1) / nblk;
    nblk = (int)((pool_pixels + ppb - 1) / ppb);
    const int64_t need = ((int64_t)nblk * 2 * c + 3 * c) * 4;
    if (need > workspace_bytes) return IIF_EINVAL;
    float* rows = (float*)workspace;
    float* coef = rows + (int64_t)nblk * 2 * c;
    hipStream_t st = as_stream(stream);
    const dim3 blk(256);
    // the sums exactly as iif_bn_backward_relu_recompute_pooled forms them (same kernel, same rows, same finalisation)
    if (dtype == IIF_F32)
        hipLaunchKernelGGL(pool_bwd_sums_kernel<float>, dim3(nblk), blk, 0, st, (const float*)g_pool, (const float*)pool_x, stats, pool_pixels, c,
                           ppb, rows);
    else
        hipLaunchKernelGGL(pool_bwd_sums_kernel<unsigned short>, dim3(nblk), blk, 0, st, (const unsigned short*)g_pool,
                           (const unsigned short*)pool_x, stats, pool_pixels, c, ppb, rows);
    IIF_LAUNCH_CHECK();
    const int cb = finalize_cb(nblk);
    const dim3 fgrid((c + cb - 1) / cb);
    if (cb == 4) hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, fgrid, blk, 0, st, rows, nblk, c, (double)m, gamma, stats, dgamma, dbeta, coef);
    else if (cb == 8) hipLaunchKernelGGL(bn_bwd_finalize_kernel<8>, fgrid, blk, 0, st, rows, nblk, c, (double)m, gamma, stats, dgamma, dbeta, coef);
    else hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, fgrid, blk, 0, st, rows, nblk, c, (double)m, gamma, stats, dgamma, dbeta, coef);
    IIF_LAUNCH_CHECK();
    const int total_rows = n * h;
    int rpb = 8;                     // rows per block: 28 pipelined steps per thread at 112 x 64 channels (2: 247 us, 4: 238, 8-28: 230)
    if ((int64_t)rpb * w * (c / v) < 2048) rpb = (int)((2048 + (int64_t)w * (c / v) - 1) / ((int64_t)w * (c / v)));
    const dim3 agrid((unsigned)((total_rows + rpb - 1) / rpb));
    if (dtype == IIF_F32)
        hipLaunchKernelGGL(pool_bn_bwd_apply_kernel<float>, agrid, blk, 0, st, (const float*)g_pool, argmax, (const float*)x, stats, coef,
                           (float*)dx, h, w, c, ho, wo, total_rows, rpb);
    else
        hipLaunchKernelGGL(pool_bn_bwd_apply_kernel<unsigned short>, agrid, blk, 0, st, (const unsigned short*)g_pool, argmax,
                           (const unsigned short*)x, stats, coef, (unsigned short*)dx, h, w, c, ho, wo, total_rows, rpb);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn_backward_partials_fused(const void* gy, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                                   const float* stats, const float* gamma, const float* partial, int n_partials, float* dgamma,
                                   float* dbeta, void* dx, void* workspace, int64_t workspace_bytes, int32_t* tickets,
                                   void* stream) {
    if (!gy || !x || !stats || !gamma || !partial || n_partials <= 0 || !dgamma || !dbeta || !dx || !workspace || m <= 0 || c <= 0)
        return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(dx)) return IIF_EUNSUPPORTED;
    if (dtype != IIF_BF16 || c % 8) return IIF_EUNSUPPORTED;
    return bn_backward_t<unsigned short>((const unsigned short*)gy, nullptr, relu_bits, (const unsigned short*)x, stats, gamma, m, c,
                                         dgamma, dbeta, (unsigned short*)dx, nullptr, (float*)workspace, workspace_bytes,
                                         as_stream(stream), partial, n_partials, false, (c + 31) / 32 <= 64 ? tickets : nullptr);
}

int iif_bn_partial_sums(const float* partial, int n_partials, int c, float* sums, void* stream) {
    if (!partial || !sums || n_partials <= 0 || c <= 0) return IIF_EINVAL;
    return launch_column_sums(partial, n_partials, c, sums, as_stream(stream));
}

int iif_bn_stats_sums(const void* x, int dtype, int64_t m, int c, float* sums, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !sums || !workspace || m <= 0 || c <= 0) return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(x)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) return c % 4 ? IIF_EUNSUPPORTED : bn_stats_sums_t<float>((const float*)x, m, c, sums, (float*)workspace, workspace_bytes, as_stream(stream));
    if (dtype == IIF_BF16) return c % 8 ? IIF_EUNSUPPORTED : bn_stats_sums_t<unsigned short>((const unsigned short*)x, m, c, sums, (float*)workspace, workspace_bytes, as_stream(stream));
    return IIF_EINVAL;
}

int iif_bn_backward_sums(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                         const float* stats, float* sums, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!gy || !x || !stats || !sums || !workspace || m <= 0 || c <= 0) return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(y_mask)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32) return c % 4 ? IIF_EUNSUPPORTED : bn_backward_sums_t<float>((const float*)gy, (const float*)y_mask, relu_bits, (const float*)x, stats, m, c, sums, (float*)workspace, workspace_bytes, as_stream(stream));
    if (dtype == IIF_BF16) return c % 8 ? IIF_EUNSUPPORTED : bn_backward_sums_t<unsigned short>((const unsigned short*)gy, (const unsigned short*)y_mask, relu_bits, (const unsigned short*)x, stats, m, c, sums, (float*)workspace, workspace_bytes, as_stream(stream));
    return IIF_EINVAL;
}

int iif_bn_backward_apply_sums(const void* gy, const void* y_mask, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                               const float* stats, const float* gamma, const float* local_sums, const float* total_sums,
                               double total_count, float* dgamma, float* dbeta, void* dx, void* gmasked, float* coef_scratch,
                               void* stream) {
    if (!gy || !x || !stats || !gamma || !local_sums || !total_sums || !dgamma || !dbeta || !dx || !coef_scratch || m <= 0 || c <= 0 ||
        total_count <= 0.0)
        return IIF_EINVAL;
    if (m > 0x7fffff00LL || bad_align(gy) || bad_align(x) || bad_align(dx) || bad_align(y_mask) || bad_align(gmasked)) return IIF_EUNSUPPORTED;
    if (dtype == IIF_F32)
        return c % 4 ? IIF_EUNSUPPORTED : bn_backward_apply_sums_t<float>((const float*)gy, (const float*)y_mask, relu_bits, (const float*)x, stats, gamma, local_sums, total_sums, total_count, m, c, dgamma, dbeta, (float*)dx, (float*)gmasked, coef_scratch, as_stream(stream));
    if (dtype == IIF_BF16)
        return c % 8 ? IIF_EUNSUPPORTED : bn_backward_apply_sums_t<unsigned short>((const unsigned short*)gy, (const unsigned short*)y_mask, relu_bits, (const unsigned short*)x, stats, gamma, local_sums, total_sums, total_count, m, c, dgamma, dbeta, (unsigned short*)dx, (unsigned short*)gmasked, coef_scratch, as_stream(stream));
    return IIF_EINVAL;
}

int iif_bn_backward_partials(const void* gy, const uint8_t* relu_bits, const void* x, int dtype, int64_t m, int c,
                             const float* stats, const float* gamma, const float* partial, int n_partials, float* dgamma,
                             float* dbeta, void* dx, void* workspace, int64_t workspace_bytes, void* stream) {
    return iif_bn_backward_partials_fused(gy, relu_bits, x, dtype, m, c, stats, gamma, partial, n_partials, dgamma, dbeta, dx,
                                          workspace, workspace_bytes, nullptr, stream);
}

}  // extern "C"
