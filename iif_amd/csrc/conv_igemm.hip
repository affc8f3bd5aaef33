// Implicit-GEMM convolution on the gfx950 matrix cores (forward and data-gradient).
//
//   dst[m, n] = sum_{tap, c} gather(src)[m, tap, c] * wgt[n, tap, c]   (+ bias[n]) (+ res[m, n])
//
// m = (image, y, x) of the DESTINATION grid, n = destination channel.  Activations
// are NHWC, weights are [n][R][S][c] so both operands are K-contiguous: a 16-byte
// piece (8 bf16 / 4 f32 channels of one tap) is the unit of every global load,
// LDS write and MFMA fragment read.  `transposed` switches the gather between the
// forward correlation (ys = y*stride - pad + r) and the data-gradient form
// (ys = (y + pad - r)/stride when divisible), so the same kernel computes
// conv2d forward, conv2d dgrad (with [cin][R][S][cout] weights), 1x1 GEMMs and
// the fully-connected layer.  Replaces the cuDNN/MIOpen calls under
// classification/resnet_pytorch.py:46-62,149-169 and resnet_cifar.py:133-138.
//
// Tiling (one 256-thread workgroup = 4 wavefronts of 64):
//   block tile 128 pixels x BN channels (BN = 128 or 64), K step = 64 bytes of
//   channels (32 bf16 / 16 f32); wave tile 64 pixels x BN/2 channels as 16x16 MFMA
//   tiles: v_mfma_f32_16x16x32_bf16 (bf16) or 4 x v_mfma_f32_16x16x4_f32 (exact
//   fp32 parity mode, k-ordered fmaf chain).  The MFMA "row" operand is the weight
//   tile, so each lane ends with 4 consecutive output channels of one pixel
//   (8-byte bf16 / 16-byte f32 stores).
//   LDS: 2 x (128 + BN) x 64 B, double buffered, register staged (global loads of
//   step k+1 are in flight while step k is multiplied).  Rows are 64 B, so the
//   four 16-byte chunks of a row are XOR-swizzled by f(row>>2) = {0,2,3,1} which
//   makes every ds_read_b128 lane group hit 16 distinct 16-byte bank slots.
//   blockIdx -> tile map is XCD-aware: the n-tiles of one pixel tile run
//   back-to-back on the same XCD so the activation tile is re-read from its L2.
#include <stdlib.h>

#include "common.h"
#ifndef IIF_CONV_AUX_SRC
#define IIF_CONV_AUX_SRC 0
#endif

namespace {

template <typename T> struct ET;
template <> struct ET<unsigned short> { static constexpr int PE = 8, KE = 32; };   // bf16 bits
template <> struct ET<float> { static constexpr int PE = 4, KE = 16; };

struct ConvArgs {
    const unsigned char* src; const unsigned char* wgt; unsigned char* dst; const unsigned char* res;
    const unsigned char* wfrag;      // nullable: the same weights as MFMA fragments (iif_conv_pack_fragments), 3x3 generation-2 kernel
    int wfrag_kind;                  // 0: [channel tile][tap][32-channel chunk] fragments; 1: the grouped 16-channel format (iif_conv_pack_fragments_g16)
    const float* bias;
    const unsigned char* res_bits;   // nullable: 1 bit per residual element (ReLU decisions); the residual is masked by it
    // nullable: the batch-norm BACKWARD partial sums of the unit whose output gradient this launch writes
    // (dst = dL/dy of that unit): bw_x its pre-normalisation output, bw_bits its ReLU decisions (nullable = no
    // ReLU), bw_stats its (mean at [c], invstd at [Cd + c]).  bn_partial then receives per tile
    // (sum g, sum g*xhat) with g = dst * [bit] instead of the forward (sum, sum of squares).
    const unsigned char* bw_x; const unsigned char* bw_bits; const float* bw_stats;
    // mask_store (round 3, the algebraic BN backward of the expanding 1x1 layers): dst is stored ALREADY gated by bw_bits
    // (every later reader gates it by those bits anyway) and bn_partial receives (sum dst, 0) per tile; bw_x is not read.
    int mask_store;
    // second source of a 1x1 launch: K continues over src2's Cs2 channels (weights rows hold Cs + Cs2 columns):
    // dst = [src | src2] * wgt^T.  sbias (nullable): per-output-channel fp32 added in the staged epilogue.
    const unsigned char* src2; int Cs2;
    const float* sbias;
    // the two-pass forward of a conv + BN unit whose raw output is never stored (round 3):
    //   no_store: the tile is rounded to bf16 and summed into bn_partial exactly as if it were stored, and dropped;
    //   aff (stats base: a at [2 Cd + c], b at [3 Cd + c]): dst = relu(fmaf(a, bf16(conv), b) + res) with the arithmetic of
    //   bn_apply_kernel, relu_out one byte of ReLU decisions per 16-byte vector (as bn_apply writes them).
    //   no_store == 2 / aff2 (round 6, register-weight kernel only): the sums come from the ACCUMULATORS (unrounded, no staging);
    //   the residual is normalised by its own BN, fmaf(a2, res, b2), aff2 laid out like aff (convolutional shortcut).
    //   rx_* (round 6, register-weight kernel only): the upstream x of the backward sums is recomputed per tile from the upstream
    //   block's a2 (rx_src2, [M, rx_k2]) and conv3 weights (rx_w3, [Cd, rx_ldw3]) instead of being read from bw_x.
    const unsigned char* rx_src2; const unsigned char* rx_w3; int rx_k2, rx_ldw3;
    //   pg_* (round 6, with rx_*): P = dst^T a2 and Gram = a2^T a2 as by-products, one fp32 slab per tile sequence (iif_regw_epilogue)
    float* pg_slab; long long pg_cap; int pg_ld; int* pg_count;
    //   pro_* (round 6, register-weight kernel only): src is the previous convolution's RAW output; its BN + ReLU (pro_stats) is applied
    //   per tile in LDS, the activation (pro_out, pro_bits; pro_csum nullable: column-sum rows) is written as a by-product.
    const float* pro_stats; unsigned char* pro_out; unsigned char* pro_bits; float* pro_csum;
    int no_store;
    const float* aff;
    const float* aff2;
    unsigned char* relu_out;
    int bn_row0;           // first partial row of this launch (launches that share one partial buffer)
    int* rows_out;         // host only: receives bn_row0 + tiles of the launch (the partial rows written so far)
    long long bn_cap;      // host only: floats available behind bn_partial
    float* bn_partial;     // nullable: [mtiles][2][Cd] per-tile (sum, sum of squares) of the stored output
    int N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, sshift, pad, transposed, ldw, M, K, mtiles, ntiles;
    int spitch, dpitch, groups;   // channels per pixel of the source / destination TENSORS (= groups * Cs / Cd)
    // uniform-tap path: explicit tap list (source displacement in pixels of the source grid, weight tap index);
    // source pixel of output (y, x) and tap t is ((y << in_shift) + tap_dy[t], (x << in_shift) + tap_dx[t])
    int ntaps, in_shift;
    signed char tap_dy[16], tap_dx[16];
    unsigned char tap_w[16];
    // destination scatter: GEMM row (n, y, x) of the (Hd, Wd) grid is written to pixel
    // (n, (y << ds_shift) + doy, (x << ds_shift) + dox) of a [N, Hfull, Wfull] tensor (stride-2 data gradients
    // run as one dense sub-convolution per output parity class)
    int scatter, ds_shift, doy, dox, Hfull, Wfull;
};

__device__ __forceinline__ int64_t dst_row(const ConvArgs& a, int m) {
    if (!a.scatter) return m;
    const int hw = a.Hd * a.Wd;
    const int n = m / hw, rem = m - n * hw;
    const int y = rem / a.Wd, x = rem - y * a.Wd;
    return ((int64_t)n * a.Hfull + ((y << a.ds_shift) + a.doy)) * a.Wfull + ((x << a.ds_shift) + a.dox);
}

// XOR key of the 16-byte chunk index of a 64-byte LDS row.  ds_read_b128 serves lanes in the groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with lane = chunk*16 + row that is 16 distinct rows per group, rows
// 0-3 and 12-15 reading chunk a and rows 4-11 chunk a^1.  Flipping chunk bit 1 on every other group of 4 rows keeps
// the 16 slots of a group distinct for ANY first row (the halo kernel reads fragments at tap-displaced rows;
// the former key {0,2,3,1}[(row>>2)&3] was conflict-free for 16-aligned fragments only: 45 % conflict cycles there).
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }

#ifdef IIF_CONV_STAMPS
// Diagnostic build only (make stamps): per-wave cycle sums of the K-step phases of the LDS-DMA kernel, written to
// a buffer of their own (cdna_hip_programming.md §7, in-kernel stamps).  Read the SHARES, not the run time.
__device__ unsigned long long* g_stamps = nullptr;
#define IIF_STAMP(var)                                                                             \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#endif

// epilogue: lane holds channels n0 + wn*BN/2 + ci*16 + fc*4 + {0..3} of pixel m0 + wm*64 + pj*16 + fr
template <typename T, int BN, bool OUTF32>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[BN / 32][4], int m0, int n0, int wm, int wn,
                                              int fr, int fc, int goff = 0) {
    constexpr int CI = BN / 32;
    using OT = typename std::conditional<OUTF32, float, T>::type;
    const bool vec_ok = ((a.Cd | a.dpitch) & 3) == 0;
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int m = m0 + wm * 64 + pj * 16 + fr;
        if (m >= a.M) continue;
        const int64_t drow = dst_row(a, m);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const int n = n0 + wn * (BN / 2) + ci * 16 + fc * 4;
            if (n >= a.Cd) continue;
            float v[4] = {acc[ci][pj].x, acc[ci][pj].y, acc[ci][pj].z, acc[ci][pj].w};
            const int64_t o = drow * a.dpitch + goff + n;
            const int cnt = a.Cd - n < 4 ? a.Cd - n : 4;
            if (a.bias)
                for (int q = 0; q < cnt; ++q) v[q] += a.bias[n + q];
            OT* dp = reinterpret_cast<OT*>(a.dst) + o;
            const OT* rp = reinterpret_cast<const OT*>(a.res) + o;
            if (vec_ok) {
                // ReLU-decision bits of the residual's 16-byte vector (4 fp32 / 8 bf16 elements per byte)
                constexpr int RV = sizeof(OT) == 4 ? 4 : 8;
                const unsigned rb = a.res_bits ? (unsigned)a.res_bits[o / RV] >> (o % RV) : 0xffu;
                if constexpr (sizeof(OT) == 4) {
                    if (a.res) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(rp);
                        v[0] += (rb & 1u) ? t.x : 0.f; v[1] += (rb & 2u) ? t.y : 0.f;
                        v[2] += (rb & 4u) ? t.z : 0.f; v[3] += (rb & 8u) ? t.w : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(dp) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    if (a.res) {
                        const u32x2 t = *reinterpret_cast<const u32x2*>(rp);
                        v[0] += (rb & 1u) ? bf16_bits_to_f32(t.x & 0xffffu) : 0.f; v[1] += (rb & 2u) ? __uint_as_float(t.x & 0xffff0000u) : 0.f;
                        v[2] += (rb & 4u) ? bf16_bits_to_f32(t.y & 0xffffu) : 0.f; v[3] += (rb & 8u) ? __uint_as_float(t.y & 0xffff0000u) : 0.f;
                    }
                    u32x2 w; w.x = pack_bf16x2(v[0], v[1]); w.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(dp) = w;
                }
            } else {
                for (int q = 0; q < cnt; ++q) {
                    if constexpr (sizeof(OT) == 4) {
                        float t = v[q]; if (a.res) t += rp[q];
                        dp[q] = t;
                    } else {
                        float t = v[q]; if (a.res) t += bf16_bits_to_f32(rp[q]);
                        dp[q] = f32_to_bf16_bits(t);
                    }
                }
            }
        }
    }
}

template <typename T, int BN, bool OUTF32>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvArgs a) {
    constexpr int BM = 128;
    constexpr int PE = ET<T>::PE, KE = ET<T>::KE;
    constexpr int NB = BN / 64;         // weight pieces per thread
    constexpr int CI = BN / 32;         // channel 16-tiles per wave
    constexpr int BUF = (BM + BN) * 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int mt = (j / a.ntiles) * 8 + xcd, nt = j % a.ntiles;
    if (mt >= a.mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging geometry: rows (tid>>2) + 64*i, chunk tid&3
    const int chunk = tid & 3, row0 = tid >> 2;
    int by[2], bx[2], ib[2];
    bool mv[2];
    const int HW = a.Hd * a.Wd;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + row0 + 64 * i;
        mv[i] = m < a.M;
        const int mm = mv[i] ? m : 0;
        const int n = mm / HW, rem = mm - n * HW;
        const int y = rem / a.Wd, x = rem - y * a.Wd;
        by[i] = a.transposed ? y + a.pad : (y << a.sshift) - a.pad;
        bx[i] = a.transposed ? x + a.pad : (x << a.sshift) - a.pad;
        ib[i] = n * a.Hs * a.Ws;
    }
    int e = chunk * PE;                 // K index of this thread's piece
    int tap = e / a.Cs;
    int c = e - tap * a.Cs;
    int r = tap / a.S;
    int s = tap - r * a.S;

    u32x4 ra[2], rb[NB];
    auto load_step = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ys, xs;
            bool ok = mv[i] && r < a.R;
            if (a.transposed) {
                const int ty = by[i] - r, tx = bx[i] - s;
                ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & a.sshift) == 0;
                ys = ty >> a.sshift; xs = tx >> a.sshift;
                ok = ok && ys < a.Hs && xs < a.Ws;
            } else {
                ys = by[i] + r; xs = bx[i] + s;
                ok = ok && (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
            }
            ra[i] = u32x4{0u, 0u, 0u, 0u};
            if (ok) {
                const int64_t off = ((int64_t)(ib[i] + ys * a.Ws + xs) * a.Cs + c) * (int64_t)sizeof(T);
                ra[i] = *reinterpret_cast<const u32x4*>(a.src + off);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + row0 + 64 * i;
            rb[i] = u32x4{0u, 0u, 0u, 0u};
            if (n < a.Cd && e < a.K) {
                const int64_t off = ((int64_t)n * a.ldw + e) * (int64_t)sizeof(T);
                rb[i] = *reinterpret_cast<const u32x4*>(a.wgt + off);
            }
        }
    };
    auto advance = [&]() {
        e += KE;
        c += KE;
        while (c >= a.Cs) { c -= a.Cs; if (++s == a.S) { s = 0; ++r; } }
    };
    auto store_step = [&](int buf) {
        unsigned char* A = smem + buf * BUF;
        unsigned char* B = A + BM * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + 64 * i;
            *reinterpret_cast<u32x4*>(A + row * 64 + ((chunk ^ swz(row)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = row0 + 64 * i;
            *reinterpret_cast<u32x4*>(B + row * 64 + ((chunk ^ swz(row)) << 4)) = rb[i];
        }
    };

    f32x4 acc[CI][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) acc[ci][pj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + KE - 1) / KE;
    const int fr = lane & 15, fc = lane >> 4;     // fragment row within a 16-tile, 16-byte chunk
    load_step();
    store_step(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const bool more = k + 1 < nk;
        if (more) { advance(); load_step(); }
        const unsigned char* A = smem + (k & 1) * BUF;
        const unsigned char* B = A + BM * 64;
        u32x4 wf[CI], xf[4];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const int row = wn * (BN / 2) + ci * 16 + fr;
            wf[ci] = *reinterpret_cast<const u32x4*>(B + row * 64 + ((fc ^ swz(row)) << 4));
        }
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int row = wm * 64 + pj * 16 + fr;
            xf[pj] = *reinterpret_cast<const u32x4*>(A + row * 64 + ((fc ^ swz(row)) << 4));
        }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                if constexpr (sizeof(T) == 2) {
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wf[ci]), __builtin_bit_cast(bf16x8, xf[pj]), acc[ci][pj], 0, 0, 0);
                } else {
                    const f32x4 wv = __builtin_bit_cast(f32x4, wf[ci]), xv = __builtin_bit_cast(f32x4, xf[pj]);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[ci][pj], 0, 0, 0);
                }
            }
        if (more) store_step((k + 1) & 1);
        __syncthreads();
    }

    conv_epilogue<T, BN, OUTF32>(a, acc, m0, n0, wm, wn, fr, fc);
}

// ------------------------------------------------------------------------------------------
// Pipelined variant: operands go HBM/L2 -> LDS by buffer_load ... lds (LDS-DMA, no staging
// registers), three LDS stages, loads of K step k+2 in flight while step k is multiplied, ONE
// raw s_barrier per step behind a counted s_waitcnt vmcnt (never 0 inside the loop).
//   * the DMA destination is lane-linear (wave base + lane*16), so the XOR swizzle is applied to
//     the per-lane SOURCE chunk: lane l of a 16-row piece fetches chunk (l&3)^f(l>>4) of row l>>2;
//     the ds_read side is unchanged;
//   * out-of-image taps, rows >= M, channels >= Cd and the ragged K tail are addressed out of the
//     buffer range: the hardware range check returns zeros, no predicated loads.
// Same tile shape, fragment maps and epilogue as the register-staged kernel above.
// LDS-staged epilogue for bf16 outputs: the wave tiles are parked in LDS as [pixel][channel] rows and
// written back with 16 B per lane, i.e. whole 128-B lines (the direct form writes 8 B per lane in 32-B
// runs).  The residual is read the same way.  With bn_partial the per-channel (sum, sum of squares) of the
// bf16-rounded tile are emitted too, so batch-norm statistics need no extra pass over the activation.
template <int BN, int BM, int NT>
__device__ __forceinline__ void staged_drain(const ConvArgs& a, unsigned char* smem, int m0, int n0, int mt, int goff);

template <int BN, int BM = 128>
__device__ __forceinline__ void conv_epilogue_staged(const ConvArgs& a, f32x4 (&acc)[BN / 32][4], unsigned char* smem,
                                                     int m0, int n0, int mt, int wm, int wn, int fr, int fc, int goff) {
    constexpr int CI = BN / 32;
    constexpr int NT = BM * 2;                  // threads of the block (4 or 8 waves)
    constexpr int PITCH = BN * 2 + 16;
    // The fp32 per-channel bias (the D W term of the BN3 algebra's data gradient) joins the ACCUMULATOR, before the one rounding
    // to bf16.  Round 3/4 added it to the staged bf16 tile and rounded again: a constant added to values on the bf16 grid loses
    // the SAME fraction of an ulp on every pixel of equal exponent - 1.3 instead of 0.35 on a column sum of 392 elements that is
    // zero in exact arithmetic (scripts/dbg_colsum.py), and d beta of the BN below is exactly that column sum.
    f32x4 sb4[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const int n = n0 + wn * (BN / 2) + ci * 16 + fc * 4;
        sb4[ci] = (a.sbias && n < a.Cd) ? *reinterpret_cast<const f32x4*>(a.sbias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                            // every wave is done reading the last stage
#pragma unroll
    for (int pj = 0; pj < 4; ++pj)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
            const int row = wm * 64 + pj * 16 + fr, ch = wn * (BN / 2) + ci * 16 + fc * 4;
            const f32x4 v = acc[ci][pj] + sb4[ci];
            u32x2 w;
            w.x = pack_bf16x2(v.x, v.y);
            w.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<u32x2*>(smem + row * PITCH + ch * 2) = w;
        }
    __syncthreads();
    staged_drain<BN, BM, NT>(a, smem, m0, n0, mt, goff);
}

// The [BM x BN] bf16 tile parked in LDS (row pitch BN * 2 + 16) goes out with 16 bytes per lane; residual, ReLU-bit masks,
// forward BN statistics / upstream BN-backward sums ride on the store loop.  Independent of the wave tiling that staged it.
template <int BN, int BM, int NT>
__device__ __forceinline__ void staged_drain(const ConvArgs& a, unsigned char* smem, int m0, int n0, int mt, int goff) {
    constexpr int PITCH = BN * 2 + 16;
    constexpr int CPR = BN / 8;                 // 16-byte chunks per row
    constexpr int RPP = NT / CPR;               // rows per pass
    const int tid = threadIdx.x;
    const int chunk = tid % CPR, r0 = tid / CPR;
    const int n = n0 + chunk * 8;
    // batch-norm partial sums of the STORED (bf16-rounded) values ride along with the store loop: every thread
    // already holds 8 channels of each row it writes
    float bs[8], bq[8], bmean[8], bistd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs[q] = 0.f; bq[q] = 0.f; bmean[q] = 0.f; bistd[q] = 0.f; }
    if (a.bw_x && n < a.Cd) {                       // (grouped: channel goff + n of the dpitch-wide upstream tensor)
#pragma unroll
        for (int q = 0; q < 8; ++q) { bmean[q] = a.bw_stats[goff + n + q]; bistd[q] = a.bw_stats[a.dpitch + goff + n + q]; }
    }
    if (a.aff && n < a.Cd) {                        // never together with bw_x: the same registers carry (a, b)
#pragma unroll
        for (int q = 0; q < 8; ++q) { bmean[q] = a.aff[2 * a.Cd + n + q]; bistd[q] = a.aff[3 * a.Cd + n + q]; }
    }
    if (n < a.Cd) {
        // Everything a batch of RBATCH rows needs from memory (residual, its ReLU bits, the upstream bits, the upstream x) is
        // requested before the first row is touched, UNCONDITIONALLY: an operand the launch does not have is read from one
        // fixed dummy address (the head of the weights: an L1 hit).  Round 3 loaded each operand under its own `if (a.res)` /
        // `if (a.bw_x)`: hipcc branches around such a load and waits for it on its own, so a thread went through its 8 rows in
        // 24-32 dependent memory round trips (20-25 us per tile under load; the data gradients with fused epilogues took
        // 12 tile rounds x 25 us at 56 x 56).  Now: one round trip per batch.
        constexpr int RT = BM / RPP;                    // rows per thread (8; 4 for the 64-channel tile)
        constexpr int RBATCH = RT < 2 ? RT : 2;        // (4 rows per batch: 133 registers in the 256-row kernel, one block per CU instead of two)
        static_assert(RT % RBATCH == 0, "row batches");
        const unsigned char* const dummy = a.wgt;
        const bool has_res = a.res != nullptr, has_rb = a.res_bits != nullptr, has_bx = a.bw_x != nullptr;
        const bool has_bb = a.bw_bits != nullptr;
        const bool res_nt = a.aff == nullptr;           // the fused block-output epilogue re-reads its residual soon: keep it cached
#pragma unroll 1
        for (int b0 = 0; b0 < RT; b0 += RBATCH) {
            int64_t ob[RBATCH];
            bool live[RBATCH];
            u32x4 rrv[RBATCH], xvv[RBATCH];
            unsigned rbv[RBATCH], mbv[RBATCH];
#pragma unroll
            for (int i = 0; i < RBATCH; ++i) {
                const int m = m0 + r0 + (b0 + i) * RPP;
                live[i] = m < a.M;
                ob[i] = (dst_row(a, live[i] ? m : a.M - 1) * a.dpitch + goff + n) * 2;
            }
            if (res_nt) {
#pragma unroll
                for (int i = 0; i < RBATCH; ++i)
#ifndef IIF_NO_NT_EPILOGUE_LOADS   // the residual's last use
                    rrv[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(has_res ? a.res + ob[i] : dummy));
#else
                    rrv[i] = *reinterpret_cast<const u32x4*>(has_res ? a.res + ob[i] : dummy);
#endif
            } else {
#pragma unroll
                for (int i = 0; i < RBATCH; ++i) rrv[i] = *reinterpret_cast<const u32x4*>(has_res ? a.res + ob[i] : dummy);
            }
#pragma unroll
            for (int i = 0; i < RBATCH; ++i) {
                rbv[i] = *(has_rb ? a.res_bits + (ob[i] >> 4) : dummy);
                mbv[i] = *(has_bb ? a.bw_bits + (ob[i] >> 4) : dummy);
#ifndef IIF_NO_NT_EPILOGUE_LOADS   // streamed once by this kernel (next reader: the BN backward, from another XCD)
                xvv[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(has_bx ? a.bw_x + ob[i] : dummy));
#else
                xvv[i] = *reinterpret_cast<const u32x4*>(has_bx ? a.bw_x + ob[i] : dummy);
#endif
            }
#pragma unroll
            for (int i = 0; i < RBATCH; ++i) {
                if (!live[i]) continue;
                const int row = r0 + (b0 + i) * RPP;
                const int64_t o = ob[i];
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * PITCH + chunk * 16);
                if (a.aff) {                            // block-uniform: BN affine + identity + ReLU of the block output
                    const u32x4 rr = rrv[i];
                    unsigned bits = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float lo = fmaf(bmean[2 * q], bf16_bits_to_f32(v[q] & 0xffffu), bistd[2 * q]);
                        float hi = fmaf(bmean[2 * q + 1], __uint_as_float(v[q] & 0xffff0000u), bistd[2 * q + 1]);
                        if (has_res) { lo += bf16_bits_to_f32(rr[q] & 0xffffu); hi += __uint_as_float(rr[q] & 0xffff0000u); }
                        bits |= (lo > 0.f ? 1u : 0u) << (2 * q);
                        bits |= (hi > 0.f ? 1u : 0u) << (2 * q + 1);
                        v[q] = pack_bf16x2(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
                    }
                    if (a.relu_out) a.relu_out[o >> 4] = (unsigned char)bits;
                } else if (has_res) {
                    const u32x4 rr = rrv[i];
                    const unsigned rb = has_rb ? rbv[i] : 0xffu;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu) + ((rb >> (2 * q)) & 1u ? bf16_bits_to_f32(rr[q] & 0xffffu) : 0.f);
                        const float hi = __uint_as_float(v[q] & 0xffff0000u) + ((rb >> (2 * q + 1)) & 1u ? __uint_as_float(rr[q] & 0xffff0000u) : 0.f);
                        v[q] = pack_bf16x2(lo, hi);
                    }
                }
                const unsigned mb = has_bb ? mbv[i] : 0xffu;
                if (a.mask_store) {                     // block-uniform: the stored gradient is gated by the upstream ReLU bits
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned lo = (mb >> (2 * q)) & 1u ? (v[q] & 0xffffu) : 0u;
                        const unsigned hi = (mb >> (2 * q + 1)) & 1u ? (v[q] & 0xffff0000u) : 0u;
                        v[q] = lo | hi;
                        if (!has_bx) { bs[2 * q] += bf16_bits_to_f32(lo); bs[2 * q + 1] += __uint_as_float(hi); }
                    }
                }
                if (!a.no_store) {
#ifndef IIF_NO_NT_CONV_STORE     // the tile is next read by another XCD (BN apply): streaming it out keeps the pixel operand's lines in L2 (-0.7 % on the step)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.dst + o));
#else
                    *reinterpret_cast<u32x4*>(a.dst + o) = v;
#endif
                }
                if (has_bx) {                           // (with mask_store v is gated already; gating it again below changes nothing)
                    const u32x4 xv = xvv[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float glo = (mb >> (2 * q)) & 1u ? bf16_bits_to_f32(v[q] & 0xffffu) : 0.f;
                        const float ghi = (mb >> (2 * q + 1)) & 1u ? __uint_as_float(v[q] & 0xffff0000u) : 0.f;
                        const float xlo = (bf16_bits_to_f32(xv[q] & 0xffffu) - bmean[2 * q]) * bistd[2 * q];
                        const float xhi = (__uint_as_float(xv[q] & 0xffff0000u) - bmean[2 * q + 1]) * bistd[2 * q + 1];
                        bs[2 * q] += glo; bq[2 * q] += glo * xlo;
                        bs[2 * q + 1] += ghi; bq[2 * q + 1] += ghi * xhi;
                    }
                } else if (a.mask_store) {              // column sums only, accumulated with the gate above
                } else if (a.bn_partial) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                        bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                        bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
                    }
                }
            }
        }
    }
    if (a.bn_partial) {
        // lanes of a wave that share the chunk (lane % CPR), then the block's waves through LDS scratch behind the
        // tile: fixed order, ONE partial row per pixel tile
        constexpr int NWV = NT / 64;
        float* scratch = reinterpret_cast<float*>(smem);                    // [NWV][2][BN], over the drained tile
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) { bs[q] += __shfl_xor(bs[q], o, 64); bq[q] += __shfl_xor(bq[q], o, 64); }
        }
        __syncthreads();                            // every thread has read its rows of the staged tile
        const int lane = tid & 63, wv = tid >> 6;
        if (lane < CPR) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                scratch[(wv * 2 + 0) * BN + lane * 8 + q] = bs[q];
                scratch[(wv * 2 + 1) * BN + lane * 8 + q] = bq[q];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cd) {
            float s2 = 0.f, q2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) { s2 += scratch[(w * 2 + 0) * BN + tid]; q2 += scratch[(w * 2 + 1) * BN + tid]; }
            float* p = a.bn_partial + (int64_t)(a.bn_row0 + mt) * 2 * a.dpitch + goff + n0 + tid;
            p[0] = s2; p[a.dpitch] = q2;
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;

// UTAP (Cs % KE == 0, <= 32 taps, not a stride-2 data gradient): every lane of a K step is in the same tap,
// so the tap / channel-chunk displacement is a wave-uniform SCALAR (the buffer instruction's soffset) and
// the per-lane voffset (pixel base + this lane's chunk) never changes: address generation costs ~3 VALU
// instructions per row per step (a tap-validity bit test), instead of the general per-piece arithmetic.
// tap list of a launch: the arrays of ConvArgs, or (several parity classes in one launch) those of the block's class
struct TapList { const signed char* dy; const signed char* dx; const unsigned char* w; };

// (TAG: a second kernel that needs the same body instantiates it under another tag — the host pass of hipcc / ROCm 7.2 refuses
// to substitute one specialisation of this template, with its static __shared__ array, into two different __global__ functions)
template <typename T, int BN, bool OUTF32, bool UTAP, int NW = 4, bool SHORTK = false, int TAG = 0>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvArgs& a, unsigned src_bytes, unsigned wgt_bytes, TapList tl, int bofs = 0) {
    // NW waves, each 64 pixels x BN/2 channels: 128 x BN (4 waves) or 256 x 128 (8 waves).  The larger tile moves
    // 12 instead of 16 KB through the vector L1 per MFLOP: the stamps (scripts/conv_stamps.py) show the 4-wave
    // kernel spending half of every K step issuing its LDS-DMA, i.e. bound by the 64 B/clk L1 path, not by MFMA.
    constexpr int BM = 32 * NW;
    constexpr int PE = ET<T>::PE, KE = ET<T>::KE;
    constexpr int NBI = BN / (16 * NW);  // weight DMA pieces per wave per stage
    static_assert(NBI >= 1, "8 waves need the 128-channel tile");
    constexpr int CI = BN / 32;
    constexpr int STAGE = (BM + BN) * 64;
    constexpr int LPS = 2 + NBI;        // DMA instructions per wave per stage
    constexpr unsigned OOB = UTAP ? 0x80000000u : 0xfffffff0u;
    // SHORTK: two LDS stages instead of three, so the block fits 4 times per CU (LDS 34 KB = the epilogue's staging
    // tile, <= 128 registers) instead of 3: one more tile's loads and stores in flight per CU.  With K <= 2 steps
    // (the 64-channel layers of the 56x56 stage) nothing is ever refilled; longer K loops pay a second barrier per step
    // (the stage that was just read is the one refilled).  Used for K <= 2304 (everything the halo / 256-row kernels do
    // not take): measured +18..21 % on the 64-channel 1x1 layers alone (scripts/bm_stream1x1.py), -15 % on 28x28
    // 128->512, -10 % on 14x14 256->1024, -8 % on the 56x56 3x3 layers, nothing slower by more than 1 %; forward + data
    // gradient launches of a ResNet50 step 11.01 -> 10.60 ms serialised.
    constexpr int NST = SHORTK ? 2 : 3;
    constexpr int STG_BYTES = BM * (BN * 2 + 16);            // conv_epilogue_staged parks the bf16 tile here
    constexpr int SMEM_BYTES = NST * STAGE > STG_BYTES || sizeof(T) != 2 || OUTF32 ? NST * STAGE : STG_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int b = (int)blockIdx.x - bofs;
    const int xcd = b & 7, j = b >> 3;
    const int mt = (j / a.ntiles) * 8 + xcd, nt = j % a.ntiles;
    if (mt >= a.mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;
    // grouped convolution: blockIdx.y selects the channel group; a.Cs / a.Cd are per-group widths
    const int grp = blockIdx.y;
    const unsigned gsrc = (unsigned)(grp * a.Cs) * (unsigned)sizeof(T);          // byte offset inside a source pixel
    const unsigned char* wgt_g = a.wgt + (int64_t)grp * a.Cd * a.ldw * (int64_t)sizeof(T);

    // ---- this lane's fixed role inside every 16-row DMA piece: row l>>2, source chunk (l&3)^f(l>>4)
    const int prow = lane >> 2;
    const int chunk = (lane & 3) ^ swz(prow);
    const int HW = a.Hd * a.Wd;
    // fast-path state
    unsigned vbase[2], vbase2[2], vmask[2], vwf[NBI];
    int shiftP = 0;                                           // bytes subtracted from the base so soffset >= 0
    // general-path state
    int by[2], bx[2], ib[2];
    bool mv[2];
    unsigned wrow[NBI];
    int e = chunk * PE, c = 0, r = 0, s = 0;
    if constexpr (UTAP) {
        // most negative tap displacement, in pixels
        int dmin = 0;
        for (int t = 0; t < a.ntaps; ++t) {
            const int dd = tl.dy[t] * a.Ws + tl.dx[t];
            dmin = dd < dmin ? dd : dmin;
        }
        shiftP = -dmin * a.spitch * (int)sizeof(T);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 16 * (2 * wave + i) + prow;
            const bool valid = m < a.M;
            const int mm = valid ? m : 0;
            const int n = mm / HW, rem = mm - n * HW;
            const int y = rem / a.Wd, x = rem - y * a.Wd;
            const int y0 = y << a.in_shift, x0 = x << a.in_shift;
            vbase[i] = ((unsigned)(n * a.Hs * a.Ws + y0 * a.Ws + x0) * (unsigned)a.spitch + (unsigned)(chunk * PE)) * (unsigned)sizeof(T) + gsrc;
            // second source of a 1x1 launch (K continues over its Cs2 channels as "tap 1"): same pixel, its own pitch
            vbase2[i] = ((unsigned)mm * (unsigned)a.Cs2 + (unsigned)(chunk * PE)) * (unsigned)sizeof(T);
            unsigned mask = 0;
            for (int t = 0; t < a.ntaps; ++t) {
                const int ys = y0 + tl.dy[t], xs = x0 + tl.dx[t];
                const bool ok = valid && (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
                mask |= (ok ? 1u : 0u) << t;
            }
            vmask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int n = n0 + 16 * (NBI * wave + i) + prow;
            vwf[i] = n < a.Cd ? ((unsigned)n * (unsigned)a.ldw + (unsigned)(chunk * PE)) * (unsigned)sizeof(T) : OOB;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 16 * (2 * wave + i) + prow;
            mv[i] = m < a.M;
            const int mm = mv[i] ? m : 0;
            const int n = mm / HW, rem = mm - n * HW;
            const int y = rem / a.Wd, x = rem - y * a.Wd;
            by[i] = a.transposed ? y + a.pad : (y << a.sshift) - a.pad;
            bx[i] = a.transposed ? x + a.pad : (x << a.sshift) - a.pad;
            ib[i] = n * a.Hs * a.Ws;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int n = n0 + 16 * (NBI * wave + i) + prow;
            wrow[i] = n < a.Cd ? (unsigned)n * (unsigned)a.ldw * (unsigned)sizeof(T) : OOB;
        }
        const int tap = e / a.Cs;
        c = e - tap * a.Cs;
        r = tap / a.S;
        s = tap - r * a.S;
    }
    // The range check is applied to voffset + soffset on gfx950 (measured: taps of the last image rows were
    // zeroed with num_records = src_bytes), so the window is widened by the base shift; the out-of-range
    // marker 0x80000000 cannot wrap below it whatever soffset is (both operands are < 2 GiB).
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src) - shiftP, 0,
                                                          src_bytes + (unsigned)shiftP + 16u, 0x00020000);
    const auto rs_wgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wgt_g), 0, wgt_bytes, 0x00020000);
    const bool two_src = UTAP && a.src2 != nullptr;                // wave-uniform
    const auto rs_src2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(two_src ? a.src2 : a.src), 0,
                                                           two_src ? (unsigned)a.M * (unsigned)a.Cs2 * (unsigned)sizeof(T) : 16u, 0x00020000);
    // wave-uniform K-step state of the uniform-tap path, kept incrementally: tap (ur, us), first channel uc,
    // tap bit index ut, source soffset usoff (bytes), weight soffset uwoff (bytes)
    int uc = 0, ut = 0;
    unsigned usoff = 0, uwoff = 0;
    // (the tap table is read with vector loads: without readfirstlane the compiler cannot tell that the two soffsets are
    // wave-uniform and wraps every LDS-DMA of the K loop in a waterfall loop - readfirstlane, compare, exec mask, branch)
    if constexpr (UTAP) {
        usoff = (unsigned)__builtin_amdgcn_readfirstlane((tl.dy[0] * a.Ws + tl.dx[0]) * a.spitch * (int)sizeof(T) + shiftP);
        uwoff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(tl.w[0] * a.Cs) * (unsigned)sizeof(T)));
    }

    auto issue = [&](auto stage_c) {
        constexpr int stage = decltype(stage_c)::value;
        unsigned char* A = smem + stage * STAGE;
        unsigned char* B = A + BM * 64;
        if constexpr (UTAP) {
            const int t = ut;
            const unsigned soff = usoff, soffw = uwoff;
            const bool second = two_src && t == 1;                 // wave-uniform: this K step reads the second source
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = ((vmask[i] >> t) & 1u) ? (second ? vbase2[i] : vbase[i]) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs_src2 : rs_src, (lds_void*)(A + (2 * wave + i) * 1024), 16, off, soff, 0, IIF_CONV_AUX_SRC);
            }
#pragma unroll
            for (int i = 0; i < NBI; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_void*)(B + (NBI * wave + i) * 1024), 16, vwf[i], soffw, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ys, xs;
                bool ok = mv[i] && r < a.R;
                if (a.transposed) {
                    const int ty = by[i] - r, tx = bx[i] - s;
                    ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & a.sshift) == 0;
                    ys = ty >> a.sshift; xs = tx >> a.sshift;
                    ok = ok && ys < a.Hs && xs < a.Ws;
                } else {
                    ys = by[i] + r; xs = bx[i] + s;
                    ok = ok && (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
                }
                const unsigned off = ok ? ((unsigned)(ib[i] + ys * a.Ws + xs) * (unsigned)a.spitch + (unsigned)c) * (unsigned)sizeof(T) + gsrc : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(A + (2 * wave + i) * 1024), 16, off, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const unsigned off = (wrow[i] != OOB && e < a.K) ? wrow[i] + (unsigned)e * (unsigned)sizeof(T) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_void*)(B + (NBI * wave + i) * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        if constexpr (UTAP) {
            uwoff += KE * (unsigned)sizeof(T);
            usoff += KE * (unsigned)sizeof(T);
            uc += KE;
            if (uc >= ((two_src && ut == 1) ? a.Cs2 : a.Cs)) {
                uc = 0; ++ut;
                const int tt = ut < a.ntaps ? ut : 0;
                usoff = (unsigned)__builtin_amdgcn_readfirstlane((tl.dy[tt] * a.Ws + tl.dx[tt]) * a.spitch * (int)sizeof(T) + shiftP);
                uwoff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(tl.w[tt] * a.Cs) * (unsigned)sizeof(T)));
            }
        } else {
            e += KE;
            c += KE;
            while (c >= a.Cs) { c -= a.Cs; if (++s == a.S) { s = 0; ++r; } }
        }
    };

    f32x4 acc[CI][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) acc[ci][pj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = UTAP ? (two_src ? (a.Cs + a.Cs2) / KE : a.ntaps * (a.Cs / KE)) : (a.K + KE - 1) / KE;
    const int fr = lane & 15, fc = lane >> 4;
    // per-lane fragment offsets inside a stage (stage bases are compile-time immediates below)
    int wofs[CI], xofs[4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const int row = wn * (BN / 2) + ci * 16 + fr;
        wofs[ci] = BM * 64 + row * 64 + ((fc ^ swz(row)) << 4);
    }
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int row = wm * 64 + pj * 16 + fr;
        xofs[pj] = row * 64 + ((fc ^ swz(row)) << 4);
    }
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // one K step on LDS stage S: retire this wave's DMA of step k (leave step k+1's in flight), meet the
    // other waves, refill the stage that was read in step k-1 with step k+2, multiply stage S
#ifdef IIF_CONV_STAMPS
    unsigned long long st_wait = 0, st_lds = 0, st_dma = 0, st_mfma = 0, st_prev = 0, st_begin = 0;
    IIF_STAMP(st_begin);
    st_prev = st_begin;
#endif
    auto step = [&](auto stage_c, auto refill_c, int k) {
        constexpr int S = decltype(stage_c)::value;
#ifdef IIF_CONV_STAMPS
        unsigned long long t0, t1, t2, t3;
        IIF_STAMP(t0);
        st_mfma += t0 - st_prev;               // previous step's MFMA issue (first step: prologue)
#endif
        if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(t1);
#endif
        const unsigned char* base = smem + S * STAGE;
        u32x4 wf[CI], xf[4];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) wf[ci] = *reinterpret_cast<const u32x4*>(base + wofs[ci]);
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) xf[pj] = *reinterpret_cast<const u32x4*>(base + xofs[pj]);
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(t2);
#endif
        // the refill DMA is issued behind the fragment reads, so its issue time covers their LDS latency
        if (k + 2 < nk) {
            if constexpr (SHORTK) {
                // two stages: the stage just read is the one refilled, so every wave must hold its fragments first
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            advance(); issue(refill_c);
        }
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(t3);
        st_wait += t1 - t0; st_lds += t2 - t1; st_dma += t3 - t2; st_prev = t3;
#endif
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                if constexpr (sizeof(T) == 2) {
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wf[ci]), __builtin_bit_cast(bf16x8, xf[pj]), acc[ci][pj], 0, 0, 0);
                } else {
                    const f32x4 wv = __builtin_bit_cast(f32x4, wf[ci]), xv = __builtin_bit_cast(f32x4, xf[pj]);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[ci][pj], 0, 0, 0);
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[ci][pj], 0, 0, 0);
                }
            }
    };
    if (nk > 0) issue(S0{});
    if (nk > 1) { advance(); issue(S1{}); }
    if constexpr (SHORTK) {                       // two stages: step k reads stage k & 1 and refills it with step k + 2
        for (int k = 0; k < nk; k += 2) {
            step(S0{}, S0{}, k);
            if (k + 1 < nk) step(S1{}, S1{}, k + 1);
        }
    } else {
        for (int k = 0; k < nk; k += 3) {
            step(S0{}, S2{}, k);
            if (k + 1 < nk) step(S1{}, S0{}, k + 1);
            if (k + 2 < nk) step(S2{}, S1{}, k + 2);
        }
    }
#ifdef IIF_CONV_STAMPS
    unsigned long long st_loop_end, st_end;
    IIF_STAMP(st_loop_end);
    st_mfma += st_loop_end - st_prev;
#endif
    if constexpr (sizeof(T) == 2 && !OUTF32) {
        if ((a.Cd & 7) == 0 && a.bias == nullptr) {     // wave-uniform
            conv_epilogue_staged<BN, BM>(a, acc, smem, m0, n0, mt, wm, wn, fr, fc, grp * a.Cd);
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(st_end);
            if (g_stamps && b < 512 && lane == 0) {
                unsigned long long* o = g_stamps + ((int64_t)b * 4 + wave) * 8;
                o[0] = st_wait; o[1] = st_lds; o[2] = st_dma; o[3] = st_mfma; o[4] = st_end - st_loop_end;
                o[5] = st_end - st_begin; o[6] = (unsigned long long)nk; o[7] = st_begin;
            }
#endif
            return;
        }
    }
    conv_epilogue<T, BN, OUTF32>(a, acc, m0, n0, wm, wn, fr, fc, grp * a.Cd);
}

// two kernel names instead of a fourth template flag (hipcc/ROCm 7.2 fails to emit the host stub of a
// __global__ template whose body differs only by such a flag)
template <typename T, int BN, bool OUTF32>
__global__ void __launch_bounds__(256) conv_igemm_dma_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv_igemm_dma_body<T, BN, OUTF32, false>(a, src_bytes, wgt_bytes, TapList{a.tap_dy, a.tap_dx, a.tap_w});
}
template <typename T, int BN, bool OUTF32>
__global__ void __launch_bounds__(256) conv_igemm_dma_utap_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv_igemm_dma_body<T, BN, OUTF32, true>(a, src_bytes, wgt_bytes, TapList{a.tap_dy, a.tap_dx, a.tap_w});
}
// K <= 64 (two K steps): 4 blocks per CU (see SHORTK)
template <int BN>
__global__ void __launch_bounds__(256, 4) conv_igemm_dma_utap_k64_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv_igemm_dma_body<unsigned short, BN, false, true, 4, true>(a, src_bytes, wgt_bytes, TapList{a.tap_dy, a.tap_dx, a.tap_w});
}
// The parity classes of a stride-2 data gradient in ONE launch.  As four launches each class offered 200-400 tiles of one or
// two K steps to 1 024 block slots (the 28x28 -> 56x56 3x3 gradient: 0.45 ms in the step for 0.46 GB = 1.0 TB/s); together
// they fill the chip.  A block finds its class by its index, takes that class's grid, tap list, scatter offsets and partial
// rows, and runs the ordinary body.
struct ConvClass {
    int Hd, Wd, M, mtiles, doy, dox, ntaps, bn_row0, bstart;
    signed char tap_dy[16], tap_dx[16];
    unsigned char tap_w[16];
};
struct ConvArgsMC { ConvArgs a; ConvClass cls[4]; int ncls; };
template <int BN>
__device__ __forceinline__ void conv_mc_body(const ConvArgsMC& p, unsigned src_bytes, unsigned wgt_bytes) {
    int k = 0;
    for (int i = 1; i < p.ncls; ++i)
        if ((int)blockIdx.x >= p.cls[i].bstart) k = i;
    const ConvClass& c = p.cls[k];
    ConvArgs a = p.a;
    a.Hd = c.Hd; a.Wd = c.Wd; a.M = c.M; a.mtiles = c.mtiles; a.doy = c.doy; a.dox = c.dox; a.ntaps = c.ntaps; a.bn_row0 = c.bn_row0;
    TapList tl;
    tl.dy = c.tap_dy; tl.dx = c.tap_dx; tl.w = c.tap_w;
    conv_igemm_dma_body<unsigned short, BN, false, true, 4, true, 1>(a, src_bytes, wgt_bytes, tl, c.bstart);
}
// plain kernel names around the body template (see the note on conv_igemm_dma_kernel: host stubs)
__global__ void __launch_bounds__(256, 4) conv_igemm_dma_utap_k64_mc128_kernel(ConvArgsMC p, unsigned src_bytes, unsigned wgt_bytes) {
    conv_mc_body<128>(p, src_bytes, wgt_bytes);
}
__global__ void __launch_bounds__(256, 4) conv_igemm_dma_utap_k64_mc64_kernel(ConvArgsMC p, unsigned src_bytes, unsigned wgt_bytes) {
    conv_mc_body<64>(p, src_bytes, wgt_bytes);
}

// ---------------------------------------------------------------- 3x3 / stride 1 / pad 1 with an LDS halo window
// The tap-by-tap kernel above pulls every source pixel through the vector L1 nine times (once per tap); the stamps
// show that path, not the MFMA, bounding it.  Here a 256-pixel tile loads the (rows + 2) x (W + 2) window of its
// source pixels ONCE per 32-channel chunk (a "halo" image in LDS, double buffered) and reads the nine taps'
// fragments out of it at displaced rows; only the weights still stream per tap (a 6-deep ring, 8 KB per step).
// L1 traffic per MFLOP drops from 16 KB (128x128 tap kernel) to ~6 KB.  Pixels are addressed in "virtual" rows
// v = n*(H+2) + y + 1 so that every image carries its own zero rows above and below: halo pixels that fall on them,
// or left/right of the image, are out-of-range lanes of the LDS-DMA (zeros, no traffic).
// bf16, 8 waves x (64 pixels x 64 channels), forward and stride-1 data gradient (tap list), staged epilogue.
// BM = 256: 8 waves, one block per CU (112 KB).  BM = 128: 4 waves, 70 KB, TWO blocks per CU: two independent barrier
// domains, so one block's fetch phase overlaps the other's MFMAs (weight ring 4 deep, 2 weight pieces per wave and step).
template <int BM>
__device__ __forceinline__ void conv3x3_halo_body(const ConvArgs& a, unsigned src_bytes, unsigned wgt_bytes) {
    constexpr int BN = 128, CI = BN / 32, NW = BM / 32;
    constexpr int HR = BM == 256 ? 512 : 304, ABUF = HR * 64;   // halo rows per buffer, 64 B (32 channels) each
    constexpr int BST = BN * 64, NBS = BM == 256 ? 6 : 4;       // weight stage and ring depth
    constexpr int NAP = BM == 256 ? 4 : 5;           // halo pieces per wave and chunk
    constexpr int NBP = 8 / NW;                      // weight pieces per wave and step
    constexpr int AHEAD = NBS - 1;                   // weight stages in flight
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * ABUF + NBS * BST + 1024];   // + dump piece

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int mt = (j / a.ntiles) * 8 + xcd, nt = j % a.ntiles;
    if (mt >= a.mtiles) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int H = a.Hd, W = a.Wd, HW = H * W, W2 = W + 2, H2 = H + 2;
    // virtual row of the tile's first / last pixel
    const int nf = m0 / HW, remf = m0 - nf * HW;
    const int vfirst = nf * H2 + remf / W + 1;
    const int mlast = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
    const int nl = mlast / HW, reml = mlast - nl * HW;
    const int vlast = nl * H2 + reml / W + 1;
    const int vbase = vfirst - 1;
    const int Hn = (vlast - vfirst + 3) * W2;        // halo rows in use (host guarantees <= HR)

    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);
    const auto rs_wgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wgt), 0, wgt_bytes, 0x00020000);

    // ---- DMA roles of this lane.  Halo piece p (16 halo rows) is issued by wave p % 8 as its (p / 8)-th piece.
    unsigned avoff[NAP];
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int hr = 16 * (wave + NW * i) + (lane >> 2);
        const int chunk = (lane & 3) ^ swz(hr);
        const int vr = vbase + hr / W2, xx = hr % W2 - 1;
        const int nn = vr / H2, yy = vr % H2 - 1;
        const bool ok = hr < Hn && hr < HR && nn < a.N && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        avoff[i] = ok ? ((unsigned)((nn * H + yy) * W + xx) * (unsigned)a.spitch + (unsigned)(chunk * 8)) * 2u : OOB;
    }
    unsigned bvoff[NBP];
#pragma unroll
    for (int q = 0; q < NBP; ++q) {
        const int row = 16 * (wave * NBP + q) + (lane >> 2);
        const int chunk = (lane & 3) ^ swz(row);
        bvoff[q] = n0 + row < a.Cd ? ((unsigned)(n0 + row) * (unsigned)a.ldw + (unsigned)(chunk * 8)) * 2u : OOB;
    }
    // ---- fragment addresses
    const int fr = lane & 15, fc = lane >> 4;
    int wofs[CI], hrow0[4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        const int row = wn * (BN / 2) + ci * 16 + fr;
        wofs[ci] = row * 64 + ((fc ^ swz(row)) << 4);
    }
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int m = m0 + wm * 64 + pj * 16 + fr;
        const int mm = m < a.M ? m : m0;
        const int n = mm / HW, rem = mm - n * HW;
        const int y = rem / W, x = rem - y * W;
        hrow0[pj] = (n * H2 + y + 1 - vbase) * W2 + x + 1;
    }
    f32x4 acc[CI][4];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) acc[ci][pj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cs / 32;
    const int G = nchunks * 9;
    unsigned char* const Bring = smem + 2 * ABUF;
    auto issue_a = [&](int c, int i) {                         // piece i of this wave for chunk c (c >= nchunks: no-op zeros)
        const int p = wave + NW * i;
        unsigned char* dst = smem + (c & 1) * ABUF + (p < HR / 16 ? p : 0) * 1024;
        const unsigned off = (c < nchunks && p < HR / 16) ? avoff[i] : OOB;     // surplus slots: zeros over piece 0's OOB lanes?
        if (p < HR / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)dst, 16, off, (unsigned)(c * 64), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(smem + 2 * ABUF + NBS * BST), 16, OOB, 0, 0, 0);
    };
    auto issue_b = [&](int g, int slot) {                       // weights of global step g = (chunk, tap)
        const int c = g / 9, t = g - c * 9;
        const unsigned so = g < G ? (unsigned)(a.tap_w[t] * a.Cs + c * 32) * 2u : 0u;
#pragma unroll
        for (int q = 0; q < NBP; ++q) {
            unsigned char* dst = Bring + slot * BST + (wave * NBP + q) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_void*)dst, 16, g < G ? bvoff[q] : OOB, so, 0, 0);
        }
    };
    // prologue: halo of chunk 0, then the first five weight stages (program order fixes the counted waits below)
#pragma unroll
    for (int i = 0; i < NAP; ++i) issue_a(0, i);
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) issue_b(g, g);

    // (Software-pipelining the fragment reads into a second register set was measured 5 % slower: 214 VGPRs.)
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = a.tap_dy[t] * W2 + a.tap_dx[t];
    int slot = 0;                                               // ring slot of the step being fetched
    // fetch(T, g, c): arrival sync of step g = (chunk c, tap T), refill DMAs, fragment reads of step g
    auto fetch = [&](auto tc, int g, int c, u32x4 (&wf)[CI], u32x4 (&xf)[4]) {
        constexpr int T = decltype(tc)::value;
        // loads issued after this step's weights: the later weight stages in flight (NBP each) + the halo pieces
        // issued in the last AHEAD steps (one per step at taps < NAP)
        constexpr int NP256[9] = {4, 5, 6, 7, 8, 8, 7, 6, 5};
        constexpr int NP128[9] = {4, 5, 6, 7, 7, 7, 6, 5, 4};
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BM == 256 ? NP256[T] : NP128[T]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the slot about to be refilled
        __builtin_amdgcn_s_barrier();
        const int refill = slot == 0 ? NBS - 1 : slot - 1;      // the slot read in the previous step
        issue_b(g + AHEAD, refill);
        if constexpr (T < NAP) issue_a(c + 1, T);
        const unsigned char* Ab = smem + (c & 1) * ABUF;
        const unsigned char* Bb = Bring + slot * BST;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) wf[ci] = *reinterpret_cast<const u32x4*>(Bb + wofs[ci]);
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int hr = hrow0[pj] + toff[T];
            xf[pj] = *reinterpret_cast<const u32x4*>(Ab + hr * 64 + ((fc ^ swz(hr)) << 4));
        }
        slot = slot == NBS - 1 ? 0 : slot + 1;
    };
    auto mma = [&](const u32x4 (&wf)[CI], const u32x4 (&xf)[4]) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj)
                acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wf[ci]), __builtin_bit_cast(bf16x8, xf[pj]), acc[ci][pj], 0, 0, 0);
    };
#ifdef IIF_CONV_STAMPS
    unsigned long long st_fetch = 0, st_mma = 0, st_a, st_b, st_c, st_begin;
    IIF_STAMP(st_begin);
#endif
    for (int c = 0; c < nchunks; ++c) {
        const int g0 = c * 9;
        u32x4 wf[CI], xf[4];
#ifdef IIF_CONV_STAMPS
#define IIF_HALO_STEP(T) IIF_STAMP(st_a); fetch(std::integral_constant<int, T>{}, g0 + T, c, wf, xf); IIF_STAMP(st_b); \
                         mma(wf, xf); IIF_STAMP(st_c); st_fetch += st_b - st_a; st_mma += st_c - st_b;
#else
#define IIF_HALO_STEP(T) fetch(std::integral_constant<int, T>{}, g0 + T, c, wf, xf); mma(wf, xf);
#endif
        IIF_HALO_STEP(0) IIF_HALO_STEP(1) IIF_HALO_STEP(2) IIF_HALO_STEP(3) IIF_HALO_STEP(4)
        IIF_HALO_STEP(5) IIF_HALO_STEP(6) IIF_HALO_STEP(7) IIF_HALO_STEP(8)
#undef IIF_HALO_STEP
    }
#ifdef IIF_CONV_STAMPS
    IIF_STAMP(st_c);
    if (g_stamps && b < 512 && lane == 0 && wave < 4) {
        unsigned long long* o = g_stamps + ((int64_t)b * 4 + wave) * 8;
        o[0] = st_fetch; o[1] = 0; o[2] = 0; o[3] = st_mma; o[4] = 0; o[5] = st_c - st_begin; o[6] = (unsigned long long)G; o[7] = st_begin;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tail's no-op DMAs still write zeros to LDS
    conv_epilogue_staged<BN, BM>(a, acc, smem, m0, n0, mt, wm, wn, fr, fc, 0);
}

// plain kernel names around the body template (see the note on conv_igemm_dma_kernel: host stubs)
__global__ void __launch_bounds__(512) conv3x3_halo_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv3x3_halo_body<256>(a, src_bytes, wgt_bytes);
}
__global__ void __launch_bounds__(256) conv3x3_halo128_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv3x3_halo_body<128>(a, src_bytes, wgt_bytes);
}

// ---------------------------------------------------------------- 3x3 / stride 1 / pad 1, generation 2 (round 3)
// What bounded the halo kernel above (26-31 % MFMA busy): eight waves of 64 x 64 meet at ONE BARRIER PER TAP (the weights of
// every tap pass through a shared LDS ring), read 8 KB of fragments per 16 MFMAs and, being in lock step, all sit in their
// fetch phase together.  Here
//   * the WEIGHTS never touch LDS: they are stored once per step as ready-made MFMA fragments (iif_conv_pack_fragments:
//     [channel tile of 16][tap][32-channel chunk] -> 1 KB, lane l's 16 bytes at l * 16), so a wave fetches the four
//     fragments of its 64 output channels with four fully coalesced 1-KB loads straight from L2 into registers, one step
//     ahead of their use.  No weight ring, no weight barrier: the only block-wide hand-over left is the halo window;
//   * the halo window ((rows + 2) x (W + 2) source pixels of a 256-pixel tile, 32 channels = 64 B per pixel, double
//     buffered, virtual rows as above) is refilled ONCE PER 32-CHANNEL CHUNK by LDS-DMA: one barrier per nine taps;
//   * a wave owns 128 pixels x 64 channels (128 accumulator registers): 12 KB of operands per 32 MFMAs instead of 8 KB per
//     16; four waves = one per SIMD, two blocks per CU (75 KB of LDS, <= 256 registers) so that one block's prologue /
//     epilogue / barrier runs under the other's MFMAs.  Layers with 64 output channels use four 64 x 64 waves along M.
// Same accumulation order per output as the halo kernel (chunk outer, tap inner, 32 channels per MFMA): bit-identical results.
// exact floor(x / d) for 0 <= x < 2^22 and 0 < d < 2^16 with rd = 1.0f / d (IEEE): (x + 0.5) / d is never closer than 0.5 / d to an
// integer, far outside the product's rounding error.  The tile geometry needs ~30 of these per lane and tile; as integer
// divisions they were ~4 us of every tile's prologue.
__device__ __forceinline__ int fdiv(int x, float rd) { return (int)(((float)x + 0.5f) * rd); }

// G16 (round 6): grouped layers whose groups are <= 16 channels wide (ResNeXt 32x4d: 4 / 8 / 16 channels per group at 56 / 28 / 14).
// A 64-channel chunk's weight matrix is then block-diagonal in 16 x 16 blocks: output tile ci (16 channels) reads the SAME 16 input
// channels only.  The dense loop spends 15 / 16 (7 / 8, 3 / 4) of its MFMAs on zeros; here K of an MFMA is TWO TAPS x the tile's 16
// input channels (lanes with K group 0-1 read tap 2p, K group 2-3 tap 2p + 1, each 8 of the 16 channels), 5 MFMAs per output tile
// and pixel fragment instead of 18: 80 per wave and tile against 288.  The 20 weight fragments of a wave (4 tiles x 5 tap pairs,
// iif_conv_pack_fragments_g16) stay in registers for the whole tile; pair 4's second tap does not exist: its weights are packed as
// zeros and its lanes re-read tap 8 (finite values).  Same halo window, same staged drain.
template <int WM, int WN, int TM, int NAP, bool G16 = false>
__device__ __forceinline__ void conv3x3_v2_body(const ConvArgs& a, unsigned src_bytes) {
    constexpr int BM = WM * TM, BN = WN * 64, NWV = WM * WN, PJ = TM / 16, CI = 4;
    static_assert(NWV == 4 && BM == 256, "four waves, 256-pixel tiles");
    // NAP halo pieces (16 rows) per wave and chunk, all of them always issued: 10 covers 56x56 windows (10 x 58 = 580 rows),
    // 8 every window of <= 512 rows (28x28 and smaller)
    constexpr int HR = NAP * NWV * 16, ABUF = HR * 64;   // halo rows per buffer
    constexpr int PITCH = BN * 2 + 16, STG = BM * PITCH;
    // two blocks per CU: the staged output tile overlays the halo buffers
    constexpr int SMEM = STG > 2 * ABUF ? STG : 2 * ABUF;
    static_assert(SMEM <= 80 * 1024, "LDS budget");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int WD = 3;                                // weight register sets: two taps ahead
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    unsigned char* const stage = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int H = a.Hd, W = a.Wd, HW = H * W, W2 = W + 2, H2 = H + 2;
    const float rHW = 1.0f / (float)HW, rW = 1.0f / (float)W, rW2 = 1.0f / (float)W2, rH2 = 1.0f / (float)H2;
    const int fr = lane & 15, fc = lane >> 4;
    const int nchunks = a.Cs / 32;
    // grouped launch (64-channel chunks, non-persistent variant): blockIdx.y = chunk; its source / destination channels start
    // at grp * Cs / grp * Cd of the spitch / dpitch wide tensors, its fragments follow those of the chunks before it
    const int grp = (int)blockIdx.y;
    const unsigned gsrc = (unsigned)(grp * a.Cs) * 2u;
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);
    // (buffer loads: ONE address register per lane = lane * 16, the fragment is selected by the scalar offset)
    const auto rs_wf = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wfrag), 0,
                                                         G16 ? (unsigned)a.groups * 20u * 1024u
                                                             : (unsigned)a.groups * (unsigned)a.Cd * 9u * (unsigned)a.Cs * 2u, 0x00020000);
    const unsigned wv = (unsigned)lane * 16u;
    const unsigned ci_stride = 9u * (unsigned)nchunks * 1024u;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = a.tap_dy[t] * W2 + a.tap_dx[t];

    // ---- per-tile state
    int m0 = 0, n0 = 0, mt = 0;
    unsigned wbase = 0;
    unsigned avoff[NAP];                                 // halo DMA roles: piece p (16 halo rows) belongs to wave p % 4
    unsigned hrow_pk[PJ / 2];                            // two halo rows (< 640) of this lane's pixel fragments per register
    // tile index -> (pixel tile, channel tile), XCD-aware: the channel tiles of a pixel tile are neighbours on one XCD
    auto set_tile = [&](int b) -> bool {
        const int xcd = b & 7, j = b >> 3;
        mt = (j / a.ntiles) * 8 + xcd;
        const int nt = j % a.ntiles;
        if (mt >= a.mtiles) return false;
        m0 = mt * BM; n0 = nt * BN;
        // virtual row of the tile's first / last pixel (every image carries its own zero rows above and below)
        const int nf = m0 / HW, remf = m0 - nf * HW;
        const int vfirst = nf * H2 + remf / W + 1;
        const int mlast = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
        const int nl = mlast / HW, reml = mlast - nl * HW;
        const int vlast = nl * H2 + reml / W + 1;
        const int vbase = vfirst - 1;
        const int Hn = (vlast - vfirst + 3) * W2;        // halo rows in use (host guarantees <= HR)
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int hr = 16 * (wave + NWV * i) + (lane >> 2);
            const int chunk = (lane & 3) ^ swz(hr);
            const int q1 = fdiv(hr, rW2);
            const int vr = vbase + q1, xx = hr - q1 * W2 - 1;
            const int nn = fdiv(vr, rH2), yy = vr - nn * H2 - 1;
            const bool ok = hr < Hn && nn < a.N && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            // branch-free: bit 31 puts the lane out of range whatever the rest is (operands are < 2 GiB)
            avoff[i] = ((((unsigned)((nn * H + yy) * W + xx) * (unsigned)a.spitch + (unsigned)(chunk * 8)) * 2u) & 0x7fffffffu) | (ok ? 0u : OOB);
        }
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj) {
            const int m = m0 + wm * TM + pj * 16 + fr;
            const int mm = m < a.M ? m : m0;
            const int n = fdiv(mm, rHW), rem = mm - n * HW;
            const int y = fdiv(rem, rW), x = rem - y * W;
            const unsigned hrw = (unsigned)((n * H2 + y + 1 - vbase) * W2 + x + 1);
            if (pj & 1) hrow_pk[pj >> 1] |= hrw << 16; else hrow_pk[pj >> 1] = hrw;
        }
        // weight fragments of this wave: channel tiles (n0 + wn * 64) / 16 + ci; fragment (ci, tap slot, chunk) is 1 KB
        wbase = (unsigned)((grp * a.Cd + n0 + wn * 64) >> 4) * 9u * (unsigned)nchunks * 1024u;
        return true;
    };
    auto issue_halo = [&](int c) {
        unsigned char* dst = smem + (c & 1) * ABUF;
#pragma unroll
        for (int i = 0; i < NAP; ++i)                         // pieces beyond the window in use are out-of-range lanes: zeros, no traffic
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(dst + (wave + NWV * i) * 1024), 16, avoff[i], (unsigned)(c * 64) + gsrc, 0, 0);
    };
    auto wload = [&](int t, int c, u32x4 (&wf)[CI]) {
        const unsigned so = wbase + ((unsigned)a.tap_w[t] * (unsigned)nchunks + (unsigned)c) * 1024u;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
            wf[ci] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wf, wv, so + ci * ci_stride, 0));
    };

    // The chunk body is a software pipeline over HALF steps (one tap, four of the wave's pixel fragments = 16 MFMAs):
    //   reads of half h + 1 are issued before the MFMAs of half h, the address arithmetic of half h + 2 (the XOR swizzle
    //   follows the tap-displaced row) sits in their shadow, the weights of tap t + 2 are requested at the start of tap t
    //   (three register sets: one tap ahead left the first MFMA of every tap waiting for L2).  The straightforward order
    //   (addresses, reads, wait, MFMAs per half) had the matrix pipe idle half the time: ~110 cycles of address VALU + ~140
    //   of LDS latency in front of every 256 cycles of MFMAs.
    constexpr int HPT = PJ / 4;                            // halves per tap (2 for 128-pixel wave tiles, 1 for 64)
    constexpr int NH = 9 * HPT;                            // halves per chunk
    const unsigned smem_base = (unsigned)(uintptr_t)smem;   // LDS byte address of the halo buffers (1 KB aligned)
    const unsigned fcs = (unsigned)fc << 4;
    // Weights: three register sets, the weights of tap t + 2 requested at the start of tap t.  (Round 3 also built a ring of nine
    // taps on a persistent 128-channel variant: level with or behind the halo kernel on every shape, removed in round 4;
    // its phase stamps are in profiles/r3_conv3x3_fragment_kernel.txt.)
    u32x4 wb[WD][CI];

#ifdef IIF_CONV_STAMPS
    unsigned long long v_begin, v_a, v_b, v_pro = 0, v_loop = 0, v_bound = 0, v_epi = 0, v_tiles = 0;
    IIF_STAMP(v_begin);
#endif
    int tile = (int)blockIdx.x;
    if (!set_tile(tile)) return;
    issue_halo(0);
    u32x4 wg[G16 ? CI : 1][G16 ? 5 : 1];                   // G16: every fragment of the wave's four output tiles
    if constexpr (G16) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pr = 0; pr < 5; ++pr)
                wg[ci][pr] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wf, wv, (unsigned)((grp * 4 + ci) * 5 + pr) * 1024u, 0));
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) wload(t, 0, wb[t]);
    }
    for (;;) {
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(v_a);
#endif
        f32x4 acc[CI][PJ];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pj = 0; pj < PJ; ++pj) acc[ci][pj] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // halo window of chunk 0 (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(v_b); v_pro += v_b - v_a; ++v_tiles;
#endif
        for (int c = 0; c < nchunks; ++c) {
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(v_a);
#endif
            // buffer (c + 1) & 1 was last read in chunk c - 1, which every wave has left (barrier at its end)
            if (c + 1 < nchunks) issue_halo(c + 1);
            if constexpr (G16) {
                // output tiles 2c and 2c + 1 read the two 16-channel halves of this 32-channel chunk
                const unsigned hbase = smem_base + (unsigned)((c & 1) * ABUF);
#pragma unroll
                for (int pr = 0; pr < 5; ++pr) {
                    const int tl = (fc >> 1) ? (2 * pr + 1 < 9 ? 2 * pr + 1 : 8) : 2 * pr;       // this lane's tap of the pair
                    const int to = (fc >> 1) ? toff[2 * pr + 1 < 9 ? 2 * pr + 1 : 8] : toff[2 * pr];
                    (void)tl;
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        u32x4 xg[PJ];
#pragma unroll
                        for (int q = 0; q < PJ; ++q) {
                            const unsigned hr = ((hrow_pk[q >> 1] >> ((q & 1) * 16)) & 0xffffu) + (unsigned)to;
                            const unsigned chk = (unsigned)(par * 2 + (fc & 1)) ^ ((hr >> 1) & 2u);
                            xg[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(hbase + (hr << 6) + (chk << 4));
                        }
                        if (2 * c + par < CI) {
#pragma unroll
                            for (int q = 0; q < PJ; ++q)
                                acc[(2 * c + par) % CI][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, wg[(2 * c + par) % CI][pr]), __builtin_bit_cast(bf16x8, xg[q]), acc[(2 * c + par) % CI][q], 0, 0, 0);
                        }
                    }
                }
                if (c + 1 < nchunks) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's halo pieces of chunk c + 1
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                continue;
            }
            const unsigned fcsb = fcs + smem_base + (unsigned)((c & 1) * ABUF);
            // (an opaque zero per chunk: without it the fragment addresses of all nine taps are hoisted out of the chunk loop as
            // loop invariants, 72 registers the weight ring needs)
            int opaque0 = 0;
            asm volatile("" : "+s"(opaque0));
            // LDS address of pixel fragment q of half h: row hr = pixel's halo row + tap shift; 64-byte row, 16-byte chunk fc ^ swz(hr)
            auto addr4 = [&](int h, unsigned (&ad)[4]) {
                const int t = h / HPT, ph = (h % HPT) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned hr = ((hrow_pk[(ph + q) >> 1] >> (((ph + q) & 1) * 16)) & 0xffffu) + (unsigned)(toff[t] + opaque0);
                    ad[q] = (hr << 6) + (fcsb ^ ((hr << 3) & 32u));
                }
            };
            auto read4 = [&](const unsigned (&ad)[4], u32x4 (&xf)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xf[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(ad[q]);
            };
            u32x4 xfb[2][4];
            unsigned ad[4];
            addr4(0, ad);
            read4(ad, xfb[0]);
            addr4(1, ad);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int t = h / HPT, ph = (h % HPT) * 4;
                if (h + 1 < NH) read4(ad, xfb[(h + 1) & 1]);
                if (h % HPT == 0) {                       // start of tap t: the weights of tap t + 2 (of the next chunk at the end)
                    const int t2 = t + 2 < 9 ? t + 2 : t + 2 - 9, c2 = t + 2 < 9 ? c : c + 1;
                    if (c2 < nchunks) wload(t2, c2, wb[(t + 2) % 3]);
                }
                __builtin_amdgcn_sched_barrier(0);        // memory requests stay in front of this half's MFMAs
                unsigned adn[4];
                if (h + 2 < NH) addr4(h + 2, adn);
#pragma unroll
                for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[ci][ph + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wb[t % WD][ci]), __builtin_bit_cast(bf16x8, xfb[h & 1][q]), acc[ci][ph + q], 0, 0, 0);
                // one MFMA, then up to two of the next-but-one half's address instructions in its shadow
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (h + 2 < NH) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ad[q] = adn[q];
                }
            }
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(v_b); v_loop += v_b - v_a;
#endif
            // chunk boundary: the last tap's slot, then: this wave's halo pieces of chunk c + 1 have landed (everything but the
            // 9 CI loads issued after them = the next chunk's weights), its reads of buffer c & 1 are done; then all waves
            if (c + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CI) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(v_a); v_bound += v_a - v_b;
#endif
        }
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(v_a);
#endif
        // ---- epilogue: park the tile in LDS, then the shared 16-byte drain
        const int dm0 = m0, dn0 = n0, dmt = mt;
        __syncthreads();                                // every wave is done reading the last halo buffer (and the stage of the previous tile)
#pragma unroll
        for (int pj = 0; pj < PJ; ++pj)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const int row = wm * TM + pj * 16 + fr, ch = wn * 64 + ci * 16 + fc * 4;
                u32x2 w;
                w.x = pack_bf16x2(acc[ci][pj].x, acc[ci][pj].y);
                w.y = pack_bf16x2(acc[ci][pj].z, acc[ci][pj].w);
                *reinterpret_cast<u32x2*>(stage + row * PITCH + ch * 2) = w;
            }
        const bool more_tiles = false;
        __syncthreads();
        staged_drain<BN, BM, 256>(a, stage, dm0, dn0, dmt, grp * a.Cd);
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(v_b); v_epi += v_b - v_a;
        if (!more_tiles && g_stamps && blockIdx.x < 512 && lane == 0) {
            unsigned long long* o = g_stamps + ((int64_t)blockIdx.x * 4 + wave) * 8;
            o[0] = v_pro; o[1] = v_loop; o[2] = v_bound; o[3] = v_epi; o[4] = v_tiles; o[5] = v_b - v_begin; o[6] = (unsigned long long)(9 * nchunks); o[7] = v_begin;
        }
#endif
        if (!more_tiles) return;
    }
}

__global__ void __launch_bounds__(256, 2) conv3x3_g16_kernel(ConvArgs a, unsigned src_bytes) {
    conv3x3_v2_body<4, 1, 64, 10, true>(a, src_bytes);
}
__global__ void __launch_bounds__(256, 2) conv3x3_v2n64_kernel(ConvArgs a, unsigned src_bytes) {
    conv3x3_v2_body<4, 1, 64, 10>(a, src_bytes);
}

// Weights as MFMA fragments.  src: [N][ld] bf16 rows of taps x K channels (forward: [cout][9 * cin]; data gradient: the
// transposed copy [cin][9 * cout]); dst fragment (n / 16, tap, k / 32): 1 KB, lane l = (row l & 15, 8 channels (l >> 4) * 8..)
// at l * 16.  One thread per 16-byte piece; a descriptor table lets one launch pack every layer of a network.
__global__ void __launch_bounds__(256) pack_fragments_kernel(const unsigned short* src_base, const iif_pack_desc* tab, int n_desc,
                                                             unsigned short* dst_base) {
    const int b = blockIdx.x;
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {                                    // last descriptor whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= b) lo = mid; else hi = mid - 1;
    }
    const iif_pack_desc d = tab[lo];
    const int64_t piece = (int64_t)(b - d.block_start) * 256 + threadIdx.x;
    const int kc_n = d.k / 32;
    const int64_t pieces = (int64_t)(d.rows / 16) * d.taps * kc_n * 64;
    if (piece >= pieces) return;
    const int lane = (int)(piece & 63);
    int64_t f = piece >> 6;
    const int kc = (int)(f % kc_n); f /= kc_n;
    const int tap = (int)(f % d.taps);
    const int nt = (int)(f / d.taps);
    const int row = nt * 16 + (lane & 15), k0 = kc * 32 + (lane >> 4) * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src_base + d.src_off + (int64_t)row * d.ld + tap * d.k + k0);
    *reinterpret_cast<u32x4*>(dst_base + d.dst_off + piece * 8) = v;
}

// The grouped 16-channel format (conv3x3_v2_body<..., G16>): src = the block-diagonal chunk matrix of iif_group_pack, rows [r][taps * 64]
// (d.k = 64, d.taps = 9, d.ld = row pitch).  Fragment ((chunk * 4 + ci) * 5 + pair): 1 KB; lane l holds output channel
// chunk * 64 + ci * 16 + (l & 15), K group l >> 4: tap 2 pair + (l >> 5), input channels ci * 16 + ((l >> 4) & 1) * 8 .. + 8 of the chunk
// (zeros for the tap that does not exist).  One thread per 16-byte piece.
__global__ void __launch_bounds__(256) pack_fragments_g16_kernel(const unsigned short* src_base, const iif_pack_desc* tab, int n_desc,
                                                                 unsigned short* dst_base) {
    const int b = blockIdx.x;
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block_start <= b) lo = mid; else hi = mid - 1;
    }
    const iif_pack_desc d = tab[lo];
    const int64_t piece = (int64_t)(b - d.block_start) * 256 + threadIdx.x;
    const int64_t pieces = (int64_t)(d.rows / 64) * 20 * 64;
    if (piece >= pieces) return;
    const int lane = (int)(piece & 63);
    int64_t f = piece >> 6;
    const int pr = (int)(f % 5); f /= 5;
    const int ci = (int)(f % 4);
    const int chunk = (int)(f / 4);
    const int row = chunk * 64 + ci * 16 + (lane & 15);
    const int tap = 2 * pr + (lane >> 5), k0 = ci * 16 + ((lane >> 4) & 1) * 8;
    u32x4 v{0u, 0u, 0u, 0u};
    if (tap < d.taps) v = *reinterpret_cast<const u32x4*>(src_base + d.src_off + (int64_t)row * d.ld + tap * d.k + k0);
    *reinterpret_cast<u32x4*>(dst_base + d.dst_off + piece * 8) = v;
}

template <typename T, bool OUTF32>
__global__ void __launch_bounds__(512) conv_igemm_dma_utap256_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    conv_igemm_dma_body<T, 128, OUTF32, true, 8>(a, src_bytes, wgt_bytes, TapList{a.tap_dy, a.tap_dx, a.tap_w});
}

// ---------------------------------------------------------------- streaming 1x1 GEMM, weights resident in LDS
// dst[M, N] = src[M, K] * wgt[N, K]^T for the bandwidth-bound 1x1 / stride-1 layers (56x56 ... 14x14 stages, K <= 512):
// forward and data gradient.  The tile kernels above spend most of a short-K tile outside the multiply (prologue, one
// exposed load latency, epilogue) and hold at most four tiles per CU; in the step these launches ran at 2.3-3.4 TB/s.  Here:
//   * persistent blocks (one per CU, 12 waves).  A block owns ONE N slice of BN output channels whose weights
//     [BN x K] stay in LDS (<= 64 KB) and walks a sequence of 128-row tiles.  The S = N / BN slices of a tile sequence
//     sit on blocks b, b + 8, ... (one XCD under round-robin placement: speed only), walk the same tiles in the same order
//     and so share each activation tile through that XCD's L2: every activation row leaves HBM once;
//   * 4 COMPUTE waves, each owning 32 rows of the 128-row tile and ALL BN channels of the slice: a wave multiplies only
//     rows its own LDS-DMA fetched, so the K loop has no barrier at all, only counted vmcnt waits on a private ring of R
//     K-slabs (32 rows x 64 B) that runs continuously across tiles: the next tiles' rows are in flight during the
//     epilogue of the current one;
//   * 8 STORE waves drain the previous tile from an LDS staging buffer (16 B per lane, whole lines; residual add,
//     ReLU-bit masks, forward BN statistics or upstream BN-backward sums as in conv_epilogue_staged) while the
//     compute waves multiply the next one.  Two block-wide barriers per tile hand the staging buffer over.
//     The epilogue's OPERANDS (residual, upstream x, ReLU bits: up to 8.1 of the 13 bytes a conv1 data gradient moves per
//     output element) do not depend on the product, so a store thread fetches them for tile j + 1 right after it has
//     drained tile j and holds them in registers across the barriers (round 2 had these loads inside the drain loop:
//     two exposed memory latencies per tile, level with the tile kernel).
// Compute waves issue no vector-memory instruction besides their DMA, so the vmcnt arithmetic is exact; the store
// waves' loads and stores live on their own counters.
constexpr int STREAM_SW = 8;
template <int BN, int KMAX>
__global__ void __launch_bounds__(64 * (4 + STREAM_SW)) gemm1x1_stream_kernel(ConvArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    constexpr int BM = 128, R = 5, CI = BN / 16, SW = STREAM_SW;      // SW store waves next to the 4 compute waves
    constexpr int WBYTES = KMAX * BN * 2;                 // resident weights: K/32 slabs of [BN rows x 64 B]
    constexpr int RING = 4 * R * 2048;                    // per compute wave: R slabs of its 32 rows x 64 B
    constexpr int PITCH = BN * 2 + 16;
    constexpr int CPR = BN / 8, RPP = 64 * SW / CPR;
    constexpr int NR = BM / RPP;                          // rows of a tile per store thread (BN / 32)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(WBYTES + RING + BM * PITCH + SW * 2 * BN * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[WBYTES + RING + BM * PITCH + SW * 2 * BN * 4];
    unsigned char* const wl = smem;
    unsigned char* const ring = smem + WBYTES;
    unsigned char* const stage = ring + RING;
    float* const scratch = reinterpret_cast<float*>(stage + BM * PITCH);          // [SW store waves][2][BN]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = a.Cs / 32;
    // block -> (tile sequence, N slice): the S slices of a sequence are S consecutive blocks OF ONE XCD (b, b + 8, ...)
    const int S = a.ntiles;
    const int xcd = (int)blockIdx.x & 7, bi = (int)blockIdx.x >> 3;
    const int slice = bi % S, seq = (bi / S) * 8 + xcd;
    const int G = (int)gridDim.x / S;                      // tile sequences (the host makes the grid a multiple of 8 S)
    const int n0 = slice * BN;
    const int my_tiles = seq < a.mtiles ? (a.mtiles - seq + G - 1) / G : 0;
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);
    const auto rs_wgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wgt), 0, wgt_bytes, 0x00020000);
    const int prow = lane >> 2;
    const int chunk = (lane & 3) ^ swz(prow);

    // ---- resident weights: slab ks, 16-row piece p -> wl + ks * BN * 64 + p * 1024; all 12 waves fetch
    {
        const int pieces = nk * (BN / 16);
        for (int q = wave; q < pieces; q += 4 + SW) {
            const int ks = q / (BN / 16), p = q - ks * (BN / 16);
            const int n = n0 + p * 16 + prow;
            const unsigned off = n < a.Cd ? ((unsigned)n * (unsigned)a.ldw + (unsigned)(ks * 32 + chunk * 8)) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lds_void*)(wl + q * 1024), 16, off, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (my_tiles <= 0) return;

    if (wave < 4) {
        // ================================================================= compute waves
        const int cw = wave;
        unsigned char* const myring = ring + cw * R * 2048;
        const int fr = lane & 15, fc = lane >> 4;
        const int total = my_tiles * nk;
        // issue state: slab gi = (tile ij, K slab iks)
        int gi = 0, ij = 0, iks = 0;
        unsigned vb[2];
        auto tile_rows = [&](int j) {
            const int m0 = (seq + j * G) * BM + cw * 32;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + i * 16 + prow;
                vb[i] = m < a.M ? ((unsigned)m * (unsigned)a.spitch + (unsigned)(chunk * 8)) * 2u : OOB;
            }
        };
        tile_rows(0);
        auto issue_next = [&]() {
            unsigned char* dstl = myring + (gi % R) * 2048;
            const unsigned soff = (unsigned)iks * 64u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(dstl + i * 1024), 16, vb[i], soff, 0, IIF_CONV_AUX_SRC);
            ++gi;
            if (++iks == nk) { iks = 0; ++ij; if (ij < my_tiles) tile_rows(ij); }
        };
        for (int q = 0; q < R - 1 && gi < total; ++q) issue_next();
        // fragment offsets
        int xo[2], wo[CI];
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) { const int row = pj * 16 + fr; xo[pj] = row * 64 + ((fc ^ swz(row)) << 4); }
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) { const int row = ci * 16 + fr; wo[ci] = row * 64 + ((fc ^ swz(row)) << 4); }
        int g = 0;
#ifdef IIF_CONV_STAMPS
        unsigned long long t_wait = 0, t_mfma = 0, t_ba = 0, t_stage = 0, t_bb = 0, t0s, t1s, t2s, t3s, t4s, t_begin;
        IIF_STAMP(t_begin);
#endif
        for (int j = 0; j < my_tiles; ++j) {
            f32x4 acc[CI][2];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) { acc[ci][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ci][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int ks = 0; ks < nk; ++ks, ++g) {
                // slabs g+1 .. g+R-2 may stay in flight (2 DMA instructions each); near the end fewer were issued
#ifdef IIF_CONV_STAMPS
                IIF_STAMP(t0s);
#endif
                if (gi >= g + R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (R - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef IIF_CONV_STAMPS
                IIF_STAMP(t1s); t_wait += t1s - t0s;
#endif
                const unsigned char* xs = myring + (g % R) * 2048;
                const unsigned char* ws = wl + ks * (BN * 64);
                const u32x4 x0 = *reinterpret_cast<const u32x4*>(xs + xo[0]);
                const u32x4 x1 = *reinterpret_cast<const u32x4*>(xs + xo[1]);
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const u32x4 wf = *reinterpret_cast<const u32x4*>(ws + wo[ci]);
                    acc[ci][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, x0), acc[ci][0], 0, 0, 0);
                    acc[ci][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, x1), acc[ci][1], 0, 0, 0);
                }
                // refill the slot read in the previous step (its fragments were consumed by that step's MFMAs)
                if (gi < total) issue_next();
#ifdef IIF_CONV_STAMPS
                IIF_STAMP(t2s); t_mfma += t2s - t1s;
#endif
            }
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(t2s);
#endif
            __builtin_amdgcn_s_barrier();                 // A: the store waves have drained tile j-1
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(t3s); t_ba += t3s - t2s;
#endif
#pragma unroll
            for (int pj = 0; pj < 2; ++pj)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const int row = cw * 32 + pj * 16 + fr, ch = ci * 16 + fc * 4;
                    u32x2 w;
                    w.x = pack_bf16x2(acc[ci][pj].x, acc[ci][pj].y);
                    w.y = pack_bf16x2(acc[ci][pj].z, acc[ci][pj].w);
                    *reinterpret_cast<u32x2*>(stage + row * PITCH + ch * 2) = w;
                }
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(t4s); t_stage += t4s - t3s;
#endif
            __builtin_amdgcn_s_barrier();                 // B: tile j is staged
#ifdef IIF_CONV_STAMPS
            IIF_STAMP(t0s); t_bb += t0s - t4s;
#endif
        }
        __builtin_amdgcn_s_barrier();                     // C: the store waves' last per-wave sums are in scratch
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(t1s);
        if (g_stamps && blockIdx.x < 256 && lane == 0 && cw == 0) {
            unsigned long long* o = g_stamps + ((int64_t)blockIdx.x * 4 + 0) * 8;
            o[0] = t_wait; o[1] = t_mfma; o[2] = t_ba; o[3] = t_stage; o[4] = t_bb; o[5] = t1s - t_begin; o[6] = (unsigned long long)my_tiles; o[7] = t_begin;
        }
#endif
        return;
    }

    // ===================================================================== store waves
    const int ts = tid - 256, sw = wave - 4;             // store-thread index 0 .. 64*SW-1
    const int chk = ts % CPR, r0 = ts / CPR;
    const int n = n0 + chk * 8;
    const bool stats = a.bn_partial != nullptr;
    const bool col_ok = n < a.Cd;
    float bmean[8], bistd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bmean[q] = 0.f; bistd[q] = 0.f; }
    if (a.bw_x && col_ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { bmean[q] = a.bw_stats[n + q]; bistd[q] = a.bw_stats[a.Cd + n + q]; }
    }
    if (a.aff && col_ok) {                                // two-pass forward, pass 2 (never with bw_x): (a, b) of the BN affine
#pragma unroll
        for (int q = 0; q < 8; ++q) { bmean[q] = a.aff[2 * a.Cd + n + q]; bistd[q] = a.aff[3 * a.Cd + n + q]; }
    }
    // combine the store waves' sums of a tile (already in scratch) and add them to this block's running sums: ONE partial row per
    // tile sequence (round 5; one per tile before: 6 272 rows at 56 x 56 sent every consumer through the two-stage reduction,
    // 18 us instead of 5 on the forward pass's dependent chain).  Tiles are added in sequence order: deterministic.
    float run_s = 0.f, run_q = 0.f;
    auto emit_partial = [&](int) {
        if (ts < BN && n0 + ts < a.Cd) {
            float s2 = 0.f, q2 = 0.f;
#pragma unroll
            for (int w = 0; w < SW; ++w) { s2 += scratch[(w * 2 + 0) * BN + ts]; q2 += scratch[(w * 2 + 1) * BN + ts]; }
            run_s += s2; run_q += q2;
        }
    };
    // epilogue operands of the NEXT tile, fetched ahead: residual rows, upstream-x rows, the two ReLU-bit bytes
    const bool has_res = a.res != nullptr, has_up = a.bw_x != nullptr;
    const bool prefetching = has_res || has_up;           // block-uniform
    u32x4 pres[NR], pupx[NR];
    unsigned pbit[NR];                                    // residual bits | upstream bits << 8
#pragma unroll
    for (int i = 0; i < NR; ++i) { pres[i] = u32x4{0u, 0u, 0u, 0u}; pupx[i] = u32x4{0u, 0u, 0u, 0u}; pbit[i] = 0xffffu; }
    const int64_t ostep = (int64_t)RPP * a.dpitch * 2;
    auto prefetch = [&](int j) {
        const int m0 = (seq + j * G) * BM;
        int64_t o = ((int64_t)(m0 + r0) * a.dpitch + n) * 2;
#pragma unroll
        for (int i = 0; i < NR; ++i, o += ostep) {
            if (m0 + r0 + i * RPP < a.M && col_ok) {
                unsigned rb = 0xffu, ub = 0xffu;
                if (has_res) {
                    pres[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.res + o));     // last use of the residual
                    if (a.res_bits) rb = a.res_bits[o >> 4];
                }
                if (has_up) {
                    pupx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.bw_x + o));    // next reader: BN backward, another XCD
                    if (a.bw_bits) ub = a.bw_bits[o >> 4];
                }
                pbit[i] = rb | (ub << 8);
            }
        }
    };
    if (prefetching) prefetch(0);
#ifdef IIF_CONV_STAMPS
    unsigned long long u_bar = 0, u_drain = 0, u_red = 0, u0, u1, u2, u3, u_begin;
    IIF_STAMP(u_begin);
#endif
    for (int j = 0; j < my_tiles; ++j) {
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(u0);
#endif
        __builtin_amdgcn_s_barrier();                     // A
        if (stats && j > 0) emit_partial(seq + (j - 1) * G);   // idle window: the compute waves stage tile j
        __builtin_amdgcn_s_barrier();                     // B
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(u1); u_bar += u1 - u0;
#endif
        const int mt = seq + j * G;
        const int m0 = mt * BM;
        float bs[8], bq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { bs[q] = 0.f; bq[q] = 0.f; }
        if (col_ok) {
            const unsigned char* sp = stage + r0 * PITCH + chk * 16;
            int64_t o = ((int64_t)(m0 + r0) * a.dpitch + n) * 2;
#pragma unroll
            for (int i = 0; i < NR; ++i, sp += RPP * PITCH, o += ostep) {
                if (m0 + r0 + i * RPP >= a.M) break;                          // only the last tile is short
                u32x4 v = *reinterpret_cast<const u32x4*>(sp);
                if (a.aff) {                              // block-uniform; the arithmetic of bn_apply_kernel
                    const u32x4 rr = pres[i];             // zeros without a residual
                    unsigned bits = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float lo = fmaf(bmean[2 * q], bf16_bits_to_f32(v[q] & 0xffffu), bistd[2 * q]);
                        float hi = fmaf(bmean[2 * q + 1], __uint_as_float(v[q] & 0xffff0000u), bistd[2 * q + 1]);
                        if (has_res) { lo += bf16_bits_to_f32(rr[q] & 0xffffu); hi += __uint_as_float(rr[q] & 0xffff0000u); }
                        bits |= (lo > 0.f ? 1u : 0u) << (2 * q);
                        bits |= (hi > 0.f ? 1u : 0u) << (2 * q + 1);
                        v[q] = pack_bf16x2(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
                    }
                    if (a.relu_out) a.relu_out[o >> 4] = (unsigned char)bits;
                } else if (has_res) {
                    const u32x4 rr = pres[i];
                    const unsigned rb = pbit[i] & 0xffu;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu) + ((rb >> (2 * q)) & 1u ? bf16_bits_to_f32(rr[q] & 0xffffu) : 0.f);
                        const float hi = __uint_as_float(v[q] & 0xffff0000u) + ((rb >> (2 * q + 1)) & 1u ? __uint_as_float(rr[q] & 0xffff0000u) : 0.f);
                        v[q] = pack_bf16x2(lo, hi);
                    }
                }
                if (!a.no_store) {
#ifndef IIF_NO_NT_CONV_STORE     // the tile is next read by another XCD (BN apply): streaming it out keeps the pixel operand's lines in L2 (-0.7 % on the step)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.dst + o));
#else
                    *reinterpret_cast<u32x4*>(a.dst + o) = v;
#endif
                }
                if (has_up) {
                    const u32x4 xv = pupx[i];
                    const unsigned mb = pbit[i] >> 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float glo = (mb >> (2 * q)) & 1u ? bf16_bits_to_f32(v[q] & 0xffffu) : 0.f;
                        const float ghi = (mb >> (2 * q + 1)) & 1u ? __uint_as_float(v[q] & 0xffff0000u) : 0.f;
                        const float xlo = (bf16_bits_to_f32(xv[q] & 0xffffu) - bmean[2 * q]) * bistd[2 * q];
                        const float xhi = (__uint_as_float(xv[q] & 0xffff0000u) - bmean[2 * q + 1]) * bistd[2 * q + 1];
                        bs[2 * q] += glo; bq[2 * q] += glo * xlo;
                        bs[2 * q + 1] += ghi; bq[2 * q + 1] += ghi * xhi;
                    }
                } else if (stats) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                        bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                        bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
                    }
                }
            }
        }
        // the next tile's operands go out behind this tile's stores and fly while the compute waves multiply and stage it
        if (prefetching && j + 1 < my_tiles) prefetch(j + 1);
#ifdef IIF_CONV_STAMPS
        IIF_STAMP(u2); u_drain += u2 - u1;
#endif
        if (stats) {
            // lanes of a wave that share the chunk, then one row of sums per store wave (combined after the next barrier)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int o = CPR; o < 64; o <<= 1) { bs[q] += __shfl_xor(bs[q], o, 64); bq[q] += __shfl_xor(bq[q], o, 64); }
            }
            if (lane < CPR) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    scratch[(sw * 2 + 0) * BN + lane * 8 + q] = bs[q];
                    scratch[(sw * 2 + 1) * BN + lane * 8 + q] = bq[q];
                }
            }
        }
    }
#ifdef IIF_CONV_STAMPS
    IIF_STAMP(u3);
    if (g_stamps && blockIdx.x < 256 && lane == 0 && sw == 0) {
        unsigned long long* o = g_stamps + ((int64_t)blockIdx.x * 4 + 1) * 8;
        o[0] = u_bar; o[1] = u_drain; o[2] = 0; o[3] = 0; o[4] = 0; o[5] = u3 - u_begin; o[6] = (unsigned long long)my_tiles; o[7] = u_begin;
    }
#endif
    __builtin_amdgcn_s_barrier();                         // C
    if (stats) {
        emit_partial(seq + (my_tiles - 1) * G);
        if (ts < BN && n0 + ts < a.Cd) {
            float* p = a.bn_partial + (int64_t)(a.bn_row0 + seq) * 2 * a.dpitch + n0 + ts;
            p[0] = run_s; p[a.dpitch] = run_q;
        }
    }
}

// Test switches of this file (the whole list: DESIGN.md, "Switches"), read from the environment ONCE; iif_conv_reload_env()
// re-reads them (tests flip them between calls).
//   IIF_CONV_REGSTAGE        every launch on the register-staged kernels (the fallback for operands >= 2 GiB)
//   IIF_CONV_NO_STREAM1X1    no launch on the persistent streaming 1x1 kernel;  IIF_CONV_STREAM1X1_FORCE: every shape it has a
//                            plan for, small grids and data gradients included (default: three forward shapes, see use_stream1x1)
//   IIF_CONV_NO_HALO / IIF_CONV_HALO_FORCE   3x3 halo kernel off / also on small grids
//   IIF_CONV_NO_V2 / IIF_CONV_V2_FORCE       3x3 fragment kernel (64 channels) off / also on small grids
struct ConvSwitches {
    bool no_stream, force_stream, regstage, no_v2, no_halo, force_halo, v2_force, no_regw, no_regw_fwdbn;
    static ConvSwitches read() {
        ConvSwitches c;
        c.no_stream = getenv("IIF_CONV_NO_STREAM1X1") != nullptr;
        c.force_stream = getenv("IIF_CONV_STREAM1X1_FORCE") != nullptr;
        c.regstage = getenv("IIF_CONV_REGSTAGE") != nullptr;
        c.no_v2 = getenv("IIF_CONV_NO_V2") != nullptr;
        c.no_halo = getenv("IIF_CONV_NO_HALO") != nullptr;
        c.v2_force = getenv("IIF_CONV_V2_FORCE") != nullptr;
        c.force_halo = getenv("IIF_CONV_HALO_FORCE") != nullptr;
        c.no_regw = getenv("IIF_CONV_NO_REGW") != nullptr;
        c.no_regw_fwdbn = getenv("IIF_CONV_NO_REGW_FWDBN") != nullptr;      // tests: the BN epilogue on the tile kernels
        return c;
    }
};
constexpr int kTwoStageK = 2304;          // the two-stage / 4-blocks-per-CU tile variant up to this K (everything the halo / 256-row kernels leave)
ConvSwitches g_sw = ConvSwitches::read();

// Which launches the streaming kernel takes, and its slice width: (K, N) -> BN columns per block with [BN x K] <= 64 KB
// resident; S = N / BN slices.  A tile sequence needs a few tiles to pipeline across: mtiles * S >= 4 * grid.
struct StreamPlan { int bn, kmax, slices; };
inline bool use_stream1x1(const ConvArgs& a, bool utap, int esz, bool outf32, StreamPlan* pl) {
    const bool force_ = g_sw.force_stream;
    if (g_sw.no_stream || !utap || esz != 2 || outf32 || a.groups > 1 || a.scatter || a.in_shift != 0 || a.sshift != 0) return false;
    if (a.R != 1 || a.S != 1 || a.pad != 0 || a.ntaps != 1 || a.bias || a.src2 || a.sbias || a.mask_store) return false;
    if ((a.no_store || a.aff) && !force_) return false;
    if (a.Hs != a.Hd || a.Ws != a.Wd || (a.Cs % 32) || a.spitch != a.Cs || a.dpitch != a.Cd) return false;
    const int K = a.Cs, N = a.Cd;
    // the shapes whose whole weight matrix is resident (measured alone AND in the step, scripts/bm_stream1x1.py): 64 -> 256,
    // 256 -> 128, 256 -> 64.  N-sliced plans (128 x 128, 64 x 512 columns x K) were level with or behind the tile kernel
    // (5.16 against 5.05 ms per step alone) and were removed in round 4.
    if (K <= 64 && N == 256) *pl = {256, 64, 1};
    else if (K > 64 && K <= 256 && N == 64 && (K >= 128 || force_)) *pl = {64, 256, 1};     // 64->64: the tile kernel wins (0.038 vs 0.060 ms)
    else if (K > 128 && K <= 256 && N % 128 == 0 && N <= 1024 && (N == 128 || force_)) *pl = {128, 256, N / 128};
    else return false;
    if (force_) return true;
    if (a.transposed || a.bw_x) return false;                               // data gradients: level with the tile kernel
    return (int64_t)((a.M + 127) / 128) * pl->slices >= 1024;               // >= 4 tiles per persistent block
}

// 256-pixel tiles: bf16 uniform-tap launches wide enough for the 128-channel tile whose 256-row grid still fills
// the chip (2 blocks per CU resident).
inline bool use_bm256(const ConvArgs& a, bool utap, int esz) {
    if (!utap || esz != 2 || a.Cd <= 64 || a.groups > 1) return false;
    // IIF_CONV_NO_BM256_2SRC=1: the two-source data gradient of the BN3 algebra route on the 34 KB / 4-blocks-per-CU tile instead
    // (it starts beside the weight-gradient blocks, profiles/r5_contention.txt; level in the step: 18.33 against 18.34 ms)
    static const bool no2 = getenv("IIF_CONV_NO_BM256_2SRC") != nullptr;
    if (no2 && a.src2) return false;
    // measured (scripts/bm_ab.sh): +13..39 % on K >= 1024 (3x3 at 128/256 channels, 1x1 from 1024 channels) when the
    // 256-row grid still offers >= 1.5 blocks per CU; short K loops and small grids are better off with 128 rows
    const int64_t tiles = (int64_t)((a.M + 255) / 256) * ((a.Cd + 127) / 128);
    return tiles >= 384 && a.ntaps * a.Cs >= 1024;
}

// partial-sum rows of this launch: [bn_row0, bn_row0 + mtiles) of the caller's buffer
inline int claim_partial_rows(const ConvArgs& a) {
    if (!a.bn_partial) return IIF_OK;
    if ((long long)(a.bn_row0 + a.mtiles) * 2 * a.dpitch > a.bn_cap) return IIF_EINVAL;
    if (a.rows_out) *a.rows_out = a.bn_row0 + a.mtiles;
    return IIF_OK;
}

// halo kernel: bf16 3x3 / stride 1 / pad 1, dense, >= 128 output channels, window of a 256-pixel tile <= 512 halo rows
inline bool use_halo(const ConvArgs& a, bool utap, int esz, bool outf32) {
    const bool off = g_sw.no_halo;
    if (off || !utap || esz != 2 || outf32 || a.groups > 1 || a.scatter || a.in_shift != 0) return false;
    if (a.ntaps != 9 || a.R != 3 || a.S != 3 || a.pad != 1 || a.Hs != a.Hd || a.Ws != a.Wd) return false;
    if (a.Cd < 128 || (a.Cd & 7) || a.bias || (a.Cs % 32)) return false;
    const int HW = a.Hd * a.Wd;
    const int span = (256 + a.Wd - 1) / a.Wd + 1 + 2 * (256 / HW + 1);       // virtual rows a tile can touch
    if ((span + 2) * (a.Wd + 2) > 512) return false;
    const int span1 = (128 + a.Wd - 1) / a.Wd + 1 + 2 * (128 / HW + 1);      // the same for a 128-pixel tile
    if ((span1 + 2) * (a.Wd + 2) > 304) return false;
    const bool force = g_sw.force_halo;                                       // tests: small grids too
    return force || (int64_t)((a.M + 255) / 256) * ((a.Cd + 127) / 128) >= 192;
}

// generation-2 3x3: fragment-packed weights supplied, bf16, stride 1 / pad 1, dense, channels in 32s / 64s, halo of a
// 256-pixel tile within 640 rows
inline bool v2_geometry_ok(int N, int H, int W, int Cs, int Cd, int groups = 1) {
    if ((Cs % 32) || (Cd % 64) || H <= 0 || W <= 0 || W > 1022) return false;
    // Measured alone (scripts/bm_conv3x3.py, bs 256): the 64-channel variant (56x56) 0.132 -> 0.102 ms forward, 0.171 -> 0.134
    // data gradient; the persistent 128-channel variant is level with or behind the halo kernel (28x28 0.078 -> 0.098,
    // 14x14 0.067 -> 0.076, 7x7 0.057 -> 0.066 ms) and was removed in round 4: layers with 128-channel multiples take the halo kernel.
    if ((Cd % 128) == 0) return false;
    // 256-pixel tiles: a small layer (CIFAR-size images) would leave most CUs without a block (IIF_CONV_V2_FORCE: tests)
    const int HW = H * W;
    if (!g_sw.v2_force && ((int64_t)N * HW + 255) / 256 * (Cd / 64) * groups < 192) return false;
    if ((int64_t)N * H * W >= (1 << 22)) return false;                    // fdiv's exact range
    const int span = (256 + W - 1) / W + 1 + 2 * (256 / HW + 1);            // virtual rows a 256-pixel tile can touch
    return (span + 2) * (W + 2) <= 640;
}
inline bool use_v2(const ConvArgs& a, bool utap, int esz, bool outf32) {
    if (!a.wfrag || g_sw.no_v2 || !utap || esz != 2 || outf32 || a.scatter || a.in_shift != 0) return false;
    if (a.ntaps != 9 || a.R != 3 || a.S != 3 || a.pad != 1 || a.Hs != a.Hd || a.Ws != a.Wd || a.bias) return false;
    if (a.groups > 1) {                 // grouped: 64-channel chunks on the 64-channel variant, blockIdx.y = chunk
        if (a.Cs != 64 || a.Cd != 64 || a.groups > 65535) return false;
    } else if (a.spitch != a.Cs || a.dpitch != a.Cd) {
        return false;
    }
    return v2_geometry_ok(a.N, a.Hd, a.Wd, a.Cs, a.Cd, a.groups);
}

template <typename T, bool OUTF32>
int launch_one(ConvArgs a, bool utap, int64_t src_bytes, int64_t wgt_bytes, hipStream_t st) {
    const bool force_v1_ = g_sw.regstage;
    if constexpr (sizeof(T) == 2 && !OUTF32) {          // 3x3 with the weights in registers (conv_regw.hip): 64 -> 64 channels
        if (!force_v1_ && !g_sw.no_regw && utap && src_bytes < 0x7f000000LL && a.ntaps == 9 && a.R == 3 && a.S == 3 && a.pad == 1 &&
            a.Hs == a.Hd && a.Ws == a.Wd && a.groups == 1 && a.Cs == a.Cd && a.spitch == a.Cs && a.dpitch == a.Cd && !a.scatter &&
            a.in_shift == 0 && !a.res && !a.res_bits && !a.bias && !a.mask_store && !a.src2 && !a.sbias && !a.no_store && !a.aff &&
            a.Cs == 64 && (!(a.bw_x || a.bw_bits) || a.bn_partial) && iif_regw3x3_ok(a.N, a.Hd, a.Wd, a.Cs)) {
            const int rc = iif_regw3x3_launch(a.src, a.wgt, a.dst, a.bn_partial, a.bn_cap, a.bn_row0, a.rows_out, a.N, a.Hd, a.Wd, a.Cs, a.ldw,
                                              a.tap_dy, a.tap_dx, a.tap_w, a.bw_x, a.bw_bits, a.bw_stats, st);
            if (rc != IIF_EUNSUPPORTED) return rc;
        }
    }
    if (!force_v1_ && src_bytes < 0x7f000000LL && use_v2(a, utap, (int)sizeof(T), OUTF32)) {
        a.mtiles = (a.M + 255) / 256;
        a.ntiles = a.Cd / 64;
        if (const int rc = claim_partial_rows(a)) return rc;
        const int64_t blocks2 = (int64_t)((a.mtiles + 7) / 8) * 8 * a.ntiles;
        if (blocks2 > 0x7fffffff) return IIF_EUNSUPPORTED;
        if (a.wfrag_kind == 1) {
            if (a.Cs != 64 || a.Cd != 64) return IIF_EINVAL;       // (the 16-channel format describes 64-channel chunks)
            hipLaunchKernelGGL(conv3x3_g16_kernel, dim3((unsigned)blocks2, (unsigned)a.groups), dim3(256), 0, st, a, (unsigned)src_bytes);
            IIF_LAUNCH_CHECK();
            return IIF_OK;
        }
        hipLaunchKernelGGL(conv3x3_v2n64_kernel, dim3((unsigned)blocks2, (unsigned)a.groups), dim3(256), 0, st, a, (unsigned)src_bytes);
        IIF_LAUNCH_CHECK();
        return IIF_OK;
    }
    if (!force_v1_ && src_bytes < 0x7f000000LL && wgt_bytes < 0x7f000000LL && use_halo(a, utap, (int)sizeof(T), OUTF32)) {
        // measured (scripts/halo_ab.sh): two 128-row blocks per CU win at 28x28 (+8 %), one 256-row block elsewhere
        const int bm = a.Wd >= 28 ? 128 : 256;
        a.mtiles = (a.M + bm - 1) / bm;
        a.ntiles = (a.Cd + 127) / 128;
        if (const int rc = claim_partial_rows(a)) return rc;
        const int64_t blocksh = (int64_t)((a.mtiles + 7) / 8) * 8 * a.ntiles;
        if (blocksh > 0x7fffffff) return IIF_EUNSUPPORTED;
        if (bm == 128) hipLaunchKernelGGL(conv3x3_halo128_kernel, dim3((unsigned)blocksh), dim3(256), 0, st, a, (unsigned)src_bytes,
                                          (unsigned)wgt_bytes);
        else hipLaunchKernelGGL(conv3x3_halo_kernel, dim3((unsigned)blocksh), dim3(512), 0, st, a, (unsigned)src_bytes,
                                (unsigned)wgt_bytes);
        IIF_LAUNCH_CHECK();
        return IIF_OK;
    }
    StreamPlan sp{0, 0, 0};
    if (!force_v1_ && src_bytes < 0x7f000000LL && wgt_bytes < 0x7f000000LL && use_stream1x1(a, utap, (int)sizeof(T), OUTF32, &sp)) {
        a.mtiles = (a.M + 127) / 128;
        a.ntiles = sp.slices;
        // one block per CU, in whole groups of 8 S (the S slices of a tile sequence on one XCD); a short layer takes fewer
        // sequences, never more than it has tiles
        const int unit = 8 * sp.slices;
        int grid = iif_persistent_grid(unit);
        const int need = (a.mtiles + 7) / 8 * unit;         // sequences in multiples of 8, S blocks each
        if (need < grid) grid = need;
        if (grid >= unit) {
            if (a.bn_partial) {                              // one row per tile sequence that has tiles (sequences 0 .. rows - 1)
                const int seqs = grid / sp.slices, rows = seqs < a.mtiles ? seqs : a.mtiles;
                if ((long long)(a.bn_row0 + rows) * 2 * a.dpitch > a.bn_cap) return IIF_EINVAL;
                if (a.rows_out) *a.rows_out = a.bn_row0 + rows;
            }
            const unsigned sb = (unsigned)src_bytes, wb = (unsigned)wgt_bytes;
            const dim3 g((unsigned)grid), blk(64 * (4 + STREAM_SW));
            if (sp.bn == 256) hipLaunchKernelGGL((gemm1x1_stream_kernel<256, 64>), g, blk, 0, st, a, sb, wb);
            else if (sp.bn == 128) hipLaunchKernelGGL((gemm1x1_stream_kernel<128, 256>), g, blk, 0, st, a, sb, wb);
            else hipLaunchKernelGGL((gemm1x1_stream_kernel<64, 256>), g, blk, 0, st, a, sb, wb);
            IIF_LAUNCH_CHECK();
            return IIF_OK;
        }
    }
    if (!force_v1_ && src_bytes < 0x7f000000LL && wgt_bytes < 0x7f000000LL && use_bm256(a, utap, (int)sizeof(T))) {
        a.mtiles = (a.M + 255) / 256;
        a.ntiles = (a.Cd + 127) / 128;
        if (const int rc = claim_partial_rows(a)) return rc;
        const int64_t blocks256 = (int64_t)((a.mtiles + 7) / 8) * 8 * a.ntiles;
        if (blocks256 > 0x7fffffff) return IIF_EUNSUPPORTED;
        hipLaunchKernelGGL((conv_igemm_dma_utap256_kernel<T, OUTF32>), dim3((unsigned)blocks256, (unsigned)a.groups), dim3(512), 0,
                           st, a, (unsigned)src_bytes, (unsigned)wgt_bytes);
        IIF_LAUNCH_CHECK();
        return IIF_OK;
    }
    a.mtiles = (a.M + 127) / 128;
    const bool narrow = a.Cd <= 64;
    const int bn = narrow ? 64 : 128;
    a.ntiles = (a.Cd + bn - 1) / bn;
    const int64_t blocks = (int64_t)((a.mtiles + 7) / 8) * 8 * a.ntiles;
    if (blocks > 0x7fffffff) return IIF_EUNSUPPORTED;
    const bool force_v1 = g_sw.regstage;
    // LDS-DMA addressing is a 32-bit byte offset with a hardware range check: both operands must be < 2 GiB
    const bool dma = !force_v1 && src_bytes < 0x7f000000LL && wgt_bytes < 0x7f000000LL;
    if (a.bn_partial && !dma) return IIF_EUNSUPPORTED;     // partial sums come out of the staged epilogue only
    if (const int rc = claim_partial_rows(a)) return rc;
    if (a.groups > 1 && !dma) return IIF_EUNSUPPORTED;     // grouped convolutions exist on the pipelined kernels only
    if (a.scatter && !(dma && utap)) return IIF_EUNSUPPORTED;
    const dim3 grid((unsigned)blocks, (unsigned)a.groups), blk(256);
    if (dma) {
        const unsigned sb = (unsigned)src_bytes, wb = (unsigned)wgt_bytes;
        if (utap) {
            // two-stage / 4-blocks-per-CU variant up to K = 2304
            // (round 5, same-call A/B: the long-K launches of small grids - the 7 x 7 stage's 392-block data gradients with 64 K steps -
            // on the three-stage kernel instead, one barrier per step and two steps of prefetch: 18.40 / 18.43 / 18.41 ms per step
            // against 18.39 / 18.42 / 18.40; for every grid of <= 1 600 blocks 18.42 / 18.42 / 18.43.  Level: knob removed.)
            const bool shortk = sizeof(T) == 2 && !OUTF32 && a.ntaps * a.Cs <= kTwoStageK && (a.Cd & 7) == 0 && !a.bias;
            if constexpr (sizeof(T) == 2 && !OUTF32) {
                if (shortk) {
                    if (narrow) hipLaunchKernelGGL((conv_igemm_dma_utap_k64_kernel<64>), grid, blk, 0, st, a, sb, wb);
                    else hipLaunchKernelGGL((conv_igemm_dma_utap_k64_kernel<128>), grid, blk, 0, st, a, sb, wb);
                    IIF_LAUNCH_CHECK();
                    return IIF_OK;
                }
            }
            if (narrow) hipLaunchKernelGGL((conv_igemm_dma_utap_kernel<T, 64, OUTF32>), grid, blk, 0, st, a, sb, wb);
            else hipLaunchKernelGGL((conv_igemm_dma_utap_kernel<T, 128, OUTF32>), grid, blk, 0, st, a, sb, wb);
        } else {
            if (narrow) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 64, OUTF32>), grid, blk, 0, st, a, sb, wb);
            else hipLaunchKernelGGL((conv_igemm_dma_kernel<T, 128, OUTF32>), grid, blk, 0, st, a, sb, wb);
        }
    } else {
        if (narrow) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, OUTF32>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<T, 128, OUTF32>), grid, blk, 0, st, a);
    }
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// Host-side planning: which addressing path, and the tap lists of the uniform-tap path.
//   forward:              source (y*stride + r - pad, x*stride + s - pad)
//   data gradient, s=1:   source (y + pad - r, x + pad - s)
//   data gradient, s=2:   one dense stride-1 sub-convolution per output parity class (py, px): only the taps
//                         with (py + pad - r) even reach that class (1, 2, 2 and 4 of the 9 taps of a 3x3),
//                         source (yy + (py + pad - r)/2, ...), destination scattered to (2*yy + py, 2*xx + px).
template <typename T, bool OUTF32>
int launch_conv(const ConvArgs& a0, int64_t src_bytes, int64_t wgt_bytes, hipStream_t st) {
    ConvArgs a = a0;
    const bool dma_ok = !g_sw.regstage && src_bytes < 0x7f000000LL && wgt_bytes < 0x7f000000LL;
    const bool utap = dma_ok && (a.Cs % ET<T>::KE) == 0 && a.R * a.S <= 16 && a.R <= 16 && a.S <= 16;
    a.scatter = 0; a.ds_shift = 0; a.doy = a.dox = 0; a.Hfull = a.Hd; a.Wfull = a.Wd; a.ntaps = 0; a.in_shift = 0;
    if constexpr (sizeof(T) == 2 && !OUTF32) {      // the stem in its space-to-depth form has a kernel of its own (conv_stem.hip)
        if (dma_ok && !a.transposed && a.sshift == 0 && a.R == 4 && a.S == 4 && a.pad == 2 && a.Cs == 16 && a.spitch == 16 &&
            a.Cd == 64 && a.dpitch == 64 && a.groups == 1 && a.ldw == 256 && a.Hs == a.Hd && a.Ws == a.Wd && !a.res && !a.bias && !a.bw_x &&
            !a.src2 && !a.sbias && !a.mask_store && !a.no_store && !a.aff && iif_stem4x4_ok(a.N, a.Hd, a.Wd))
        {
            const int rc = iif_stem4x4_launch(a.src, a.wgt, a.dst, a.bn_partial, a.bn_cap, a.bn_row0, a.rows_out, a.N, a.Hd, a.Wd, st);
            if (rc != IIF_EUNSUPPORTED) return rc;
        }
        // the two passes of the never-stored forward (conv_regw.hip: statistics from the accumulators / BN epilogue)
        const bool geo1x1 = dma_ok && !g_sw.no_regw && a.R == 1 && a.S == 1 && a.sshift == 0 && a.pad == 0 && a.groups == 1 && a.spitch == a.Cs &&
                            a.dpitch == a.Cd && a.Hs == a.Hd && a.Ws == a.Wd && !a.bias && !a.src2 && !a.sbias && !a.transposed;
        const bool geo1x1x = dma_ok && !g_sw.no_regw && a.R == 1 && a.S == 1 && a.sshift == 0 && a.pad == 0 && a.groups == 1 && a.spitch == a.Cs &&
                             a.dpitch == a.Cd && a.Hs == a.Hd && a.Ws == a.Wd && !a.bias && !a.src2 && !a.sbias;
        const iif_regw_prologue pro{a.pro_stats, a.pro_out, a.pro_bits, a.pro_csum};
        if (a.no_store == 2) {
            if (!geo1x1 || !a.bn_partial || a.res || a.bw_x || a.mask_store || a.aff) return IIF_EUNSUPPORTED;
            return iif_regw1x1_stats_launch(a.src, a.wgt, a.bn_partial, a.bn_cap, a.bn_row0, a.rows_out, a.M, a.Cs, a.Cd, a.spitch, a.ldw, a.dpitch, st,
                                            a.pro_stats ? &pro : nullptr);
        }
        if (a.pro_stats) {                                    // plain forward + statistics with the prologue: no other kernel has it
            if (!geo1x1 || !a.bn_partial || a.res || a.bw_x || a.mask_store || a.aff || a.no_store || !iif_regw1x1_pro_ok(a.M, a.Cs, a.Cd)) return IIF_EUNSUPPORTED;
            return iif_regw1x1_launch(a.src, a.wgt, a.dst, a.bn_partial, a.bn_cap, a.bn_row0, a.rows_out, a.M, a.Cs, a.Cd, a.spitch, a.ldw, a.dpitch,
                                      nullptr, 0, st, &pro);
        }
        if (a.aff && geo1x1 && a.relu_out && !a.bn_partial && !a.res_bits && !a.bw_x && !a.mask_store && !g_sw.no_regw_fwdbn &&
            iif_regw1x1_fwdbn_ok(a.M, a.Cs, a.Cd)) {
            const int rc = iif_regw1x1_fwdbn_launch(a.src, a.wgt, a.dst, a.M, a.Cs, a.Cd, a.spitch, a.ldw, a.dpitch, a.res, a.aff, a.aff2, a.relu_out, st);
            if (rc != IIF_EUNSUPPORTED) return rc;
        }
        if (a.aff2) return IIF_EUNSUPPORTED;                  // (the tile kernels' BN epilogue takes a plain residual only)
        if (a.rx_src2 && !(geo1x1x && iif_regw1x1_rx_ok(a.M, a.Cs, a.Cd, a.rx_k2))) return IIF_EUNSUPPORTED;
        // narrow -> wide 1x1 layers: weights in registers (conv_regw.hip)
        const bool epi = a.res || a.res_bits || a.bw_x || a.bw_bits || a.mask_store;
        if (dma_ok && !g_sw.no_regw && a.R == 1 && a.S == 1 && a.sshift == 0 && a.pad == 0 && a.groups == 1 && a.spitch == a.Cs && a.dpitch == a.Cd &&
            a.Hs == a.Hd && a.Ws == a.Wd && !a.bias && !a.src2 && !a.sbias && !a.aff && (!epi || (a.bn_partial && !a.no_store)) &&
            iif_regw1x1_ok(a.M, a.Cs, a.Cd, epi))
        {
            const iif_regw_epilogue e{a.res, a.res_bits, a.bw_x, a.bw_bits, a.bw_stats, a.mask_store, a.rx_src2, a.rx_w3, a.rx_k2, a.rx_ldw3,
                                      a.pg_slab, a.pg_cap, a.pg_ld, a.pg_count};
            const int rc = iif_regw1x1_launch(a.src, a.wgt, a.dst, a.bn_partial, a.bn_cap, a.bn_row0, a.rows_out, a.M, a.Cs, a.Cd, a.spitch,
                                              a.ldw, a.dpitch, epi ? &e : nullptr, a.no_store, st);
            if (rc != IIF_EUNSUPPORTED) return rc;
        }
        if (a.rx_src2) return IIF_EUNSUPPORTED;               // (no other kernel recomputes the upstream x)
    }
    if (a.rx_src2 || a.pro_stats) return IIF_EUNSUPPORTED;
    if (!utap) return (a.src2 || a.sbias || a.mask_store || a.no_store || a.aff) ? IIF_EUNSUPPORTED : launch_one<T, OUTF32>(a, false, src_bytes, wgt_bytes, st);
    if (!a.transposed || a.sshift == 0) {
        for (int r = 0; r < a.R; ++r)
            for (int s = 0; s < a.S; ++s) {
                const int t = a.ntaps++;
                a.tap_dy[t] = (signed char)(a.transposed ? a.pad - r : r - a.pad);
                a.tap_dx[t] = (signed char)(a.transposed ? a.pad - s : s - a.pad);
                a.tap_w[t] = (unsigned char)(r * a.S + s);
            }
        a.in_shift = a.transposed ? 0 : a.sshift;
        if (a.src2) {                                        // K continues over the second source as "tap 1" of a 1x1 launch
            if (a.ntaps != 1 || (a.Cs2 % ET<T>::KE) || a.sshift != 0 || a.groups > 1) return IIF_EUNSUPPORTED;
            a.tap_dy[1] = 0; a.tap_dx[1] = 0; a.tap_w[1] = 1; a.ntaps = 2;
        }
        return launch_one<T, OUTF32>(a, true, src_bytes, wgt_bytes, st);
    }
    // stride-2 data gradient: 4 parity classes of the destination grid
    const int H = a.Hd, W = a.Wd;
    ConvArgs cls[4];
    int ncls = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvArgs c = a;
            c.Hd = (H - py + 1) / 2; c.Wd = (W - px + 1) / 2;
            if (c.Hd <= 0 || c.Wd <= 0) continue;
            c.M = a.N * c.Hd * c.Wd;
            c.scatter = 1; c.ds_shift = 1; c.doy = py; c.dox = px; c.Hfull = H; c.Wfull = W;
            c.ntaps = 0;
            for (int r = 0; r < a.R; ++r) {
                if ((py + a.pad - r) & 1) continue;
                for (int s = 0; s < a.S; ++s) {
                    if ((px + a.pad - s) & 1) continue;
                    const int t = c.ntaps++;
                    c.tap_dy[t] = (signed char)((py + a.pad - r) / 2);      // exact: the numerator is even
                    c.tap_dx[t] = (signed char)((px + a.pad - s) / 2);
                    c.tap_w[t] = (unsigned char)(r * a.S + s);
                }
            }
            // a class without taps receives no contribution (dst += 0): skipped, unless the backward sums of the
            // upstream unit ride on this launch (every pixel of dst has to be visited once)
            if (c.ntaps == 0 && c.res == c.dst && !c.bw_x) continue;
            cls[ncls++] = c;
        }
    // all classes in one launch where each of them would take the 4-blocks-per-CU tile kernel
    if constexpr (sizeof(T) == 2 && !OUTF32) {
        const bool shortk = a.R * a.S * a.Cs <= kTwoStageK && (a.Cd & 7) == 0 && !a.bias && a.groups == 1 && ncls > 1 && !g_sw.regstage;
        if (shortk) {
            ConvArgsMC p{};
            p.a = a;
            p.a.scatter = 1; p.a.ds_shift = 1; p.a.Hfull = H; p.a.Wfull = W; p.a.in_shift = 0;
            const bool narrow = a.Cd <= 64;
            const int bn = narrow ? 64 : 128;
            p.a.ntiles = (a.Cd + bn - 1) / bn;
            p.ncls = ncls;
            int64_t bstart = 0;
            int row0 = a.rows_out ? *a.rows_out : a.bn_row0;
            for (int i = 0; i < ncls; ++i) {
                ConvClass& k = p.cls[i];
                k.Hd = cls[i].Hd; k.Wd = cls[i].Wd; k.M = cls[i].M; k.mtiles = (cls[i].M + 127) / 128;
                k.doy = cls[i].doy; k.dox = cls[i].dox; k.ntaps = cls[i].ntaps; k.bn_row0 = row0; k.bstart = (int)bstart;
                for (int t = 0; t < 16; ++t) { k.tap_dy[t] = cls[i].tap_dy[t]; k.tap_dx[t] = cls[i].tap_dx[t]; k.tap_w[t] = cls[i].tap_w[t]; }
                bstart += (int64_t)((k.mtiles + 7) / 8) * 8 * p.a.ntiles;
                row0 += k.mtiles;
            }
            if (bstart <= 0x7fffffff) {
                if (a.bn_partial) {
                    if ((long long)row0 * 2 * a.dpitch > a.bn_cap) return IIF_EINVAL;
                    if (a.rows_out) *a.rows_out = row0;
                }
                const dim3 grid((unsigned)bstart), blk(256);
                if (narrow) hipLaunchKernelGGL(conv_igemm_dma_utap_k64_mc64_kernel, grid, blk, 0, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes);
                else hipLaunchKernelGGL(conv_igemm_dma_utap_k64_mc128_kernel, grid, blk, 0, st, p, (unsigned)src_bytes, (unsigned)wgt_bytes);
                IIF_LAUNCH_CHECK();
                return IIF_OK;
            }
        }
    }
    for (int i = 0; i < ncls; ++i) {
        ConvArgs& c = cls[i];
        if (c.rows_out) c.bn_row0 = *c.rows_out;
        const int rc = launch_one<T, OUTF32>(c, true, src_bytes, wgt_bytes, st);
        if (rc != IIF_OK) return rc;
    }
    return IIF_OK;
}

}  // namespace

namespace {
struct ConvExtra { int mask_store; const void* src2; int cs2; const float* sbias; int no_store; const float* aff; unsigned char* relu_out; const float* aff2;
                   const void* rx_src2; const void* rx_w3; int rx_k2, rx_ldw3;
                   const float* pro_stats; void* pro_out; unsigned char* pro_bits; float* pro_csum;
                   float* pg_slab; long long pg_cap; int pg_ld; int* pg_count; };
int conv_entry(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
               const unsigned char* res_bits, const float* bias, float* bn_partial, int64_t bn_partial_floats,
               int32_t* n_partials, void* stream, const void* bw_x = nullptr, const unsigned char* bw_bits = nullptr,
               const float* bw_stats = nullptr, const ConvExtra* ex = nullptr);
}

extern "C" int iif_conv_pack_fragments(const void* src_base, const iif_pack_desc* table, int n_desc, int total_blocks,
                                       void* dst_base, void* stream) {
    if (!src_base || !table || !dst_base || n_desc <= 0 || total_blocks <= 0) return IIF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src_base) | reinterpret_cast<uintptr_t>(dst_base)) & 15) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(pack_fragments_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                       (const unsigned short*)src_base, table, n_desc, (unsigned short*)dst_base);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_conv_pack_fragments_g16(const void* src_base, const iif_pack_desc* table, int n_desc, int total_blocks,
                                           void* dst_base, void* stream) {
    if (!src_base || !table || !dst_base || n_desc <= 0 || total_blocks <= 0) return IIF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src_base) | reinterpret_cast<uintptr_t>(dst_base)) & 15) return IIF_EUNSUPPORTED;
    hipLaunchKernelGGL(pack_fragments_g16_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                       (const unsigned short*)src_base, table, n_desc, (unsigned short*)dst_base);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

extern "C" int iif_conv3x3_frag_ok(const iif_conv_desc* d) {
    if (!d || g_sw.no_v2 || g_sw.regstage) return 0;
    if (d->dtype != IIF_BF16 || d->dst_dtype != IIF_BF16 || d->r != 3 || d->s != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->hs != d->hd || d->ws != d->wd) return 0;
    const int g = d->groups > 1 ? d->groups : 1;
    if (g > 1 && (d->cs != 64 || d->cd != 64)) return 0;
    if ((int64_t)d->n * d->hs * d->ws * d->cs * g * 2 >= 0x7f000000LL) return 0;
    return v2_geometry_ok(d->n, d->hd, d->wd, d->cs, d->cd, g) ? 1 : 0;
}

extern "C" int iif_conv_reload_env(void) {
    g_sw = ConvSwitches::read();
    return IIF_OK;
}

extern "C" int iif_conv_igemm_dgrad_bnbwd(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                          const unsigned char* res_bits, const void* up_x, const unsigned char* up_bits,
                                          const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                                          void* stream) {
    if (!d || !up_x || !up_stats || !partial || !n_partials || !d->transposed) return IIF_EINVAL;
    if (res_bits && (!res || d->stride != 1)) return IIF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(up_x) & 15)) return IIF_EUNSUPPORTED;
    return conv_entry(d, src, wgt, dst, res, res_bits, nullptr, partial, partial_floats, n_partials, stream, up_x, up_bits, up_stats);
}

extern "C" int iif_conv_igemm_dgrad_masksum(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                            const unsigned char* res_bits, const void* up_x, const unsigned char* up_bits,
                                            const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                                            void* stream) {
    if (!d || !up_bits || !partial || !n_partials || !d->transposed) return IIF_EINVAL;
    if (res_bits && !res) return IIF_EINVAL;
    if (up_x && !up_stats) return IIF_EINVAL;
    const ConvExtra ex{1, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, dst, res, res_bits, nullptr, partial, partial_floats, n_partials, stream, up_x, up_bits,
                      up_x ? up_stats : nullptr, &ex);
}

extern "C" int iif_conv_dgrad_rx_ok(const iif_conv_desc* d, int c2) {
    if (!d || g_sw.no_regw || g_sw.regstage || !d->transposed) return 0;
    if (d->dtype != IIF_BF16 || d->dst_dtype != IIF_BF16 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0 || d->groups > 1) return 0;
    if (d->hs != d->hd || d->ws != d->wd) return 0;
    return iif_regw1x1_rx_ok(d->n * d->hd * d->wd, d->cs, d->cd, c2) ? 1 : 0;
}

extern "C" int iif_conv_igemm_dgrad_masksum_rx(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                               const unsigned char* res_bits, const void* up_a2, int up_c2, const void* up_w3, int up_ldw3,
                                               const unsigned char* up_bits, const float* up_stats, float* partial, int64_t partial_floats,
                                               int32_t* n_partials, void* stream) {
    if (!d || !up_bits || !partial || !n_partials || !d->transposed || !up_a2 || !up_w3 || !up_stats) return IIF_EINVAL;
    if (res_bits && !res) return IIF_EINVAL;
    if (!iif_conv_dgrad_rx_ok(d, up_c2) || ((reinterpret_cast<uintptr_t>(up_a2) | reinterpret_cast<uintptr_t>(up_w3)) & 15)) return IIF_EUNSUPPORTED;
    const ConvExtra ex{1, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, up_a2, up_w3, up_c2, up_ldw3, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, dst, res, res_bits, nullptr, partial, partial_floats, n_partials, stream, nullptr, up_bits, up_stats, &ex);
}

extern "C" int iif_conv_dgrad_rx_pg_ok(const iif_conv_desc* d, int c2) {
    if (!iif_conv_dgrad_rx_ok(d, c2)) return 0;
    return iif_regw1x1_pg_ok(d->n * d->hd * d->wd, d->cs, d->cd, c2) ? 1 : 0;
}

extern "C" int iif_conv_igemm_dgrad_masksum_rx_pg(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                                  const unsigned char* res_bits, const void* up_a2, int up_c2, const void* up_w3, int up_ldw3,
                                                  const unsigned char* up_bits, const float* up_stats, float* partial, int64_t partial_floats,
                                                  int32_t* n_partials, float* pg_slabs, int64_t pg_floats, int pg_ld, int32_t* n_slabs,
                                                  void* stream) {
    if (!d || !up_bits || !partial || !n_partials || !d->transposed || !up_a2 || !up_w3 || !up_stats || !pg_slabs || !n_slabs) return IIF_EINVAL;
    if (res_bits && !res) return IIF_EINVAL;
    if (!iif_conv_dgrad_rx_pg_ok(d, up_c2) || ((reinterpret_cast<uintptr_t>(up_a2) | reinterpret_cast<uintptr_t>(up_w3)) & 15)) return IIF_EUNSUPPORTED;
    int count = 0;
    const ConvExtra ex{1, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, up_a2, up_w3, up_c2, up_ldw3, nullptr, nullptr, nullptr, nullptr,
                       pg_slabs, (long long)pg_floats, pg_ld, &count};
    const int rc = conv_entry(d, src, wgt, dst, res, res_bits, nullptr, partial, partial_floats, n_partials, stream, nullptr, up_bits, up_stats, &ex);
    *n_slabs = count;
    return rc;
}

extern "C" int iif_conv_igemm_dgrad2_bnbwd(const iif_conv_desc* d, const void* src, const void* src2, int cs2, const void* wgt,
                                           const float* bias, void* dst, const void* up_x, const unsigned char* up_bits,
                                           const float* up_stats, float* partial, int64_t partial_floats, int32_t* n_partials,
                                           void* stream) {
    if (!d || !src2 || !d->transposed) return IIF_EINVAL;
    if (up_x && (!up_stats || !partial || !n_partials)) return IIF_EINVAL;
    const ConvExtra ex{0, src2, cs2, bias, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, dst, nullptr, nullptr, nullptr, up_x ? partial : nullptr, partial_floats, n_partials, stream, up_x,
                      up_bits, up_stats, &ex);
}

extern "C" int iif_conv_igemm_stats_only(const iif_conv_desc* d, const void* src, const void* wgt, float* bn_partial,
                                         int64_t bn_partial_floats, int32_t* n_partials, void* stream) {
    if (!d || !bn_partial || !n_partials || d->transposed) return IIF_EINVAL;
    const ConvExtra ex{0, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    // dst is never written; the source pointer stands in for the non-null / alignment checks
    return conv_entry(d, src, wgt, const_cast<void*>(src), nullptr, nullptr, nullptr, bn_partial, bn_partial_floats, n_partials, stream,
                      nullptr, nullptr, nullptr, &ex);
}

extern "C" int iif_conv_igemm_bn_relu(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                      const float* stats, unsigned char* relu_bits, void* stream) {
    if (!d || !stats || d->transposed) return IIF_EINVAL;
    const ConvExtra ex{0, nullptr, 0, nullptr, 0, stats, relu_bits, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, dst, res, nullptr, nullptr, nullptr, 0, nullptr, stream, nullptr, nullptr, nullptr, &ex);
}

extern "C" int iif_conv_fwdbn_ok(const iif_conv_desc* d) {
    if (!d || g_sw.no_regw || g_sw.regstage || g_sw.no_regw_fwdbn) return 0;
    if (d->dtype != IIF_BF16 || d->dst_dtype != IIF_BF16 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0 || d->groups > 1 || d->transposed) return 0;
    if (d->hs != d->hd || d->ws != d->wd) return 0;
    return iif_regw1x1_fwdbn_ok(d->n * d->hd * d->wd, d->cs, d->cd) ? 1 : 0;
}

extern "C" int iif_conv_igemm_stats_acc(const iif_conv_desc* d, const void* src, const void* wgt, float* bn_partial,
                                        int64_t bn_partial_floats, int32_t* n_partials, void* stream) {
    if (!d || !bn_partial || !n_partials || d->transposed) return IIF_EINVAL;
    if (!iif_conv_fwdbn_ok(d)) return IIF_EUNSUPPORTED;
    const ConvExtra ex{0, nullptr, 0, nullptr, 2, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, const_cast<void*>(src), nullptr, nullptr, nullptr, bn_partial, bn_partial_floats, n_partials, stream,
                      nullptr, nullptr, nullptr, &ex);
}

extern "C" int iif_conv_igemm_bn_relu2(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                                       const float* res_stats, const float* stats, unsigned char* relu_bits, void* stream) {
    if (!d || !stats || !relu_bits || d->transposed || (res_stats && !res)) return IIF_EINVAL;
    if (res_stats && !iif_conv_fwdbn_ok(d)) return IIF_EUNSUPPORTED;
    const ConvExtra ex{0, nullptr, 0, nullptr, 0, stats, relu_bits, res_stats, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return conv_entry(d, src, wgt, dst, res, nullptr, nullptr, nullptr, 0, nullptr, stream, nullptr, nullptr, nullptr, &ex);
}

extern "C" int iif_conv_pro_ok(const iif_conv_desc* d, int stats_only) {
    if (!d || g_sw.no_regw || g_sw.regstage) return 0;
    if (d->dtype != IIF_BF16 || d->dst_dtype != IIF_BF16 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0 || d->groups > 1 || d->transposed) return 0;
    if (d->hs != d->hd || d->ws != d->wd) return 0;
    const int M = d->n * d->hd * d->wd;
    return (stats_only ? (!g_sw.no_regw_fwdbn && iif_regw1x1_fwdbn_ok(M, d->cs, d->cd)) : iif_regw1x1_pro_ok(M, d->cs, d->cd)) ? 1 : 0;
}

extern "C" int iif_conv_igemm_bnstats_pro(const iif_conv_desc* d, const void* src_raw, const float* src_stats, void* act_out,
                                          unsigned char* act_bits, float* act_csum, const void* wgt, void* dst /* NULL: statistics only */,
                                          float* bn_partial, int64_t bn_partial_floats, int32_t* n_partials, void* stream) {
    if (!d || !src_raw || !src_stats || !act_out || !act_bits || !bn_partial || !n_partials || d->transposed) return IIF_EINVAL;
    if (!iif_conv_pro_ok(d, dst == nullptr) || ((reinterpret_cast<uintptr_t>(act_out)) & 15)) return IIF_EUNSUPPORTED;
    const ConvExtra ex{0, nullptr, 0, nullptr, dst ? 0 : 2, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, src_stats, act_out, act_bits, act_csum};
    return conv_entry(d, src_raw, wgt, dst ? dst : const_cast<void*>(src_raw), nullptr, nullptr, nullptr, bn_partial, bn_partial_floats, n_partials,
                      stream, nullptr, nullptr, nullptr, &ex);
}

extern "C" int iif_conv_igemm_bnstats(const iif_conv_desc* d, const void* src, const void* wgt, void* dst,
                                      const void* res, const float* bias, float* bn_partial, int64_t bn_partial_floats,
                                      int32_t* n_partials, void* stream) {
    return conv_entry(d, src, wgt, dst, res, nullptr, bias, bn_partial, bn_partial_floats, n_partials, stream);
}

extern "C" int iif_conv_igemm_masked_res(const iif_conv_desc* d, const void* src, const void* wgt, void* dst,
                                         const void* res, const unsigned char* res_bits, void* stream) {
    if (!res || !res_bits) return IIF_EINVAL;
    // the bit bytes follow the residual's 16-byte vectors: rows must be whole vectors, stores vectorised
    const int v = d && d->dst_dtype == IIF_F32 ? 4 : 8;
    if (!d || d->transposed == 0 || d->stride != 1 || (d->cd * (d->groups > 1 ? d->groups : 1)) % v) return IIF_EUNSUPPORTED;
    return conv_entry(d, src, wgt, dst, res, res_bits, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int iif_conv_igemm(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
                              const float* bias, void* stream) {
    return iif_conv_igemm_bnstats(d, src, wgt, dst, res, bias, nullptr, 0, nullptr, stream);
}

namespace {
int conv_entry(const iif_conv_desc* d, const void* src, const void* wgt, void* dst, const void* res,
               const unsigned char* res_bits, const float* bias, float* bn_partial, int64_t bn_partial_floats,
               int32_t* n_partials, void* stream, const void* bw_x, const unsigned char* bw_bits, const float* bw_stats,
               const ConvExtra* ex) {
    if (!d || !src || !wgt || !dst) return IIF_EINVAL;
    if (d->n <= 0 || d->hs <= 0 || d->ws <= 0 || d->cs <= 0 || d->hd <= 0 || d->wd <= 0 || d->cd <= 0 ||
        d->r <= 0 || d->s <= 0 || d->pad < 0)
        return IIF_EINVAL;
    if (d->stride != 1 && d->stride != 2) return IIF_EUNSUPPORTED;
    if (d->dtype != IIF_F32 && d->dtype != IIF_BF16) return IIF_EINVAL;
    if (d->dst_dtype != d->dtype && d->dst_dtype != IIF_F32) return IIF_EINVAL;
    const int pe = d->dtype == IIF_F32 ? 4 : 8;
    if (d->cs % pe != 0 || d->ldw % pe != 0 || d->ldw < d->r * d->s * d->cs + (ex && ex->src2 ? ex->cs2 : 0)) return IIF_EUNSUPPORTED;
    if (ex && (ex->src2 || ex->sbias || ex->mask_store || ex->no_store || ex->aff || ex->pro_stats)) {
        // round-3 epilogue / operand options: bf16 1x1 stride-1 launches on the LDS-staged epilogue only
        if (d->dtype != IIF_BF16 || d->dst_dtype != IIF_BF16 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0 || d->groups > 1 ||
            (d->cd % 8) || bias || (d->cs % 32))
            return IIF_EUNSUPPORTED;
        if (ex->src2 && ((ex->cs2 % 32) || ex->cs2 <= 0 || (reinterpret_cast<uintptr_t>(ex->src2) & 15))) return IIF_EUNSUPPORTED;
        if (ex->mask_store && (!bw_bits || !bn_partial)) return IIF_EINVAL;
        if (ex->no_store && (!bn_partial || res || bw_x)) return IIF_EINVAL;
        if (ex->aff && (bn_partial || res_bits || bw_x || ex->no_store)) return IIF_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(wgt) | reinterpret_cast<uintptr_t>(dst) |
         reinterpret_cast<uintptr_t>(res)) & 15)
        return IIF_EUNSUPPORTED;
    const int64_t M = (int64_t)d->n * d->hd * d->wd;
    if (M > 0x7fffff00LL || (int64_t)d->n * d->hs * d->ws > 0x7fffff00LL) return IIF_EUNSUPPORTED;
    ConvArgs a{};
    a.src = (const unsigned char*)src; a.wgt = (const unsigned char*)wgt; a.dst = (unsigned char*)dst;
    a.res = (const unsigned char*)res; a.bias = bias;
    a.wfrag = (const unsigned char*)d->wgt_frag;
    a.wfrag_kind = d->wgt_frag ? d->wgt_frag_kind : 0;
    a.res_bits = res_bits;
    a.bn_partial = nullptr;
    if (n_partials) *n_partials = 0;
    int rows = 0;
    if (bn_partial) {
        // fused statistics (forward) / backward sums of the upstream unit need the LDS-staged bf16 epilogue of the
        // pipelined kernels; one partial row per pixel tile, counted where the tile height is chosen (launch_one)
        const int64_t esz0 = 2;
        const bool ok = d->dtype == IIF_BF16 && d->dst_dtype == IIF_BF16 && (d->cd % 8) == 0 && !bias && (bw_x || (ex && ex->mask_store) || !res) &&
                        !g_sw.regstage &&
                        (int64_t)d->n * d->hs * d->ws * d->cs * (d->groups > 1 ? d->groups : 1) * esz0 < 0x7f000000LL;
        if (!ok) return IIF_EUNSUPPORTED;
        a.bn_partial = bn_partial;
        a.bn_cap = bn_partial_floats;
        a.rows_out = &rows;
        a.bw_x = (const unsigned char*)bw_x; a.bw_bits = bw_bits; a.bw_stats = bw_stats;
    }
    if (ex) {
        a.mask_store = ex->mask_store; a.src2 = (const unsigned char*)ex->src2; a.Cs2 = ex->cs2; a.sbias = ex->sbias;
        a.no_store = ex->no_store; a.aff = ex->aff; a.relu_out = ex->relu_out; a.aff2 = ex->aff2;
        a.rx_src2 = (const unsigned char*)ex->rx_src2; a.rx_w3 = (const unsigned char*)ex->rx_w3; a.rx_k2 = ex->rx_k2; a.rx_ldw3 = ex->rx_ldw3;
        a.pro_stats = ex->pro_stats; a.pro_out = (unsigned char*)ex->pro_out; a.pro_bits = ex->pro_bits; a.pro_csum = ex->pro_csum;
        a.pg_slab = ex->pg_slab; a.pg_cap = ex->pg_cap; a.pg_ld = ex->pg_ld; a.pg_count = ex->pg_count;
    }
    a.N = d->n; a.Hs = d->hs; a.Ws = d->ws; a.Cs = d->cs; a.Hd = d->hd; a.Wd = d->wd; a.Cd = d->cd;
    a.R = d->r; a.S = d->s; a.sshift = d->stride - 1; a.pad = d->pad; a.transposed = d->transposed ? 1 : 0;
    a.ldw = d->ldw; a.M = (int)M; a.K = d->r * d->s * d->cs;
    a.groups = d->groups > 1 ? d->groups : 1;
    a.spitch = a.groups * d->cs; a.dpitch = a.groups * d->cd;
    if (a.groups > 65535) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t esz = d->dtype == IIF_F32 ? 4 : 2;
    const int64_t src_bytes = (int64_t)d->n * d->hs * d->ws * a.spitch * esz;
    const int64_t wgt_bytes = (int64_t)d->cd * d->ldw * esz;          // one group's weight matrix
    int rc;
    if (d->dtype == IIF_BF16) {
        if (d->dst_dtype == IIF_F32) rc = launch_conv<unsigned short, true>(a, src_bytes, wgt_bytes, st);
        else rc = launch_conv<unsigned short, false>(a, src_bytes, wgt_bytes, st);
    } else {
        rc = launch_conv<float, true>(a, src_bytes, wgt_bytes, st);
    }
    if (n_partials) *n_partials = rows;
    return rc;
}
}  // namespace

#ifdef IIF_CONV_STAMPS
extern "C" int iif_debug_set_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? IIF_OK : IIF_ELAUNCH;
}
#endif
