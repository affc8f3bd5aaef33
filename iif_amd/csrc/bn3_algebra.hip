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

// ---------------------------------------------------------------------------------------------------------------
// Everything between the producer's partial rows and the stacked-weights data gradient in ONE launch (+ the slab sum below):
// round 3 ran four dependent launches here (slice sums, coefficients, c x c products, slab sum: 50-90 us per block inside the
// step, with 40-85 us idle gaps behind them on the compute stream).
// Grid (C / 64 channel groups, S row slices).  Phase 1, every block: the slice's column sums of its group's 128 columns
// (sum g~ and sum g~ xhat of 64 channels), published with agent-scope atomic exchanges, one ticket per group (the fence-free
// protocol of bn_reduce_finalize_kernel).  Phase 2, the LAST block of a group: the S slices in order -> sums; (A, B, D), dgamma,
// dbeta of its 64 channels; the g~ half of the stacked weights wt[j][ch] = bf16(A[ch] W[ch][j]); and the group's share of the a2
// half, slab[g][jo][i] = sum_{ch in g} bf16(B[ch] W[ch][jo]) W[ch][i] on the matrix pipe (K = 64 channels: two
// v_mfma_f32_16x16x32_bf16 per 16 x 16 tile, fragments by ds_read_b64_tr_b16 from two swizzled [64][c] LDS tiles), plus
// bias_slab[g][jo] = sum_ch D[ch] W[ch][jo].  c in {64, 128, 256}.
typedef __attribute__((address_space(3))) const unsigned char lds_cu8;
__device__ __forceinline__ s16x4 tr_read_lds(const unsigned char* p) {       // (inline asm: see tr_read in conv_wgrad.hip)
    s16x4 v;
    const unsigned a = (unsigned)(unsigned long long)(lds_cu8*)(p);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// halves of a fragment stay separate values until they are joined BEHIND the fence (TrPair / tr_join in conv_wgrad.hip)
struct TrPair { s16x4 lo, hi; };
__device__ __forceinline__ s16x8 tr_join(TrPair& p) {
    asm volatile("" : "+v"(p.lo), "+v"(p.hi));
    return s16x8{p.lo.x, p.lo.y, p.lo.z, p.lo.w, p.hi.x, p.hi.y, p.hi.z, p.hi.w};
}
// byte offset of logical byte column colb of row `row` in a [rows][rb] tile: conflict-free for the transposing reads
__device__ __forceinline__ int swz_addr(int row, int colb, int rb) {
    const int key = rb >= 256 ? (row & 7) : ((row >> 1) & 3);
    return row * rb + ((((colb >> 5) ^ key)) << 5) + (colb & 31);
}

constexpr int kGC = 64;                                  // channels per group
// (c is a template parameter: with a run-time trip count hipcc branches around every load of the batched requests below and
// waits for each one separately)
template <int c>
__global__ void __launch_bounds__(256) bn3_prep_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* partial,
                                                       int nrows, int rows_per_slice, const float* stats, const float* gamma, int C,
                                                       double count, float* slices, int* tickets, float* coef, float* dgamma, float* dbeta,
                                                       unsigned short* wt, int ldwt, float* slab, float* bias_slab, float* corr_slab) {
    __shared__ __attribute__((aligned(1024))) unsigned char tiles[2 * kGC * 512];      // W and B o W, [64][c] bf16 each
    __shared__ double shd[256];
    __shared__ float cA[kGC], cB[kGC], cD[kGC], cE[kGC], cM[kGC], ssum[2 * kGC];
    __shared__ int last;
    const int g = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
    const int ch0 = g * kGC, C2 = 2 * C;
    const int t = threadIdx.x;
    // ---- phase 1: this slice's sums of the group's 128 columns (two threads per column, interleaved rows, fixed order)
    {
        const int col = t & 127, half = t >> 7;
        const int ch = ch0 + (col & 63);
        const int gcol = (col < 64 ? 0 : C) + ch;
        const int r0 = sl * rows_per_slice;
        int r1 = r0 + rows_per_slice; if (r1 > nrows) r1 = nrows;
        double a = 0.0;
        if (ch < C) {
            // every dependent round trip costs 3-5 us while the other streams saturate the memory system: 32 rows in flight
            // per thread, i.e. one or two batches per slice
            for (int rb_ = r0 + half; rb_ < r1; rb_ += 64) {
                float v[32];
                // (unconditional loads from a clamped row, masked afterwards: a conditional load is branched around and waited
                // for on its own - 32 dependent round trips instead of one)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = rb_ + 2 * i;
                    v[i] = partial[(int64_t)(r < r1 ? r : r1 - 1) * C2 + gcol];
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) a += (rb_ + 2 * i < r1) ? (double)v[i] : 0.0;
            }
        }
        shd[t] = a;
        __syncthreads();
        float keep = 0.f;
        if (t < 128 && ch < C)
            keep = __hip_atomic_exchange(slices + (int64_t)sl * C2 + gcol, (float)(shd[t] + shd[t + 128]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(keep) : "memory");        // this wave's exchanges have returned
        __syncthreads();
        if (t == 0) {
            const int tk = __hip_atomic_fetch_add(tickets + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (tk == S - 1);
        }
        __syncthreads();
        if (!last) return;
    }
    // ---- phase 2: the last block of channel group g.  Everything it needs from memory is requested in ONE batch (the slices,
    // its rows of W and P, the statistics): two dependent round trips to the coefficients instead of six
    constexpr int rb = c * 2, cpr = c / 8;                 // row bytes, 16-byte chunks per row
    constexpr int nchunk = kGC * cpr / 256;                 // W chunks per thread: 2 / 4 / 8
    u32x4 wreg[nchunk];
#pragma unroll
    for (int k = 0; k < nchunk; ++k) {
        const int idx = t + 256 * k, row = idx / cpr, q = idx - row * cpr;
        wreg[k] = *reinterpret_cast<const u32x4*>(W + (int64_t)(ch0 + row < C ? ch0 + row : 0) * ldw + q * 8);
        if (ch0 + row >= C) wreg[k] = u32x4{0u, 0u, 0u, 0u};
    }
    // P rows for sum_j P[ch][j] W[ch][j] (only with P; without it the same loads read the head of `partial` and are dropped):
    // 4 threads per channel, a contiguous quarter of the row each
    const int lc4 = t >> 2, sub4 = t & 3;
    const bool chv = ch0 + lc4 < C;
    constexpr int pq = c / 16;                              // float4 pieces of P per thread: 4 / 8 / 16
    const float* prow = (P && chv) ? P + (int64_t)(ch0 + lc4) * ldp + sub4 * (c / 4) : partial;
    f32x4 preg[pq];
#pragma unroll
    for (int k = 0; k < pq; ++k) preg[k] = *reinterpret_cast<const f32x4*>(prow + 4 * k);
    const int chs = chv ? ch0 + lc4 : 0;
    const float mu_ = stats[chs], invstd_ = stats[C + chs], gamma_ = gamma[chs];
    {
        // slices: threads t and t + 128 take the halves [0, 32) and [32, 64) of column t & 127, all loads in flight
        const int col = t & 127, hf = t >> 7;
        const int ch = ch0 + (col & 63);
        const int gcol = (col < 64 ? 0 : C) + (ch < C ? ch : 0);
        const int j0 = hf * 32;
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = __hip_atomic_load(slices + (int64_t)(j0 + i < S ? j0 + i : S - 1) * C2 + gcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) a += (j0 + i < S && ch < C) ? (double)v[i] : 0.0;
        shd[t] = a;
    }
    if (t == 0) __hip_atomic_store(tickets + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // W rows of the group -> LDS (swizzled)
    unsigned char* const WT = tiles;
    unsigned char* const BT = tiles + kGC * rb;
#pragma unroll
    for (int k = 0; k < nchunk; ++k) {
        const int idx = t + 256 * k, row = idx / cpr, q = idx - row * cpr;
        *reinterpret_cast<u32x4*>(WT + swz_addr(row, q * 16, rb)) = wreg[k];
    }
    __syncthreads();
    if (t < 128) ssum[t] = (float)(shd[t] + shd[t + 128]);
    float pw = 0.f;
    if (P) {
#pragma unroll
        for (int k = 0; k < pq; ++k) {
            const int j = sub4 * (c / 4) + 4 * k;
            const u32x2 wv = *reinterpret_cast<const u32x2*>(WT + swz_addr(lc4, j * 2, rb));
            pw += preg[k].x * bf16_bits_to_f32(wv.x & 0xffffu); pw += preg[k].y * __uint_as_float(wv.x & 0xffff0000u);
            pw += preg[k].z * bf16_bits_to_f32(wv.y & 0xffffu); pw += preg[k].w * __uint_as_float(wv.y & 0xffff0000u);
        }
    }
    pw += __shfl_xor(pw, 1, 64); pw += __shfl_xor(pw, 2, 64);
    __syncthreads();
    if (sub4 == 0) {
        const int lc = lc4, ch = ch0 + lc;
        float A = 0.f, B = 0.f, D = 0.f, E = 0.f, Mu = 0.f;
        if (ch < C) {
            const float mu = mu_, invstd = invstd_;
            const float s1 = ssum[lc];
            const float s2 = P ? invstd * (pw - mu * s1) : ssum[kGC + lc];        // sum g~ xhat
            A = gamma_ * invstd;
            B = -A * (float)((double)s2 / count) * invstd;
            E = -A * (float)((double)s1 / count);
            D = E - B * mu;
            Mu = mu;
            dbeta[ch] = s1; dgamma[ch] = s2;
            coef[ch] = A; coef[C + ch] = B; coef[2 * C + ch] = D;
        }
        cA[lc] = A; cB[lc] = B; cD[lc] = D; cE[lc] = E; cM[lc] = Mu;
    }
    __syncthreads();
    // B o W beside it
#pragma unroll
    for (int k = 0; k < nchunk; ++k) {
        const int idx = t + 256 * k, row = idx / cpr, q = idx - row * cpr;
        const u32x4 w = wreg[k];
        const float b = cB[row];
        u32x4 bw;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bw[e] = pack_bf16x2(b * bf16_bits_to_f32(w[e] & 0xffffu), b * __uint_as_float(w[e] & 0xffff0000u));
        *reinterpret_cast<u32x4*>(BT + swz_addr(row, q * 16, rb)) = bw;
    }
    __syncthreads();
    // the g~ half of the stacked weights: wt[j][ch0 + lane] = bf16(A W[ch][j]); a wave writes 128 contiguous bytes per j
    {
        const int lane = t & 63, wv = t >> 6;
        if (ch0 + lane < C) {
            const float A = cA[lane];
            for (int j = wv; j < c; j += 4) {
                const unsigned short wb = *reinterpret_cast<const unsigned short*>(WT + swz_addr(lane, j * 2, rb));
                wt[(int64_t)j * ldwt + ch0 + lane] = f32_to_bf16_bits(A * bf16_bits_to_f32(wb));
            }
        }
        // bias_slab[g][jo] = sum_ch D[ch] W[ch][jo], D = E - B mu, with the B W product taken AS THE MATRIX PIPE SEES IT
        // (bf16(B W), the BT tile): the a2 half of the stacked weights is sum_ch bf16(B W[ch][jo]) W[ch][i], and only with the
        // same rounded factor does  sum_i colsum(a2)[i] G[jo][i] + M bias[jo]  cancel (mu[ch] = W[ch] . colsum(a2) / M); the
        // rounding of B W times mu - the same sign on every pixel - otherwise lands in the data gradient's column sums
        // corr_slab[g][jo] = sum_ch sum(g~)[ch] * (bf16(A W[ch][jo]) - A W[ch][jo]): what the rounding of the g~ half of the
        // stacked weights (the loop above stores exactly these bf16 values) adds to the COLUMN SUM of the data gradient
        // (bn3_gm_finish_kernel); a thread owns a column and walks the group's channels, as for the bias
        if (t < c) {
            float bsum = 0.f, esum = 0.f;
            for (int ch = 0; ch < kGC; ++ch) {
                const float w = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(WT + swz_addr(ch, t * 2, rb)));
                const float bw = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(BT + swz_addr(ch, t * 2, rb)));
                bsum += cE[ch] * w - cM[ch] * bw;
                const float v = cA[ch] * w;
                esum += ssum[ch] * (bf16_bits_to_f32(f32_to_bf16_bits(v)) - v);
            }
            bias_slab[(int64_t)g * c + t] = bsum;
            corr_slab[(int64_t)g * c + t] = esum;
        }
    }
    // slab[g][jo][i] on the matrix pipe: A operand = W (rows i), B operand = B o W (columns jo), K = the group's 64 channels
    {
        const int lane = t & 63, wv = t >> 6;
        const int gq = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
        float* const out = slab + (int64_t)g * c * c;
        constexpr int nb = c / 16;
        for (int ib = wv; ib < nb; ib += 4) {
            TrPair ap[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                ap[kb].lo = tr_read_lds(WT + swz_addr(kb * 32 + 4 * gq + q, (ib * 16 + 4 * pp) * 2, rb));
                ap[kb].hi = tr_read_lds(WT + swz_addr(kb * 32 + 16 + 4 * gq + q, (ib * 16 + 4 * pp) * 2, rb));
            }
            lds_fence();
            const s16x8 af[2] = {tr_join(ap[0]), tr_join(ap[1])};
            for (int jb = 0; jb < nb; ++jb) {
                TrPair bp[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bp[kb].lo = tr_read_lds(BT + swz_addr(kb * 32 + 4 * gq + q, (jb * 16 + 4 * pp) * 2, rb));
                    bp[kb].hi = tr_read_lds(BT + swz_addr(kb * 32 + 16 + 4 * gq + q, (jb * 16 + 4 * pp) * 2, rb));
                }
                lds_fence();
                const s16x8 bf[2] = {tr_join(bp[0]), tr_join(bp[1])};
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[0]), __builtin_bit_cast(bf16x8, bf[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[1]), __builtin_bit_cast(bf16x8, bf[1]), acc, 0, 0, 0);
                *reinterpret_cast<f32x4*>(out + (int64_t)(jb * 16 + li) * c + ib * 16 + 4 * gq) = acc;
            }
        }
    }
}

// sum of the channel-group slabs of bn3_prep_kernel in a fixed order (deterministic), bf16 conversion into the stacked weights
// (G = the slab count as a template parameter: with a run-time trip count every slab was one dependent round trip - load, wait, add -
// i.e. 16 of them at C = 1024 on the compute stream's critical path, 12 us on average and 45 at worst inside the step; now one.)
//
// Column-sum compensation (round 5).  In exact arithmetic the data gradient da = [g~ | a2] [A o W ; W^T diag(B) W]^T + D W sums
// to ZERO over the pixels in every column (BN backward has sum_m dy = 0 per channel).  The stacked weights are bf16: each
// rounding error delta is multiplied by the SAME-signed column sums of its operand (a2 is a ReLU output: all >= 0; g~ is gated),
// so sum_m da[m][j] picks up  sum_i colsum(a2)[i] delta2[j][i] + sum_ch colsum(g~)[ch] delta1[j][ch]  - coherent over all
// pixels, and what is left of the upstream column sums (d beta of the BN below, the stem's above all) after cancellation is
// small: 32 % on bn1's bias gradient through the shortcut of layer1.0.  Both column sums are known here, so the fp32 bias
// absorbs the error:  bias[j] = D W[j] - (those two sums) / M.  colsum2 == nullptr: no compensation (the round-3 / 4 behaviour).
// Grid c * c / 256 blocks; a block owns whole rows j (256 / c of them).
template <int G>
__global__ void __launch_bounds__(256) bn3_gm_finish_kernel(const float* slab, const float* bias_slab, const float* corr_slab, int slices,
                                                            int C, int c, const float* colsum2, float inv_count, unsigned short* wt,
                                                            int ldwt, float* bias) {
    __shared__ float red[4];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int cc = c * c;
    const int j = q / c, i = q - j * c;                    // (c * c is a multiple of 256: every thread has an element)
    float v[G], vb[G], vc[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {                                   // unconditional, clamped: all in flight together
        const int kk = k < slices ? k : 0;
        v[k] = slab[(int64_t)kk * cc + q];
        vb[k] = bias_slab[(int64_t)kk * c + j];
        vc[k] = corr_slab[(int64_t)kk * c + j];
    }
    const float cs = colsum2 ? colsum2[i] : 0.f;
    float s = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) { s += k < slices ? v[k] : 0.f; sb += k < slices ? vb[k] : 0.f; sc += k < slices ? vc[k] : 0.f; }
    const unsigned short r = f32_to_bf16_bits(s);
    wt[(int64_t)j * ldwt + C + i] = r;
    // row sum of colsum2[i] * (rounded - exact): lanes of a wave, then the c / 64 waves of the row (fixed order)
    const float e = wsum(cs * (bf16_bits_to_f32(r) - s));
    const int wave = threadIdx.x >> 6, wpr = c >> 6;       // waves per row: 1 / 2 / 4
    if ((threadIdx.x & 63) == 0) red[wave] = e;
    __syncthreads();
    if (i == 0) {
        float es = 0.f;
        for (int w = 0; w < wpr; ++w) es += red[wave + w];
        bias[j] = colsum2 ? sb - (es + sc) * inv_count : sb;
    }
}

// dW[ch][j] = A P[ch][j] + B sum_i W[ch][i] Gram[i][j] + D csum[j].  Grid (C / 64, c / 32); c in {64, 128, 256}.
// Round 4's form staged K through LDS 64 at a time with element-wise global loads (a lane per matrix ROW: 64 lines per wave
// instruction) - four dependent memory round trips for a few MFLOP, 39 us alone and 13 launches per step on the weight-gradient
// stream.  Here a block requests EVERYTHING it will read before it waits once: its 64 x c tile of W (16 bytes per lane along
// the rows), the c x 32 panel of Gram, its P values, coefficients and column sums; both operands go to LDS as they are (W rows
// with a pitch of c + 8 halves: the four row groups of a wave read four different bank quarters), and a thread multiplies
// 4 channels x 2 columns four k at a time (one 8-byte read per row, one per Gram row).  Same order of additions as before (i
// ascending, one multiply and one add per term): bit-identical values.
template <int c>
__global__ void __launch_bounds__(256) bn3_dw_kernel(const float* P, int ldp, const unsigned short* W, int ldw, const float* gram,
                                                     int ldg, const float* csum, const float* coef, int C, float* dW, int lddw) {
    constexpr int WP = c + 8;                             // W tile pitch in halves
    constexpr int CPR = c / 8;                            // 16-byte pieces per W row
    constexpr int NW = 64 * CPR / 256;                    // pieces per thread: 2 / 4 / 8
    constexpr int NG = c * 8 / 256;                       // Gram panel (c rows x 128 bytes): 2 / 4 / 8 pieces per thread
    __shared__ __attribute__((aligned(16))) unsigned short wsm[64 * WP];
    __shared__ __attribute__((aligned(16))) float gsm[c * 32];
    const int t = threadIdx.x, ch0 = blockIdx.x * 64, j0 = blockIdx.y * 32;
    u32x4 wv[NW], gv[NG];
#pragma unroll
    for (int p = 0; p < NW; ++p) {
        const int q = t + 256 * p, row = q / CPR, col = q % CPR;
        const int ch = ch0 + row < C ? ch0 + row : C - 1;
        wv[p] = *reinterpret_cast<const u32x4*>(W + (int64_t)ch * ldw + col * 8);
    }
#pragma unroll
    for (int p = 0; p < NG; ++p) {
        const int q = t + 256 * p, i = q >> 3, col = q & 7;
        gv[p] = *reinterpret_cast<const u32x4*>(gram + (int64_t)i * ldg + j0 + col * 4);
    }
    const int tch = (t >> 4) * 4, tj = (t & 15) * 2;
    float2 pv[4];
    float A[4], B[4], D[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = ch0 + tch + r < C ? ch0 + tch + r : C - 1;
        pv[r] = *reinterpret_cast<const float2*>(P + (int64_t)ch * ldp + j0 + tj);
        A[r] = coef[ch]; B[r] = coef[C + ch]; D[r] = coef[2 * C + ch];
    }
    const float2 cs = *reinterpret_cast<const float2*>(csum + j0 + tj);
#pragma unroll
    for (int p = 0; p < NW; ++p) {
        const int q = t + 256 * p, row = q / CPR, col = q % CPR;
        *reinterpret_cast<u32x4*>(wsm + row * WP + col * 8) = wv[p];
    }
#pragma unroll
    for (int p = 0; p < NG; ++p) {
        const int q = t + 256 * p;
        *reinterpret_cast<u32x4*>(gsm + (q >> 3) * 32 + (q & 7) * 4) = gv[p];
    }
    __syncthreads();
    float acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; }
#pragma unroll 4
    for (int i = 0; i < c; i += 4) {
        float2 g2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g2[q] = *reinterpret_cast<const float2*>(gsm + (i + q) * 32 + tj);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint2 w4 = *reinterpret_cast<const uint2*>(wsm + (tch + r) * WP + i);
            const float wf[4] = {bf16_bits_to_f32(w4.x & 0xffffu), __uint_as_float(w4.x & 0xffff0000u),
                                 bf16_bits_to_f32(w4.y & 0xffffu), __uint_as_float(w4.y & 0xffff0000u)};
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[r][0] += wf[q] * g2[q].x; acc[r][1] += wf[q] * g2[q].y; }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = ch0 + tch + r;
        if (ch < C)
            *reinterpret_cast<float2*>(dW + (int64_t)ch * lddw + j0 + tj) =
                float2{A[r] * pv[r].x + B[r] * acc[r][0] + D[r] * cs.x, A[r] * pv[r].y + B[r] * acc[r][1] + D[r] * cs.y};
    }
}

}  // namespace

extern "C" {

int64_t iif_bn3_algebra_prep_scratch_floats(int C, int c) {
    const int64_t G = (C + kGC - 1) / kGC;
    return (int64_t)64 * 2 * C + G * ((int64_t)c * c + 2 * c);
}

int iif_bn3_algebra_prep(const float* P, int ldp, const void* w_bf16, int ldw, const float* partial, int n_partials,
                         const float* stats, const float* gamma, int C, int c, int64_t m, float* coef, float* dgamma, float* dbeta,
                         void* wt, int ldwt, float* bias, float* scratch, int64_t scratch_floats, int32_t* tickets, const float* colsum2,
                         void* stream) {
    if (!w_bf16 || !partial || !stats || !gamma || !coef || !dgamma || !dbeta || !wt || !bias || !scratch || !tickets || C <= 0 ||
        c <= 0 || m <= 0 || n_partials <= 0)
        return IIF_EINVAL;
    if ((c != 64 && c != 128 && c != 256) || (P && ldp < c) || ldw < c || (ldw % 8) || ldwt < C + c) return IIF_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(w_bf16) & 15) != 0) return IIF_EUNSUPPORTED;
    if (scratch_floats < iif_bn3_algebra_prep_scratch_floats(C, c)) return IIF_EINVAL;
    const int G = (C + kGC - 1) / kGC;
    if (G > 64) return IIF_EUNSUPPORTED;                  // tickets: int32[64]
    const int slices = n_partials < 64 ? n_partials : 64;
    const int rps = (n_partials + slices - 1) / slices;
    const int S = (n_partials + rps - 1) / rps;
    float* slab = scratch + (int64_t)64 * 2 * C;
    float* bias_slab = slab + (int64_t)G * c * c;
    float* corr_slab = bias_slab + (int64_t)G * c;
    hipStream_t st = as_stream(stream);
#define IIF_PREP(CC) hipLaunchKernelGGL(bn3_prep_kernel<CC>, dim3(G, S), dim3(256), 0, st, P, ldp, (const unsigned short*)w_bf16, ldw, partial, \
                                        n_partials, rps, stats, gamma, C, (double)m, scratch, tickets, coef, dgamma, dbeta, (unsigned short*)wt, ldwt, slab, bias_slab, corr_slab)
    if (c == 64) IIF_PREP(64); else if (c == 128) IIF_PREP(128); else IIF_PREP(256);
#undef IIF_PREP
    IIF_LAUNCH_CHECK();
#define IIF_GMF(GG) hipLaunchKernelGGL(bn3_gm_finish_kernel<GG>, dim3(c * c / 256), dim3(256), 0, st, slab, bias_slab, corr_slab, G, C, c, \
                                       colsum2, (float)(1.0 / (double)m), (unsigned short*)wt, ldwt, bias)
    if (G <= 4) IIF_GMF(4); else if (G <= 8) IIF_GMF(8); else if (G <= 16) IIF_GMF(16); else if (G <= 32) IIF_GMF(32); else IIF_GMF(64);
#undef IIF_GMF
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_bn3_algebra_dw(const float* P, int ldp, const void* w_bf16, int ldw, const float* gram, int ldg, const float* csum,
                       const float* coef, int C, int c, float* dW, int lddw, void* stream) {
    if (!P || !w_bf16 || !gram || !csum || !coef || !dW || C <= 0 || c <= 0) return IIF_EINVAL;
    if ((c != 64 && c != 128 && c != 256) || ldp < c || ldw < c || ldg < c || lddw < c) return IIF_EUNSUPPORTED;
    // 16-byte pieces of W and Gram, 8-byte pairs of P / dW / csum
    if ((ldw % 8) || (ldg % 4) || (ldp % 2) || (lddw % 2) || (reinterpret_cast<uintptr_t>(w_bf16) & 15) || (reinterpret_cast<uintptr_t>(gram) & 15) ||
        (reinterpret_cast<uintptr_t>(P) & 7) || (reinterpret_cast<uintptr_t>(dW) & 7) || (reinterpret_cast<uintptr_t>(csum) & 7))
        return IIF_EUNSUPPORTED;
    const dim3 grid((C + 63) / 64, c / 32), blk(256);
    hipStream_t st = as_stream(stream);
#define IIF_DW(CC) hipLaunchKernelGGL(bn3_dw_kernel<CC>, grid, blk, 0, st, P, ldp, (const unsigned short*)w_bf16, ldw, gram, ldg, csum, coef, C, dW, lddw)
    if (c == 64) IIF_DW(64); else if (c == 128) IIF_DW(128); else IIF_DW(256);
#undef IIF_DW
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
