// 1x1 / stride 1 convolution forward (bf16) for the narrow -> wide layers of a bottleneck (conv3: K = c -> N = 4c;
// classification/resnet_pytorch.py:160-161) with the WEIGHTS IN REGISTERS and the batch-norm partial sums of the stored output.
//
// Why (round 4): the 128 x 128 tile kernel runs these launches at 3.3 TB/s.  A block of it lives ~12 us for
// 32 KB in and 32 KB out, re-fetches its weight tile from L2 through the LDS-DMA path (one weight byte per activation byte) and
// re-reads the activation tile once per N tile.  Here a persistent block of eight waves owns ALL (or a 256-column slice) of the
// output channels: wave w keeps the MFMA fragments of its CW columns x K in registers for the whole launch (K x CW x 2 B / 64
// lanes = 64 VGPRs), so the DMA path carries the activation rows only, each exactly once; every wave multiplies the whole
// 64-row tile (fragment reads out of a double-buffered LDS tile, one barrier per tile, the next tile in flight) and drains its own
// 64 x CW block through a wave-private LDS transpose into 16-byte stores; the per-channel (sum, sum of squares) accumulate in
// registers over the block's tiles: ONE partial row per tile sequence.  The N slices of a sequence sit on blocks b, b + 8, ... of
// one XCD and share the activation tile through that L2.  Accumulation order over K is the tile kernel's (K steps of 32,
// ascending): the stored values are bit-identical to it.
#include "common.h"
// Tile stores of the 1x1 kernels: a wave owns CW = 16 / 32 columns, i.e. 32 / 64 bytes of a row - HALF or a QUARTER of a 128-byte
// line per store, the rest of the line coming from the neighbouring waves a little later.  As nontemporal (streaming) stores those
// pieces left the L2 before they met: WRITE_SIZE read 1.2x (CW = 32) to 1.5x (CW = 16) the bytes of the tensor, 1.2 GB per step
// (profiles/r6_e_hbm_per_launch.txt against r6_d); as plain write-back stores the L2 joins them: exact bytes, 5-25 % off the
// kernels' alone time.  -DIIF_REGW_NT_STORE: the round-5 form.
#ifdef IIF_REGW_NT_STORE
#define IIF_REGW_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define IIF_REGW_ST(v, p) (*(p) = (v))
#endif

namespace {
typedef __attribute__((address_space(3))) void lds_void;

struct RegwArgs {
    const unsigned char* src; const unsigned char* wgt; unsigned char* dst; float* bn_partial;
    int M, mtiles, spitch, ldw, Cd, dpitch, bn_row0, S;
    // EPI kernels only (the epilogue options of conv_igemm.hip's staged_drain, same arithmetic, same meaning):
    const unsigned char* res; const unsigned char* res_bits; const unsigned char* bw_x; const unsigned char* bw_bits;
    const float* bw_stats; int mask_store;
    int no_store;          // forward only: the tile is rounded and summed exactly as if it were stored, and dropped (two-pass forward)
    // EPI 2 (forward BN epilogue, round 6): dst = relu(fma(a, bf16(conv), b) + r) with (a, b) at aff[2 Cd + c] / aff[3 Cd + c], r the
    // residual `res` as it is (aff2 null) or normalised by ITS batch norm, fma(a2, res, b2) with aff2 laid out like aff (the
    // convolutional shortcut of a downsample block); relu_out: one byte of ReLU decisions per 16-byte vector.  bn_apply_kernel's arithmetic.
    const float* aff; const float* aff2; unsigned char* relu_out;
    // EPI 4 (round 6): the data-gradient epilogue of EPI 1 whose upstream x (bw_x: the raw output of the upstream block's conv3)
    // is RECOMPUTED per tile instead of read: x = src2 (that block's a2, [M, KK2]) times w3^T (its conv3 weights as the forward
    // multiplied them, [Cd, ldw3] bf16), rounded to bf16 as the stored tensor would have been.  The upstream output need not exist.
    const unsigned char* src2; const unsigned char* w3; int spitch2, ldw3; unsigned src2_bytes;
    // PRO (round 6): `src` is the RAW output of the previous convolution; its batch norm + ReLU (pro_stats: a at [2 K + c], b at
    // [3 K + c], bn_apply_kernel's arithmetic) is applied to the tile in LDS before the MFMA phase, and the activated tile is written
    // out as a by-product (pro_out [M, K] bf16, pro_bits one byte per 16-byte vector) by slice 0: the bn_apply launch and one
    // pass over the activation disappear.  pro_csum (nullable): per sequence one row [2][K] (column sums of the activated tiles, zeros).
    const float* pro_stats; unsigned char* pro_out; unsigned char* pro_bits; float* pro_csum;
    // PG (round 6, EPI 4 with one N slice and KK2 = 64): the two small matrices the algebraic BN3 backward of the UPSTREAM block needs
    // come out of this launch - P = g~^T a2 ([Cd, KK2]: the block holds g~'s tile in its staging buffers and a2's in LDS) and
    // Gram = a2^T a2 ([KK2, KK2]) - one fp32 slab [(Cd + KK2), pg_ld] per tile sequence (P rows first), summed afterwards in
    // sequence order (iif_slab_sum): the stacked weight-gradient launch that re-read g~ and a2 (0.5 GB at 56 x 56) is not needed.
    float* pg_slab; int pg_ld;
};

// transposing LDS read for the PG products (the discipline of conv_wgrad.hip: asm so that the compiler's wait-count pass does not
// drain the LDS-DMA in flight in front of it; halves joined BEHIND the fence)
typedef __attribute__((address_space(3))) const unsigned char lds_cu8r;
__device__ __forceinline__ s16x4 tr_read_r(const unsigned char* p) {
    s16x4 v;
    const unsigned a = (unsigned)(unsigned long long)(lds_cu8r*)(p);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
struct TrPairR { s16x4 lo, hi; };
__device__ __forceinline__ s16x8 tr_join_r(TrPairR& p) {
    asm volatile("" : "+v"(p.lo), "+v"(p.hi));
    return s16x8{p.lo.x, p.lo.y, p.lo.z, p.lo.w, p.hi.x, p.hi.y, p.hi.z, p.hi.w};
}

__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 2; }      // as conv_igemm.hip's swz: 64-byte LDS rows

// EPI: the data-gradient epilogue (residual gated by its ReLU bits, store gated by the upstream block's ReLU bits, upstream
// BN-backward sums with or without the upstream x).  Everything a tile's epilogue reads from memory is requested ONE TILE AHEAD
// into registers (the MFMA phase of a tile is a fraction of a microsecond: nothing to hide a load behind), unconditionally (an
// operand the launch does not have is read from one dummy line), 10 registers per staged 16-byte vector.
// EPI 0: plain forward (+ sums of the stored tile); 1: the data-gradient epilogue; 2: the forward BN epilogue (pass 2 of the
// two-pass forward); 3: statistics only, taken from the ACCUMULATORS (pass 1: no staging, no store; sums of the unrounded tile);
// 4: EPI 1 with the upstream x recomputed from (src2, w3) over KK2 channels
template <int KK, int CW, int MT, int EPI, int KK2 = 0, bool PRO = false, bool PG = false>
__global__ void __launch_bounds__(512, 1) gemm1x1_regw_kernel(RegwArgs a, unsigned src_bytes) {
    constexpr bool RX = EPI == 4, DG = EPI == 1 || EPI == 4;
    static_assert(!PG || (RX && KK2 == 64 && CW == 32 && MT % 32 == 0), "P / Gram by-product: one 64-channel second source");
    constexpr int VPR = KK / 8, NPV = PRO ? MT * VPR / 512 : 0;         // PRO: vectors per row of the tile, vectors per thread
    static_assert(!PRO || ((MT * VPR) % 512 == 0 && 512 % VPR == 0 && (EPI == 0 || EPI == 3)), "prologue shape");
    constexpr int NK2 = RX ? KK2 / 32 : 1, TILE2 = RX ? NK2 * MT * 64 : 0, NAP2 = RX ? NK2 * (MT / 16) / 8 : 0;
    static_assert(!RX || (KK2 >= 64 && (NK2 * (MT / 16)) % 8 == 0), "second source");
    constexpr int NK = KK / 32, CB = CW / 16, RB = MT / 16;
    constexpr int SLAB = MT * 64, TILE = NK * SLAB;       // NK slabs of [MT rows x 64 B]
    constexpr int NAP = NK * RB / 8;                       // 1-KB DMA pieces (16 rows of a slab) per wave and tile
    constexpr int PITCH = CW * 2 + 16, STG = MT * PITCH;
    constexpr int LPR = CW / 8, RPI = 64 / LPR, NST = MT / RPI;      // lanes per staged row, rows per store instruction, stores per tile
    constexpr unsigned OOB = 0x80000000u;
    constexpr int STGB = EPI == 3 ? 0 : 8 * STG;
    static_assert((NK * RB) % 8 == 0 && MT % RPI == 0 && 2 * TILE + STGB + 2 * TILE2 <= 160 * 1024, "shape");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * TILE + STGB + 2 * TILE2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int S = a.S;
    const int xcd = (int)blockIdx.x & 7, bi = (int)blockIdx.x >> 3;
    const int slice = bi % S, seq = (bi / S) * 8 + xcd;
    const int G = (int)gridDim.x / S;                      // tile sequences
    const int n0 = slice * 8 * CW + wave * CW;             // this wave's first output channel
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);

    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int q = wave + 8 * i, ks = q / RB, p = q % RB;
            const int row = p * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ swz64(row);
            const int m = t * MT + row;
            const unsigned off = m < a.M ? ((unsigned)m * (unsigned)a.spitch + (unsigned)(ks * 32 + chunk * 8)) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(smem + buf * TILE + ks * SLAB + p * 1024), 16, off, 0, 0, 0);
        }
    };
    const auto rs_src2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(RX ? a.src2 : a.src), 0, RX ? a.src2_bytes : src_bytes, 0x00020000);
    unsigned char* const smem2 = smem + 2 * TILE + STGB;       // EPI 4: the two [MT x KK2] tiles of the second source
    auto issue_tile2 = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NAP2; ++i) {
            const int q = wave + 8 * i, ks = q / RB, p = q % RB;
            const int row = p * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ swz64(row);
            const int m = t * MT + row;
            const unsigned off = m < a.M ? ((unsigned)m * (unsigned)a.spitch2 + (unsigned)(ks * 32 + chunk * 8)) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src2, (lds_void*)(smem2 + buf * TILE2 + ks * SLAB + p * 1024), 16, off, 0, 0, 0);
        }
    };
    // epilogue operands of one tile: vector k of this lane = row k * RPI + lane / LPR, channels n0 + (lane % LPR) * 8 ..
    const unsigned char* const dummy = a.wgt;
    const bool has_res = (DG || EPI == 2) && a.res != nullptr, has_rb = DG && a.res_bits != nullptr;
    const bool has_bx = EPI == 1 && a.bw_x != nullptr, has_bb = DG && a.bw_bits != nullptr;
    const bool has_aff2 = EPI == 2 && a.aff2 != nullptr;
    constexpr int NOP = (DG || EPI == 2) ? NST : 1;
    constexpr int NOX = EPI == 1 ? NST : 1;
    constexpr int NOB = DG ? NST : 1;
    const unsigned char* const p_res = has_res ? a.res : dummy; const size_t m_res = has_res ? ~(size_t)0 : 0;
    const unsigned char* const p_bx = has_bx ? a.bw_x : dummy; const size_t m_bx = has_bx ? ~(size_t)0 : 0;
    const unsigned char* const p_rb = has_rb ? a.res_bits : dummy; const size_t m_rb = has_rb ? ~(size_t)0 : 0;
    const unsigned char* const p_bb = has_bb ? a.bw_bits : dummy; const size_t m_bb = has_bb ? ~(size_t)0 : 0;
    auto load_ops = [&](int t, u32x4 (&r)[NOP], u32x4 (&x)[NOX], unsigned (&rbv)[NOB], unsigned (&mbv)[NOB]) {
        if constexpr (EPI == 2) {                      // the residual rows only (last forward use of the block input: streamed)
            const size_t base = ((size_t)t * MT * a.dpitch + n0) * 2;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const size_t o = base + ((size_t)(k * RPI + lane / LPR) * a.dpitch + (lane % LPR) * 8) * 2;
                r[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p_res + (o & m_res)));      // (straight-line: see below)
            }
        }
        if constexpr (DG) {
            // base pointer and offset mask per operand instead of a select per load: the loads stay straight-line code (a branchy
            // form - some paths with fewer loads - makes the compiler's own wait in front of the first use of these registers a
            // vmcnt(0), i.e. a wait for the tile's STORES: seen in the EPI 4 instances, 373 instead of ~200 us at 56 x 56)
            const size_t base = ((size_t)t * MT * a.dpitch + n0) * 2;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const size_t o = base + ((size_t)(k * RPI + lane / LPR) * a.dpitch + (lane % LPR) * 8) * 2;
                r[k] = *reinterpret_cast<const u32x4*>(p_res + (o & m_res));
                if constexpr (EPI == 1) x[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p_bx + (o & m_bx)));
                rbv[k] = *(p_rb + ((o >> 4) & m_rb));
                mbv[k] = *(p_bb + ((o >> 4) & m_bb));
            }
        }
    };
    u32x4 c_res[NOP], c_x[NOX], n_res[NOP], n_x[NOX];
    unsigned c_rb[NOB], c_mb[NOB], n_rb[NOB], n_mb[NOB];
    float bmean[8], bistd[8];
    float ra[8], rb2[8];                               // EPI 2: bmean / bistd hold (a, b) of the unit, ra / rb2 those of the residual's BN
#pragma unroll
    for (int q = 0; q < 8; ++q) { bmean[q] = 0.f; bistd[q] = 0.f; }
    if (has_bx || RX) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { bmean[q] = a.bw_stats[n0 + (lane % LPR) * 8 + q]; bistd[q] = a.bw_stats[a.dpitch + n0 + (lane % LPR) * 8 + q]; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ra[q] = 1.f; rb2[q] = 0.f; }
    if constexpr (EPI == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ch = n0 + (lane % LPR) * 8 + q;
            bmean[q] = a.aff[2 * a.Cd + ch]; bistd[q] = a.aff[3 * a.Cd + ch];
            if (has_aff2) { ra[q] = a.aff2[2 * a.Cd + ch]; rb2[q] = a.aff2[3 * a.Cd + ch]; }
        }
    }
    int tile = seq;
    if (tile < a.mtiles) { issue_tile(tile, 0); issue_tile2(tile, 0); load_ops(tile, c_res, c_x, c_rb, c_mb); }
    u32x4 w3reg[NK2][CB];                              // EPI 4: this wave's columns of the upstream conv3, all KK2 input channels
    if constexpr (RX) {
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                w3reg[ks][cb] = *reinterpret_cast<const u32x4*>(a.w3 + ((size_t)(n0 + cb * 16 + fr) * a.ldw3 + ks * 32 + fc * 8) * 2);
    }
    u32x4 wreg[NK][CB];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            wreg[ks][cb] = *reinterpret_cast<const u32x4*>(a.wgt + ((size_t)(n0 + cb * 16 + fr) * a.ldw + ks * 32 + fc * 8) * 2);
    // The weights (and the first tile's epilogue operands) are made to ARRIVE here: left pending, the compiler's wait-count pass
    // merges "weights still in flight" into the loop and waits for them in front of every MFMA group - vmcnt(15) ... vmcnt(0),
    // which in steady state are waits for the NEXT tile's prefetch.
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+v"(wreg[ks][cb]));
    if constexpr (EPI == 1) {
#pragma unroll
        for (int k = 0; k < NST; ++k) asm volatile("" : "+v"(c_res[k]), "+v"(c_x[k]), "+v"(c_rb[k]), "+v"(c_mb[k]));
    }
    if constexpr (RX) {
#pragma unroll
        for (int ks = 0; ks < NK2; ++ks)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+v"(w3reg[ks][cb]));
#pragma unroll
        for (int k = 0; k < NST; ++k) asm volatile("" : "+v"(c_res[k]), "+v"(c_rb[k]), "+v"(c_mb[k]));
    }
    if constexpr (EPI == 2) {
#pragma unroll
        for (int k = 0; k < NST; ++k) asm volatile("" : "+v"(c_res[k]));
    }
    int xo[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { const int row = rb * 16 + fr; xo[rb] = row * 64 + ((fc ^ swz64(row)) << 4); }
    float bs[8], bq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs[q] = 0.f; bq[q] = 0.f; }
    // EPI 3: this lane's running sums of channels n0 + cb * 16 + fc * 4 + {0..3} over its rows (fr + 16 rb of every tile)
    f32x4 as[EPI == 3 ? CB : 1], aq[EPI == 3 ? CB : 1];
#pragma unroll
    for (int cb = 0; cb < (EPI == 3 ? CB : 1); ++cb) { as[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; aq[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    unsigned char* const stg = smem + 2 * TILE + wave * STG;
    // PRO: this thread's vectors of a tile: vector tid + 512 i = row (tid + 512 i) / VPR, channel vector tid % VPR (the same for every
    // i: VPR divides 512), i.e. K step pcv / 4, chunk pcv % 4 of the row's 64-byte slab line
    const int pcv = PRO ? tid % VPR : 0, prow0 = PRO ? tid / VPR : 0;
    float pa[8], pb[8], pcs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { pa[q] = 0.f; pb[q] = 0.f; pcs[q] = 0.f; }
    if constexpr (PRO) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { pa[q] = a.pro_stats[2 * KK + pcv * 8 + q]; pb[q] = a.pro_stats[3 * KK + pcv * 8 + q]; }
    }
    // PG: P^T tiles of this wave's 32 g~ columns x the 64 a2 channels, and two 16 x 16 tiles of Gram (row block wave & 3, column
    // blocks 2 (wave >> 2) + {0, 1}); lane holds [column li][rows 4 g4 .. + 3] of a tile, i.e. four consecutive a2 channels of one row
    f32x4 pacc[PG ? CB : 1][PG ? 4 : 1], gacc[PG ? 2 : 1];
#pragma unroll
    for (int cb = 0; cb < (PG ? CB : 1); ++cb)
#pragma unroll
        for (int nb = 0; nb < (PG ? 4 : 1); ++nb) pacc[cb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (PG ? 2 : 1); ++j) gacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    // The first tile (and everything the prologue asked for) has landed; inside the loop the wait for the NEXT tile sits at the
    // bottom of the body, behind the stores it counts over (round 6: at the top behind a `first` flag before - two paths into the
    // loop body made the compiler's own waits conservative, vmcnt(0), in some instances).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; tile < a.mtiles; tile += G, buf ^= 1) {
        __builtin_amdgcn_s_barrier();
        if (tile + G < a.mtiles) { issue_tile(tile + G, buf ^ 1); issue_tile2(tile + G, buf ^ 1); load_ops(tile + G, n_res, n_x, n_rb, n_mb); }
        if constexpr (PRO) {
            // normalise + ReLU the raw tile in place (every slice block does, for its own MFMAs), slice 0 writes the activation out
            unsigned char* Aw = smem + buf * TILE;
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                const int row = prow0 + (512 / VPR) * i;
                unsigned char* lp = Aw + (pcv >> 2) * SLAB + row * 64 + (((pcv & 3) ^ swz64(row)) << 4);
                u32x4 v = *reinterpret_cast<const u32x4*>(lp);
                unsigned bits = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = fmaf(pa[2 * q], bf16_bits_to_f32(v[q] & 0xffffu), pb[2 * q]);
                    const float hi = fmaf(pa[2 * q + 1], __uint_as_float(v[q] & 0xffff0000u), pb[2 * q + 1]);
                    bits |= (lo > 0.f ? 1u : 0u) << (2 * q);
                    bits |= (hi > 0.f ? 1u : 0u) << (2 * q + 1);
                    v[q] = pack_bf16x2(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
                    pcs[2 * q] += bf16_bits_to_f32(v[q] & 0xffffu); pcs[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u);
                }
                *reinterpret_cast<u32x4*>(lp) = v;
                if (slice == 0) {
                    const size_t o = ((size_t)(tile * MT + row) * KK + pcv * 8) * 2;
                    *reinterpret_cast<u32x4*>(a.pro_out + o) = v;
                    a.pro_bits[o >> 4] = (unsigned char)bits;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // the tile is the activation for every wave
        }
        const unsigned char* Ab = smem + buf * TILE;
        f32x4 acc[CB][RB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            u32x4 xf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) xf[rb] = *reinterpret_cast<const u32x4*>(Ab + ks * SLAB + xo[rb]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[ks][cb]),
                                                                         __builtin_bit_cast(bf16x8, xf[rb]), acc[cb][rb], 0, 0, 0);
        }
        // lane holds channels cb * 16 + fc * 4 + {0..3} of row rb * 16 + fr
        if constexpr (EPI == 3) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { as[cb][j] += acc[cb][rb][j]; aq[cb][j] = fmaf(acc[cb][rb][j], acc[cb][rb][j], aq[cb][j]); }
            // the next tile (behind it only the prologue's stores of slice 0, which may stay in flight)
            if (PRO && slice == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPV) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                u32x2 w;
                w.x = pack_bf16x2(acc[cb][rb].x, acc[cb][rb].y);
                w.y = pack_bf16x2(acc[cb][rb].z, acc[cb][rb].w);
                *reinterpret_cast<u32x2*>(stg + (rb * 16 + fr) * PITCH + (cb * 16 + fc * 4) * 2) = w;
            }
        unsigned char* const dcol = a.dst + ((size_t)tile * MT * a.dpitch + n0) * 2;
        u32x4 gv[RX ? NST : 1];
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int row = k * RPI + lane / LPR, chunk = lane % LPR;
            u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * PITCH + chunk * 16);
            if constexpr (EPI == 2) {                   // (the arithmetic of bn_apply_kernel, element for element)
                const u32x4 rr = c_res[k];
                unsigned bits = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float lo = fmaf(bmean[2 * q], bf16_bits_to_f32(v[q] & 0xffffu), bistd[2 * q]);
                    float hi = fmaf(bmean[2 * q + 1], __uint_as_float(v[q] & 0xffff0000u), bistd[2 * q + 1]);
                    if (has_aff2) {
                        lo += fmaf(ra[2 * q], bf16_bits_to_f32(rr[q] & 0xffffu), rb2[2 * q]);
                        hi += fmaf(ra[2 * q + 1], __uint_as_float(rr[q] & 0xffff0000u), rb2[2 * q + 1]);
                    } else if (has_res) {
                        lo += bf16_bits_to_f32(rr[q] & 0xffffu); hi += __uint_as_float(rr[q] & 0xffff0000u);
                    }
                    bits |= (lo > 0.f ? 1u : 0u) << (2 * q);
                    bits |= (hi > 0.f ? 1u : 0u) << (2 * q + 1);
                    v[q] = pack_bf16x2(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
                }
                const size_t ob = ((size_t)row * a.dpitch + chunk * 8) * 2;
                IIF_REGW_ST(v, reinterpret_cast<u32x4*>(dcol + ob));
                a.relu_out[(((size_t)tile * MT * a.dpitch + n0) * 2 + ob) >> 4] = (unsigned char)bits;
            } else if constexpr (DG) {                  // (the arithmetic of staged_drain, element for element)
                if (has_res) {
                    const u32x4 rr = c_res[k];
                    const unsigned rb = has_rb ? c_rb[k] : 0xffu;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu) + ((rb >> (2 * q)) & 1u ? bf16_bits_to_f32(rr[q] & 0xffffu) : 0.f);
                        const float hi = __uint_as_float(v[q] & 0xffff0000u) + ((rb >> (2 * q + 1)) & 1u ? __uint_as_float(rr[q] & 0xffff0000u) : 0.f);
                        v[q] = pack_bf16x2(lo, hi);
                    }
                }
                const unsigned mb = has_bb ? c_mb[k] : 0xffu;
                if (a.mask_store) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned lo = (mb >> (2 * q)) & 1u ? (v[q] & 0xffffu) : 0u;
                        const unsigned hi = (mb >> (2 * q + 1)) & 1u ? (v[q] & 0xffff0000u) : 0u;
                        v[q] = lo | hi;
                        if (!has_bx && !RX) { bs[2 * q] += bf16_bits_to_f32(lo); bs[2 * q + 1] += __uint_as_float(hi); }
                    }
                }
                IIF_REGW_ST(v, reinterpret_cast<u32x4*>(dcol + ((size_t)row * a.dpitch + chunk * 8) * 2));
                if constexpr (RX) {
                    gv[k] = v;                          // meets the recomputed x below
                    if constexpr (PG) *reinterpret_cast<u32x4*>(stg + row * PITCH + chunk * 16) = v;      // g~ as stored, for P (same lane, same slot)
                } else if (has_bx) {
                    const u32x4 xv = c_x[k];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float glo = (mb >> (2 * q)) & 1u ? bf16_bits_to_f32(v[q] & 0xffffu) : 0.f;
                        const float ghi = (mb >> (2 * q + 1)) & 1u ? __uint_as_float(v[q] & 0xffff0000u) : 0.f;
                        const float xlo = (bf16_bits_to_f32(xv[q] & 0xffffu) - bmean[2 * q]) * bistd[2 * q];
                        const float xhi = (__uint_as_float(xv[q] & 0xffff0000u) - bmean[2 * q + 1]) * bistd[2 * q + 1];
                        bs[2 * q] += glo; bq[2 * q] += glo * xlo;
                        bs[2 * q + 1] += ghi; bq[2 * q + 1] += ghi * xhi;
                    }
                } else if (!a.mask_store) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                        bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                        bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
                    }
                }
            } else {
                if (!a.no_store) IIF_REGW_ST(v, reinterpret_cast<u32x4*>(dcol + ((size_t)row * a.dpitch + chunk * 8) * 2));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                    bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                    bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
                }
            }
        }
        if constexpr (PG) {
            // P^T += a2_tile^T g~_tile and Gram += a2_tile^T a2_tile over the tile's MT rows, 32 per MFMA: fragments by transposing
            // reads (16 lanes fetch 4 rows x 16 columns; lane (q, p) supplies row 4 g4 + q, columns 4 p .. + 3) - g~ from this wave's
            // staging slot (row-major, pitch PITCH), a2 from the tile's 32-channel slabs (64-byte rows, chunk ^ swz64(row))
            const unsigned char* A2p = smem2 + buf * TILE2;
            const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
            const int gm = wave & 3, gh = wave >> 2;
#pragma unroll
            for (int kk = 0; kk < MT / 32; ++kk) {
                TrPairR ap[CB], bp[4];
                const int r0 = kk * 32 + 4 * g4 + tq, r1 = r0 + 16;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const unsigned char* base = stg + r0 * PITCH + (cb * 16 + 4 * tp) * 2;
                    ap[cb].lo = tr_read_r(base); ap[cb].hi = tr_read_r(base + 16 * PITCH);
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int ch = nb * 16 + 4 * tp, ks = ch >> 5, chunk = (ch & 31) >> 3, within = (ch & 7) * 2;
                    bp[nb].lo = tr_read_r(A2p + ks * SLAB + r0 * 64 + ((chunk ^ swz64(r0)) << 4) + within);
                    bp[nb].hi = tr_read_r(A2p + ks * SLAB + r1 * 64 + ((chunk ^ swz64(r1)) << 4) + within);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                s16x8 af[CB], bf[4];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) af[cb] = tr_join_r(ap[cb]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bf[nb] = tr_join_r(bp[nb]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        pacc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[nb]), __builtin_bit_cast(bf16x8, af[cb]),
                                                                              pacc[cb][nb], 0, 0, 0);
                // Gram tile (row block gm, column block 2 gh + j): wave-uniform selects of the fragments already read
                const s16x8 grow = gm == 0 ? bf[0] : (gm == 1 ? bf[1] : (gm == 2 ? bf[2] : bf[3]));
                const s16x8 gc0 = gh == 0 ? bf[0] : bf[2], gc1 = gh == 0 ? bf[1] : bf[3];
                gacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gc0), __builtin_bit_cast(bf16x8, grow), gacc[0], 0, 0, 0);
                gacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gc1), __builtin_bit_cast(bf16x8, grow), gacc[1], 0, 0, 0);
            }
        }
        if constexpr (RX) {
            // the upstream block's conv3 on this tile: x[row][n0 + ..] = a2[row][:] . w3[n0 + ..][:], through the same wave-private
            // staging transpose (LDS operations of a wave execute in order: the reads above are done), then sum g~ and sum g~ xhat
            const unsigned char* A2 = smem2 + buf * TILE2;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK2; ++ks) {
                u32x4 xf[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) xf[rb] = *reinterpret_cast<const u32x4*>(A2 + ks * SLAB + xo[rb]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        acc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w3reg[ks][cb]),
                                                                             __builtin_bit_cast(bf16x8, xf[rb]), acc[cb][rb], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    u32x2 w;
                    w.x = pack_bf16x2(acc[cb][rb].x, acc[cb][rb].y);
                    w.y = pack_bf16x2(acc[cb][rb].z, acc[cb][rb].w);
                    *reinterpret_cast<u32x2*>(stg + (rb * 16 + fr) * PITCH + (cb * 16 + fc * 4) * 2) = w;
                }
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int row = k * RPI + lane / LPR, chunk = lane % LPR;
                const u32x4 xv = *reinterpret_cast<const u32x4*>(stg + row * PITCH + chunk * 16);
                const u32x4 v = gv[k];
                const unsigned mb = has_bb ? c_mb[k] : 0xffu;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float glo = (mb >> (2 * q)) & 1u ? bf16_bits_to_f32(v[q] & 0xffffu) : 0.f;
                    const float ghi = (mb >> (2 * q + 1)) & 1u ? __uint_as_float(v[q] & 0xffff0000u) : 0.f;
                    const float xlo = (bf16_bits_to_f32(xv[q] & 0xffffu) - bmean[2 * q]) * bistd[2 * q];
                    const float xhi = (__uint_as_float(xv[q] & 0xffff0000u) - bmean[2 * q + 1]) * bistd[2 * q + 1];
                    bs[2 * q] += glo; bq[2 * q] += glo * xlo;
                    bs[2 * q + 1] += ghi; bq[2 * q + 1] += ghi * xhi;
                }
            }
        }
        if constexpr (EPI == 1) {
#pragma unroll
            for (int k = 0; k < NST; ++k) { c_res[k] = n_res[k]; c_x[k] = n_x[k]; c_rb[k] = n_rb[k]; c_mb[k] = n_mb[k]; }
        }
        if constexpr (RX) {
#pragma unroll
            for (int k = 0; k < NST; ++k) { c_res[k] = n_res[k]; c_rb[k] = n_rb[k]; c_mb[k] = n_mb[k]; }
        }
        if constexpr (EPI == 2) {
#pragma unroll
            for (int k = 0; k < NST; ++k) c_res[k] = n_res[k];
        }
        // the next tile's DMA pieces (and epilogue operands) of THIS wave have landed; this tile's stores may stay in flight
        if (EPI == 0 && a.no_store) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no stores behind the DMA to count over)
        else if (PRO && slice == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + 2 * NPV) : "memory");
        else if (EPI == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");     // NST vectors + NST bit bytes
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    }
    if constexpr (PRO) {
        if (a.pro_csum != nullptr && slice == 0 && seq < G) {
            // threads tid, tid + VPR, ... hold the same channel vector: fixed-order sum through LDS (the tile buffers are free now)
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int q = 0; q < 8; ++q) red[tid * 8 + q] = pcs[q];
            __syncthreads();
            if (tid < KK) {
                const int cv = tid >> 3, q = tid & 7;
                float s = 0.f;
                for (int t = cv; t < 512; t += VPR) s += red[t * 8 + q];
                a.pro_csum[(size_t)seq * 2 * KK + tid] = s;                 // rows laid out like BN partial rows: (sums, zeros)
                a.pro_csum[(size_t)seq * 2 * KK + KK + tid] = 0.f;
            }
        }
    }
    if constexpr (PG) {
        if (a.pg_slab != nullptr && seq < G) {
            // lane holds P[n0 + cb 16 + li][nb 16 + 4 g4 .. + 3] and Gram[gm 16 + li][(2 gh + j) 16 + 4 g4 .. + 3]
            const int g4 = lane >> 4, li = lane & 15;
            float* const sl = a.pg_slab + (size_t)seq * (size_t)(a.Cd + KK2) * a.pg_ld;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    *reinterpret_cast<f32x4*>(sl + (size_t)(n0 + cb * 16 + li) * a.pg_ld + nb * 16 + 4 * g4) = pacc[cb][nb];
            const int gm = wave & 3, gh = wave >> 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(sl + (size_t)(a.Cd + gm * 16 + li) * a.pg_ld + (2 * gh + j) * 16 + 4 * g4) = gacc[j];
        }
    }
    if (a.bn_partial == nullptr || seq >= G) return;
    if constexpr (EPI == 3) {
        // rows of a channel sit on the 16 lanes that share fc: butterfly over fr, lane fr == 0 writes its four channels
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { as[cb][j] += __shfl_xor(as[cb][j], o, 64); aq[cb][j] += __shfl_xor(aq[cb][j], o, 64); }
            }
        if (fr == 0) {
            float* p = a.bn_partial + (int64_t)(a.bn_row0 + seq) * 2 * a.dpitch + n0 + fc * 4;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                *reinterpret_cast<f32x4*>(p + cb * 16) = as[cb];
                *reinterpret_cast<f32x4*>(p + a.dpitch + cb * 16) = aq[cb];
            }
        }
        return;
    }
    // lanes that share the channel chunk (lane % LPR); every wave owns its own columns of the sequence's partial row
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) { bs[q] += __shfl_xor(bs[q], o, 64); bq[q] += __shfl_xor(bq[q], o, 64); }
    }
    if (lane < LPR) {
        float* p = a.bn_partial + (int64_t)(a.bn_row0 + seq) * 2 * a.dpitch + n0 + lane * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) { p[q] = bs[q]; p[a.dpitch + q] = bq[q]; }
    }
}
}  // namespace

// (K, N) -> (columns per wave, rows per tile): the block's 8 waves x CW columns are one N slice, S = N / (8 CW) slices.
// epi: launches with epilogue operands take 32 columns per wave at most (10 registers per staged vector, two tiles' worth).
static bool regw_plan(int K, int N, bool epi, int* cw, int* mt) {
    if (K == 64 && N == 256 && epi) { *cw = 32; *mt = 64; return true; }
    if (K == 128 && N == 512) { *cw = epi ? 32 : 64; *mt = 64; return true; }
    if (K == 256 && N == 1024) { *cw = 32; *mt = 64; return true; }
    // K = 512: 32 columns per wave with 32-row tiles (64-row tiles do not fit the LDS beside the staging buffers): half the
    // LDS-DMA bytes per flop of the 16-column form, which is what bounds these launches (IIF_REGW_K512_CW16: the round-5 form)
    static const bool k512_cw16 = getenv("IIF_REGW_K512_CW16") != nullptr;
    // (with epilogue operands at 7 x 7, M = 12 544: 77 against 69 us with 16 columns, +0.05 ms per step with 32 - the tile kernel keeps it)
    if (K == 512 && N == 2048 && !epi) { *cw = k512_cw16 ? 16 : 32; *mt = k512_cw16 ? 64 : 32; return true; }
    // (the wide -> narrow shapes are level with the tile kernels alone, 59.3 / 60.1 and 43.1 / 42.8 us, and level to +0.05 ms in the step)
    // ResNeXt's conv3 (width -> 2 x width; resnet_pytorch.py:141-143 with groups 32, base width 4): the same kernels, fewer slices
    static const bool no_x2 = getenv("IIF_REGW_NO_X2") != nullptr;      // read once, like every other switch
    if (!no_x2) {
        if (K == 128 && N == 256) { *cw = 32; *mt = 64; return true; }
        if (K == 256 && N == 512) { *cw = 32; *mt = 64; return true; }
        // (the data-gradient epilogue too - ResNeXt-101's 23 producers at 14 x 14: 21.09 -> 20.94 ms; IIF_REGW_K512_NO_EPI: off)
        static const bool k512_no_epi = getenv("IIF_REGW_K512_NO_EPI") != nullptr;
        if (K == 512 && N == 1024 && (!epi || !k512_no_epi)) { *cw = k512_cw16 && !epi ? 16 : 32; *mt = k512_cw16 && !epi ? 64 : 32; return true; }
        if (K == 1024 && N == 2048 && !epi) { *cw = 16; *mt = 32; return true; }
    }
    if (K == 512 && N == 128 && !epi) { *cw = 16; *mt = 64; return true; }
    if (K == 1024 && N == 256 && !epi) { *cw = 16; *mt = 32; return true; }
    // (1024 -> 512, ResNeXt-101's conv1 / conv3 data gradient at 14 x 14: level forward, +0.1 ms with the epilogue; tile kernels keep them)
    return false;
}

// the data-gradient epilogue with the upstream x recomputed over k2 channels: the (K, k2) pairs that have an instance
bool iif_regw1x1_rx_ok(int M, int K, int N, int k2) {
    const bool pair = (K == 64 && k2 == 64) || (K == 128 && (k2 == 64 || k2 == 128)) || (K == 256 && k2 == 128);
    // (M >= 1024: at least eight partial rows' worth of 128-row groups, as for the statistics pass - there is no other kernel to fall back to)
    return pair && M >= 1024 && iif_regw1x1_ok(M, K, N, 1);
}

// ... and P / Gram of the upstream block as by-products: the block must hold every column of g~ (one N slice of 8 x 32 columns)
bool iif_regw1x1_pg_ok(int M, int K, int N, int k2) {
    return k2 == 64 && N == 256 && (K == 64 || K == 128) && iif_regw1x1_rx_ok(M, K, N, k2);
}

bool iif_regw1x1_ok(int M, int K, int N, int epi) {
    int cw, mt;
    if (M <= 0 || (int64_t)M * K * 2 >= 0x7f000000LL || !regw_plan(K, N, epi != 0, &cw, &mt)) return false;
    return M % mt == 0;
}

// the plain forward instances that exist with the BN + ReLU prologue (conv3 of the blocks that store its output: 14 x 14, 7 x 7)
bool iif_regw1x1_pro_ok(int M, int K, int N) {
    int cw = 0, mt = 0;
    if (!iif_regw1x1_ok(M, K, N, 0) || !regw_plan(K, N, false, &cw, &mt)) return false;
    return M >= 1024 && ((K == 256 && cw == 32 && mt == 64) || (K == 512 && ((cw == 16 && mt == 64) || (cw == 32 && mt == 32))));
}

int iif_regw1x1_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int M, int K, int N, int spitch, int ldw, int dpitch, const iif_regw_epilogue* e, int no_store, hipStream_t st,
                       const iif_regw_prologue* pro) {
    const bool epi = e != nullptr;
    if (epi && no_store) return IIF_EUNSUPPORTED;
    if (pro && (epi || no_store || !bn_partial || spitch != K || !pro->stats || !pro->out || !pro->bits || !iif_regw1x1_pro_ok(M, K, N))) return IIF_EUNSUPPORTED;
    if (!src || !wgt || !dst || !iif_regw1x1_ok(M, K, N, epi)) return IIF_EUNSUPPORTED;
    int cw = 0, mt = 0;
    regw_plan(K, N, epi, &cw, &mt);
    const int S = N / (8 * cw);
    RegwArgs a{(const unsigned char*)src, (const unsigned char*)wgt, (unsigned char*)dst, bn_partial, M, M / mt, spitch, ldw, N, dpitch,
               bn_row0, S, nullptr, nullptr, nullptr, nullptr, nullptr, 0, no_store};
    int k2 = 0;
    if (epi) {
        a.res = (const unsigned char*)e->res; a.res_bits = e->res_bits; a.bw_x = (const unsigned char*)e->bw_x; a.bw_bits = e->bw_bits;
        a.bw_stats = e->bw_stats; a.mask_store = e->mask_store;
        if (e->rx_src2) {                                  // the upstream x recomputed from (a2, w3)
            k2 = e->rx_k2;
            if (!iif_regw1x1_rx_ok(M, K, N, k2) || !e->rx_w3 || !e->bw_stats || e->bw_x || e->rx_ldw3 < k2 || bn_partial == nullptr) return IIF_EUNSUPPORTED;
            if ((int64_t)M * k2 * 2 >= 0x7f000000LL) return IIF_EUNSUPPORTED;
            a.src2 = (const unsigned char*)e->rx_src2; a.w3 = (const unsigned char*)e->rx_w3; a.spitch2 = k2; a.ldw3 = e->rx_ldw3;
            a.src2_bytes = (unsigned)((int64_t)M * k2 * 2);
            if (e->pg_slab) {
                if (!iif_regw1x1_pg_ok(M, K, N, k2) || e->pg_ld < k2 || (e->pg_ld & 3) || !e->pg_count ||
                    (reinterpret_cast<uintptr_t>(e->pg_slab) & 15)) return IIF_EUNSUPPORTED;
                a.pg_slab = e->pg_slab; a.pg_ld = e->pg_ld;
            }
        } else if (e->pg_slab) {
            return IIF_EUNSUPPORTED;
        }
    }
    if (pro) { a.pro_stats = pro->stats; a.pro_out = (unsigned char*)pro->out; a.pro_bits = pro->bits; a.pro_csum = pro->csum; }
    const int unit = 8 * S;
    int grid = iif_persistent_grid(unit);
    const int need = (a.mtiles + 7) / 8 * unit;
    if (need < grid) grid = need;
    // one partial row per tile sequence, never more rows than the tile kernels write (ceil(M / 128): what callers size for)
    const int rows128 = (M + 127) / 128;
    if (bn_partial && grid / S > rows128) grid = rows128 / 8 * unit;
    if (grid < unit) return IIF_EUNSUPPORTED;
    const int G = grid / S;
    if (bn_partial) {
        if ((long long)(bn_row0 + G) * 2 * dpitch > bn_cap) return IIF_EINVAL;
        if (rows_out) *rows_out = bn_row0 + G;
    }
    if (a.pg_slab) {
        if ((long long)G * (N + k2) * a.pg_ld > e->pg_cap) return IIF_EINVAL;
        *e->pg_count = G;
    }
    const unsigned sb = (unsigned)((int64_t)M * spitch * 2);
    const dim3 g((unsigned)grid), b(512);
#define IIF_REGW(KK, CW, MT, EP) hipLaunchKernelGGL((gemm1x1_regw_kernel<KK, CW, MT, EP>), g, b, 0, st, a, sb)
#define IIF_REGWX(KK, K2) hipLaunchKernelGGL((gemm1x1_regw_kernel<KK, 32, 64, 4, K2>), g, b, 0, st, a, sb)
    if (k2 && a.pg_slab) {
        if (K == 64) hipLaunchKernelGGL((gemm1x1_regw_kernel<64, 32, 64, 4, 64, false, true>), g, b, 0, st, a, sb);
        else hipLaunchKernelGGL((gemm1x1_regw_kernel<128, 32, 64, 4, 64, false, true>), g, b, 0, st, a, sb);
    } else if (k2) {
        if (K == 64 && k2 == 64) IIF_REGWX(64, 64);
        else if (K == 128 && k2 == 64) IIF_REGWX(128, 64);
        else if (K == 128 && k2 == 128) IIF_REGWX(128, 128);
        else if (K == 256 && k2 == 128) IIF_REGWX(256, 128);
        else return IIF_EUNSUPPORTED;
    } else if (epi) {
        if (K == 64) IIF_REGW(64, 32, 64, 1);
        else if (K == 128) IIF_REGW(128, 32, 64, 1);
        else if (K == 256) IIF_REGW(256, 32, 64, 1);
        else IIF_REGW(512, 32, 32, 1);
    } else if (pro) {
        if (K == 256) hipLaunchKernelGGL((gemm1x1_regw_kernel<256, 32, 64, 0, 0, true>), g, b, 0, st, a, sb);
        else if (cw == 32) hipLaunchKernelGGL((gemm1x1_regw_kernel<512, 32, 32, 0, 0, true>), g, b, 0, st, a, sb);
        else hipLaunchKernelGGL((gemm1x1_regw_kernel<512, 16, 64, 0, 0, true>), g, b, 0, st, a, sb);
    } else {
        if (K == 128 && cw == 64) IIF_REGW(128, 64, 64, 0);
        else if (K == 128) IIF_REGW(128, 32, 64, 0);
        else if (K == 256) IIF_REGW(256, 32, 64, 0);
        else if (K == 512 && cw == 32) IIF_REGW(512, 32, 32, 0);
        else if (K == 512) IIF_REGW(512, 16, 64, 0);
        else IIF_REGW(1024, 16, 32, 0);
    }
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// ---- the two passes of the never-stored conv + BN (+ shortcut) + ReLU forward (round 6): 32 columns per wave, N / 256 slices
static bool regw_plan2(int K, int N) { return (K == 64 || K == 128 || K == 256) && N >= 256 && N <= 2048 && (N % 256) == 0; }

bool iif_regw1x1_fwdbn_ok(int M, int K, int N) {
    // (at least eight 128-row groups: one partial row per tile sequence, never more rows than ceil(M / 128), sequences in eights)
    return M >= 1024 && (M % 64) == 0 && (int64_t)M * K * 2 < 0x7f000000LL && regw_plan2(K, N);
}

// mode 2: forward BN epilogue; mode 3: statistics from the accumulators (dst / res / aff unused)
static int regw_launch2(int mode, const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                        int M, int K, int N, int spitch, int ldw, int dpitch, const void* res, const float* aff, const float* aff2,
                        unsigned char* relu_out, hipStream_t st, const iif_regw_prologue* pro = nullptr) {
    if (!src || !wgt || !iif_regw1x1_fwdbn_ok(M, K, N) || spitch != K || dpitch != N) return IIF_EUNSUPPORTED;
    if (pro && (mode != 3 || !pro->stats || !pro->out || !pro->bits)) return IIF_EINVAL;
    if (mode == 2 && (!dst || !aff || !relu_out || (aff2 && !res))) return IIF_EINVAL;
    if (mode == 3 && !bn_partial) return IIF_EINVAL;
    const int S = N / 256;
    RegwArgs a{(const unsigned char*)src, (const unsigned char*)wgt, (unsigned char*)dst, bn_partial, M, M / 64, spitch, ldw, N, dpitch,
               bn_row0, S, (const unsigned char*)res, nullptr, nullptr, nullptr, nullptr, 0, 0, aff, aff2, relu_out};
    if (pro) { a.pro_stats = pro->stats; a.pro_out = (unsigned char*)pro->out; a.pro_bits = pro->bits; a.pro_csum = pro->csum; }
    const int unit = 8 * S;
    int grid = iif_persistent_grid(unit);
    const int need = (a.mtiles + 7) / 8 * unit;
    if (need < grid) grid = need;
    const int rows128 = (M + 127) / 128;
    if (bn_partial && grid / S > rows128) grid = rows128 / 8 * unit;
    if (grid < unit) return IIF_EUNSUPPORTED;
    const int G = grid / S;
    if (bn_partial) {
        if ((long long)(bn_row0 + G) * 2 * dpitch > bn_cap) return IIF_EINVAL;
        if (rows_out) *rows_out = bn_row0 + G;
    }
    const unsigned sb = (unsigned)((int64_t)M * spitch * 2);
    const dim3 g((unsigned)grid), b(512);
    if (mode == 2) {
        if (K == 64) IIF_REGW(64, 32, 64, 2);
        else if (K == 128) IIF_REGW(128, 32, 64, 2);
        else IIF_REGW(256, 32, 64, 2);
    } else if (pro) {
        if (K == 64) hipLaunchKernelGGL((gemm1x1_regw_kernel<64, 32, 64, 3, 0, true>), g, b, 0, st, a, sb);
        else if (K == 128) hipLaunchKernelGGL((gemm1x1_regw_kernel<128, 32, 64, 3, 0, true>), g, b, 0, st, a, sb);
        else hipLaunchKernelGGL((gemm1x1_regw_kernel<256, 32, 64, 3, 0, true>), g, b, 0, st, a, sb);
    } else {
        if (K == 64) IIF_REGW(64, 32, 64, 3);
        else if (K == 128) IIF_REGW(128, 32, 64, 3);
        else IIF_REGW(256, 32, 64, 3);
    }
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}
#undef IIF_REGW

int iif_regw1x1_fwdbn_launch(const void* src, const void* wgt, void* dst, int M, int K, int N, int spitch, int ldw, int dpitch,
                             const void* res, const float* aff, const float* aff2, unsigned char* relu_out, hipStream_t st) {
    return regw_launch2(2, src, wgt, dst, nullptr, 0, 0, nullptr, M, K, N, spitch, ldw, dpitch, res, aff, aff2, relu_out, st);
}

int iif_regw1x1_stats_launch(const void* src, const void* wgt, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                             int M, int K, int N, int spitch, int ldw, int dpitch, hipStream_t st, const iif_regw_prologue* pro) {
    return regw_launch2(3, src, wgt, nullptr, bn_partial, bn_cap, bn_row0, rows_out, M, K, N, spitch, ldw, dpitch, nullptr, nullptr, nullptr,
                        nullptr, st, pro);
}

// =====================================================================================================================
// 3x3 / stride 1 / pad 1, 64 -> 64 channels (conv2 of the 56 x 56 bottlenecks, forward and data gradient;
// classification/resnet_pytorch.py:156-158) with the weights in registers.
//
// The window kernels run these layers at ~20 % of the matrix pipe: conv3x3_v2n64 fetches four weight fragments per K step
// straight from L2 (46 cycles each to a lone wave) and, like every raster-order tile, pulls a window of 2.3-3.6 x the tile's
// pixels through the LDS-DMA path (a 128-pixel run of a 56-wide image touches 4 rows + 2 halo rows of 58), which is what bounds it:
// the CU's DMA fill rate, ~8.5 B/clk.  Here
//   * a tile is an 8 x 8 pixel SQUARE: its window is 10 x 10 pixels (1.56 x), one 1-KB DMA piece per wave and 32-channel chunk;
//     window rows are a compile-time function of the lane (no per-tile divisions), padding = out-of-range DMA lanes;
//   * the nine taps' fragments of a wave's 32 output channels x all 64 input channels stay in registers for the whole launch
//     (9 x 64 x 32 x 2 B / 64 lanes = 144 VGPRs): the only memory instructions of the tap loop are the fragment reads, 0.5 KB of
//     LDS per MFMA; four waves = 2 column slices x 2 pixel groups of 32, TWO blocks per CU;
//   * persistent blocks, a tile's whole window two tiles ahead in a three-slot ring (counted vmcnt, always the same number of
//     instructions per tile), two barriers per tile; the tile leaves through a block-wide staged transpose (whole 128-byte lines,
//     16 B per lane) with the BN sums - or, for the data gradient, the upstream BN-backward sums - on the way; ONE partial row
//     per block.  K order = chunk-major, tap-minor, as conv3x3_v2n64.
//   Measured alone at [256, 56, 56, 64] (scripts/bm_regw3.py): forward + sums 67 us (conv3x3_v2n64: 102), data gradient + upstream
//   sums 128 (134); stages on the way: raster 128-pixel tiles 98 us (window DMA-bound), 8 x 8 tiles with eight waves 84 -> 78 us.
namespace {
struct Regw3Args {
    const unsigned char* src; const unsigned char* wgt; unsigned char* dst; float* bn_partial;
    const unsigned char* bw_x; const unsigned char* bw_bits; const float* bw_stats;
    int N, H, W, M, ntiles, tiles_x, tiles_per_image, ldw, bn_row0;
    unsigned m_tpi, m_tx;          // magic multipliers of tiles_per_image and tiles_x
    signed char tap_dy[9], tap_dx[9]; unsigned char tap_w[9];
};

__device__ __forceinline__ int fdiv22(int x, float rd) { return (int)(((float)x + 0.5f) * rd); }      // exact for x < 2^22

// magic-number division of a wave-uniform index on the scalar unit (q = floor(t / d) for t < 2^32 / d): the float-reciprocal form costs
// four VALU instructions per use even for uniform operands, and the tile bookkeeping below was a third of this kernel's VALU time
__device__ __forceinline__ unsigned udiv_magic(unsigned t, unsigned m) { return m ? __umulhi(t, m) : t; }      // m = 0: divisor 1

template <bool EPI>
__global__ void __launch_bounds__(256, 2) conv3x3_regw64_kernel(Regw3Args a, unsigned src_bytes) {
    constexpr int C = 64, CW = 32, NCH = 2, CB = 2, PB = 2, TS = 8, WS = TS + 2;      // tile side, window side
    constexpr int HR = 128, CHB = HR * 64, SLOT = NCH * CHB, NSLOT = 3;         // window: 100 pixels in 8 pieces of 16 per 32-channel chunk
    constexpr int PITCH = C * 2 + 16, STG = 64 * PITCH;
    constexpr int NDMA = 2 * NCH;                                               // DMA instructions per wave and tile (pieces w and w + 4)
    constexpr int NV = 2;                                                       // staged 16-byte vectors per thread and tile
    constexpr int NOPS = EPI ? 2 * NV : 0;                                      // epilogue operand loads per thread and tile
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * SLOT + STG + 4 * 2 * C * 4 + 2 * C * 4];
    unsigned char* const stage = smem + NSLOT * SLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int slice = wave & 1, pgrp = wave >> 1;                               // 32 output channels x 32 pixels (four tile rows)
    const int n0 = slice * CW, px0 = pgrp * 32;
    const int H = a.H, W = a.W;
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);

    // tiles of this block: a contiguous range per XCD
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t8 = (a.ntiles + 7) >> 3;
    const int tend = (xcd + 1) * t8 < a.ntiles ? (xcd + 1) * t8 : a.ntiles;
    int tile = xcd * t8 + jb;
    auto tile_origin = [&](int t, int& n, int& y0, int& x0) {                  // scalar unit
        const unsigned tt = (unsigned)__builtin_amdgcn_readfirstlane(t);
        const unsigned nn = udiv_magic(tt, a.m_tpi), r = tt - nn * (unsigned)a.tiles_per_image;
        const unsigned ty = udiv_magic(r, a.m_tx), tx = r - ty * (unsigned)a.tiles_x;
        n = (int)nn; y0 = (int)ty * TS; x0 = (int)tx * TS;
    };

    // this lane's two window pixels (pieces w and w + 4: pixels 16 w .. and 64 + 16 w ..; the last piece is all padding)
    int wyx[2][2], dch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hrl = 16 * (wave + 4 * i) + (lane >> 2);
        wyx[i][0] = hrl < WS * WS ? hrl / WS : -100;              // (a window row nobody has: out of range for every tile)
        wyx[i][1] = hrl - (hrl / WS) * WS;
        dch[i] = ((lane & 3) ^ swz64(hrl)) * 16;
    }
    // The window of tile t goes out two tiles ahead, ALWAYS NDMA instructions per wave (a tile past the range is all out-of-range
    // lanes: zeros into a slot nobody reads), because the waits below count instructions.
    auto issue_window = [&](int t, int slot) {
        int n, y0, x0;
        tile_origin(t < tend ? t : tile, n, y0, x0);
        const bool livet = t < tend;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yy = y0 - 1 + wyx[i][0], xx = x0 - 1 + wyx[i][1];
            const bool ok = livet && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = ok ? (unsigned)((n * H + yy) * W + xx) * (unsigned)(C * 2) + (unsigned)dch[i] : OOB;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(smem + slot * SLOT + c * CHB + (wave + 4 * i) * 1024), 16, off,
                                                         (unsigned)(c * 64), 0, 0);
        }
    };
    issue_window(tile, 0);
    issue_window(tile + per_xcd, 1);
    u32x4 wreg[9][NCH][CB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                wreg[t][c][cb] = *reinterpret_cast<const u32x4*>(a.wgt + ((size_t)(n0 + cb * 16 + fr) * a.ldw + a.tap_w[t] * C + c * 32 + fc * 8) * 2);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+v"(wreg[t][c][cb]));      // arrive here (see gemm1x1_regw_kernel)
    // fragment addresses: pixel p of the tile sits at window row (p / 8 + 1) * 10 + p % 8 + 1; a tap adds dy * 10 + dx
    int faddr[PB][9];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int p = px0 + pb * 16 + fr, hb = ((p >> 3) + 1) * WS + (p & 7) + 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hr = hb + a.tap_dy[t] * WS + a.tap_dx[t];
            faddr[pb][t] = hr * 64 + ((fc ^ swz64(hr)) << 4);
        }
    }
    float bs[8], bq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs[q] = 0.f; bq[q] = 0.f; }
    const unsigned char* const dummy = a.wgt;
    const bool has_bx = EPI && a.bw_x != nullptr, has_bb = EPI && a.bw_bits != nullptr;
    const int vchunk = tid & 7, vrow = tid >> 3;                    // this thread's staged vectors: tile pixels vrow and vrow + 32
    // the upstream statistics wait in LDS (behind the partial-sum scratch): 16 registers less in a kernel that has none to spare
    float* const ustat = reinterpret_cast<float*>(stage + STG + 4 * 2 * C * 4);          // [2][C]: mean, invstd
    if (has_bx && tid < 2 * C) ustat[tid] = a.bw_stats[tid];
    int slot = 0;
    bool first = true;
    for (; tile < tend; tile += per_xcd, slot = slot == NSLOT - 1 ? 0 : slot + 1) {
        // Vector-memory instructions per wave, in issue order (D = a tile's NDMA window pieces, ops = its NOPS operand loads, st = its
        // NV stores):  .. (j-2): D(j) ops(j-2) st(j-2) | (j-1): D(j+1) ops(j-1) st(j-1) | (j): ..   D(j) has NDMA + 2 NOPS + 2 NV younger.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + 2 * NOPS + 2 * NV) : "memory");
        first = false;
        __builtin_amdgcn_s_barrier();                              // the window has landed for every wave; the slot of tile j - 1 is free
        issue_window(tile + 2 * per_xcd, slot == 0 ? NSLOT - 1 : slot - 1);
        int n, y0, x0;
        tile_origin(tile, n, y0, x0);
        size_t vo[NV];
        u32x4 ox[NV];
        unsigned ob[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int p = vrow + 32 * i;
            vo[i] = ((size_t)((n * H + y0 + (p >> 3)) * W + x0 + (p & 7)) * C + vchunk * 8) * 2;
            if (EPI) {                                             // the tile's epilogue operands, unconditionally
                ox[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(has_bx ? a.bw_x + vo[i] : dummy));
                ob[i] = *(has_bb ? a.bw_bits + (vo[i] >> 4) : dummy);
            }
        }
        f32x4 acc[CB][PB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned char* Ab = smem + slot * SLOT + c * CHB;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                u32x4 xf[PB];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const u32x4*>(Ab + faddr[pb][t]);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[t][c][cb]),
                                                                             __builtin_bit_cast(bf16x8, xf[pb]), acc[cb][pb], 0, 0, 0);
            }
        }
        // ---- the tile leaves through the block's staging buffer: lane holds channels n0 + cb*16 + fc*4 + {0..3} of tile pixel px0 + pb*16 + fr
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                u32x2 w;
                w.x = pack_bf16x2(acc[cb][pb].x, acc[cb][pb].y);
                w.y = pack_bf16x2(acc[cb][pb].z, acc[cb][pb].w);
                *reinterpret_cast<u32x2*>(stage + (px0 + pb * 16 + fr) * PITCH + (n0 + cb * 16 + fc * 4) * 2) = w;
            }
        // (not __syncthreads(): that also waits for vmcnt(0), i.e. for the windows just requested two tiles ahead)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (vrow + 32 * i) * PITCH + vchunk * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.dst + vo[i]));      // (whole 128-byte lines: streamed)
            if (EPI && has_bx) {
                const unsigned mb = has_bb ? ob[i] : 0xffu;
                float bmean[8], bistd[8];
#pragma unroll
                for (int q = 0; q < 8; q += 4) {
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(ustat + vchunk * 8 + q), s4 = *reinterpret_cast<const f32x4*>(ustat + C + vchunk * 8 + q);
                    bmean[q] = m4.x; bmean[q + 1] = m4.y; bmean[q + 2] = m4.z; bmean[q + 3] = m4.w;
                    bistd[q] = s4.x; bistd[q + 1] = s4.y; bistd[q + 2] = s4.z; bistd[q + 3] = s4.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float glo = (mb >> (2 * q)) & 1u ? bf16_bits_to_f32(v[q] & 0xffffu) : 0.f;
                    const float ghi = (mb >> (2 * q + 1)) & 1u ? __uint_as_float(v[q] & 0xffff0000u) : 0.f;
                    const float xlo = (bf16_bits_to_f32(ox[i][q] & 0xffffu) - bmean[2 * q]) * bistd[2 * q];
                    const float xhi = (__uint_as_float(ox[i][q] & 0xffff0000u) - bmean[2 * q + 1]) * bistd[2 * q + 1];
                    bs[2 * q] += glo; bq[2 * q] += glo * xlo;
                    bs[2 * q + 1] += ghi; bq[2 * q + 1] += ghi * xhi;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                    bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                    bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
                }
            }
        }
    }
    if (a.bn_partial == nullptr) return;
    // lanes of a wave that share the channel chunk (lane % 8), then the four waves through LDS: fixed order, one row per block
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { bs[q] += __shfl_xor(bs[q], o, 64); bq[q] += __shfl_xor(bq[q], o, 64); }
    }
    float* const scratch = reinterpret_cast<float*>(stage + STG);           // [4 waves][2][C]
    __syncthreads();
    if (lane < 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            scratch[(wave * 2 + 0) * C + lane * 8 + q] = bs[q];
            scratch[(wave * 2 + 1) * C + lane * 8 + q] = bq[q];
        }
    }
    __syncthreads();
    if (tid < C) {
        float s2 = 0.f, q2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s2 += scratch[(w * 2 + 0) * C + tid]; q2 += scratch[(w * 2 + 1) * C + tid]; }
        float* p = a.bn_partial + (int64_t)(a.bn_row0 + blockIdx.x) * 2 * C + tid;
        p[0] = s2; p[C] = q2;
    }
}
}  // namespace

bool iif_regw3x3_ok(int N, int H, int W, int C) {
    if (C != 64 || N <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return false;
    return (int64_t)N * H * W < (1 << 22) && (int64_t)N * H * W * C * 2 < 0x7f000000LL;
}

int iif_regw3x3_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int N, int H, int W, int C, int ldw, const signed char* tap_dy, const signed char* tap_dx, const unsigned char* tap_w,
                       const void* bw_x, const unsigned char* bw_bits, const float* bw_stats, hipStream_t st) {
    if (!src || !wgt || !dst || !iif_regw3x3_ok(N, H, W, C)) return IIF_EUNSUPPORTED;
    const int cus = iif_persistent_cus();
    Regw3Args a{};
    a.src = (const unsigned char*)src; a.wgt = (const unsigned char*)wgt; a.dst = (unsigned char*)dst; a.bn_partial = bn_partial;
    a.bw_x = (const unsigned char*)bw_x; a.bw_bits = bw_bits; a.bw_stats = bw_stats;
    a.N = N; a.H = H; a.W = W; a.M = N * H * W; a.tiles_x = W / 8; a.tiles_per_image = (H / 8) * (W / 8);
    a.ntiles = N * a.tiles_per_image; a.ldw = ldw; a.bn_row0 = bn_row0;
    for (int t = 0; t < 9; ++t) { a.tap_dy[t] = tap_dy[t]; a.tap_dx[t] = tap_dx[t]; a.tap_w[t] = tap_w[t]; }
    a.m_tpi = a.tiles_per_image == 1 ? 0u : (unsigned)(((1ull << 32) + a.tiles_per_image - 1) / a.tiles_per_image);
    a.m_tx = a.tiles_x == 1 ? 0u : (unsigned)(((1ull << 32) + a.tiles_x - 1) / a.tiles_x);
    int grid = 2 * cus / 8 * 8;                                          // two four-wave blocks per CU
    const int need = (a.ntiles + 7) / 8 * 8;
    if (need < grid) grid = need;
    const int rows128 = (a.M + 127) / 128;
    if (bn_partial && grid > rows128) grid = rows128 / 8 * 8;            // never more partial rows than the tile kernels' ceil(M / 128)
    if (grid < 8) return IIF_EUNSUPPORTED;
    if (bn_partial) {
        if ((long long)(bn_row0 + grid) * 2 * C > bn_cap) return IIF_EINVAL;
        if (rows_out) *rows_out = bn_row0 + grid;
    }
    const unsigned sb = (unsigned)((int64_t)a.M * C * 2);
    const bool epi = bw_x != nullptr || bw_bits != nullptr;
    const dim3 g((unsigned)grid), b(256);
    if (epi) hipLaunchKernelGGL(conv3x3_regw64_kernel<true>, g, b, 0, st, a, sb);
    else hipLaunchKernelGGL(conv3x3_regw64_kernel<false>, g, b, 0, st, a, sb);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}
