// Weight gradient of a convolution as an MFMA contraction over pixels (gfx950).
//
//   dW[k][r][s][c] = sum_m dY[m, k] * gather(X)[m; r,s,c]        m = (image, y, x)
//
// Both operands have the reduction index (the pixel) as their slow dimension, so
// the tiles are staged pixel-major ([pixel][channel], exactly as they sit in HBM:
// 16-byte channel pieces, coalesced) and the MFMA fragments are formed by the
// gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) or by ds_read_b32 column
// reads (exact-fp32 mode).  Row pitches (cols*2+32 B bf16, cols*4+64 B f32) make
// both read patterns bank-conflict free.
//
// Block tile: BC output channels x 128 (tap,channel) columns, 32 (bf16) / 16 (f32)
// pixels per K step, double-buffered LDS, register-staged prefetch.  The pixel
// range is split across gridDim.y (split-K) into fp32 slabs that a second kernel
// sums in fixed order: deterministic, no float atomics.
// Replaces the cuDNN/MIOpen wgrad calls autograd issues for
// classification/resnet_pytorch.py:46-62 and resnet_cifar.py:112-115.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#ifndef IIF_WG_AUX_X
#define IIF_WG_AUX_X 0
#endif
#ifndef IIF_WG_AUX_Y
#define IIF_WG_AUX_Y 0
#endif

namespace {

struct WgArgs {
    const unsigned char* x; const unsigned char* dy; float* out;
    int N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, sshift, pad, ldw, M, K;
    int ktiles, ntiles, steps_per_split, nsteps, nsplits;
    int xpitch, ypitch, groups;   // channels per pixel of the x / dy TENSORS (= groups * Cs / Cd)
    int64_t slab;          // elements between split slabs (0 when writing dW directly)
};

template <typename T> struct WT;
template <> struct WT<unsigned short> {
    static constexpr int PE = 8, ROWS = 32;
    static constexpr int pitch(int cols) { return cols * 2 + 32; }
};
template <> struct WT<float> {
    static constexpr int PE = 4, ROWS = 16;
    static constexpr int pitch(int cols) { return cols * 4 + 64; }
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Transposing LDS read, issued as inline asm.  With the builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16) hipcc cannot tell
// which LDS bytes the read touches and puts `s_waitcnt vmcnt(0)` in front of the first read of every K step: ALL LDS-DMA in
// flight - the stages fetched ahead included - drained before every multiply, i.e. no prefetch at all (every kernel of this
// file ran one memory round trip per 32-pixel step: ~2 900 cycles for 256 cycles of MFMA, whatever the ring depth).
// The asm form is invisible to that pass; the price is that its completion is invisible too: tr_fence<N>() =
// `s_waitcnt lgkmcnt(N)` pinned in place must stand between the reads and their first use.
typedef __attribute__((address_space(3))) const unsigned char lds_cu8;
__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
    s16x4 v;
    const unsigned a = (unsigned)(unsigned long long)(lds_cu8*)(p);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void tr_fence() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// The two 64-bit halves of one MFMA fragment, as the transposing reads return them.  The compiler does not know that the
// registers are still being written, so NOTHING may touch them (not even the copy that assembles the 128-bit fragment)
// before the covering tr_fence: the halves stay separate values until tr_join(), which is called AFTER the fence and routes
// them through an empty asm - the fragment is then assembled from values that are defined behind the `s_waitcnt`, and any
// register copy the allocator needs for it reads completed data.  tests/test_cabi.py::test_tr_read_results_are_waited_for
// checks the emitted ISA for exactly this (no instruction names a tr-read destination before the wait that covers it).
struct TrPair { s16x4 lo, hi; };
__device__ __forceinline__ s16x8 tr_join(TrPair& p) {
    asm volatile("" : "+v"(p.lo), "+v"(p.hi));
    return s16x8{p.lo.x, p.lo.y, p.lo.z, p.lo.w, p.hi.x, p.hi.y, p.hi.z, p.hi.w};
}

template <typename T, int BC>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgArgs a) {
    constexpr int PE = WT<T>::PE, ROWS = WT<T>::ROWS;
    constexpr int BNW = 128;
    constexpr int PX = WT<T>::pitch(BNW), PY = WT<T>::pitch(BC);
    constexpr int XB = ROWS * PX, YB = ROWS * PY, BUF = XB + YB;
    constexpr int PPRX = BNW / PE;                 // pieces per X-tile row
    constexpr int RPPX = 256 / PPRX;               // rows covered per pass (X): 16 (bf16) / 8 (f32)
    constexpr int PPRY = BC / PE;
    constexpr int RPPY = 256 / PPRY;
    constexpr int NPY = ROWS / RPPY;               // passes for the dY tile: 2 (BC=128) / 1 (BC=64)
    constexpr int KJ = BC / 32;                    // cout 16-tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;       // wave tile: 64 columns x BC/2 channels
    const int kt = blockIdx.x % a.ktiles, ntile = blockIdx.x / a.ktiles;
    const int k0 = kt * BC, n0 = ntile * BNW;
    const int split = blockIdx.y;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;

    // ---- X gather geometry: this thread's column piece is fixed, its 2 pixel rows advance by ROWS
    const int xchunk = tid % PPRX, xrow0 = tid / PPRX;
    const int ncol = n0 + xchunk * PE;
    const bool nvalid = ncol < a.K;
    int tap = ncol / a.Cs;
    const int cch = ncol - tap * a.Cs;
    const int tr = tap / a.S, ts = tap - tr * a.S;
    int py[2], px[2], pb[2], pm[2];
    const int HW = a.Hd * a.Wd, HWs = a.Hs * a.Ws;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = step0 * ROWS + xrow0 + RPPX * i;
        pm[i] = m;
        const int n = m / HW, rem = m - n * HW;
        py[i] = rem / a.Wd; px[i] = rem - py[i] * a.Wd; pb[i] = n * HWs;
    }
    const int ychunk = tid % PPRY, yrow0 = tid / PPRY;
    const int kcol = k0 + ychunk * PE;
    const bool kvalid = kcol < a.Cd;

    u32x4 rx[2], ry[NPY];
    auto load_step = [&](int step) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ys = (py[i] << a.sshift) - a.pad + tr, xs = (px[i] << a.sshift) - a.pad + ts;
            rx[i] = u32x4{0u, 0u, 0u, 0u};
            if (nvalid && pm[i] < a.M && (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws) {
                const int64_t off = ((int64_t)(pb[i] + ys * a.Ws + xs) * a.Cs + cch) * (int64_t)sizeof(T);
                rx[i] = *reinterpret_cast<const u32x4*>(a.x + off);
            }
        }
#pragma unroll
        for (int i = 0; i < NPY; ++i) {
            const int m = step * ROWS + yrow0 + RPPY * i;
            ry[i] = u32x4{0u, 0u, 0u, 0u};
            if (kvalid && m < a.M) {
                const int64_t off = ((int64_t)m * a.Cd + kcol) * (int64_t)sizeof(T);
                ry[i] = *reinterpret_cast<const u32x4*>(a.dy + off);
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pm[i] += ROWS;
            px[i] += ROWS;
            while (px[i] >= a.Wd) {
                px[i] -= a.Wd;
                if (++py[i] == a.Hd) { py[i] = 0; pb[i] += HWs; }
            }
        }
    };
    auto store_step = [&](int buf) {
        unsigned char* X = smem + buf * BUF;
        unsigned char* Y = X + XB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(X + (xrow0 + RPPX * i) * PX + xchunk * 16) = rx[i];
#pragma unroll
        for (int i = 0; i < NPY; ++i)
            *reinterpret_cast<u32x4*>(Y + (yrow0 + RPPY * i) * PY + ychunk * 16) = ry[i];
    };

    f32x4 acc[4][KJ];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj) acc[ni][kj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    if (step0 < step1) {
        load_step(step0);
        store_step(0);
    }
    __syncthreads();
    for (int step = step0; step < step1; ++step) {
        const bool more = step + 1 < step1;
        if (more) { advance(); load_step(step + 1); }
        const unsigned char* X = smem + ((step - step0) & 1) * BUF;
        const unsigned char* Y = X + XB;
        if constexpr (sizeof(T) == 2) {
            // fragment element j<4 <-> LDS row 4g+j, j>=4 <-> row 16+4g+(j-4) (same map for both operands)
            const int q = li >> 2, p = li & 3;
            TrPair xp[4], yp[KJ];
            s16x8 xf[4], yf[KJ];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const unsigned char* base = X + (4 * g + q) * PX + (wn * 64 + ni * 16 + 4 * p) * 2;
                xp[ni].lo = tr_read(base); xp[ni].hi = tr_read(base + 16 * PX);
            }
#pragma unroll
            for (int kj = 0; kj < KJ; ++kj) {
                const unsigned char* base = Y + (4 * g + q) * PY + (wk * (BC / 2) + kj * 16 + 4 * p) * 2;
                yp[kj].lo = tr_read(base); yp[kj].hi = tr_read(base + 16 * PY);
            }
            tr_fence<0>();
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) xf[ni] = tr_join(xp[ni]);
#pragma unroll
            for (int kj = 0; kj < KJ; ++kj) yf[kj] = tr_join(yp[kj]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj)
                    acc[ni][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, xf[ni]), __builtin_bit_cast(bf16x8, yf[kj]), acc[ni][kj], 0, 0, 0);
        } else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float xf[4], yf[KJ];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    xf[ni] = *reinterpret_cast<const float*>(X + (4 * qq + g) * PX + (wn * 64 + ni * 16 + li) * 4);
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj)
                    yf[kj] = *reinterpret_cast<const float*>(Y + (4 * qq + g) * PY + (wk * (BC / 2) + kj * 16 + li) * 4);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int kj = 0; kj < KJ; ++kj)
                        acc[ni][kj] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[ni], yf[kj], acc[ni][kj], 0, 0, 0);
            }
        }
        if (more) store_step((step + 1 - step0) & 1);
        __syncthreads();
    }

    // D[row = column n'][col = channel k]: lane holds 4 consecutive n' for channel k0 + .. + li
    float* out = a.out + (int64_t)split * a.slab;
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) {
        const int k = k0 + wk * (BC / 2) + kj * 16 + li;
        if (k >= a.Cd) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + g * 4;
            if (n >= a.K) continue;
            *reinterpret_cast<f32x4*>(out + (int64_t)k * a.ldw + n) = acc[ni][kj];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pipelined variant: both tiles arrive by buffer_load ... lds (LDS-DMA), three LDS stages, one
// raw s_barrier per step behind a counted vmcnt.  The DMA destination is lane-linear, so rows are
// contiguous (no padding) and bank conflicts are removed by an XOR swizzle applied to the per-lane
// SOURCE chunk and undone in the read address:
//   bf16 (ds_read_b64_tr_b16, 8 rows x 32 B per half-wave): 32-B slot ^= row&7 (256-B rows) or
//        (row>>1)&3 (128-B rows);   f32 (ds_read_b32 column reads): 64-B group ^= row&1.
// Out-of-image taps / pixels >= M / channels >= Cd are addressed out of range (hardware zeros).
typedef __attribute__((address_space(3))) void lds_void_w;

#ifdef IIF_CONV_STAMPS
__device__ unsigned long long* g_wstamps = nullptr;
#define IIF_WSTAMP(var)                                                                            \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#endif

template <typename T, int COLS> struct Swz {
    static constexpr int RB = COLS * (int)sizeof(T);      // row bytes
    // physical 16-byte chunk p of LDS row `row` holds logical chunk:
    static __device__ __forceinline__ int logical(int p, int row) {
        if constexpr (sizeof(T) == 2) {
            const int key = RB >= 256 ? (row & 7) : ((row >> 1) & 3);
            return (((p >> 1) ^ key) << 1) | (p & 1);
        } else {
            return (((p >> 2) ^ (row & 1)) << 2) | (p & 3);
        }
    }
    // byte address (within the tile) of logical byte column `colb` of row `row`
    static __device__ __forceinline__ int addr(int row, int colb) {
        if constexpr (sizeof(T) == 2) {
            const int key = RB >= 256 ? (row & 7) : ((row >> 1) & 3);
            return row * RB + ((((colb >> 5) ^ key)) << 5) + (colb & 31);
        } else {
            return row * RB + ((((colb >> 6) ^ (row & 1))) << 6) + (colb & 63);
        }
    }
};

// WK wave groups along the output channels (2 x WK waves): 128 kernel columns x BC channels, BC = 64 | 128 with
// WK = 2, or 256 with WK = 4 (8 waves: 12 instead of 16 KB through the vector L1 per MFLOP, see conv_igemm.hip)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NST LDS stages: NST - 1 K steps are in flight while one is multiplied.  Measured (round 4, each shape alone): 4 and 6 stages
// are level with 3 once the prefetch really is in flight (tr_read above), and their LDS footprint costs the step 0.2-0.3 ms.
template <typename T, int BC, int WK = 2, int NST = 3>
__global__ void __launch_bounds__(128 * WK) conv_wgrad_dma_kernel(WgArgs a, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int PE = WT<T>::PE, ROWS = WT<T>::ROWS;
    constexpr int BNW = 128;
    constexpr int NWV = 2 * WK;          // waves
    constexpr int CW = BC / WK;          // output channels per wave
    using SX = Swz<T, BNW>;
    using SY = Swz<T, BC>;
    constexpr int XB = ROWS * SX::RB, YB = ROWS * SY::RB, STAGE = XB + YB;
    constexpr int LPRX = SX::RB / 16, RPIX = 64 / LPRX, NIX = ROWS / RPIX / NWV;   // X: lanes/row, rows/instr, instr/wave
    constexpr int LPRY = SY::RB / 16, RPIY = 64 / LPRY, NIY = ROWS / RPIY / NWV;
    constexpr int LPS = NIX + NIY;
    constexpr int KJ = CW / 16;
    constexpr unsigned OOB = 0xfffffff0u;
    static_assert((NIX == 1 || NIX == 2) && (NIY == 1 || NIY == 2), "tile geometry");
    static_assert(NST >= 3 && NST <= 7 && (NST - 2) * LPS <= 63, "ring depth");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    // XCD-aware map (workgroups are dealt round-robin over the 8 XCDs): all tiles of one pixel split run
    // back-to-back on ONE XCD, so the split's pixel rows of x / dy are fetched from HBM once and re-read
    // from that XCD's L2 by the other tiles.
    const int tiles = a.ktiles * a.ntiles;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int split = (jj / tiles) * 8 + xcd, tile = jj % tiles;
    if (split >= a.nsplits) return;
    const int kt = tile % a.ktiles, ntile = tile / a.ktiles;
    const int k0 = kt * BC, n0 = ntile * BNW;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x), 0, x_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.dy), 0, dy_bytes, 0x00020000);
    const int grp = blockIdx.y;                              // channel group (grouped convolution)
    const unsigned gx = (unsigned)(grp * a.Cs) * (unsigned)sizeof(T), gy = (unsigned)(grp * a.Cd) * (unsigned)sizeof(T);

    // ---- X pieces of this lane: instr i covers tile rows RPIX*(NIX*wave+i) .. +RPIX
    int xr[NIX], py[NIX], px[NIX], pb[NIX], pm[NIX], ttr[NIX], tts[NIX], tch[NIX];
    bool nval[NIX];
    const int HW = a.Hd * a.Wd, HWs = a.Hs * a.Ws;
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        xr[i] = RPIX * (NIX * wave + i) + lane / LPRX;
        const int lc = SX::logical(lane % LPRX, xr[i]);
        const int ncol = n0 + lc * PE;
        nval[i] = ncol < a.K;
        const int tap = ncol / a.Cs;
        tch[i] = ncol - tap * a.Cs;
        ttr[i] = tap / a.S; tts[i] = tap - ttr[i] * a.S;
        const int m = step0 * ROWS + xr[i];
        pm[i] = m;
        const int n = m / HW, rem = m - n * HW;
        py[i] = rem / a.Wd; px[i] = rem - py[i] * a.Wd; pb[i] = n * HWs;
    }
    int yr[NIY];
    unsigned ycol[NIY];
#pragma unroll
    for (int i = 0; i < NIY; ++i) {
        yr[i] = RPIY * (NIY * wave + i) + lane / LPRY;
        const int lc = SY::logical(lane % LPRY, yr[i]);
        const int kcol = k0 + lc * PE;
        ycol[i] = kcol < a.Cd ? (unsigned)kcol * (unsigned)sizeof(T) : OOB;
    }

    auto issue = [&](int stage, int step) {
        unsigned char* X = smem + stage * STAGE;
        unsigned char* Y = X + XB;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int ys = (py[i] << a.sshift) - a.pad + ttr[i], xs = (px[i] << a.sshift) - a.pad + tts[i];
            const bool ok = nval[i] && pm[i] < a.M && (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
            const unsigned off = ok ? ((unsigned)(pb[i] + ys * a.Ws + xs) * (unsigned)a.xpitch + (unsigned)tch[i]) * (unsigned)sizeof(T) + gx : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_w*)(X + (NIX * wave + i) * 1024), 16, off, 0, 0, IIF_WG_AUX_X);
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int m = step * ROWS + yr[i];
            const unsigned off = (ycol[i] != OOB && m < a.M) ? (unsigned)m * (unsigned)a.ypitch * (unsigned)sizeof(T) + ycol[i] + gy : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_w*)(Y + (NIY * wave + i) * 1024), 16, off, 0, 0, IIF_WG_AUX_Y);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            pm[i] += ROWS;
            px[i] += ROWS;
            while (px[i] >= a.Wd) {
                px[i] -= a.Wd;
                if (++py[i] == a.Hd) { py[i] = 0; pb[i] += HWs; }
            }
        }
    };

    f32x4 acc[4][KJ];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj) acc[ni][kj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    const int nst = step1 - step0;
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
        if (s_ < nst) {
            if (s_ > 0) advance();
            issue(s_, step0 + s_);
        }
    int stage = 0;
#ifdef IIF_CONV_STAMPS
    unsigned long long w_wait = 0, w_issue = 0, w_rest = 0, w0, w1, w2, w3, wb;
    IIF_WSTAMP(wb); w3 = wb;
#endif
    for (int t = 0; t < nst; ++t) {
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w0); w_rest += w0 - w3;
#endif
        {   // stage t has landed once at most min(NST - 2, steps after t) younger stages are outstanding
            const int rem = nst - 1 - t;
            if (rem >= NST - 2) wait_vmcnt<(NST - 2) * LPS>();
            else if (NST > 3 && rem == 1) wait_vmcnt<LPS>();
            else if (NST > 4 && rem == 2) wait_vmcnt<2 * LPS>();
            else if (NST > 5 && rem == 3) wait_vmcnt<3 * LPS>();
            else if (NST > 6 && rem == 4) wait_vmcnt<4 * LPS>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w1);
#endif
        if (t + NST - 1 < nst) { advance(); issue(stage == 0 ? NST - 1 : stage - 1, step0 + t + NST - 1); }
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w2); w_wait += w1 - w0; w_issue += w2 - w1; w3 = w2;
#endif
        const unsigned char* X = smem + stage * STAGE;
        const unsigned char* Y = X + XB;
        if constexpr (sizeof(T) == 2) {
            const int q = li >> 2, p = li & 3;
            // LDS returns in order: the dy fragments first, then the x fragments one 16-column block at a time; the MFMAs of
            // block ni start as soon as its two reads are back (6 / 4 / 2 / 0 younger reads still outstanding)
            TrPair xp[4], yp[KJ];
            s16x8 yf[KJ];
#pragma unroll
            for (int kj = 0; kj < KJ; ++kj) {
                const int colb = (wk * CW + kj * 16 + 4 * p) * 2;
                yp[kj].lo = tr_read(Y + SY::addr(4 * g + q, colb)); yp[kj].hi = tr_read(Y + SY::addr(16 + 4 * g + q, colb));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int colb = (wn * 64 + ni * 16 + 4 * p) * 2;
                xp[ni].lo = tr_read(X + SX::addr(4 * g + q, colb)); xp[ni].hi = tr_read(X + SX::addr(16 + 4 * g + q, colb));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (ni == 0) tr_fence<6>(); else if (ni == 1) tr_fence<4>(); else if (ni == 2) tr_fence<2>(); else tr_fence<0>();
                if (ni == 0) {
#pragma unroll
                    for (int kj = 0; kj < KJ; ++kj) yf[kj] = tr_join(yp[kj]);
                }
                const s16x8 xf = tr_join(xp[ni]);
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj)
                    acc[ni][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, yf[kj]), acc[ni][kj], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float xf[4], yf[KJ];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    xf[ni] = *reinterpret_cast<const float*>(X + SX::addr(4 * qq + g, (wn * 64 + ni * 16 + li) * 4));
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj)
                    yf[kj] = *reinterpret_cast<const float*>(Y + SY::addr(4 * qq + g, (wk * CW + kj * 16 + li) * 4));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int kj = 0; kj < KJ; ++kj)
                        acc[ni][kj] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[ni], yf[kj], acc[ni][kj], 0, 0, 0);
            }
        }
        stage = stage == NST - 1 ? 0 : stage + 1;
    }

#ifdef IIF_CONV_STAMPS
    IIF_WSTAMP(w0); w_rest += w0 - w3;
    if (g_wstamps && blockIdx.x < 512 && lane == 0 && wave < 4) {
        unsigned long long* o = g_wstamps + ((int64_t)blockIdx.x * 4 + wave) * 8;
        o[0] = w_wait; o[1] = 0; o[2] = w_issue; o[3] = w_rest; o[4] = 0; o[5] = w0 - wb; o[6] = (unsigned long long)nst; o[7] = wb;
    }
#endif
    float* out = a.out + (int64_t)split * a.slab + (int64_t)grp * a.Cd * a.ldw;
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) {
        const int k = k0 + wk * CW + kj * 16 + li;
        if (k >= a.Cd) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + g * 4;
            if (n >= a.K) continue;
            *reinterpret_cast<f32x4*>(out + (int64_t)k * a.ldw + n) = acc[ni][kj];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 weight gradient (bf16): dW[Cd][Cs] = dy^T x, a plain [M, Cd]^T [M, Cs] contraction over the pixels.
// Same tile (BNW x-columns x BC dy-channels), 32-pixel steps, three LDS stages and split-K slabs as conv_wgrad_dma_kernel,
// with the per-step instruction stream of that kernel (measured ~1 900 cycles per step for 256 cycles of MFMA once its
// prefetch really was in flight) cut to what a 1x1 layer needs:
//   * a lane's DMA source offset only ever advances by 32 pixel rows: one v_add per piece and step (invalid columns keep a
//     fixed out-of-range offset with increment 0; rows >= M fall off the end of the buffer by themselves);
//   * the K loop is unrolled over the three stages, so every fragment read is  lane offset (loop invariant register) +
//     immediate (stage base, +16 rows for the upper half): no address arithmetic at all in front of the 16 LDS reads;
//   * BNW = 64 for layers with <= 64 input channels (the 128-column tile multiplied zeros in half of its MFMAs).
// NWV waves = (BNW / 64) along the x columns x WK along the channels, CW = BC / WK channels per wave.
struct W1Args {
    const unsigned char* x; const unsigned char* dy; float* out;
    int M, Cs, Cd, ldw, ktiles, ntiles, steps_per_split, nsteps, nsplits;
    int64_t slab;
    // stacked dy (iif_wgrad1x1_stacked): output rows >= Cd1 contract the second tensor dy2 [M][Cd2]; Cd = Cd1 + Cd2, Cd1 % BC == 0
    const unsigned char* dy2; int Cd1, Cd2;
};

template <int OFF> __device__ __forceinline__ s16x4 tr_read_at(unsigned a) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}

template <int BC, int BNW, int NWV>
__device__ __forceinline__ void wgrad1x1_body(const W1Args& a, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int NWN = BNW / 64, WK = NWV / NWN, CW = BC / WK, KJ = CW / 16;
    using SX = Swz<unsigned short, BNW>;
    using SY = Swz<unsigned short, BC>;
    constexpr int RBX = SX::RB, RBY = SY::RB;                     // row bytes
    constexpr int XB = 32 * RBX, YB = 32 * RBY, STAGE = XB + YB;
    constexpr int NPX = XB / 1024, NPY = YB / 1024;               // 1-KB DMA pieces per step
    // YPAD: fewer dy pieces than waves (64 channels x 32 pixels = 4 KB under 8 waves): the waves without one issue an
    // out-of-range piece into a dump kilobyte, so that every wave's load count per step - what the counted waits rely on - is the same
    constexpr bool YPAD = NPY < NWV;
    constexpr int NIX = NPX / NWV, NIY = YPAD ? 1 : NPY / NWV, LPS = NIX + NIY;
    constexpr int LPRX = RBX / 16, RPIX = 64 / LPRX, LPRY = RBY / 16, RPIY = 64 / LPRY;
    constexpr unsigned OOB = 0xfffffff0u;
    static_assert(NIX >= 1 && NIY >= 1 && NIX * NWV == NPX && (YPAD || NIY * NWV == NPY) && KJ >= 1, "tile geometry");
    static_assert(2 * STAGE + 16 * RBY < 65536 && 2 * STAGE + 16 * RBX < 65536, "fragment reads use 16-bit immediates");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * STAGE + (YPAD ? 1024 : 0)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int tiles = a.ktiles * a.ntiles;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;           // all tiles of one pixel split back to back on ONE XCD
    const int split = (jj / tiles) * 8 + xcd, tile = jj % tiles;
    if (split >= a.nsplits) return;
    const int k0 = (tile % a.ktiles) * BC, n0 = (tile / a.ktiles) * BNW;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;
    const int nst = step1 - step0;

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x), 0, x_bytes, 0x00020000);
    // (block-uniform) which dy tensor this block's channel tile lives in, its row pitch in channels and the tile's first channel in it
    const bool second = a.dy2 != nullptr && k0 >= a.Cd1;
    const int ycols = a.dy2 == nullptr ? a.Cd : (second ? a.Cd2 : a.Cd1);
    const int yk0 = second ? k0 - a.Cd1 : k0;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(second ? a.dy2 : a.dy), 0,
                                                        second ? (unsigned)a.M * (unsigned)a.Cd2 * 2u : dy_bytes, 0x00020000);

    // ---- DMA: piece p of a step covers tile rows RPI * p ...; this wave issues pieces NI * wave + i
    unsigned offx[NIX], incx[NIX], offy[NIY], incy[NIY];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int r = RPIX * (NIX * wave + i) + lane / LPRX;
        const int col = n0 + SX::logical(lane % LPRX, r) * 8;
        const bool ok = col < a.Cs;
        offx[i] = ok ? ((unsigned)(step0 * 32 + r) * (unsigned)a.Cs + (unsigned)col) * 2u : OOB;
        incx[i] = ok ? 64u * (unsigned)a.Cs : 0u;
    }
#pragma unroll
    for (int i = 0; i < NIY; ++i) {
        const int r = RPIY * (NIY * wave + i) + lane / LPRY;
        const int col = yk0 + SY::logical(lane % LPRY, r) * 8;
        const bool ok = col < ycols && (!YPAD || wave < NPY);
        offy[i] = ok ? ((unsigned)(step0 * 32 + r) * (unsigned)ycols + (unsigned)col) * 2u : OOB;
        incy[i] = ok ? 64u * (unsigned)ycols : 0u;
    }
    auto issue = [&](auto stage_c) {                       // the next step's tiles into stage S; offsets move on by 32 rows
        constexpr int S = decltype(stage_c)::value;
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_w*)(smem + S * STAGE + (NIX * wave + i) * 1024), 16, offx[i], 0, 0, IIF_WG_AUX_X);
            offx[i] += incx[i];
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            unsigned char* dst = smem + S * STAGE + XB + (NIY * wave + i) * 1024;
            if (YPAD && wave >= NPY) dst = smem + 3 * STAGE;                         // (wave-uniform)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_w*)dst, 16, offy[i], 0, 0, IIF_WG_AUX_Y);
            offy[i] += incy[i];
        }
    };

    // ---- fragment reads: lane offsets inside a stage (the +16-row half and the stage base are immediates)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    unsigned xo[4], yo[KJ];
    const unsigned sbase = (unsigned)(unsigned long long)(lds_cu8*)smem;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) xo[ni] = sbase + (unsigned)SX::addr(4 * g + q, (wn * 64 + ni * 16 + 4 * pp) * 2);
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) yo[kj] = sbase + (unsigned)(XB + SY::addr(4 * g + q, (wk * CW + kj * 16 + 4 * pp) * 2));

    f32x4 acc[4][KJ];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj) acc[ni][kj] = f32x4{0.f, 0.f, 0.f, 0.f};

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // one K step on stage S: this wave's pieces of step t have landed (the next step's stay in flight), meet the other
    // waves, refill the stage read in step t - 1 with step t + 2, multiply
    auto step = [&](auto stage_c, auto refill_c, int t) {
        constexpr int S = decltype(stage_c)::value;
        if (t + 1 < nst) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nst) issue(refill_c);
        TrPair xp[4], yp[KJ];
        s16x8 yf[KJ];
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj) {
            yp[kj].lo = tr_read_at<S * STAGE>(yo[kj]); yp[kj].hi = tr_read_at<S * STAGE + 16 * RBY>(yo[kj]);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            xp[ni].lo = tr_read_at<S * STAGE>(xo[ni]); xp[ni].hi = tr_read_at<S * STAGE + 16 * RBX>(xo[ni]);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni == 0) tr_fence<6>(); else if (ni == 1) tr_fence<4>(); else if (ni == 2) tr_fence<2>(); else tr_fence<0>();
            if (ni == 0) {
#pragma unroll
                for (int kj = 0; kj < KJ; ++kj) yf[kj] = tr_join(yp[kj]);
            }
            const s16x8 xf = tr_join(xp[ni]);
#pragma unroll
            for (int kj = 0; kj < KJ; ++kj)
                acc[ni][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, yf[kj]),
                                                                      acc[ni][kj], 0, 0, 0);
        }
    };
    if (nst > 0) issue(S0{});
    if (nst > 1) issue(S1{});
    for (int t = 0; t < nst; t += 3) {
        step(S0{}, S2{}, t);
        if (t + 1 < nst) step(S1{}, S0{}, t + 1);
        if (t + 2 < nst) step(S2{}, S1{}, t + 2);
    }

    float* out = a.out + (int64_t)split * a.slab;
#pragma unroll
    for (int kj = 0; kj < KJ; ++kj) {
        const int k = k0 + wk * CW + kj * 16 + li;
        if (k >= a.Cd) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + g * 4;
            if (n >= a.Cs) continue;
            *reinterpret_cast<f32x4*>(out + (int64_t)k * a.ldw + n) = acc[ni][kj];
        }
    }
}

// plain kernel names around the body template (hipcc / ROCm 7.2 does not emit the host stub of a __global__ template whose
// body holds generic lambdas: the same workaround as in conv_igemm.hip)
__global__ void __launch_bounds__(512) wgrad1x1_256x128_kernel(W1Args a, unsigned xb, unsigned yb) { wgrad1x1_body<256, 128, 8>(a, xb, yb); }
__global__ void __launch_bounds__(256) wgrad1x1_128x128_kernel(W1Args a, unsigned xb, unsigned yb) { wgrad1x1_body<128, 128, 4>(a, xb, yb); }
__global__ void __launch_bounds__(256) wgrad1x1_64x128_kernel(W1Args a, unsigned xb, unsigned yb) { wgrad1x1_body<64, 128, 4>(a, xb, yb); }
__global__ void __launch_bounds__(256) wgrad1x1_64x64_kernel(W1Args a, unsigned xb, unsigned yb) { wgrad1x1_body<64, 64, 4>(a, xb, yb); }
// <= 64 dy channels x 256 x columns: conv1 of the 56 x 56 bottlenecks (256 -> 64).  As two 64 x 128 tiles every block streamed
// the dy tensor again (617 MB for 514, 342 us in the step, 1.8 TB/s); one tile per block reads both operands once.
__global__ void __launch_bounds__(512) wgrad1x1_64x256_kernel(W1Args a, unsigned xb, unsigned yb) { wgrad1x1_body<64, 256, 8>(a, xb, yb); }

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient with ALL NINE TAPS per block ("halo window", bf16).
//
// The tap-per-tile kernel above moves (128 + BC) * 2 B through the LDS-DMA path per pixel for 128 * BC MACs
// (43 MAC/B at BC = 256) and is bound by DMA issue.  Here a block owns a 64 (ci) x 64 (co) tile of ALL taps:
// 9 * 64 * 64 MACs per pixel for (64 + 64) * 2 B = 144 MAC/B.  The contraction runs over VIRTUAL pixels of the
// zero-padded image, v = n*(H+2)*(W+2) + yp*(W+2) + xp: padding pixels are out-of-range DMA lanes (zeros in LDS,
// free), so tap (r, s) is a plain row shift (r-1)*(W+2) + (s-1) of the x window with no border masks and no
// wrap-around.  x rows live in a 256-row LDS ring (each 32-row block is loaded ONCE and read by the nine taps at
// nine shifts), dy rows in a 128-row ring; per K step every wave issues one 1-KB piece of each, two steps ahead.
// Wave w owns input channels 16w..16w+15: 9 taps x 4 output-channel blocks = 36 MFMAs and 144 accumulator
// registers, 18 + 8 transposing LDS reads per step.  Output: the same split-K slabs as above.
struct WgHaloArgs {
    const unsigned char* x; const unsigned char* dy; float* out;
    int N, H, W, Cs, Cd, ldw;       // Cs / Cd: channels per pixel of the x / dy TENSORS
    int ci_tiles, co_tiles, steps_per_split, nsteps, nsplits, HB;     // HB: ring lead in rows (32 | 64)
    int grouped;                    // grouped convolution in 64-channel chunks: tile t = chunk t (x channels 64 t .., dy channels 64 t ..),
                                    // dW rows 64 t .., columns tap * 64 + ci (the packed block-diagonal layout of the chunk matrices)
    int64_t slab;
};

// XB / YB: 32-row blocks in the x / dy rings, D: (x, dy) block pairs in flight ahead of the step being multiplied.
// A new block may only land on a slot last read two steps ago: D <= XB - 2 - LEAD and D <= YB - 2.
template <int XB, int YB, int D>
__global__ void __launch_bounds__(256) conv3x3_wgrad_halo_kernel(WgHaloArgs a, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int RB = 128;                       // row bytes of both tiles (64 bf16)
    constexpr int XR = 32 * XB, YR = 32 * YB;     // ring rows
    constexpr unsigned OOB = 0xfffffff0u;
    using SW = Swz<unsigned short, 64>;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(XR + YR) * RB];
    unsigned char* const XS = smem;
    unsigned char* const YS = smem + XR * RB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = a.ci_tiles * a.co_tiles;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int split = (jj / tiles) * 8 + xcd, tile = jj % tiles;
    if (split >= a.nsplits) return;
    const int ci0 = a.grouped ? tile * 64 : (tile % a.ci_tiles) * 64, co0 = a.grouped ? tile * 64 : (tile / a.ci_tiles) * 64;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;
    const int nst = step1 - step0;
    const int Wp = a.W + 2, PV = (a.H + 2) * Wp, HW = a.H * a.W;
    const int V0 = step0 * 32, LEAD = a.HB >> 4;          // x blocks ahead of the dy block of the same step (2*HB/32)

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x), 0, x_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.dy), 0, dy_bytes, 0x00020000);

    // ---- DMA lanes: this lane's row inside a 32-row block and its 16-byte chunk
    const int lrow = 8 * wave + (lane >> 3);
    // (image, padded row, padded column) of the lane's next pixel, advanced by 32 virtual pixels per block: cheaper
    // than recomputing the decomposition by reciprocal multiplication (measured: fp32 / fp64 reciprocals 10 % slower)
    struct Cur { int n, yp, xp; };
    // For v < 0 (before the first image) the truncating divisions leave a non-canonical but linearly consistent
    // (n < 0, yp <= 0, xp <= 0): such pixels are out of range anyway, and advance() walks back into canonical form.
    auto start = [&](int v) {
        Cur c; const int t = v + PV; c.n = t / PV - 1; const int rem = t - (c.n + 1) * PV; c.yp = rem / Wp; c.xp = rem - c.yp * Wp;
        return c;
    };
    auto advance = [&](Cur& c) {
        c.xp += 32;
        while (c.xp >= Wp) { c.xp -= Wp; if (++c.yp == a.H + 2) { c.yp = 0; ++c.n; } }
    };
    Cur cx = start(V0 - a.HB + lrow), cy = start(V0 + lrow);
    int xblk = 0, yblk = 0;                                   // next block to load (ring position)
    auto issue_x = [&]() {
        const int prow = (xblk * 32 + lrow) & (XR - 1);
        const int lc = SW::logical(lane & 7, prow);
        const bool ok = (unsigned)cx.n < (unsigned)a.N && (unsigned)(cx.yp - 1) < (unsigned)a.H && (unsigned)(cx.xp - 1) < (unsigned)a.W;
        const unsigned off = ok ? ((unsigned)(cx.n * HW + (cx.yp - 1) * a.W + cx.xp - 1) * (unsigned)a.Cs + (unsigned)(ci0 + lc * 8)) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_w*)(XS + ((xblk * 32 + 8 * wave) & (XR - 1)) * RB), 16, off, 0, 0, IIF_WG_AUX_X);
        advance(cx); ++xblk;
    };
    auto issue_y = [&]() {
        const int prow = (yblk * 32 + lrow) & (YR - 1);
        const int lc = SW::logical(lane & 7, prow);
        const bool ok = (unsigned)cy.n < (unsigned)a.N && (unsigned)(cy.yp - 1) < (unsigned)a.H && (unsigned)(cy.xp - 1) < (unsigned)a.W;
        const unsigned off = ok ? ((unsigned)(cy.n * HW + (cy.yp - 1) * a.W + cy.xp - 1) * (unsigned)a.Cd + (unsigned)(co0 + lc * 8)) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_w*)(YS + ((yblk * 32 + 8 * wave) & (YR - 1)) * RB), 16, off, 0, 0, IIF_WG_AUX_Y);
        advance(cy); ++yblk;
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) acc[t][kj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: the x lead, then the (x, dy) pairs of steps 0 .. D-1 -- every later step issues exactly one pair
    for (int j = 0; j < LEAD; ++j) issue_x();
#pragma unroll
    for (int j = 0; j < D; ++j) { issue_x(); issue_y(); }

    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // LDS byte offsets of this lane's fragment reads at step 0.  A step moves every row by 32: the swizzle key
    // ((row >> 1) & 3) is unchanged, so the offset just advances by 32 rows modulo the ring.
    int xo[9], yo[4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        xo[tap] = SW::addr((a.HB + 4 * g + q + (tap / 3 - 1) * Wp + (tap % 3 - 1)) & (XR - 1), (16 * wave + 4 * p) * 2);
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) yo[kj] = SW::addr(4 * g + q, (16 * kj + 4 * p) * 2);
#ifdef IIF_CONV_STAMPS
    unsigned long long w_wait = 0, w_issue = 0, w_rest = 0, w0, w1, w2, w3, wb;
    IIF_WSTAMP(wb); w3 = wb;
#endif
    for (int t = 0; t < nst; ++t) {
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w0); w_rest += w0 - w3;
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");       // this step's pair has landed, D-1 pairs still in flight
        __builtin_amdgcn_s_barrier();
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w1);
#endif
        issue_x(); issue_y();                                   // blocks t+LEAD+D / t+D: their ring slots were last read at step t-2 or earlier
#ifdef IIF_CONV_STAMPS
        IIF_WSTAMP(w2); w_wait += w1 - w0; w_issue += w2 - w1; w3 = w2;
#endif
        const int xs = t * (32 * RB), ys = t * (32 * RB);
        TrPair yp[4], xp[9];
        s16x8 yf[4];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            yp[kj].lo = tr_read(YS + ((yo[kj] + ys) & (YR * RB - 1))); yp[kj].hi = tr_read(YS + ((yo[kj] + ys + 16 * RB) & (YR * RB - 1)));
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            xp[tap].lo = tr_read(XS + ((xo[tap] + xs) & (XR * RB - 1))); xp[tap].hi = tr_read(XS + ((xo[tap] + xs + 16 * RB) & (XR * RB - 1)));
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // in-order LDS returns: tap `tap` may start once at most 2 * (8 - tap) younger reads are outstanding (the counter has 4 bits)
            if (tap == 0) tr_fence<15>(); else if (tap == 1) tr_fence<14>(); else if (tap == 2) tr_fence<12>();
            else if (tap == 3) tr_fence<10>(); else if (tap == 4) tr_fence<8>(); else if (tap == 5) tr_fence<6>();
            else if (tap == 6) tr_fence<4>(); else if (tap == 7) tr_fence<2>(); else tr_fence<0>();
            if (tap == 0) {
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) yf[kj] = tr_join(yp[kj]);
            }
            const s16x8 xf = tr_join(xp[tap]);
#pragma unroll
            for (int kj = 0; kj < 4; ++kj)
                acc[tap][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, yf[kj]),
                                                                        acc[tap][kj], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the trailing prefetches write LDS: drain before exit
#ifdef IIF_CONV_STAMPS
    IIF_WSTAMP(w0); w_rest += w0 - w3;
    if (g_wstamps && blockIdx.x < 512 && lane == 0 && wave < 4) {
        unsigned long long* o = g_wstamps + ((int64_t)blockIdx.x * 4 + wave) * 8;
        o[0] = w_wait; o[1] = 0; o[2] = w_issue; o[3] = w_rest; o[4] = 0; o[5] = w0 - wb; o[6] = (unsigned long long)nst; o[7] = wb;
    }
#endif

    float* out = a.out + (int64_t)split * a.slab;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        const int k = co0 + kj * 16 + li;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int n = a.grouped ? tap * 64 + wave * 16 + g * 4 : tap * a.Cs + ci0 + wave * 16 + g * 4;
            *reinterpret_cast<f32x4*>(out + (int64_t)k * a.ldw + n) = acc[tap][kj];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the space-to-depth stem (4x4 / stride 1 / pad 2 top-left, 1 bottom-right over [N, H, W, 16] ->
// [N, H, W, 64], bf16): the same virtual-pixel scheme as conv3x3_wgrad_halo_kernel with ALL SIXTEEN TAPS per block.
// An x pixel is only 32 bytes, so the tap-per-tile kernel gathers 16-byte pieces sixteen times over (21 MAC per DMA
// byte); here a 32-pixel block of x is ONE 1-KB piece, loaded once into a 512-row ring and read at sixteen row
// shifts (r-2)*(W+3) + (s-2): 16*16*64 MACs per pixel for 32 + 128 B = 102 MAC/B.  Wave w owns tap row r = w:
// 4 taps x 4 output-channel blocks = 16 MFMAs into 64 accumulator registers per step.  This launch is the tail of
// the backward pass (nothing left to overlap it with), so its duration is exposed one to one.
struct WgStemArgs {
    const unsigned char* x; const unsigned char* dy; float* out;
    int N, H, W, ldw, steps_per_split, nsteps, nsplits, LB, LEAD;     // LB: ring blocks before the step's own, LEAD: blocks ahead in total
    int64_t slab;
};

__global__ void __launch_bounds__(256) conv4x4_s2d_wgrad_kernel(WgStemArgs a, unsigned x_bytes, unsigned dy_bytes) {
    constexpr int XRB = 32, YRB = 128;            // row bytes: 16 / 64 bf16
    constexpr int XR = 512, YR = 128;             // ring rows (16 + 4 blocks of 32)
    constexpr int D = 2;                          // block pairs in flight; needs LEAD + D <= 14 (host-checked)
    constexpr unsigned OOB = 0xfffffff0u;
    using SW = Swz<unsigned short, 64>;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XR * XRB + YR * YRB];
    unsigned char* const XS = smem;
    unsigned char* const YS = smem + XR * XRB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int split = jj * 8 + xcd;
    if (split >= a.nsplits) return;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.nsteps) step1 = a.nsteps;
    const int nst = step1 - step0;
    const int Wp = a.W + 3, Hp = a.H + 3, PV = Hp * Wp, HW = a.H * a.W;
    const int V0 = step0 * 32;

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.x), 0, x_bytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.dy), 0, dy_bytes, 0x00020000);

    // cursors over virtual pixels (see conv3x3_wgrad_halo_kernel); the pixel at padded (yp, xp) is (yp - 2, xp - 2)
    struct Cur { int n, yp, xp; };
    auto start = [&](int v) {
        const int t = v + 32 * PV;                       // v >= -32 * LB > -32 * PV
        Cur c; c.n = t / PV - 32; const int rem = t - (c.n + 32) * PV; c.yp = rem / Wp; c.xp = rem - c.yp * Wp;
        return c;
    };
    auto advance = [&](Cur& c) {
        c.xp += 32;
        while (c.xp >= Wp) { c.xp -= Wp; if (++c.yp == Hp) { c.yp = 0; ++c.n; } }
    };
    const int xrow = lane >> 1, yrow = 8 * wave + (lane >> 3);      // the lane's row inside a 32-row block (x: wave 0 only)
    Cur cx = start(V0 - 32 * a.LB + xrow), cy = start(V0 + yrow);
    int xblk = 0, yblk = 0;
    auto issue_x = [&]() {
        const bool ok = (unsigned)cx.n < (unsigned)a.N && (unsigned)(cx.yp - 2) < (unsigned)a.H && (unsigned)(cx.xp - 2) < (unsigned)a.W;
        const unsigned off = ok ? ((unsigned)(cx.n * HW + (cx.yp - 2) * a.W + cx.xp - 2) * 16u + (unsigned)((lane & 1) * 8)) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_w*)(XS + ((xblk * 32) & (XR - 1)) * XRB), 16, off, 0, 0, IIF_WG_AUX_X);
        advance(cx); ++xblk;
    };
    auto issue_y = [&]() {
        const int prow = (yblk * 32 + yrow) & (YR - 1);
        const int lc = SW::logical(lane & 7, prow);
        const bool ok = (unsigned)cy.n < (unsigned)a.N && (unsigned)(cy.yp - 2) < (unsigned)a.H && (unsigned)(cy.xp - 2) < (unsigned)a.W;
        const unsigned off = ok ? ((unsigned)(cy.n * HW + (cy.yp - 2) * a.W + cy.xp - 2) * 64u + (unsigned)(lc * 8)) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_w*)(YS + ((yblk * 32 + 8 * wave) & (YR - 1)) * YRB), 16, off, 0, 0, IIF_WG_AUX_Y);
        advance(cy); ++yblk;
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) acc[t][kj] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave == 0) for (int j = 0; j < a.LEAD; ++j) issue_x();
#pragma unroll
    for (int j = 0; j < D; ++j) { if (wave == 0) issue_x(); issue_y(); }

    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    int xo[4], yo[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
        xo[s_] = ((32 * a.LB + 4 * g + q + (wave - 2) * Wp + (s_ - 2)) & (XR - 1)) * XRB + 4 * p * 2;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) yo[kj] = SW::addr(4 * g + q, (16 * kj + 4 * p) * 2);
    for (int t = 0; t < nst; ++t) {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0) issue_x();
        issue_y();
        const int xs = t * (32 * XRB), ys = t * (32 * YRB);
        TrPair yp[4], xp[4];
        s16x8 yf[4];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            yp[kj].lo = tr_read(YS + ((yo[kj] + ys) & (YR * YRB - 1))); yp[kj].hi = tr_read(YS + ((yo[kj] + ys + 16 * YRB) & (YR * YRB - 1)));
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            xp[s_].lo = tr_read(XS + ((xo[s_] + xs) & (XR * XRB - 1))); xp[s_].hi = tr_read(XS + ((xo[s_] + xs + 16 * XRB) & (XR * XRB - 1)));
        }
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            if (s_ == 0) tr_fence<6>(); else if (s_ == 1) tr_fence<4>(); else if (s_ == 2) tr_fence<2>(); else tr_fence<0>();
            if (s_ == 0) {
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) yf[kj] = tr_join(yp[kj]);
            }
            const s16x8 xf = tr_join(xp[s_]);
#pragma unroll
            for (int kj = 0; kj < 4; ++kj)
                acc[s_][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, yf[kj]),
                                                                       acc[s_][kj], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float* out = a.out + (int64_t)split * a.slab;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        const int k = kj * 16 + li;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int n = (wave * 4 + s_) * 16 + g * 4;
            *reinterpret_cast<f32x4*>(out + (int64_t)k * a.ldw + n) = acc[s_][kj];
        }
    }
}

// dw[k][n] = sum_s slab[s][k][n] for n < K (pad columns untouched), fixed order.  blockIdx.y selects a chunk
// of `chunk` consecutive slabs; with gridDim.y > 1 the chunk sums go to out + blockIdx.y*slab (second stage
// then runs with the chunk sums as its slabs).
#ifdef IIF_NT_SLABS
#define IIF_SLAB_LD(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))
#else
#define IIF_SLAB_LD(p) (*reinterpret_cast<const f32x4*>(p))
#endif
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* ws, int splits, int chunk, int64_t slab, int Cd,
                                                           int ldw, int K, float* out, int64_t out_stride) {
    const int64_t total4 = (int64_t)Cd * ldw / 4;
    const int s0 = blockIdx.y * chunk;
    int s1 = s0 + chunk; if (s1 > splits) s1 = splits;
    float* dst = out + (int64_t)blockIdx.y * out_stride;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int n = (int)(e % ldw);
        if (n >= K) continue;
        // 16 slabs in flight, requested unconditionally (a slab past the chunk reads the chunk's first one again and is not
        // added), summed in slab order.  Round 4 kept 4 in flight under a run-time trip count: a 32-slab sum was 8 dependent
        // memory round trips of 2-5 us each on the weight-gradient stream; it is 2 now, with the same additions in the same order.
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = s0; j < s1; j += 16) {
            f32x4 t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = IIF_SLAB_LD(ws + (int64_t)(j + q < s1 ? j + q : s0) * slab + e);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 u = s + t[q];
                if (j + q < s1) s = (j + q == s0) ? t[q] : u;
            }
        }
        *reinterpret_cast<f32x4*>(dst + e) = s;
    }
}

// sum the split-K slabs into dw (fixed order)
// splits_req < 0: an automatic round sized for 1/|splits_req| of the device - the caller runs that many weight gradients
// side by side on streams of their own.  Every block writes its accumulator tile once whatever the tiling, so the split-K
// slab traffic of a launch is (blocks in the round) x (tile bytes): 75 MB written + read again by the nine-tap kernel, 32 MB
// by the 256 x 128 1x1 kernel, ~4 GB of the step's 73 (profiles/r5_d_pmc_hbm_traffic.csv).  Two half rounds side by side
// keep the device as full with half of that.
inline int share_of(int splits_req) { return splits_req < 0 ? -splits_req : 1; }

inline int reduce_slabs(float* ws, int64_t ws_bytes, int splits, int64_t slab, int rows, int ldw, int K, float* dw, hipStream_t st) {
    const int64_t total4 = slab / 4;
    const int blocks = (int)(cdiv64(total4, 256) < 2048 ? cdiv64(total4, 256) : 2048);
    // few elements x many slabs: sum chunks of 16 slabs in parallel first (into the slab area itself:
    // chunk c writes slab c, which only chunk 0 reads, and chunk 0's own slab 0 is read before written
    // by the same thread), then one pass over the <= ceil(splits/16) chunk sums
    if (splits > 32 && blocks * 256 < 65536) {
        const int chunk = 16, nch = (splits + chunk - 1) / chunk;
        float* stage = ws + (int64_t)splits * slab;       // needs nch more slabs of workspace
        if ((int64_t)(splits + nch) * slab * 4 <= ws_bytes) {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, nch), dim3(256), 0, st, ws, splits, chunk, slab, rows, ldw, K, stage, slab);
            IIF_LAUNCH_CHECK();
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, 1), dim3(256), 0, st, stage, nch, nch, slab, rows, ldw, K, dw, (int64_t)0);
            IIF_LAUNCH_CHECK();
            return IIF_OK;
        }
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, 1), dim3(256), 0, st, ws, splits, splits, slab, rows, ldw, K, dw, (int64_t)0);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

// all nine taps per block (conv3x3_wgrad_halo_kernel): 3x3 / stride 1 / pad 1, dense, bf16, channels in 64s, W <= 61
inline int launch_wgrad_halo(const WgArgs& a, float* dw, float* ws, int64_t ws_bytes, int splits_req, int64_t x_bytes,
                             int64_t dy_bytes, hipStream_t st) {
    WgHaloArgs h{};
    h.x = a.x; h.dy = a.dy; h.N = a.N; h.H = a.Hd; h.W = a.Wd; h.Cs = a.xpitch; h.Cd = a.ypitch; h.ldw = a.ldw;
    h.grouped = a.groups > 1;
    h.ci_tiles = h.grouped ? a.groups : a.Cs / 64; h.co_tiles = h.grouped ? 1 : a.Cd / 64;
    h.HB = a.Wd + 3 <= 32 ? 32 : 64;
    const int64_t vt = (int64_t)a.N * (a.Hd + 2) * (a.Wd + 2);
    if (vt > 0x7fff0000LL) return -100;            // beyond the 32-bit virtual index: caller falls back
    h.nsteps = (int)((vt + 31) / 32);
    const int tiles = h.ci_tiles * h.co_tiles;
    int splits = splits_req;
    if (splits <= 0) {
        const int per_cu = 2;
        // grouped layers (ResNeXt: the weight gradient is block-diagonal, 8 chunks x 147 KB at 14 x 14 against 51 MB of operands) take
        // a quarter of a round: ResNeXt-101 20.80 / 20.87 (full) -> 20.37 (half) -> 20.36 / 20.29 (quarter) -> 21.0 (eighth) ms per step;
        // dense layers keep the full round (ResNet-50: 16.57 -> 16.65 with half)
        static const int halo_div = [] { const char* e = getenv("IIF_WGRAD_HALO_DIV"); return e ? atoi(e) : 1; }();
        static const int halo_gdiv = [] { const char* e = getenv("IIF_WGRAD_HALO_GROUPED_DIV"); return e ? atoi(e) : 4; }();
        const int hdiv = h.grouped ? halo_gdiv : halo_div;
        splits = 256 * per_cu / share_of(splits_req) / tiles / (hdiv > 0 ? hdiv : 1);
        if (splits < 1) splits = 1;
        const int max_by_work = h.nsteps / 8 > 0 ? h.nsteps / 8 : 1;
        if (splits > max_by_work) splits = max_by_work;
    }
    const int64_t slab = (int64_t)a.groups * a.Cd * a.ldw;
    const int64_t fit = ws ? ws_bytes / (slab * 4) : 0;
    if (splits > fit) splits = (int)fit;
    if (splits < 1) splits = 1;
    if (splits > h.nsteps) splits = h.nsteps;
    if (splits > 65535) splits = 65535;
    h.steps_per_split = (h.nsteps + splits - 1) / splits;
    splits = (h.nsteps + h.steps_per_split - 1) / h.steps_per_split;
    h.slab = splits > 1 ? slab : 0;
    h.out = splits > 1 ? ws : dw;
    h.nsplits = splits;
    const dim3 grid((unsigned)(tiles * ((splits + 7) / 8) * 8));
    // 8-block x ring, 4-block dy ring, two block pairs in flight (48 KB, two blocks per CU).  Deeper rings were
    // measured: <8,8,4> equal at W <= 29, <16,8,6> (one block per CU) 10 % slower at 56x56 -- the issue phase is
    // bound by the DMA path, not by latency.
    hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<8, 4, 2>), grid, dim3(256), 0, st, h, (unsigned)x_bytes, (unsigned)dy_bytes);
    IIF_LAUNCH_CHECK();
    if (splits > 1) return reduce_slabs(ws, ws_bytes, splits, slab, a.groups * a.Cd, a.ldw, a.K, dw, st);
    return IIF_OK;
}

// all sixteen taps per block for the space-to-depth stem (conv4x4_s2d_wgrad_kernel)
inline int launch_wgrad_stem(const WgArgs& a, float* dw, float* ws, int64_t ws_bytes, int splits_req, int64_t x_bytes,
                             int64_t dy_bytes, hipStream_t st) {
    WgStemArgs h{};
    h.x = a.x; h.dy = a.dy; h.N = a.N; h.H = a.Hd; h.W = a.Wd; h.ldw = a.ldw;
    const int Wp = a.Wd + 3;
    h.LB = (2 * Wp + 2 + 31) / 32;
    h.LEAD = h.LB + (31 + Wp + 1) / 32;
    if (h.LEAD + 2 > 14) return -100;                // the 16-block x ring cannot hold the window: caller falls back
    const int64_t vt = (int64_t)a.N * (a.Hd + 3) * Wp;
    if (vt > 0x7fff0000LL) return -100;
    h.nsteps = (int)((vt + 31) / 32);
    int splits = splits_req;
    if (splits <= 0) {
        const int per_cu = 2;
        splits = 256 * per_cu / share_of(splits_req);
        const int max_by_work = h.nsteps / 8 > 0 ? h.nsteps / 8 : 1;
        if (splits > max_by_work) splits = max_by_work;
    }
    const int64_t slab = (int64_t)a.Cd * a.ldw;
    const int64_t fit = ws ? ws_bytes / (slab * 4) : 0;
    if (splits > fit) splits = (int)fit;
    if (splits < 1) splits = 1;
    if (splits > h.nsteps) splits = h.nsteps;
    if (splits > 65535) splits = 65535;
    h.steps_per_split = (h.nsteps + splits - 1) / splits;
    splits = (h.nsteps + h.steps_per_split - 1) / h.steps_per_split;
    h.slab = splits > 1 ? slab : 0;
    h.out = splits > 1 ? ws : dw;
    h.nsplits = splits;
    hipLaunchKernelGGL(conv4x4_s2d_wgrad_kernel, dim3((unsigned)(((splits + 7) / 8) * 8)), dim3(256), 0, st, h, (unsigned)x_bytes,
                       (unsigned)dy_bytes);
    IIF_LAUNCH_CHECK();
    if (splits > 1) return reduce_slabs(ws, ws_bytes, splits, slab, a.Cd, a.ldw, a.K, dw, st);
    return IIF_OK;
}

// 1x1 / stride 1 (wgrad1x1_body): tile by the channel counts, one co-resident round of splits as below
inline int launch_wgrad_1x1(const WgArgs& g, float* dw, float* ws, int64_t ws_bytes, int splits_req, int64_t x_bytes, int64_t dy_bytes,
                            hipStream_t st, const unsigned char* dy2 = nullptr, int cd2 = 0) {
    W1Args a{};
    a.x = g.x; a.dy = g.dy; a.M = g.M; a.Cs = g.Cs; a.Cd = g.Cd; a.ldw = g.ldw;
    a.dy2 = dy2; a.Cd1 = dy2 ? g.Cd - cd2 : g.Cd; a.Cd2 = cd2;
    // (stacked: the first tensor's channels fill whole tiles; the second's last tile may be partly empty)
    const int bc = dy2 ? (a.Cd1 % 256 == 0 ? 256 : 128) : ((g.Cd >= 256 && g.Cd % 256 == 0) ? 256 : (g.Cd <= 64 ? 64 : 128));
    static const bool no_wide = getenv("IIF_WGRAD_NO_64X256") != nullptr;
    const int bnw = (g.Cs <= 64 && bc == 64) ? 64 : ((bc == 64 && g.Cs > 128 && g.Cs % 256 == 0 && !no_wide) ? 256 : 128);      // (256 x 64 as two 4-wave blocks per CU was slower than 256 x 128 with half of its columns empty: 0.115 against 0.106 ms at 56 x 56)
    a.ktiles = (g.Cd + bc - 1) / bc;
    a.ntiles = (g.Cs + bnw - 1) / bnw;
    a.nsteps = (g.M + 31) / 32;
    const int tiles = a.ktiles * a.ntiles;
    // blocks per CU by LDS (3 stages of 32 x (bnw + bc) x 2 bytes) and registers: 256 x 128: one 8-wave block (72 KB);
    // 128 x 128: three (48 KB); 64 x {128, 64}: four
    const int per_cu = bc == 256 ? 1 : (bc == 128 ? 3 : (bnw == 256 ? 2 : 4));
    int splits = splits_req;
    const int64_t slab = (int64_t)g.Cd * g.ldw;
    if (splits <= 0) {
        splits = 256 * per_cu / share_of(splits_req) / tiles;
        if (splits < 1) splits = 1;
        const int max_by_work = a.nsteps / 8 > 0 ? a.nsteps / 8 : 1;
        if (splits > max_by_work) splits = max_by_work;
        // Small outputs write splits x their own size in slabs: a Gram matrix a2^T a2 (16-256 KB) 16-48 MB, P = g~^T a2 of the
        // 56 x 56 stage (64 KB) 16 MB - more than the operands at 14 x 14.  Round 5, same-call A/B of the step (ms): default splits
        // 18.02; Gram 1/4 + other outputs <= 256 KB 1/2: 17.78; Gram 1/8: 17.77; 1/2 for EVERY 1x1 weight gradient 17.86 and for
        // the nine-tap kernel 17.90 (their streams become the critical path).  Gram (x == dy) is forward data a block ahead on the
        // shortcut stream: nothing waits for it.  IIF_WGRAD_GRAM_DIV / IIF_WGRAD_SMALL_DIV = 1 restore the full round of splits.
        static const int small_div = [] { const char* e = getenv("IIF_WGRAD_SMALL_DIV"); return e ? atoi(e) : 2; }();
        static const int gram_div = [] { const char* e = getenv("IIF_WGRAD_GRAM_DIV"); return e ? atoi(e) : 8; }();
        const bool gram = g.x == g.dy && dy2 == nullptr;
        static const int small_kb = [] { const char* e = getenv("IIF_WGRAD_SMALL_KB"); return e ? atoi(e) : 256; }();
        // (late round 6: outputs up to 4 MB - the 14 x 14 / 7 x 7 1x1 layers - take half a round too: 16.75 / 16.80 / 16.84 -> 16.73 / 16.71 /
        // 16.78 ms per step in one call, their slabs halved; a quarter round: 17.2.  IIF_WGRAD_MID_DIV=1 restores the full round)
        static const int mid_div = [] { const char* e = getenv("IIF_WGRAD_MID_DIV"); return e ? atoi(e) : 2; }();
        const int div = gram ? gram_div : (slab * 4 <= ((int64_t)small_kb << 10) ? small_div : (slab * 4 <= (4 << 20) ? mid_div : 1));
        if (div > 1 && splits > 8) { splits /= div; if (splits < 8) splits = 8; }       // (never more splits than before)
    }
    const int64_t fit = ws ? ws_bytes / (slab * 4) : 0;
    if (splits > fit) splits = (int)fit;
    if (splits < 1) splits = 1;
    if (splits > a.nsteps) splits = a.nsteps > 0 ? a.nsteps : 1;
    if (splits > 65535) splits = 65535;
    a.steps_per_split = (a.nsteps + splits - 1) / splits;
    splits = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
    if (splits < 1) splits = 1;
    a.slab = splits > 1 ? slab : 0;
    a.out = splits > 1 ? ws : dw;
    a.nsplits = splits;
    const dim3 grid((unsigned)(tiles * ((splits + 7) / 8) * 8));
    const unsigned xb = (unsigned)x_bytes, yb = (unsigned)dy_bytes;
    if (bc == 256) hipLaunchKernelGGL(wgrad1x1_256x128_kernel, grid, dim3(512), 0, st, a, xb, yb);
    else if (bc == 128) hipLaunchKernelGGL(wgrad1x1_128x128_kernel, grid, dim3(256), 0, st, a, xb, yb);
    else if (bnw == 256) hipLaunchKernelGGL(wgrad1x1_64x256_kernel, grid, dim3(512), 0, st, a, xb, yb);
    else if (bnw == 128) hipLaunchKernelGGL(wgrad1x1_64x128_kernel, grid, dim3(256), 0, st, a, xb, yb);
    else hipLaunchKernelGGL(wgrad1x1_64x64_kernel, grid, dim3(256), 0, st, a, xb, yb);
    IIF_LAUNCH_CHECK();
    if (splits > 1) return reduce_slabs(ws, ws_bytes, splits, slab, g.Cd, g.ldw, g.Cs, dw, st);
    return IIF_OK;
}

template <typename T>
int launch_wgrad(WgArgs a, float* dw, float* ws, int64_t ws_bytes, int splits_req, int64_t x_bytes, int64_t dy_bytes,
                 hipStream_t st) {
    constexpr int ROWS = WT<T>::ROWS;
    // 256-channel tiles (8 waves): bf16, dense, >= 256 output channels, the LDS-DMA path
    // (IIF_CONV_REGSTAGE: every launch on the register-staged kernel, the fallback for operands >= 2 GiB; tests)
    const bool force_v1 = getenv("IIF_CONV_REGSTAGE") != nullptr;       // (read per call: a test flips it)
    const bool dma_ok = !force_v1 && x_bytes < 0x7ffffff0LL && dy_bytes < 0x7ffffff0LL;
    if constexpr (sizeof(T) == 2) {
        // dense: channels in 64s; grouped: 64-channel chunks (the nine-tap tile IS a chunk)
        const bool shape = a.groups == 1 ? (a.Cs % 64 == 0 && a.Cd % 64 == 0 && a.xpitch == a.Cs && a.ypitch == a.Cd)
                                         : (a.Cs == 64 && a.Cd == 64 && a.ldw >= 576);
        const bool halo = dma_ok && shape && a.R == 3 && a.S == 3 && a.sshift == 0 && a.pad == 1 && a.Hs == a.Hd &&
                          a.Ws == a.Wd && a.Wd + 3 <= 64;
        if (halo) {
            const int rc = launch_wgrad_halo(a, dw, ws, ws_bytes, splits_req, x_bytes, dy_bytes, st);
            if (rc != -100) return rc;
        }
        const bool stem = dma_ok && a.groups == 1 && a.R == 4 && a.S == 4 && a.sshift == 0 && a.pad == 2 && a.Hs == a.Hd &&
                          a.Ws == a.Wd && a.Cs == 16 && a.Cd == 64 && a.xpitch == 16 && a.ypitch == 64;
        if (stem) {
            const int rc = launch_wgrad_stem(a, dw, ws, ws_bytes, splits_req, x_bytes, dy_bytes, st);
            if (rc != -100) return rc;
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (dma_ok && a.R == 1 && a.S == 1 && a.sshift == 0 && a.pad == 0 && a.groups == 1 && a.Hs == a.Hd && a.Ws == a.Wd &&
            a.xpitch == a.Cs && a.ypitch == a.Cd)
            return launch_wgrad_1x1(a, dw, ws, ws_bytes, splits_req, x_bytes, dy_bytes, st);
    }
    bool wide = sizeof(T) == 2 && dma_ok && a.groups == 1 && a.Cd >= 256 && a.Cd % 256 == 0;
    const int bc = wide ? 256 : (a.Cd <= 64 ? 64 : 128);
    a.ktiles = (a.Cd + bc - 1) / bc;
    a.ntiles = (a.K + 127) / 128;
    a.nsteps = (a.M + ROWS - 1) / ROWS;
    const int tiles = a.ktiles * a.ntiles;
    int splits = splits_req;
    if (splits <= 0) {
        // one full co-resident round: 256 CUs x (3 | 4) workgroups (LDS 48 | 36 KB each), so every
        // workgroup gets the same number of steps and there is no tail round
        // blocks per CU: 4 (64-channel tile), 3 (128), and ONE 8-wave block for the 256-channel tile: two per CU write twice
        // the split-K slabs for nothing (step 20.67 -> 20.62 ms at one; 0.75 / 1.25 per CU leave a tail round: 21.2 / 20.8;
        // round 3, scripts removed).
        const int slots = 256 * (bc == 256 ? 1 : (bc == 128 ? 3 : 4)) / share_of(splits_req);
        splits = slots / (tiles * a.groups);
        if (splits < 1) splits = 1;
        const int max_by_work = a.nsteps / 8 > 0 ? a.nsteps / 8 : 1;
        if (splits > max_by_work) splits = max_by_work;
    }
    const int64_t slab = (int64_t)a.groups * a.Cd * a.ldw;
    const int64_t fit = ws ? ws_bytes / (slab * 4) : 0;
    if (splits > fit) splits = (int)fit;
    if (splits < 1) splits = 1;
    if (splits > a.nsteps) splits = a.nsteps > 0 ? a.nsteps : 1;
    if (splits > 65535) splits = 65535;
    a.steps_per_split = (a.nsteps + splits - 1) / splits;
    splits = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;
    if (splits < 1) splits = 1;
    a.slab = splits > 1 ? slab : 0;
    a.out = splits > 1 ? ws : dw;
    a.nsplits = splits;
    const dim3 grid(tiles, splits);
    const dim3 grid1d((unsigned)(tiles * ((splits + 7) / 8) * 8), (unsigned)a.groups);
    if (a.groups > 1 && !dma_ok)
        return IIF_EUNSUPPORTED;
    const bool dma = dma_ok;
    if (dma) {
        const unsigned xb = (unsigned)x_bytes, yb = (unsigned)dy_bytes;
        if (bc == 64) hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 64>), grid1d, dim3(256), 0, st, a, xb, yb);
        else if (bc == 256) {
            if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 256, 4>), grid1d, dim3(512), 0, st, a, xb, yb);
        } else hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 128>), grid1d, dim3(256), 0, st, a, xb, yb);
    } else {
        if (bc == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<T, 128>), grid, dim3(256), 0, st, a);
    }
    IIF_LAUNCH_CHECK();
    if (splits > 1) return reduce_slabs(ws, ws_bytes, splits, slab, a.groups * a.Cd, a.ldw, a.K, dw, st);
    return IIF_OK;
}

}  // namespace

extern "C" int iif_conv_wgrad(const iif_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                              int64_t workspace_bytes, int splits, void* stream) {
    if (!d || !x || !dy || !dw) return IIF_EINVAL;
    if (d->n <= 0 || d->hs <= 0 || d->ws <= 0 || d->cs <= 0 || d->hd <= 0 || d->wd <= 0 || d->cd <= 0 ||
        d->r <= 0 || d->s <= 0 || d->pad < 0 || workspace_bytes < 0)
        return IIF_EINVAL;
    if (d->stride != 1 && d->stride != 2) return IIF_EUNSUPPORTED;
    if (d->dtype != IIF_F32 && d->dtype != IIF_BF16) return IIF_EINVAL;
    if (d->transposed) return IIF_EINVAL;
    const int pe = d->dtype == IIF_F32 ? 4 : 8;
    if (d->cs % pe != 0 || d->cd % pe != 0 || d->ldw % 4 != 0 || d->ldw < d->r * d->s * d->cs) return IIF_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dw) |
         reinterpret_cast<uintptr_t>(workspace)) & 15)
        return IIF_EUNSUPPORTED;
    const int64_t M = (int64_t)d->n * d->hd * d->wd;
    if (M > 0x7fffff00LL || (int64_t)d->n * d->hs * d->ws > 0x7fffff00LL) return IIF_EUNSUPPORTED;
    WgArgs a{};
    a.x = (const unsigned char*)x; a.dy = (const unsigned char*)dy;
    a.N = d->n; a.Hs = d->hs; a.Ws = d->ws; a.Cs = d->cs; a.Hd = d->hd; a.Wd = d->wd; a.Cd = d->cd;
    a.R = d->r; a.S = d->s; a.sshift = d->stride - 1; a.pad = d->pad; a.ldw = d->ldw; a.M = (int)M;
    a.K = d->r * d->s * d->cs;
    a.groups = d->groups > 1 ? d->groups : 1;
    a.xpitch = a.groups * d->cs; a.ypitch = a.groups * d->cd;
    if (a.groups > 65535) return IIF_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t esz = d->dtype == IIF_F32 ? 4 : 2;
    const int64_t x_bytes = (int64_t)d->n * d->hs * d->ws * a.xpitch * esz, dy_bytes = M * a.ypitch * esz;
    if (d->dtype == IIF_BF16)
        return launch_wgrad<unsigned short>(a, dw, (float*)workspace, workspace_bytes, splits, x_bytes, dy_bytes, st);
    return launch_wgrad<float>(a, dw, (float*)workspace, workspace_bytes, splits, x_bytes, dy_bytes, st);
}

extern "C" int iif_wgrad1x1_stacked(const void* x, const void* dy, const void* dy2, int64_t m, int cs, int cd1, int cd2, int ldw,
                                    float* out, void* workspace, int64_t workspace_bytes, int splits, void* stream) {
    if (!x || !dy || !dy2 || !out || m <= 0 || cs <= 0 || cd1 <= 0 || cd2 <= 0 || workspace_bytes < 0) return IIF_EINVAL;
    if (cs % 8 || cd1 % 128 || cd2 % 8 || ldw % 4 || ldw < cs) return IIF_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dy2) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return IIF_EUNSUPPORTED;
    if (m > 0x7fffff00LL || m * cs * 2 >= 0x7f000000LL || m * cd1 * 2 >= 0x7f000000LL || m * cd2 * 2 >= 0x7f000000LL) return IIF_EUNSUPPORTED;
    WgArgs a{};
    a.x = (const unsigned char*)x; a.dy = (const unsigned char*)dy;
    a.N = 1; a.Hs = 1; a.Ws = (int)m; a.Cs = cs; a.Hd = 1; a.Wd = (int)m; a.Cd = cd1 + cd2;
    a.R = 1; a.S = 1; a.sshift = 0; a.pad = 0; a.ldw = ldw; a.M = (int)m; a.K = cs; a.groups = 1;
    a.xpitch = cs; a.ypitch = cd1;
    return launch_wgrad_1x1(a, out, (float*)workspace, workspace_bytes, splits, m * cs * 2, m * cd1 * 2, as_stream(stream),
                            (const unsigned char*)dy2, cd2);
}

// sum of n fp32 slabs [rows, ld] (columns < cols) in slab order into out - the split-K reduction of this file for slabs written by
// another kernel (the P / Gram by-product of iif_conv_igemm_dgrad_masksum_rx_pg).  slab_floats: floats available behind `slabs`;
// the two-stage form needs ceil(n / 16) slabs of room behind the n slabs and is used when there is.
extern "C" int iif_slab_sum(float* slabs, int64_t slab_floats, int n, int rows, int ld, int cols, float* out, void* stream) {
    if (!slabs || !out || n < 1 || rows < 1 || ld < cols || cols < 1 || (ld & 3)) return IIF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) return IIF_EINVAL;
    const int64_t slab = (int64_t)rows * ld;
    if ((int64_t)n * slab > slab_floats) return IIF_EINVAL;
    return reduce_slabs(slabs, slab_floats * 4, n, slab, rows, ld, cols, out, (hipStream_t)stream);
}

#ifdef IIF_CONV_STAMPS
extern "C" int iif_debug_set_wgrad_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), &buf, sizeof(buf)) == hipSuccess ? IIF_OK : IIF_ELAUNCH;
}
#endif
