// Fused IIF classifier-head kernels for gfx950 (MI355X).
//
// One 64-lane wavefront owns one row of logits.  For C <= 2048 (every class
// count on the reference's path: 100 / 365 / 1000 / 1204) the whole row lives
// in registers: ONE read of the logits, wave-shuffle max / sum-exp, loss and
// gradient written in the same pass (classification/custom.py:28-36 does the
// same work as >= 6 elementwise/reduction launches plus autograd).
// The kernels are HBM/latency-bound; there is no GEMM shape here, so no MFMA.
//
// Arithmetic: the softmax runs in base 2 on the hardware transcendentals (v_exp_f32 / v_log_f32, ~1 ulp):
// z2 = x * (iif * log2 e), p = 2^(z2 - max z2) / sum, lse = ln 2 * (max z2 + log2 sum).  The lane keeps
// iif * log2 e in registers next to iif, so the base change costs no instruction per element; a full-precision
// expf() here made the kernel VALU-bound (1.9-3.0 TB/s at [65536, 1000]).  The one-hot term of the gradient is
// patched by the one lane that owns the target column instead of being compared in every lane.
// The scalar loss comes out of the SAME launch when the caller passes a ticket word: the last block to finish
// (atomic ticket, agent-scope fence) sums the per-row losses in a fixed order — deterministic, no second launch.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float fast_log2(float v) { return __builtin_amdgcn_logf(v); }

// ---------------------------------------------------------------- element I/O
template <typename T> struct Io;
template <> struct Io<float> {
    static __device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct Io<unsigned short> {  // bf16 bits
    static __device__ __forceinline__ f32x4 load4(const unsigned short* p) {
        u32x2 w = *reinterpret_cast<const u32x2*>(p);
        f32x4 r;
        r.x = bf16_bits_to_f32(w.x & 0xffffu); r.y = __uint_as_float(w.x & 0xffff0000u);
        r.z = bf16_bits_to_f32(w.y & 0xffffu); r.w = __uint_as_float(w.y & 0xffff0000u);
        return r;
    }
    static __device__ __forceinline__ void store4(unsigned short* p, f32x4 v) {
        u32x2 w; w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<u32x2*>(p) = w;
    }
    static __device__ __forceinline__ float load1(const unsigned short* p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void store1(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};

// One lane's V consecutive columns of a row as V floats: 16 bytes per lane for both element types (4 fp32 / 8 bf16), so
// every wave instruction moves whole 1-KB runs.  `nv` (0 < nv <= V, multiple of 4) is the number of columns that exist:
// the bf16 row tail of a class count that is 4 but not 0 modulo 8 (1204) is an 8-byte access.
template <typename T> struct RowIo;
// A row is fetched as RAW 16-byte vectors and unpacked where it is used, so rows in flight cost 4 registers per lane and
// chunk whatever the element type: DEPTH rows are kept in flight per wave (bf16 rows are half the bytes, so two of them
// give a CU the same bytes in flight as one fp32 row: with one, [65536, 1000] bf16 ran at 3.2 TB/s against 4.9 for fp32).
template <> struct RowIo<float> {
    static constexpr int V = 4, DEPTH = 1;
    using Raw = f32x4;
    static __device__ __forceinline__ Raw load_raw(const float* p, int) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void unpack(const Raw& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, int, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
};
template <> struct RowIo<unsigned short> {
    static constexpr int V = 8, DEPTH = 2;
    using Raw = u32x4;
    // 8-byte aligned 16-byte vectors: the rows of a CONTIGUOUS [B, C] matrix with C % 8 == 4 (LVIS: 1204) start on alternate
    // 8-byte boundaries; gfx950 runs with unaligned access enabled, so the access stays one dwordx4 (round 3 / 4 advice: such a
    // matrix used to fall back to the streaming kernel)
    typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
    static __device__ __forceinline__ Raw load_raw(const unsigned short* p, int nv) {
        u32x4 w = u32x4{0u, 0u, 0u, 0u};
        if (nv == 8) w = *reinterpret_cast<const u32x4_a8*>(p);
        else { const u32x2 h = *reinterpret_cast<const u32x2*>(p); w.x = h.x; w.y = h.y; }
        return w;
    }
    static __device__ __forceinline__ void unpack(const Raw& w, float (&v)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = bf16_bits_to_f32(w[q] & 0xffffu); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(unsigned short* p, int nv, const float (&v)[8]) {
        u32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
        if (nv == 8) *reinterpret_cast<u32x4_a8*>(p) = w;
        else *reinterpret_cast<u32x2*>(p) = u32x2{w.x, w.y};
    }
};

struct CeArgs {
    const void* x; int64_t ldx;
    const float* tab;
    const int64_t* ta; const int64_t* tb;
    float lam;
    const float* roww; const float* clsw;
    int64_t ignore; float scale;
    int B, C;
    float* loss_row;
    void* dx; int64_t lddx;
    int32_t* status;
    float* loss_out;       // scalar loss = scale * sum(loss_row), written by the last block when ticket != nullptr
    int32_t* ticket;       // zero on entry, zero again on exit
};

// Single-launch loss reduce.  workspace = int32 ticket (zero on entry / exit) followed by one float per block.
// Every wave adds up the losses of the rows it walked (fixed order), the block combines its waves (fixed order) into
// partial[blockIdx]; the last block to take a ticket sums the <= 2048 partials in a fixed tree.  Deterministic for a
// given (B, C, dtype): the grid depends on nothing else.
__device__ __forceinline__ void finish_with_ticket(const CeArgs& a, float wave_loss) {
    if (a.ticket == nullptr) return;                // block-uniform
    __shared__ float sh[256];
    __shared__ int last;
    float* partial = reinterpret_cast<float*>(a.ticket + 1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if (lane == 0) sh[w] = wave_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < wpb; ++i) acc += sh[i];
        // No release fence here: an agent-scope fence writes back the whole XCD L2, which at this point is full of
        // the gradient rows just stored (it cost 40 % of the kernel).  The partial goes out as an agent-scope atomic
        // exchange (performed at the coherence point; a returning atomic has completed when its value is back), the
        // ticket is taken only after that value has returned, and the last block reads the partials with
        // agent-scope atomic loads: the same ordering without touching the ordinary stores.
        const float prev = __hip_atomic_exchange(partial + blockIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(prev) : "memory");
        const int t = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    float acc = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
        acc += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *a.loss_out = sh[0] * a.scale;
        __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Per-row scalars shared by the register and the streaming variants.
struct RowCoef {
    int64_t ta, tb;      // -1 when the term is inactive
    float wa, wb;        // lam * class_weight terms (0 when inactive)
    float rw;
};

__device__ __forceinline__ RowCoef row_coef(const CeArgs& a, int row) {
    RowCoef r;
    r.rw = a.roww ? a.roww[row] : 1.0f;
    int64_t ta = a.ta[row];
    int64_t tb = a.tb ? a.tb[row] : a.ignore;
    float la = a.tb ? a.lam : 1.0f, lb = a.tb ? 1.0f - a.lam : 0.0f;
    bool bad = false;
    if (ta == a.ignore) { ta = -1; } else if (ta < 0 || ta >= a.C) { ta = -1; bad = true; }
    if (!a.tb || tb == a.ignore) { tb = -1; } else if (tb < 0 || tb >= a.C) { tb = -1; bad = true; }
    if (bad && a.status && threadIdx.x % IIF_WAVE == 0) atomicExch(a.status, 1);
    r.ta = ta; r.tb = tb;
    r.wa = ta >= 0 ? la * (a.clsw ? a.clsw[ta] : 1.0f) : 0.0f;
    r.wb = tb >= 0 ? lb * (a.clsw ? a.clsw[tb] : 1.0f) : 0.0f;
    return r;
}

// ------------------------------------------------ register-resident row (C%4==0)
// MODE 0: CE loss + gradient, MODE 1: softmax output (fp32), MODE 2: MODE 0 without mixup, row / class weights and half vectors
// A wave walks rows wave_id, wave_id + n_waves, ...  The IIF table sits in LDS (one copy per block, <= 8 KB) and the
// next row's logits are already in flight while the current row is reduced and stored (6-8 waves per SIMD at
// C = 1000 / 1204; a register-held table cost 110 VGPRs = half the occupancy).
// Round 4 (the bf16 row loop was instruction-bound: 374 VALU instructions and 5 dependent loads per row, 148 branches in the
// kernel): only the LAST chunk of a row can be ragged, so chunks 0 .. NCH-2 carry no column test at all; the lanes of the last
// chunk that own no column hold -inf (and the table 1.0) there, so the arithmetic needs no per-element select either; the row's
// target comes from a scalar load one row ahead, and the target logit from the row's own registers (v_readlane) instead
// of a dependent global load after the reduction (plain CE: no mixup, no row / class weights - otherwise row_coef).
// (launch bound: 16 row values per lane = C <= 1024 fit 80 registers, i.e. 6 waves per SIMD, when the allocator is told so; at 7 the
// round-4 loop spills 9-14 registers)
template <typename T, int NCH, int MODE>
__global__ void __launch_bounds__(256, (NCH * RowIo<T>::V <= 16 ? 6 : 1)) row_reg_kernel(CeArgs a, float* sm_out, int64_t ld_sm) {
    constexpr int V = RowIo<T>::V;                     // columns per lane and chunk: 16 bytes of T
    constexpr int EPD = V / 4;                         // elements per dword of the raw vector
    constexpr unsigned NEG = sizeof(T) == 2 ? 0xFF80FF80u : 0xFF800000u;       // -inf in every element of a dword
    constexpr int JL = NCH - 1;                        // the only chunk that can be ragged
    __shared__ __attribute__((aligned(16))) float tab_s[NCH * 64 * V];
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nwaves = gridDim.x * wpb;
    int row = blockIdx.x * wpb + (threadIdx.x >> 6);
    constexpr int DEPTH = RowIo<T>::DEPTH;            // rows in flight per wave
    typename RowIo<T>::Raw xr[DEPTH][NCH];
    // columns of the last chunk that exist for this lane: V, or 4 on the tail of a bf16 row with C % 8 == 4, or 0
    const int left = a.C - (JL * 64 + lane) * V;
    const int nl = left >= V ? V : (left > 0 ? left : 0);
    const bool half_tail = MODE != 2 && (a.C % V) != 0;     // block-uniform: some lane's last vector is an 8-byte half (bf16, C % 8 == 4)
    auto load_row = [&](int r, typename RowIo<T>::Raw (&dst)[NCH]) {
        const T* xp = static_cast<const T*>(a.x) + (int64_t)r * a.ldx;
#pragma unroll
        for (int j = 0; j < JL; ++j) dst[j] = RowIo<T>::load_raw(xp + (j * 64 + lane) * V, V);
        if (half_tail) {                               // rare shape: per-lane access width
            if (nl > 0) dst[JL] = RowIo<T>::load_raw(xp + (JL * 64 + lane) * V, nl);
        } else {                                       // unconditional: a lane without columns re-reads the row's first vector
            dst[JL] = RowIo<T>::load_raw(xp + (nl > 0 ? (JL * 64 + lane) * V : 0), V);
        }
    };
    auto mask_tail = [&](typename RowIo<T>::Raw& t) {  // -inf into the columns that do not exist
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q * EPD >= nl) t[q] = __builtin_bit_cast(decltype(t[q] + t[q]), NEG);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (row + d * nwaves < a.B) load_row(row + d * nwaves, xr[d]);        // wave-uniform: in flight while the table is staged
    // table -> LDS in 16-byte pieces, all loads of a thread issued before the first LDS write (a scalar loop cost a
    // single-wave block 20 dependent round trips to L2: 22 us at [1024, 1204]); columns >= C hold 1.0 (-inf * 1 = -inf)
    {
        constexpr int V4 = NCH * 16 * V;                   // float4 pieces of the padded table
        f32x4 tv[(V4 + 63) / 64];
#pragma unroll
        for (int q = 0; q < (V4 + 63) / 64; ++q) {
            const int i = threadIdx.x + q * blockDim.x;
            tv[q] = (i < V4 && i * 4 < a.C) ? *reinterpret_cast<const f32x4*>(a.tab + i * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int q = 0; q < (V4 + 63) / 64; ++q) {
            const int i = threadIdx.x + q * blockDim.x;
            if (i < V4) *reinterpret_cast<f32x4*>(tab_s + i * 4) = tv[q];
        }
    }
    __syncthreads();
    constexpr bool plain = MODE == 2;                   // the training loss of classification/train.py
    // the row's target one row ahead, through the scalar cache (the row index is wave-uniform)
    int64_t ta_next = 0;
    if (plain && row < a.B) ta_next = a.ta[__builtin_amdgcn_readfirstlane(row)];
    float wave_loss = 0.f;
    for (; row < a.B; row += nwaves) {
        asm volatile("" ::: "memory");            // keep the table reads in LDS: hoisted into registers they halve the occupancy
        const T* x = static_cast<const T*>(a.x) + (int64_t)row * a.ldx;
        const int64_t ta_cur = ta_next;
        if (plain && row + nwaves < a.B) ta_next = a.ta[__builtin_amdgcn_readfirstlane(row + nwaves)];
        mask_tail(xr[0][JL]);
        // plain CE: the target logit out of the row's registers (lane, chunk and element are wave-uniform)
        float xt = 0.f;
        bool t_ok = false;
        if (plain) {
            t_ok = ta_cur != a.ignore && ta_cur >= 0 && ta_cur < a.C;
            const int t = t_ok ? (int)ta_cur : 0;
            const int tj = __builtin_amdgcn_readfirstlane(t / (64 * V)), tl = __builtin_amdgcn_readfirstlane((t / V) & 63);
            const int te = __builtin_amdgcn_readfirstlane(t % V);
            unsigned w = 0;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (tj == j && te / EPD == q) {
                        const auto el = xr[0][j][q];          // by value: bit_cast of a vector-element lvalue reads element 0
                        w = __builtin_amdgcn_readlane(__builtin_bit_cast(unsigned, el), tl);
                    }
            if constexpr (sizeof(T) == 2) xt = (te & 1) ? __uint_as_float(w & 0xffff0000u) : bf16_bits_to_f32(w & 0xffffu);
            else xt = __uint_as_float(w);
        }
        float z[NCH][V];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const float* tp = tab_s + (j * 64 + lane) * V;
            float xv[V];
            RowIo<T>::unpack(xr[0][j], xv);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                z[j][e] = xv[e] * tp[e];
                m = fmaxf(m, z[j][e]);
            }
        }
        // The next row's loads go out BEFORE this row's stores: vmcnt retires in order, so a load issued after the
        // stores could only be waited for together with them (write latency + read latency per row, 1.9 TB/s).
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < NCH; ++j) xr[d][j] = xr[d + 1][j];
        if (row + DEPTH * nwaves < a.B) load_row(row + DEPTH * nwaves, xr[DEPTH - 1]);   // wave-uniform
        m = wave_max(m);
        const float m2 = m * kLog2e;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {           // 2^(z*log2e - m*log2e): one fma + v_exp_f32 per element
#pragma unroll
            for (int e = 0; e < V; ++e) z[j][e] = fast_exp2(__builtin_fmaf(z[j][e], kLog2e, -m2));
#pragma unroll
            for (int e = 0; e < V; e += 4) s += (z[j][e] + z[j][e + 1]) + (z[j][e + 2] + z[j][e + 3]);
        }
        s = wave_sum(s);
        const float inv_s = 1.0f / s;
        if (MODE == 1) {
            float* o = sm_out + (int64_t)row * ld_sm;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c0 = (j * 64 + lane) * V;
                const int n = j < JL ? V : nl;
#pragma unroll
                for (int e = 0; e < V; e += 4)
                    if (e < n) *reinterpret_cast<f32x4*>(o + c0 + e) = f32x4{z[j][e] * inv_s, z[j][e + 1] * inv_s, z[j][e + 2] * inv_s, z[j][e + 3] * inv_s};
            }
            continue;
        }
        const float lse = m + kLn2 * fast_log2(s);
        RowCoef rc;
        float r = 0.f;
        if (plain) {
            rc.rw = 1.0f; rc.tb = -1; rc.wb = 0.f;
            rc.ta = t_ok ? ta_cur : -1;
            rc.wa = t_ok ? 1.0f : 0.0f;
            if (!t_ok && ta_cur != a.ignore && a.status && lane == 0) atomicExch(a.status, 1);
            if (t_ok) r = lse - xt * tab_s[(int)ta_cur];
        } else {
            rc = row_coef(a, row);
            if (rc.ta >= 0) r += rc.wa * (lse - Io<T>::load1(x + rc.ta) * tab_s[rc.ta]);
            if (rc.tb >= 0) r += rc.wb * (lse - Io<T>::load1(x + rc.tb) * tab_s[rc.tb]);
        }
        if (lane == 0) a.loss_row[row] = rc.rw * r;
        wave_loss += rc.rw * r;
        if (a.dx == nullptr) continue;
        T* dx = static_cast<T*>(a.dx) + (int64_t)row * a.lddx;
        const float g = a.scale * rc.rw;
        const float gs = g * (rc.wa + rc.wb) * inv_s, ga = g * rc.wa, gb = g * rc.wb;
        const int ia = (int)rc.ta, ib = (int)rc.tb;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c0 = (j * 64 + lane) * V;
            const int n = j < JL ? V : nl;
            if (j < JL || n > 0) {
                const float* tp = tab_s + c0;
                float p[V];
#pragma unroll
                for (int e = 0; e < V; ++e) p[e] = z[j][e] * gs;
                const unsigned da = (unsigned)(ia - c0), db = (unsigned)(ib - c0);
                if ((da < (unsigned)V) | (db < (unsigned)V)) {      // only the lane(s) owning a target column
#pragma unroll
                    for (int e = 0; e < V; ++e) p[e] -= (da == (unsigned)e ? ga : 0.f) + (db == (unsigned)e ? gb : 0.f);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) p[e] *= tp[e];
                RowIo<T>::store(dx + c0, n, p);
            }
        }
    }
    if (MODE != 1) finish_with_ticket(a, wave_loss);
}

// -------------------------------------------- streaming row (any C / alignment)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) row_stream_kernel(CeArgs a, float* sm_out, int64_t ld_sm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float wave_loss = 0.f;
    if (row < a.B) {                              // wave-uniform
        const T* x = static_cast<const T*>(a.x) + (int64_t)row * a.ldx;
        float m = -INFINITY;
        for (int c = lane; c < a.C; c += 64) m = fmaxf(m, Io<T>::load1(x + c) * (a.tab[c] * kLog2e));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += fast_exp2(Io<T>::load1(x + c) * (a.tab[c] * kLog2e) - m);
        s = wave_sum(s);
        const float inv_s = 1.0f / s;
        if (MODE == 1) {
            float* o = sm_out + (int64_t)row * ld_sm;
            for (int c = lane; c < a.C; c += 64) o[c] = fast_exp2(Io<T>::load1(x + c) * (a.tab[c] * kLog2e) - m) * inv_s;
        } else {
            const float lse = kLn2 * (m + fast_log2(s));
            const RowCoef rc = row_coef(a, row);
            float r = 0.f;
            if (rc.ta >= 0) r += rc.wa * (lse - Io<T>::load1(x + rc.ta) * a.tab[rc.ta]);
            if (rc.tb >= 0) r += rc.wb * (lse - Io<T>::load1(x + rc.tb) * a.tab[rc.tb]);
            if (lane == 0) a.loss_row[row] = rc.rw * r;
            wave_loss = rc.rw * r;
            if (a.dx != nullptr) {
                T* dx = static_cast<T*>(a.dx) + (int64_t)row * a.lddx;
                const float g = a.scale * rc.rw;
                const float gs = g * (rc.wa + rc.wb) * inv_s, ga = g * rc.wa, gb = g * rc.wb;
                for (int c = lane; c < a.C; c += 64) {
                    const float tc = a.tab[c];
                    float p = fast_exp2(Io<T>::load1(x + c) * (tc * kLog2e) - m) * gs;
                    if (c == rc.ta) p -= ga;
                    if (c == rc.tb) p -= gb;
                    Io<T>::store1(dx + c, p * tc);
                }
            }
        }
    }
    if (MODE != 1) finish_with_ticket(a, wave_loss);
}

// fixed-order sum of the per-row losses: one 256-thread block, deterministic
__global__ void __launch_bounds__(256) loss_reduce_kernel(const float* rows, int B, float scale, float* out) {
    __shared__ float sh[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += rows[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0] * scale;
}

template <typename T>
__global__ void __launch_bounds__(256) scale_rows_kernel(const T* x, int64_t ldx, const float* tab, int B, int C,
                                                         T* out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    const T* xr = x + (int64_t)row * ldx;
    T* o = out + (int64_t)row * ldo;
    for (int c = lane; c < C; c += 64) Io<T>::store1(o + c, Io<T>::load1(xr + c) * tab[c]);
}

template <typename T>
__global__ void __launch_bounds__(256) topk_hits_kernel(const T* x, int64_t ldx, const float* tab,
                                                        const int64_t* tgt, int B, int C, int k0, int k1, int k2,
                                                        int k3, int nk, int32_t* hits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    const int64_t t = tgt[row];
    if (t < 0 || t >= C) return;
    const T* xr = x + (int64_t)row * ldx;
    const float zt = Io<T>::load1(xr + t) * (tab ? tab[t] : 1.0f);
    int cnt = 0;
    for (int c = lane; c < C; c += 64) {
        const float z = Io<T>::load1(xr + c) * (tab ? tab[c] : 1.0f);
        cnt += (z > zt) || (z == zt && c < (int)t);
    }
    cnt = wave_sum_i(cnt);
    if (lane == 0) {
        const int ks[4] = {k0, k1, k2, k3};
        for (int j = 0; j < nk; ++j)
            if (cnt < ks[j]) atomicAdd(&hits[j], 1);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) scale_by_scalar_kernel(const T* x, int64_t n, const float* s, T* out) {
    const float f = *s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        Io<T>::store1(out + i, Io<T>::load1(x + i) * f);
}

inline bool aligned(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

constexpr unsigned kMaxRowBlocks = 2048;      // = partial slots of the single-launch loss workspace

// *inline_reduce (in/out): the caller wants the scalar loss reduced by this launch; cleared when the launch
// configuration cannot do it (streaming kernel with more blocks than workspace slots).
template <typename T, int MODE>
int launch_rows(CeArgs a, float* sm_out, int64_t ld_sm, hipStream_t st, bool* inline_reduce = nullptr) {
    // waves per block: every block stages the table once and takes one ticket, so small batches use fewer, fatter blocks.
    // finish_with_ticket's tree needs a power of two <= 4 (sh[256], __launch_bounds__(256)): anything else falls back to 4
    int wpb = 4;                // measured 1 / 2 / 4: [1024, 1204] 23.2 / 13.3 / 9.5 us, [256, 1000] 9.1 / 7.6 / 6.6 us
    const dim3 grid((a.B + wpb - 1) / wpb), block(64 * wpb);
    // register-row kernel: at most 256 CUs x 8 blocks; beyond that a wave walks several rows.  Never more blocks than the
    // single-launch loss workspace has partial slots.
    // blocks of the register-row kernel, round-4 loop at [65536, 1000] (512 / 768 / 1024 / 1536 / 2048 blocks):
    //   bf16 3.55 / 4.32 / 4.60 / 4.84 / 4.67 TB/s, fp32 5.22 / 5.01 / 4.98 / 4.93 / 4.82 TB/s ([8192, 1204] fp32: 512 best as well)
    unsigned maxb = sizeof(T) == 2 ? 1536u : 512u;
    if (maxb < 1u) maxb = 1u;
    if (maxb > kMaxRowBlocks) maxb = kMaxRowBlocks;
    const dim3 pgrid(grid.x < maxb ? grid.x : maxb);
    // rows in 16-byte lane vectors: 4 fp32 / 8 bf16 columns; a bf16 row may end on a half vector (C % 8 == 4: 1204)
    constexpr int V = RowIo<T>::V;
    // (bf16 rows need 8-byte alignment only - RowIo<unsigned short>: a contiguous [B, 1204] matrix qualifies)
    constexpr int RA = sizeof(T) == 2 ? 4 : V;            // row pitch granule in elements, = 8 / 16 bytes
    constexpr int PA = sizeof(T) == 2 ? 8 : 16;
    bool vec = (a.C % 4 == 0) && (a.C <= 2048) && (a.ldx % RA == 0) && aligned(a.x, PA) && aligned(a.tab, 16);
    if (MODE != 1 && a.dx) vec = vec && (a.lddx % RA == 0) && aligned(a.dx, PA);
    if (MODE == 1) vec = vec && (ld_sm % 4 == 0) && aligned(sm_out, 16);
    if constexpr (MODE == 0) {       // plain CE on whole 16-byte vectors: the loop without mixup / weights / half tails
        if (vec && !a.tb && !a.roww && !a.clsw && a.C % V == 0) return launch_rows<T, 2>(a, sm_out, ld_sm, st, inline_reduce);
    }
    if (vec) {
        const int nch = (a.C + 64 * V - 1) / (64 * V);
        if (nch <= 1) hipLaunchKernelGGL((row_reg_kernel<T, 1, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);
        else if (nch <= 2) hipLaunchKernelGGL((row_reg_kernel<T, 2, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);
        else if (nch <= 3) hipLaunchKernelGGL((row_reg_kernel<T, 3, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);     // bf16: C = 1204 (LVIS head)
        else if (nch <= 4) hipLaunchKernelGGL((row_reg_kernel<T, 4, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);
        else {
            if constexpr (V == 4) {
                if (nch <= 5) hipLaunchKernelGGL((row_reg_kernel<T, 5, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);     // fp32: C = 1204 (LVIS head)
                else if (nch <= 6) hipLaunchKernelGGL((row_reg_kernel<T, 6, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);
                else hipLaunchKernelGGL((row_reg_kernel<T, 8, MODE>), pgrid, block, 0, st, a, sm_out, ld_sm);
            } else {
                return IIF_EUNSUPPORTED;           // unreachable: C <= 2048 is at most 4 chunks of 512 bf16 columns
            }
        }
    } else {
        if (grid.x > kMaxRowBlocks) {
            a.ticket = nullptr;
            if (inline_reduce) *inline_reduce = false;
        }
        hipLaunchKernelGGL((row_stream_kernel<T, MODE>), grid, block, 0, st, a, sm_out, ld_sm);
    }
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // namespace

extern "C" {

int iif_ce_fwd_bwd(const void* logits, int dtype, int64_t ld_logits, const float* table,
                   const int64_t* targets_a, const int64_t* targets_b, float lam,
                   const float* row_weight, const float* class_weight, int64_t ignore_index,
                   float scale, int B, int C, float* loss_per_row, float* loss_out, void* dlogits,
                   int64_t ld_dlogits, int32_t* d_status, void* d_workspace, void* stream) {
    if (B < 0 || C <= 0) return IIF_EINVAL;
    if (dtype != IIF_F32 && dtype != IIF_BF16) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (B == 0) {  // empty batch: loss 0 (mmdet heads can see zero RoIs)
        if (loss_out) {
            if (hipMemsetAsync(loss_out, 0, sizeof(float), st) != hipSuccess) return IIF_ELAUNCH;
        }
        return IIF_OK;
    }
    if (!logits || !table || !targets_a || !loss_per_row) return IIF_EINVAL;
    if (ld_logits < C || (dlogits && ld_dlogits < C)) return IIF_EINVAL;
    bool one_launch = loss_out != nullptr && d_workspace != nullptr;
    CeArgs a{logits, ld_logits, table, targets_a, targets_b, lam, row_weight, class_weight,
             ignore_index, scale, B, C, loss_per_row, dlogits, ld_dlogits, d_status,
             one_launch ? loss_out : nullptr, one_launch ? static_cast<int32_t*>(d_workspace) : nullptr};
    int rc = dtype == IIF_F32 ? launch_rows<float, 0>(a, nullptr, 0, st, &one_launch)
                              : launch_rows<unsigned short, 0>(a, nullptr, 0, st, &one_launch);
    if (rc != IIF_OK) return rc;
    if (loss_out && !one_launch) {
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, loss_per_row, B, scale, loss_out);
        IIF_LAUNCH_CHECK();
    }
    return IIF_OK;
}

int iif_softmax(const void* logits, int dtype, int64_t ld_logits, const float* table, int B, int C,
                float* out, int64_t ld_out, void* stream) {
    if (B < 0 || C <= 0) return IIF_EINVAL;
    if (B == 0) return IIF_OK;
    if (!logits || !table || !out || ld_logits < C || ld_out < C) return IIF_EINVAL;
    CeArgs a{};
    a.x = logits; a.ldx = ld_logits; a.tab = table; a.B = B; a.C = C;
    if (dtype == IIF_F32) return launch_rows<float, 1>(a, out, ld_out, as_stream(stream));
    if (dtype == IIF_BF16) return launch_rows<unsigned short, 1>(a, out, ld_out, as_stream(stream));
    return IIF_EINVAL;
}

int iif_scale_logits(const void* logits, int dtype, int64_t ld_logits, const float* table, int B, int C,
                     void* out, int64_t ld_out, void* stream) {
    if (B < 0 || C <= 0) return IIF_EINVAL;
    if (B == 0) return IIF_OK;
    if (!logits || !table || !out || ld_logits < C || ld_out < C) return IIF_EINVAL;
    const int wpb = 4;
    const dim3 grid((B + wpb - 1) / wpb), block(64 * wpb);
    if (dtype == IIF_F32)
        hipLaunchKernelGGL(scale_rows_kernel<float>, grid, block, 0, as_stream(stream),
                           (const float*)logits, ld_logits, table, B, C, (float*)out, ld_out);
    else if (dtype == IIF_BF16)
        hipLaunchKernelGGL(scale_rows_kernel<unsigned short>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)logits, ld_logits, table, B, C, (unsigned short*)out, ld_out);
    else
        return IIF_EINVAL;
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_topk_hits(const void* logits, int dtype, int64_t ld_logits, const float* table,
                  const int64_t* targets, int B, int C, const int32_t* k_host, int nk, int32_t* hits,
                  void* stream) {
    if (B < 0 || C <= 0 || nk <= 0 || nk > 4 || !k_host) return IIF_EINVAL;
    if (B == 0) return IIF_OK;
    if (!logits || !targets || !hits || ld_logits < C) return IIF_EINVAL;
    int k[4] = {0, 0, 0, 0};
    for (int j = 0; j < nk; ++j) k[j] = k_host[j];
    const int wpb = 4;
    const dim3 grid((B + wpb - 1) / wpb), block(64 * wpb);
    if (dtype == IIF_F32)
        hipLaunchKernelGGL(topk_hits_kernel<float>, grid, block, 0, as_stream(stream), (const float*)logits,
                           ld_logits, table, targets, B, C, k[0], k[1], k[2], k[3], nk, hits);
    else if (dtype == IIF_BF16)
        hipLaunchKernelGGL(topk_hits_kernel<unsigned short>, grid, block, 0, as_stream(stream),
                           (const unsigned short*)logits, ld_logits, table, targets, B, C, k[0], k[1], k[2],
                           k[3], nk, hits);
    else
        return IIF_EINVAL;
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_scale_by_device_scalar(const void* x, int dtype, int64_t n, const float* d_scalar, void* out, void* stream) {
    if (n < 0 || !d_scalar) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!x || !out) return IIF_EINVAL;
    const int blocks = (int)(cdiv64(n, 256) < 2048 ? cdiv64(n, 256) : 2048);
    if (dtype == IIF_F32)
        hipLaunchKernelGGL(scale_by_scalar_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           (const float*)x, n, d_scalar, (float*)out);
    else if (dtype == IIF_BF16)
        hipLaunchKernelGGL(scale_by_scalar_kernel<unsigned short>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           (const unsigned short*)x, n, d_scalar, (unsigned short*)out);
    else
        return IIF_EINVAL;
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
