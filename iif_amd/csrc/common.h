// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/iif_amd.h"

#define IIF_WAVE 64

#define IIF_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return IIF_ELAUNCH;  \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int lo16) {
    return __uint_as_float(lo16 << 16);
}
// round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32; keeps NaN a NaN)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return (unsigned int)f32_to_bf16_bits(lo) | ((unsigned int)f32_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// conv_stem.hip: the space-to-depth stem (4 x 4 taps over 16 padded channels -> 64, bf16) with BN partial sums; the
// convolution entry routes to it when the geometry fits (IIF_EUNSUPPORTED otherwise: the general kernel runs)
bool iif_stem4x4_ok(int N, int H, int W);
int iif_stem4x4_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int N, int H, int W, hipStream_t st);

// conv_regw.hip: 1x1 / stride 1 forward or data gradient with the weights in registers (narrow -> wide layers), bf16, BN partial
// sums; with an epilogue descriptor the data-gradient options of staged_drain (residual / its bits / gated store / upstream sums)
struct iif_regw_epilogue {
    const void* res; const unsigned char* res_bits; const void* bw_x; const unsigned char* bw_bits; const float* bw_stats; int mask_store;
    // round 6: instead of bw_x, the operands it is recomputed from per tile: the upstream block's a2 [M, rx_k2] and its conv3 weights [N, rx_ldw3]
    const void* rx_src2; const void* rx_w3; int rx_k2, rx_ldw3;
    // with rx_*: P = dst^T a2 and Gram = a2^T a2 as by-products, one fp32 slab [(N + rx_k2), pg_ld] per tile sequence (pg_cap floats
    // available, *pg_count receives the slab count; iif_slab_sum adds them up).  nullptr: not produced.
    float* pg_slab; long long pg_cap; int pg_ld; int* pg_count;
};
// round 6: `src` is the raw output of the previous convolution; its BN + ReLU (stats laid out as iif_bn_finalize_stats writes them) is
// applied to each tile in LDS and the activation written out as a by-product (out [M, K] bf16, bits one byte per 16-byte vector;
// csum nullable: one row [2][K] = (column sums of the activation, zeros) per partial row of the launch: iif_bn_partial_sums reduces them)
struct iif_regw_prologue { const float* stats; void* out; unsigned char* bits; float* csum; };
bool iif_regw1x1_ok(int M, int K, int N, int epi);
bool iif_regw1x1_rx_ok(int M, int K, int N, int k2);
bool iif_regw1x1_pg_ok(int M, int K, int N, int k2);      // ... with the P / Gram by-product (one N slice, k2 = 64)
bool iif_regw1x1_pro_ok(int M, int K, int N);
int iif_regw1x1_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int M, int K, int N, int spitch, int ldw, int dpitch, const iif_regw_epilogue* e, int no_store, hipStream_t st,
                       const iif_regw_prologue* pro = nullptr);
// the two passes of the never-stored conv + BN (+ identity / normalised shortcut) + ReLU forward (K in {64, 128, 256}, N a multiple
// of 256): statistics from the accumulators (no store), and the convolution with bn_apply's arithmetic in its epilogue
bool iif_regw1x1_fwdbn_ok(int M, int K, int N);
int iif_regw1x1_fwdbn_launch(const void* src, const void* wgt, void* dst, int M, int K, int N, int spitch, int ldw, int dpitch,
                             const void* res, const float* aff, const float* aff2, unsigned char* relu_out, hipStream_t st);
int iif_regw1x1_stats_launch(const void* src, const void* wgt, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                             int M, int K, int N, int spitch, int ldw, int dpitch, hipStream_t st, const iif_regw_prologue* pro = nullptr);
// 3x3 / stride 1 / pad 1, C -> C channels (64, 128), forward or data gradient (explicit tap list), optional upstream BN-backward sums
bool iif_regw3x3_ok(int N, int H, int W, int C);
int iif_regw3x3_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int N, int H, int W, int C, int ldw, const signed char* tap_dy, const signed char* tap_dx, const unsigned char* tap_w,
                       const void* bw_x, const unsigned char* bw_bits, const float* bw_stats, hipStream_t st);

// Compute units a persistent grid (one or two resident blocks per CU: conv_regw.hip, conv_stem.hip, the streaming 1x1 kernel)
// sizes itself to: the device's count, or the budget set by iif_set_cu_budget() when that is smaller (a rank that overlaps
// RCCL's reduction kernels with backward leaves them a few CUs instead of making their blocks queue behind a persistent grid).
int iif_persistent_cus();
int iif_persistent_grid(int unit);      // blocks of a grid launched in whole units of `unit` blocks, within the budget (see iif_host.cpp)
