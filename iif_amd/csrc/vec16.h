// 16-byte channel vectors of an NHWC row: 4 fp32 or 8 bf16 values, widened to fp32 in registers.
#pragma once
#include "common.h"

namespace {
template <typename T> struct VT;
template <> struct VT<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    // Non-temporal load: the line will not be used again for a long time (an activation that is next read in backward, an
    // operand at its last use).  The hint keeps it from displacing what the NEXT kernel reads back (the tensor this kernel
    // writes) in L2 / the Infinity Cache: -2.2 % on the ResNet50 step from the two BN apply kernels alone.
    static __device__ __forceinline__ void load_nt(const float* p, float (&v)[4]) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    // the 16 raw bytes first, the conversion later: a batch of loads can be requested before the first value is touched
    static __device__ __forceinline__ u32x4 raw_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
    static __device__ __forceinline__ u32x4 raw(const float* p) { return *reinterpret_cast<const u32x4*>(p); }
    static __device__ __forceinline__ void unpack(const u32x4& t, float (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
    }
    static __device__ __forceinline__ float round_trip(float v) { return v; }       // a value as the storage type holds it
    template <bool NT> static __device__ __forceinline__ void store_as(float* p, const float (&v)[4]) {
        const f32x4 t{v[0], v[1], v[2], v[3]};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = t;
    }
};
template <> struct VT<unsigned short> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void load_nt(const unsigned short* p, float (&v)[8]) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
#ifdef IIF_NT_STORE
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
#else
        *reinterpret_cast<u32x4*>(p) = t;
#endif
    }
    static __device__ __forceinline__ u32x4 raw_nt(const unsigned short* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
    static __device__ __forceinline__ u32x4 raw(const unsigned short* p) { return *reinterpret_cast<const u32x4*>(p); }
    static __device__ __forceinline__ void unpack(const u32x4& t, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ float round_trip(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    template <bool NT> static __device__ __forceinline__ void store_as(unsigned short* p, const float (&v)[8]) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = t;
    }
};
}  // namespace
