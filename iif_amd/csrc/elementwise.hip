// Small streaming kernels (HBM-bound, 16 B per lane where alignment allows).
#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    static __device__ __forceinline__ float load1(const float* p) { return *p; }
    static __device__ __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct Vec<unsigned short> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        u32x4 t = *reinterpret_cast<const u32x4*>(p);
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

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) mix_rows_kernel(const T* x, const int64_t* perm, float lam, int64_t n, T* out) {
    const int b = blockIdx.y;
    const T* xa = x + (int64_t)b * n;
    const T* xb = x + perm[b] * n;
    T* o = out + (int64_t)b * n;
    const float mu = 1.0f - lam;
    constexpr int N = Vec<T>::N;
    if (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * N; i < n; i += (int64_t)gridDim.x * 256 * N) {
            float a[N], c[N];
            Vec<T>::load(xa + i, a); Vec<T>::load(xb + i, c);
#pragma unroll
            for (int k = 0; k < N; ++k) a[k] = lam * a[k] + mu * c[k];
            Vec<T>::store(o + i, a);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
            Vec<T>::store1(o + i, lam * Vec<T>::load1(xa + i) + mu * Vec<T>::load1(xb + i));
    }
}

}  // namespace

extern "C" int iif_mix_rows(const void* x, int dtype, const int64_t* perm, float lam, int B, int64_t n, void* out,
                            void* stream) {
    if (B < 0 || n < 0) return IIF_EINVAL;
    if (B == 0 || n == 0) return IIF_OK;
    if (!x || !perm || !out || B > 65535) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const bool al = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (dtype == IIF_F32) {
        const bool vec = al && (n % 4 == 0);
        const int gx = (int)(cdiv64(n, vec ? 1024 : 256) < 64 ? cdiv64(n, vec ? 1024 : 256) : 64);
        if (vec) hipLaunchKernelGGL((mix_rows_kernel<float, true>), dim3(gx, B), dim3(256), 0, st, (const float*)x, perm, lam, n, (float*)out);
        else hipLaunchKernelGGL((mix_rows_kernel<float, false>), dim3(gx, B), dim3(256), 0, st, (const float*)x, perm, lam, n, (float*)out);
    } else if (dtype == IIF_BF16) {
        const bool vec = al && (n % 8 == 0);
        const int gx = (int)(cdiv64(n, vec ? 2048 : 256) < 64 ? cdiv64(n, vec ? 2048 : 256) : 64);
        if (vec) hipLaunchKernelGGL((mix_rows_kernel<unsigned short, true>), dim3(gx, B), dim3(256), 0, st, (const unsigned short*)x, perm, lam, n, (unsigned short*)out);
        else hipLaunchKernelGGL((mix_rows_kernel<unsigned short, false>), dim3(gx, B), dim3(256), 0, st, (const unsigned short*)x, perm, lam, n, (unsigned short*)out);
    } else {
        return IIF_EINVAL;
    }
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}
