// What a 2-read / 1-write bf16 streaming pass (the shape of bn_apply + residual: y = relu(a*x + b + r), one ReLU bit per element)
// can reach on this chip, by loop form: vectors per thread and trip (loads in flight), grid size (grid-stride over a resident
// grid or one vector batch per thread), non-temporal loads / stores, with / without the bit bytes.  Tensors far beyond the
// Infinity Cache (3 x 411 MB), so nothing is served on-die.  Rates = algorithmic bytes / HIP-event time.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_rw scripts/micro/stream_rw.hip && gpurun_out/stream_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack(float a, float b) {
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
    ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) & 0xffff0000u;
    return ua | ub;
}

// U vectors per thread and trip, all 2U loads requested before the first use.  CONTIG: a block's U vectors per thread are
// U consecutive 4-KB rows (coalesced per instruction); the grid strides over the tensor.
template <int U, bool NTL, bool NTS, bool BITS, bool RES>
__global__ void __launch_bounds__(256) k_stream(const u32x4* __restrict__ x, const u32x4* __restrict__ r, u32x4* __restrict__ y,
                                                unsigned char* __restrict__ bits, int64_t nvec, float a, float b) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < nvec; base += stride) {
        u32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            const int64_t ii = i < nvec ? i : base;
            xv[u] = NTL ? __builtin_nontemporal_load(x + ii) : x[ii];
            if (RES) rv[u] = NTL ? __builtin_nontemporal_load(r + ii) : r[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            u32x4 o;
            unsigned bb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t0 = fmaf(a, lo(xv[u][k]), b), t1 = fmaf(a, hi(xv[u][k]), b);
                if (RES) { t0 += lo(rv[u][k]); t1 += hi(rv[u][k]); }
                bb |= (t0 > 0.f ? 1u : 0u) << (2 * k);
                bb |= (t1 > 0.f ? 1u : 0u) << (2 * k + 1);
                o[k] = pack(fmaxf(t0, 0.f), fmaxf(t1, 0.f));
            }
            if (i < nvec) {
                if (NTS) __builtin_nontemporal_store(o, y + i); else y[i] = o;
                if (BITS) bits[i] = (unsigned char)bb;
            }
        }
    }
}

// the bit bytes gathered through LDS so that a wave stores them as 16-byte lanes (4 lanes x 16 B = its 64 bytes) instead of 64 x 1 B
template <int U, bool NTL>
__global__ void __launch_bounds__(256) k_stream_bits16(const u32x4* __restrict__ x, const u32x4* __restrict__ r, u32x4* __restrict__ y,
                                                       unsigned char* __restrict__ bits, int64_t nvec, float a, float b) {
    __shared__ unsigned char sb[U * 256];
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t blk = (int64_t)blockIdx.x * 256 * U; blk < nvec; blk += stride) {
        const int64_t base = blk + threadIdx.x;
        u32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            const int64_t ii = i < nvec ? i : base;
            xv[u] = NTL ? __builtin_nontemporal_load(x + ii) : x[ii];
            rv[u] = NTL ? __builtin_nontemporal_load(r + ii) : r[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            u32x4 o;
            unsigned bb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t0 = fmaf(a, lo(xv[u][k]), b) + lo(rv[u][k]), t1 = fmaf(a, hi(xv[u][k]), b) + hi(rv[u][k]);
                bb |= (t0 > 0.f ? 1u : 0u) << (2 * k);
                bb |= (t1 > 0.f ? 1u : 0u) << (2 * k + 1);
                o[k] = pack(fmaxf(t0, 0.f), fmaxf(t1, 0.f));
            }
            if (i < nvec) y[i] = o;
            sb[u * 256 + threadIdx.x] = (unsigned char)bb;
        }
        __syncthreads();
        if (threadIdx.x < U * 16) {
            const int64_t o = blk + threadIdx.x * 16;
            if (o + 16 <= nvec) *reinterpret_cast<u32x4*>(bits + o) = *reinterpret_cast<const u32x4*>(sb + threadIdx.x * 16);
        }
        __syncthreads();
    }
}


// the same pass with per-channel coefficients read from memory (what bn_apply really does): a thread's U vectors lie 256 vectors
// apart, i.e. on the same channel vector whenever C / 8 divides 256, so its 16 coefficients are loaded once per trip; ONE trip per
// thread (grid = nvec / (256 U)) or a grid-stride loop (any smaller grid)
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_bn(const u32x4* __restrict__ x, const u32x4* __restrict__ r, u32x4* __restrict__ y,
                                            unsigned char* __restrict__ bits, int64_t nvec, const float* __restrict__ ca,
                                            const float* __restrict__ cb, int cv) {
    const int c0 = (int)(threadIdx.x % cv) * 8;
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = ca[c0 + k]; b[k] = cb[c0 + k]; }
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < nvec; base += stride) {
        u32x4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            const int64_t ii = i < nvec ? i : base;
            xv[u] = NTL ? __builtin_nontemporal_load(x + ii) : x[ii];
            rv[u] = NTL ? __builtin_nontemporal_load(r + ii) : r[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256;
            u32x4 o;
            unsigned bb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t0 = fmaf(a[2 * k], lo(xv[u][k]), b[2 * k]) + lo(rv[u][k]), t1 = fmaf(a[2 * k + 1], hi(xv[u][k]), b[2 * k + 1]) + hi(rv[u][k]);
                bb |= (t0 > 0.f ? 1u : 0u) << (2 * k);
                bb |= (t1 > 0.f ? 1u : 0u) << (2 * k + 1);
                o[k] = pack(fmaxf(t0, 0.f), fmaxf(t1, 0.f));
            }
            if (i < nvec) {
                if (NTS) __builtin_nontemporal_store(o, y + i); else y[i] = o;
                bits[i] = (unsigned char)bb;
            }
        }
    }
}

// plain copy / read-only / write-only references
__global__ void __launch_bounds__(256) k_copy(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) y[i] = x[i];
}
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ x, unsigned* sink, int64_t nvec) {
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) acc ^= x[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ y, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) y[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}

template <typename F> float timeit(F f, int reps = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int64_t M = 256LL * 56 * 56, C = 256, nvec = M * C / 8;
    const size_t bytes = (size_t)nvec * 16;
    u32x4 *x, *r, *y; unsigned char* bits; unsigned* sink;
    hipMalloc(&x, bytes); hipMalloc(&r, bytes); hipMalloc(&y, bytes); hipMalloc(&bits, nvec); hipMalloc(&sink, 64);
    hipMemset(x, 0x3c, bytes); hipMemset(r, 0x3d, bytes);
    const double gb3 = (3.0 * bytes + nvec) / 1e9, gb3nb = 3.0 * bytes / 1e9, gb2 = 2.0 * bytes / 1e9;
    printf("tensors: %.0f MB each\n", bytes / 1e6);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        printf("grid %5d: copy %6.0f GB/s  read %6.0f  write %6.0f\n", grid,
               gb2 / timeit([&] { k_copy<<<grid, 256>>>(x, y, nvec); }) * 1e3,
               gb2 / 2 / timeit([&] { k_read<<<grid, 256>>>(x, sink, nvec); }) * 1e3,
               gb2 / 2 / timeit([&] { k_write<<<grid, 256>>>(y, nvec); }) * 1e3);
    }
#define RUN(U, NTL, NTS, BITS, G) \
    printf("U=%d ntl=%d nts=%d bits=%d grid=%6d: %6.0f GB/s\n", U, NTL, NTS, BITS, G, \
           (BITS ? gb3 : gb3nb) / timeit([&] { k_stream<U, NTL, NTS, BITS, true><<<G, 256>>>(x, r, y, bits, nvec, 0.5f, 0.1f); }) * 1e3)
    const int full1 = (int)((nvec + 255) / 256);
    for (int g : {2048, 4096, 8192, full1}) { RUN(1, true, false, true, g); }
    for (int g : {1024, 2048, 4096, full1 / 2}) { RUN(2, true, false, true, g); }
    for (int g : {512, 1024, 2048, 4096, full1 / 4}) { RUN(4, true, false, true, g); }
    for (int g : {512, 1024, 2048, full1 / 8}) { RUN(8, true, false, true, g); }
    for (int g : {2048, 4096}) { RUN(1, false, false, true, g); RUN(1, true, true, true, g); RUN(1, true, false, false, g); RUN(1, false, false, false, g); }
    for (int g : {1024, 2048}) { RUN(4, false, false, true, g); RUN(4, true, true, true, g); RUN(4, true, false, false, g); RUN(4, false, false, false, g); RUN(4, false, true, false, g); }
#define RUNB(U, NTL, G) \
    printf("bits16 U=%d ntl=%d grid=%6d: %6.0f GB/s\n", U, NTL, G, gb3 / timeit([&] { k_stream_bits16<U, NTL><<<G, 256>>>(x, r, y, bits, nvec, 0.5f, 0.1f); }) * 1e3)
    for (int g : {1024, 2048, 4096}) { RUNB(1, true, g); RUNB(2, true, g); RUNB(4, true, g); }
    // two-tensor form (bn_apply without residual): 1 read + 1 write
    printf("no residual: U=1 grid 4096 %6.0f GB/s, U=4 grid 1024 %6.0f, U=4 grid 2048 %6.0f\n",
           (gb2 + nvec / 1e9) / timeit([&] { k_stream<1, true, false, true, false><<<4096, 256>>>(x, r, y, bits, nvec, 0.5f, 0.1f); }) * 1e3,
           (gb2 + nvec / 1e9) / timeit([&] { k_stream<4, true, false, true, false><<<1024, 256>>>(x, r, y, bits, nvec, 0.5f, 0.1f); }) * 1e3,
           (gb2 + nvec / 1e9) / timeit([&] { k_stream<4, true, false, true, false><<<2048, 256>>>(x, r, y, bits, nvec, 0.5f, 0.1f); }) * 1e3);

    float *ca, *cb;
    hipMalloc(&ca, 4096 * 4); hipMalloc(&cb, 4096 * 4);
    hipMemset(ca, 0x3c, 4096 * 4); hipMemset(cb, 0, 4096 * 4);
#define RUNC(U, NTL, NTS, G) \
    printf("coef U=%d ntl=%d nts=%d grid=%6d: %6.0f GB/s\n", U, NTL, NTS, (int)(G), \
           gb3 / timeit([&] { k_bn<U, NTL, NTS><<<(int)(G), 256>>>(x, r, y, bits, nvec, ca, cb, 32); }) * 1e3)
    RUNC(1, true, false, 4096); RUNC(1, true, false, full1); RUNC(1, true, true, full1); RUNC(1, false, false, full1); RUNC(1, false, true, full1);
    RUNC(2, true, false, full1 / 2); RUNC(2, true, true, full1 / 2); RUNC(2, false, false, full1 / 2);
    RUNC(4, true, false, full1 / 4); RUNC(4, true, true, full1 / 4); RUNC(4, false, false, full1 / 4); RUNC(4, false, true, full1 / 4);
    RUNC(4, true, true, 1024); RUNC(4, true, true, 2048); RUNC(2, true, true, 2048); RUNC(2, true, true, 4096);
    RUNC(8, true, false, full1 / 8); RUNC(8, true, true, full1 / 8);
    return 0;
}
