// Issue cost of `buffer_load ... lds` (LDS-DMA) against `buffer_load_dwordx4` (to registers) on gfx950.
// Every wave issues ITER 1-KB pieces (64 lanes x 16 B) from an L2-resident window and waits once at the end;
// reports cycles per piece per wave for 1..4 waves per SIMD (256..1024 threads per block, one block per CU).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dma_issue scripts/micro/dma_issue.hip && gpurun_out/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
constexpr int ITER = 256;

template <int MODE>   // 0: LDS-DMA, 1: registers, 2: registers + ds_write_b128
__global__ void __launch_bounds__(1024) k(const unsigned char* src, unsigned bytes, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, bytes, 0x00020000);
    unsigned off = ((blockIdx.x * 16 + wave) * 65536u + lane * 16u) % bytes;
    u32x4 acc = {0, 0, 0, 0};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 8
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + ((wave * 4 + (i & 3)) * 1024)), 16, off, 0, 0, 0);
        } else {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            if (MODE == 2) *reinterpret_cast<u32x4*>(smem + (wave * 4 + (i & 3)) * 1024 + lane * 16) = v;
            else acc ^= v;
        }
        off += 1024; if (off >= bytes) off -= bytes;
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");     // issue time only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");     // + drain
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = t1 - t0; out[(blockIdx.x * 16 + wave) * 2 + 1] = t2 - t0; }
    if (MODE != 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (MODE != 1 && smem[threadIdx.x] == 0x77 && threadIdx.x == 12345) sink[1] = 1;
}

int main() {
    const unsigned bytes = 2u << 20;      // stays in L2 / MALL
    unsigned char* src; unsigned long long* out; unsigned* sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    hipMalloc(&out, 256 * 16 * 2 * 8); hipMalloc(&sink, 8);
    const char* names[3] = {"buffer_load ... lds      ", "buffer_load_dwordx4      ", "buffer_load_dwordx4+ds_wr"};
    for (int mode = 0; mode < 3; ++mode)
        for (int waves = 4; waves <= 16; waves *= 2)
            for (int blocks : {1, 256}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves), 0, 0, src, bytes, out, sink);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves), 0, 0, src, bytes, out, sink);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64 * waves), 0, 0, src, bytes, out, sink);
                }
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(blocks * 16 * 2);
                hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
                double a = 0, b = 0; int n = 0;
                for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < waves; ++w) { a += h[(bl * 16 + w) * 2]; b += h[(bl * 16 + w) * 2 + 1]; ++n; }
                printf("%s  %2d waves/CU (%d per SIMD), %3d CUs: issue %.1f cycles/piece/wave, with drain %.1f  -> %.1f B/clk/CU\n", names[mode], waves,
                       waves / 4, blocks, a / n / ITER, b / n / ITER, 1024.0 * waves / (b / n / ITER));
            }
    return 0;
}
