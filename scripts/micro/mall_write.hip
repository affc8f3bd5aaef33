// Is freshly WRITTEN data served from the 256 MiB Infinity Cache?  A writer kernel fills S megabytes with store policy P
// (0 default, 1 sc0, 2 nt, 3 sc0+nt, 16 sc1, 17 sc0+sc1), then a reader kernel streams the same S megabytes; for comparison the
// reader runs a second time right behind the first (what it has just READ).  Rates in GB/s.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mall_write scripts/micro/mall_write.hip && gpurun_out/mall_write
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int AUX>
__global__ void __launch_bounds__(256) writer(unsigned char* p, unsigned bytes) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
    const unsigned n = bytes / 16;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        u32x4 v = {i, i + 1, i + 2, i + 3};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, i * 16u, 0, AUX);
    }
}
template <bool REV = false>
__global__ void __launch_bounds__(256) reader(const unsigned char* p, unsigned bytes, unsigned* sink) {
    const unsigned n = bytes / 16;
    u32x4 acc = {0, 0, 0, 0};
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc ^= *reinterpret_cast<const u32x4*>(p + (size_t)(REV ? n - 1 - i : i) * 16);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int AUX, bool REV = false>
void run(unsigned char* buf, unsigned char* flush, unsigned mb, unsigned* sink) {
    hipEvent_t e0, e1, e2, e3;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    const unsigned bytes = mb << 20;
    float tw = 0, tr1 = 0, tr2 = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(flush, r, 1u << 30, 0);                 // push everything else out of the caches
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(writer<AUX>, dim3(4096), dim3(256), 0, 0, buf, bytes);
        hipEventRecord(e1, 0);
        if (REV) hipLaunchKernelGGL(reader<true>, dim3(4096), dim3(256), 0, 0, buf, bytes, sink);
        else hipLaunchKernelGGL(reader<false>, dim3(4096), dim3(256), 0, 0, buf, bytes, sink);
        hipEventRecord(e2, 0);
        hipLaunchKernelGGL(reader<false>, dim3(4096), dim3(256), 0, 0, buf, bytes, sink);
        hipEventRecord(e3, 0);
        hipEventSynchronize(e3);
        float a, b, c;
        hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3);
        if (r) { tw += a; tr1 += b; tr2 += c; }
    }
    const double g = bytes / 1e6 * (reps - 1);
    printf("S %4u MB policy %2d%s: write %6.0f GB/s   read after write %6.0f GB/s   read after read %6.0f GB/s\n", mb, AUX, REV ? " reader from the END" : "", g / tw, g / tr1, g / tr2);
}
int main() {
    unsigned char *buf, *flush; unsigned* sink;
    hipMalloc(&buf, 1u << 30); hipMalloc(&flush, 1u << 30); hipMalloc(&sink, 64);
    for (unsigned mb : {206u, 280u, 320u, 411u, 820u}) {
        run<0>(buf, flush, mb, sink); run<0, true>(buf, flush, mb, sink); run<2>(buf, flush, mb, sink);
    }
    return 0;
}
