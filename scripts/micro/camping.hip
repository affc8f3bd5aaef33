// Does a split of a streamed tensor into CONTIGUOUS per-block ranges (what a split-K weight gradient does)
// read slower than the same bytes dealt round-robin in small chunks (what every streaming kernel does)?
// Every block reads `per` chunks of CH bytes (256 threads x 16 B x U loads in flight):
//   mode 0: chunk id = it * NB + b          (neighbouring blocks read neighbouring chunks)
//   mode 1: chunk id = b * per + it         (block b owns one contiguous range; stride between blocks = per * CH)
//   mode 2: as 1 with the per-block range length chosen so that the stride is an odd multiple of 4 KB
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/camping scripts/micro/camping.hip && gpurun_out/camping
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U>
__global__ void __launch_bounds__(256) rd(const unsigned char* __restrict__ src, long long nchunks, int per, int mode, long long stride_chunks,
                                          unsigned* sink) {
    const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < per; ++it) {
        long long c = mode == 0 ? (long long)it * nb + b : (long long)b * stride_chunks + it;
        if (c >= nchunks) break;
        const unsigned char* p = src + c * (long long)(256 * 16 * U) + t * 16;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + u * 4096));
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    const long long bytes = 1024LL << 20;
    unsigned char* src; unsigned* sink;
    hipMalloc(&src, bytes + (64 << 20)); hipMemset(src, 1, bytes + (64 << 20)); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    constexpr int U = 4;
    const long long CH = 256 * 16 * U;
    const long long nchunks = bytes / CH;
    printf("1 GiB read, chunk %lld B, U=%d loads in flight per thread\n", CH, U);
    for (int nb : {256, 512, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 3; ++mode) {
            int per = (int)((nchunks + nb - 1) / nb);
            long long stride = per;
            if (mode == 2) { stride = per | 1; }          // odd number of 16-KB chunks
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(rd<U>, dim3(nb), dim3(256), 0, 0, src, mode == 0 ? nchunks : nchunks + (64 << 20) / CH, per, mode, stride, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("blocks %5d  mode %d (%s)  per-block %6d chunks  stride %8lld KB : %.3f ms  %.2f TB/s\n", nb, mode,
                   mode == 0 ? "round-robin " : mode == 1 ? "contiguous  " : "contig, odd ", per, stride * CH / 1024, best,
                   (double)per * nb * CH / best / 1e9);
        }
    }
    // the weight-gradient pattern itself: 256 blocks, block b streams pixels [b*3136, (b+1)*3136) of a 128-B-row tensor and of a
    // 512-B-row tensor, 32 pixels per step (4 KB + 16 KB)
    return 0;
}
