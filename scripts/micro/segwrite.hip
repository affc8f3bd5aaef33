// Write (and read) bandwidth of a [M][N] bf16 matrix produced tile by tile, as a GEMM epilogue does it: block (mt, nt) writes a
// 128-row x SEG-byte tile (16 B per lane, whole 128-B lines) of rows whose pitch is N * 2 bytes.  SEG = 256 B (128-channel
// tiles), 512 B, 1 KB or the whole row.  Blocks are mapped as the convolution kernels map them (the N tiles of an M tile back
// to back on one XCD).  Does a narrow segment cost HBM efficiency?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/segwrite scripts/micro/segwrite.hip && gpurun_out/segwrite
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT, bool RD>
__global__ void __launch_bounds__(256) k(unsigned char* dst, int M, int rowbytes, int seg, int ntiles, int mtiles, unsigned* sink) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int mt = (j / ntiles) * 8 + xcd, nt = j % ntiles;
    if (mt >= mtiles) return;
    const int cpr = seg / 16, rpp = 256 / cpr;              // 16-B chunks per tile row, rows per pass
    const int chunk = threadIdx.x % cpr, r0 = threadIdx.x / cpr;
    u32x4 v = {1u, 2u, 3u, (unsigned)b};
    u32x4 acc = {0, 0, 0, 0};
    for (int row = r0; row < 128; row += rpp) {
        const long long m = (long long)mt * 128 + row;
        if (m >= M) break;
        unsigned char* p = dst + m * rowbytes + (long long)nt * seg + chunk * 16;
        if (RD) acc ^= *reinterpret_cast<const u32x4*>(p);
        else if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
        else *reinterpret_cast<u32x4*>(p) = v;
    }
    if (RD && (acc.x ^ acc.y) == 0x12345u) sink[0] = 1;
}

int main() {
    unsigned char* buf; unsigned* sink;
    const long long cap = 1LL << 30;
    hipMalloc(&buf, cap); hipMemset(buf, 0, cap); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int M, N; } shapes[] = {{50176, 1024}, {200704, 512}, {802816, 256}, {12544, 2048}};
    for (auto s : shapes) {
        const int rowbytes = s.N * 2;
        for (int seg = 256; seg <= rowbytes && seg <= 4096; seg *= 2) {
            const int ntiles = rowbytes / seg, mtiles = (s.M + 127) / 128;
            const int blocks = ntiles * ((mtiles + 7) / 8) * 8;
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemsetAsync(buf + (512 << 20), rep, 400 << 20, 0);      // push the previous pass out of the caches
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL((k<false, false>), dim3(blocks), dim3(256), 0, 0, buf, s.M, rowbytes, seg, ntiles, mtiles, sink);
                    if (mode == 1) hipLaunchKernelGGL((k<true, false>), dim3(blocks), dim3(256), 0, 0, buf, s.M, rowbytes, seg, ntiles, mtiles, sink);
                    if (mode == 2) hipLaunchKernelGGL((k<false, true>), dim3(blocks), dim3(256), 0, 0, buf, s.M, rowbytes, seg, ntiles, mtiles, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("M %6d N %4d  segment %4d B (%d N tiles, %5d blocks)  %s: %.3f ms  %.2f TB/s\n", s.M, s.N, seg, ntiles, blocks,
                       mode == 0 ? "store   " : mode == 1 ? "store nt" : "load    ", best, (double)s.M * rowbytes / best / 1e9);
            }
        }
    }
    return 0;
}
