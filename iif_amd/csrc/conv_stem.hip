// The stem convolution in its space-to-depth form (classification/resnet_pytorch.py:196-197: 7 x 7 / stride 2 / pad 3 over
// 3 channels = 4 x 4 / stride 1 over the 2 x 2 sub-pixel image with 12 -> 16 padded channels; resnet_engine.py packs both),
// bf16, 64 output channels, with the batch-norm partial sums of the stored output.
//
// Why a kernel of its own (round 4): a pixel of the sub-pixel image is 32 bytes, so the general LDS-DMA kernel pulls every
// source byte through the vector memory path 16 times (once per tap): 1.6 GB of DMA for 103 MB of input, 336 us in the step
// = the DMA fill rate of the CUs, not the matrix pipe (105 GFLOP) and not HBM (0.5 GB).  Here
//   * a 128-pixel tile loads the (rows + 3) x (W + 3) window of its source pixels ONCE (LDS halo image, double buffered,
//     zeros outside the image come from out-of-range DMA lanes): 172 B of DMA per output pixel instead of 512;
//   * the whole weight matrix (64 x 256 bf16 = 32 KB) lives in registers as MFMA fragments (128 VGPRs per wave), so a K step
//     (two horizontally adjacent taps = 64 contiguous bytes of the halo image) costs two ds_read_b128 per eight MFMAs;
//   * a pixel stride of 32 B makes the fragment reads bank-conflict free without a swizzle (16 lanes x 16 B at stride 32 B
//     interleave with the other half-group's +16 B);
//   * blocks are persistent (one barrier per tile, the next tile's window in flight under the MFMAs), two 4-wave blocks per CU
//     so that one block's epilogue (bf16 pack, wave-private LDS transpose, 16-byte stores, column sums) runs under the other's
//     MFMAs; the per-channel (sum, sum of squares) of the stored values accumulate in registers over all tiles of a block:
//     ONE partial row per block.
#include "common.h"

namespace {
typedef __attribute__((address_space(3))) void lds_void;

struct StemArgs {
    const unsigned char* src; const unsigned char* wgt; unsigned char* dst; float* bn_partial;
    int N, H, W, M, ntiles, dpitch, bn_row0;
};

constexpr int SBM = 128;                      // pixels per tile (32 per wave)
constexpr int SHR = 768;                      // halo pixels per buffer (host: (rows + 3) * (W + 3) <= SHR)
constexpr int SABUF = SHR * 32;
constexpr int SNAP = SHR / 32 / 4;            // DMA pieces (32 halo pixels = 1 KB) per wave and tile
constexpr int SPITCH = 64 * 2 + 16;           // staged output row
constexpr int SSTG = 32 * SPITCH;             // per wave
constexpr unsigned OOB = 0x80000000u;

__global__ void __launch_bounds__(256, 2) stem4x4_kernel(StemArgs a, unsigned src_bytes) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SABUF + 4 * SSTG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fc = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W, W2 = W + 3;
    const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src), 0, src_bytes, 0x00020000);

    // tiles of this block: the XCD (blockIdx % 8) owns a contiguous range of tiles, so that neighbouring windows share an L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t8 = (a.ntiles + 7) >> 3;
    const int tend = (xcd + 1) * t8 < a.ntiles ? (xcd + 1) * t8 : a.ntiles;
    int tile = xcd * t8 + j;

    auto issue_halo = [&](int t, int buf) {
        const int m0 = t * SBM;
        const int n = m0 / HW, rem = m0 - n * HW, y0 = rem / W;
        const int ylast = (rem + SBM - 1) / W;                       // tiles never cross images (host: HW % 128 == 0)
        const int Hn = (ylast - y0 + 4) * W2;
#pragma unroll
        for (int i = 0; i < SNAP; ++i) {
            const int p = wave + 4 * i;
            const int hr = 32 * p + (lane >> 1);
            const int ry = hr / W2, rx = hr - ry * W2;
            const int yy = y0 - 2 + ry, xx = rx - 2;
            const bool ok = hr < Hn && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = ok ? (unsigned)((n * H + yy) * W + xx) * 32u + (unsigned)(lane & 1) * 16u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lds_void*)(smem + buf * SABUF + p * 1024), 16, off, 0, 0, 0);
        }
    };

    if (tile < tend) issue_halo(tile, 0);
    // the weights as MFMA A fragments: step s = (tap row s / 2, tap pair s % 2), 32 consecutive K of row (ci * 16 + fr)
    u32x4 wreg[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            wreg[s][ci] = *reinterpret_cast<const u32x4*>(a.wgt + ((size_t)((ci * 16 + fr) * 256 + s * 32 + fc * 8)) * 2);

    float bs[8], bq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { bs[q] = 0.f; bq[q] = 0.f; }
    unsigned char* const stg = smem + 2 * SABUF + wave * SSTG;
    const int rowb = W2 * 32;
    int buf = 0;
    bool first = true;
    for (; tile < tend; tile += per_xcd, buf ^= 1) {
        // this tile's window: issued before the previous tile's four stores (vmcnt retires in order)
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();                   // every wave's pieces have landed; nobody reads the other buffer any more
        if (tile + per_xcd < tend) issue_halo(tile + per_xcd, buf ^ 1);
        const int m0 = tile * SBM;
        const int n = m0 / HW, rem = m0 - n * HW, y0 = rem / W;
        const unsigned char* Ab = smem + buf * SABUF;
        int xoff[2];
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
            const int r = rem + wave * 32 + pj * 16 + fr;
            const int y = r / W, x = r - y * W;
            xoff[pj] = ((y - y0) * W2 + x) * 32 + fc * 16;
        }
        f32x4 acc[4][2];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int pj = 0; pj < 2; ++pj) acc[ci][pj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            u32x4 xf[2];
#pragma unroll
            for (int pj = 0; pj < 2; ++pj)
                xf[pj] = *reinterpret_cast<const u32x4*>(Ab + xoff[pj] + (s >> 1) * rowb + (s & 1) * 64);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int pj = 0; pj < 2; ++pj)
                    acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[s][ci]),
                                                                         __builtin_bit_cast(bf16x8, xf[pj]), acc[ci][pj], 0, 0, 0);
        }
        // epilogue: lane holds channels ci * 16 + fc * 4 + {0..3} of pixel pj * 16 + fr -> wave-private staged rows
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int pj = 0; pj < 2; ++pj) {
                u32x2 w;
                w.x = pack_bf16x2(acc[ci][pj].x, acc[ci][pj].y);
                w.y = pack_bf16x2(acc[ci][pj].z, acc[ci][pj].w);
                *reinterpret_cast<u32x2*>(stg + (pj * 16 + fr) * SPITCH + (ci * 16 + fc * 4) * 2) = w;
            }
        unsigned char* const drow = a.dst + ((size_t)(m0 + wave * 32) * a.dpitch) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = k * 8 + (lane >> 3), chunk = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * SPITCH + chunk * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(drow + ((size_t)row * a.dpitch + chunk * 8) * 2));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = bf16_bits_to_f32(v[q] & 0xffffu), hi = __uint_as_float(v[q] & 0xffff0000u);
                bs[2 * q] += lo; bq[2 * q] = fmaf(lo, lo, bq[2 * q]);
                bs[2 * q + 1] += hi; bq[2 * q + 1] = fmaf(hi, hi, bq[2 * q + 1]);
            }
        }
    }
    if (a.bn_partial == nullptr) return;
    // lanes that share the channel chunk (lane % 8), then the four waves through LDS: fixed order, one row per block
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { bs[q] += __shfl_xor(bs[q], o, 64); bq[q] += __shfl_xor(bq[q], o, 64); }
    }
    __syncthreads();
    float* scratch = reinterpret_cast<float*>(smem);                    // [4][2][64]
    if (lane < 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            scratch[(wave * 2 + 0) * 64 + lane * 8 + q] = bs[q];
            scratch[(wave * 2 + 1) * 64 + lane * 8 + q] = bq[q];
        }
    }
    __syncthreads();
    if (tid < 64) {
        float s2 = 0.f, q2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s2 += scratch[(w * 2 + 0) * 64 + tid]; q2 += scratch[(w * 2 + 1) * 64 + tid]; }
        float* p = a.bn_partial + (int64_t)(a.bn_row0 + blockIdx.x) * 2 * a.dpitch + tid;
        p[0] = s2; p[a.dpitch] = q2;
    }
}
}  // namespace

// Geometry the kernel covers: tiles of 128 pixels never cross an image and their window fits the halo buffer.
bool iif_stem4x4_ok(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || (int64_t)N * H * W >= (1 << 26)) return false;
    if (((int64_t)H * W) % SBM) return false;
    const int span = (SBM + W - 2) / W + 1;            // image rows a tile can touch
    return (span + 3) * (W + 3) <= SHR;
}

int iif_stem4x4_launch(const void* src, const void* wgt, void* dst, float* bn_partial, long long bn_cap, int bn_row0, int* rows_out,
                       int N, int H, int W, hipStream_t st) {
    if (!src || !wgt || !dst || !iif_stem4x4_ok(N, H, W)) return IIF_EUNSUPPORTED;
    const int cus = iif_persistent_cus();
    StemArgs a{(const unsigned char*)src, (const unsigned char*)wgt, (unsigned char*)dst, bn_partial, N, H, W, N * H * W,
               N * H * W / SBM, 64, bn_row0};
    int grid = 2 * cus / 8 * 8;                          // two blocks per CU, whole groups of 8 (one per XCD)
    const int need = (a.ntiles + 7) / 8 * 8;
    if (need < grid) grid = need;                       // (never more partial rows than the tile kernels' ceil(M / 128) = ntiles)
    if (bn_partial && grid > a.ntiles) grid = a.ntiles / 8 * 8;
    if (grid < 8) return IIF_EUNSUPPORTED;
    if (bn_partial) {
        if ((long long)(bn_row0 + grid) * 2 * a.dpitch > bn_cap) return IIF_EINVAL;
        if (rows_out) *rows_out = bn_row0 + grid;
    }
    hipLaunchKernelGGL(stem4x4_kernel, dim3((unsigned)grid), dim3(256), 0, st, a, (unsigned)((int64_t)N * H * W * 32));
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}
