// Cosine / normed classifier heads (row-normalisation kernels, gfx950).
//   cosine (resnet_cifar.py:68-78):  ex = scale * x / (1 + |x|),  ew = W / |W_row|,  logits = ex @ ew^T
//   normed (resnet_cifar.py:46-48):  ex = x / max(|x|, eps),      ew = W / |W_col|,  logits = ex @ ew
// The GEMMs run on iif_conv_igemm / iif_conv_wgrad; these kernels are the per-row maps and their
// backward.  One 64-lane wave per row, fp32 math, shuffle reductions; HBM/latency-bound, tiny.
#include "common.h"

namespace {

template <typename T> struct HIO;
template <> struct HIO<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct HIO<unsigned short> {
    static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }
};

// mode 0: out = x * scale/(1+n)      (cosine feature map)
// mode 1: out = x / max(n, eps)      (F.normalize; rows of zeros stay zero)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) rowmap_fwd_kernel(const TI* x, int rows, int cols, int64_t ldx, int mode, float scale,
                                                         float eps, TO* out, int64_t ldo, float* norms) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + (int64_t)row * ldx;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = HIO<TI>::ld(xr + c); ss = fmaf(v, v, ss); }
    const float n = sqrtf(wave_sum(ss));
    const float coef = mode == 0 ? scale / (1.0f + n) : (n > 0.f ? 1.0f / fmaxf(n, eps) : 0.f);
    TO* o = out + (int64_t)row * ldo;
    for (int c = lane; c < cols; c += 64) HIO<TO>::st(o + c, HIO<TI>::ld(xr + c) * coef);
    if (lane == 0 && norms) norms[row] = n;
}

// backward of the row map: g = dL/d(out) (fp32 or T), x the forward input, n its stored norm
//   mode 0: dx = scale * ( g/(1+n) - x * <g,x> / (n (1+n)^2) )
//   mode 1: dx = g/m - x * <g,x> / m^3,  m = max(n, eps)
template <typename TI, typename TG, typename TO>
__global__ void __launch_bounds__(256) rowmap_bwd_kernel(const TI* x, const float* norms, const TG* g, int rows, int cols,
                                                         int64_t ldx, int64_t ldg, int mode, float scale, float eps, TO* dx,
                                                         int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + (int64_t)row * ldx;
    const TG* gr = g + (int64_t)row * ldg;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot = fmaf(HIO<TG>::ld(gr + c), HIO<TI>::ld(xr + c), dot);
    dot = wave_sum(dot);
    const float n = norms[row];
    float a, b;
    if (mode == 0) {
        a = scale / (1.0f + n);
        b = n > 0.f ? scale * dot / (n * (1.0f + n) * (1.0f + n)) : 0.f;
    } else {
        const float m = fmaxf(n, eps);
        a = n > 0.f ? 1.0f / m : 0.f;
        b = n > 0.f ? dot / (m * m * m) : 0.f;
    }
    TO* o = dx + (int64_t)row * lddx;
    for (int c = lane; c < cols; c += 64) HIO<TO>::st(o + c, HIO<TG>::ld(gr + c) * a - HIO<TI>::ld(xr + c) * b);
}


// mmdet normed predictors (instance_segmentation/mmdet/models/utils/normed_predictor.py:34-40, 67-73, 104-112):
//   v = r[row] * x[row]  (r = per-row IIF weight, or 1);  n = |v|;  out = v * scale / (n^power + eps)
// backward, g = dL/d(out):  dv = g*a - v*b,  a = scale/den,  b = scale*power*n^(power-2)*<g,v>/den^2;  dx = r*dv
__global__ void __launch_bounds__(256) rownorm_fwd_kernel(const float* x, const float* rscale, int rows, int cols, int64_t ldx,
                                                          float power, float scale, float eps, float* out, int64_t ldo,
                                                          float* norms) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    const float r = rscale ? rscale[row] : 1.0f;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = xr[c] * r; ss = fmaf(v, v, ss); }
    const float n = sqrtf(wave_sum(ss));
    const float den = (power == 1.0f ? n : powf(n, power)) + eps;
    const float coef = scale / den;
    float* o = out + (int64_t)row * ldo;
    for (int c = lane; c < cols; c += 64) o[c] = xr[c] * r * coef;
    if (lane == 0) norms[row] = n;
}

__global__ void __launch_bounds__(256) rownorm_bwd_kernel(const float* x, const float* rscale, const float* norms, const float* g,
                                                          int rows, int cols, int64_t ldx, int64_t ldg, float power, float scale,
                                                          float eps, float* dx, int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    const float* gr = g + (int64_t)row * ldg;
    const float r = rscale ? rscale[row] : 1.0f;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot = fmaf(gr[c], xr[c] * r, dot);
    dot = wave_sum(dot);
    const float n = norms[row];
    const float den = (power == 1.0f ? n : powf(n, power)) + eps;
    const float a = scale / den;
    // d(n^p)/dv = p * n^(p-2) * v
    const float b = n > 0.f ? scale * power * (power == 1.0f ? 1.0f / n : powf(n, power - 2.0f)) * dot / (den * den) : 0.f;
    float* o = dx + (int64_t)row * lddx;
    for (int c = lane; c < cols; c += 64) o[c] = r * (gr[c] * a - xr[c] * r * b);
}

__global__ void __launch_bounds__(256) transpose_f32_kernel(const float* in, int rows, int cols, int64_t ldi, float* out,
                                                            int64_t ldo) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(int64_t)(by + j) * ldi + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < cols && by + tx < rows) out[(int64_t)(bx + j) * ldo + by + tx] = tile[tx][j];
}

// out[0] = alpha * sum_i a[i]*b[i]  over a [rows, cols] window (fixed-order, one block)
__global__ void __launch_bounds__(256) dot_window_kernel(const float* a, const float* b, int rows, int cols, int64_t lda,
                                                         int64_t ldb, float alpha, const float* alpha_div, float* out) {
    __shared__ double sh[256];
    double acc = 0.0;
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        acc += (double)a[(int64_t)r * lda + c] * (double)b[(int64_t)r * ldb + c];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] * (double)alpha / (alpha_div ? (double)*alpha_div : 1.0));
}

}  // namespace

extern "C" {

int iif_rowmap_forward(const void* x, int x_dtype, int rows, int cols, int64_t ldx, int mode, float scale, float eps,
                       void* out, int out_dtype, int64_t ldo, float* norms, void* stream) {
    if (rows < 0 || cols <= 0 || (mode != 0 && mode != 1)) return IIF_EINVAL;
    if (rows == 0) return IIF_OK;
    if (!x || !out || ldx < cols || ldo < cols) return IIF_EINVAL;
    const dim3 grid((rows + 3) / 4), blk(256);
    hipStream_t st = as_stream(stream);
#define IIF_RM(TI, TO) hipLaunchKernelGGL((rowmap_fwd_kernel<TI, TO>), grid, blk, 0, st, (const TI*)x, rows, cols, ldx, mode, scale, eps, (TO*)out, ldo, norms)
    if (x_dtype == IIF_F32 && out_dtype == IIF_F32) IIF_RM(float, float);
    else if (x_dtype == IIF_F32 && out_dtype == IIF_BF16) IIF_RM(float, unsigned short);
    else if (x_dtype == IIF_BF16 && out_dtype == IIF_BF16) IIF_RM(unsigned short, unsigned short);
    else if (x_dtype == IIF_BF16 && out_dtype == IIF_F32) IIF_RM(unsigned short, float);
    else return IIF_EINVAL;
#undef IIF_RM
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_rowmap_backward(const void* x, int x_dtype, const float* norms, const void* g, int g_dtype, int rows, int cols,
                        int64_t ldx, int64_t ldg, int mode, float scale, float eps, void* dx, int dx_dtype, int64_t lddx,
                        void* stream) {
    if (rows < 0 || cols <= 0 || (mode != 0 && mode != 1)) return IIF_EINVAL;
    if (rows == 0) return IIF_OK;
    if (!x || !norms || !g || !dx || ldx < cols || ldg < cols || lddx < cols) return IIF_EINVAL;
    const dim3 grid((rows + 3) / 4), blk(256);
    hipStream_t st = as_stream(stream);
#define IIF_RB(TI, TG, TO) hipLaunchKernelGGL((rowmap_bwd_kernel<TI, TG, TO>), grid, blk, 0, st, (const TI*)x, norms, (const TG*)g, rows, cols, ldx, ldg, mode, scale, eps, (TO*)dx, lddx)
    if (x_dtype == IIF_F32 && g_dtype == IIF_F32 && dx_dtype == IIF_F32) IIF_RB(float, float, float);
    else if (x_dtype == IIF_BF16 && g_dtype == IIF_BF16 && dx_dtype == IIF_BF16) IIF_RB(unsigned short, unsigned short, unsigned short);
    else if (x_dtype == IIF_BF16 && g_dtype == IIF_F32 && dx_dtype == IIF_BF16) IIF_RB(unsigned short, float, unsigned short);
    else if (x_dtype == IIF_F32 && g_dtype == IIF_BF16 && dx_dtype == IIF_F32) IIF_RB(float, unsigned short, float);
    else return IIF_EINVAL;
#undef IIF_RB
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_rownorm_forward(const float* x, const float* row_scale, int rows, int cols, int64_t ldx, float power, float scale,
                        float eps, float* out, int64_t ldo, float* norms, void* stream) {
    if (rows < 0 || cols <= 0 || !(power > 0.f)) return IIF_EINVAL;
    if (rows == 0) return IIF_OK;
    if (!x || !out || !norms || ldx < cols || ldo < cols) return IIF_EINVAL;
    hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, row_scale, rows, cols, ldx,
                       power, scale, eps, out, ldo, norms);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_rownorm_backward(const float* x, const float* row_scale, const float* norms, const float* g, int rows, int cols,
                         int64_t ldx, int64_t ldg, float power, float scale, float eps, float* dx, int64_t lddx, void* stream) {
    if (rows < 0 || cols <= 0 || !(power > 0.f)) return IIF_EINVAL;
    if (rows == 0) return IIF_OK;
    if (!x || !norms || !g || !dx || ldx < cols || ldg < cols || lddx < cols) return IIF_EINVAL;
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, row_scale, norms, g, rows,
                       cols, ldx, ldg, power, scale, eps, dx, lddx);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_transpose_f32(const float* in, int rows, int cols, int64_t ldi, float* out, int64_t ldo, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || ldi < cols || ldo < rows) return IIF_EINVAL;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, as_stream(stream), in,
                       rows, cols, ldi, out, ldo);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_dot_window_f32(const float* a, const float* b, int rows, int cols, int64_t lda, int64_t ldb, float alpha,
                       const float* d_alpha_div, float* out, void* stream) {
    if (!a || !b || !out || rows <= 0 || cols <= 0 || lda < cols || ldb < cols) return IIF_EINVAL;
    hipLaunchKernelGGL(dot_window_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, b, rows, cols, lda, ldb, alpha,
                       d_alpha_div, out);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
