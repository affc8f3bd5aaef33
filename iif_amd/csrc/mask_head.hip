// Mask-side class-channel selection of the Mask R-CNN head, gfx950.
//
// Reference: the class channel of every RoI is picked by its (IIF-derived) label,
//   inference  FCNMaskHead.get_seg_masks   mask_pred[range(N), labels]            (fcn_mask_head.py:289-290)
//   training   mask_cross_entropy          pred[inds, label] -> BCE-with-logits, mean   (cross_entropy_loss.py:158-162)
// The reference materialises the [N, H, W] slice, and its autograd scatters the slice gradient into a zero
// [N, C, H, W] tensor.  Here the selected channel is read in place: one block per RoI streams H*W elements
// (HBM-bound, 4 B read + 4 B target read + 4 B gradient write per element), per-RoI loss partials are reduced
// by one block in a fixed order (deterministic).
#include "common.h"

namespace {

template <typename T> struct LD;
template <> struct LD<float> { static __device__ __forceinline__ float ld(const float* p) { return *p; } };
template <> struct LD<unsigned short> { static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_bits_to_f32(*p); } };

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;               // valid in thread 0
}

template <typename T>
__global__ void __launch_bounds__(256) mask_gather_kernel(const T* pred, const int64_t* labels, int C, int hw, float* out,
                                                          int* status) {
    const int n = blockIdx.x;
    const int64_t lb = labels[n];
    if (lb < 0 || lb >= C) { if (threadIdx.x == 0) atomicOr(status, 1); return; }
    const T* p = pred + ((int64_t)n * C + lb) * hw;
    float* o = out + (int64_t)n * hw;
    for (int i = threadIdx.x; i < hw; i += 256) o[i] = LD<T>::ld(p + i);
}

// loss_i = max(x,0) - x*t + log1p(exp(-|x|))   (F.binary_cross_entropy_with_logits);  d/dx = sigmoid(x) - t
template <typename T>
__global__ void __launch_bounds__(256) mask_bce_kernel(const T* pred, const float* target, const int64_t* labels, int C, int hw,
                                                       float gscale, float* row_loss, float* dpred, int* status) {
    __shared__ float sh[4];
    const int n = blockIdx.x;
    const int64_t lb = labels[n];
    if (lb < 0 || lb >= C) {
        if (threadIdx.x == 0) { atomicOr(status, 1); row_loss[n] = 0.f; }
        return;
    }
    const int64_t base = ((int64_t)n * C + lb) * hw;
    const float* t = target + (int64_t)n * hw;
    float acc = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float x = LD<T>::ld(pred + base + i), y = t[i];
        acc += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        if (dpred) {
            const float s = 1.0f / (1.0f + expf(-x));
            dpred[base + i] = (s - y) * gscale;
        }
    }
    const float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) row_loss[n] = tot;
}

__global__ void __launch_bounds__(256) mask_loss_reduce_kernel(const float* row_loss, int n, double inv_count, float* loss) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)row_loss[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] * inv_count);
}

}  // namespace

extern "C" {

int iif_mask_gather(const void* pred, int dtype, const int64_t* labels, int n, int c, int hw, float* out, int* status,
                    void* stream) {
    if (n < 0 || c <= 0 || hw <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16)) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!pred || !labels || !out || !status) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    if (dtype == IIF_F32) hipLaunchKernelGGL(mask_gather_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)pred, labels, c, hw, out, status);
    else hipLaunchKernelGGL(mask_gather_kernel<unsigned short>, dim3(n), dim3(256), 0, st, (const unsigned short*)pred, labels, c, hw, out, status);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_mask_bce_fwd_bwd(const void* pred, int dtype, const float* target, const int64_t* labels, int n, int c, int hw,
                         float grad_scale, float* row_loss, float* loss, float* dpred, int* status, void* stream) {
    if (n <= 0 || c <= 0 || hw <= 0 || (dtype != IIF_F32 && dtype != IIF_BF16)) return IIF_EINVAL;
    if (!pred || !target || !labels || !row_loss || !loss || !status) return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    const double inv = 1.0 / ((double)n * (double)hw);
    const float gs = grad_scale * (float)inv;
    if (dtype == IIF_F32) hipLaunchKernelGGL(mask_bce_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)pred, target, labels, c, hw, gs, row_loss, dpred, status);
    else hipLaunchKernelGGL(mask_bce_kernel<unsigned short>, dim3(n), dim3(256), 0, st, (const unsigned short*)pred, target, labels, c, hw, gs, row_loss, dpred, status);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mask_loss_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, n, inv, loss);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
