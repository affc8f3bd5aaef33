// FASA side outputs of the IIF classifier loss, gfx950.
//
// Reference (instance_segmentation/mmdet):
//   FasaIIFLoss.forward, use_cums   losses/fasa_iif_loss.py:154-160   per-class sums of the row losses and counts
//   fa_update / fa_update_push      roi_heads/bbox_heads/fasa_bbox_head.py:118-147   per-class feature mean and
//                                   unbiased variance of the positive RoI embeddings, exponential moving average
//   fa_generate                     fasa_bbox_head.py:149-172        virtual embeddings mean + sqrt(var) * N(0,1)
// The reference loops over torch.unique(labels) on the host (one sync per class).  Here every class is one
// thread / one block that scans the labels in row order: no host round trip, deterministic sums.
#include "common.h"

namespace {

// thread c: cum_labels[c] += #{i: labels[i] == c};  cum_losses[c] += sum of rows[i] over those i (row order)
__global__ void __launch_bounds__(256) class_accumulate_kernel(const float* rows, const int64_t* labels, int n, int C,
                                                               float* cum_losses, float* cum_labels) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        if (labels[i] == (int64_t)c) { s += rows[i]; ++cnt; }
    if (cnt) { cum_losses[c] += s; cum_labels[c] += (float)cnt; }
}

// block c, thread -> feature dims d = tid, tid+256, ...: two-pass mean / variance over the rows of class c
__global__ void __launch_bounds__(256) fasa_update_kernel(const float* emb, const int64_t* labels, int n, int D, int64_t lde,
                                                          float decay, float* fmean, float* fvar, float* fused) {
    __shared__ int wcnt[4];
    const int c = blockIdx.x;
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 256) mine += labels[i] == (int64_t)c ? 1 : 0;
    mine = wave_sum_i(mine);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (total == 0) return;                                  // block-uniform
    const bool seen = fused[c] > 0.f;
    __syncthreads();                                         // every thread has read fused[c] before thread 0 bumps it
    for (int d = threadIdx.x; d < D; d += 256) {
        float sum = 0.f;
        for (int i = 0; i < n; ++i)
            if (labels[i] == (int64_t)c) sum += emb[(int64_t)i * lde + d];
        const float mean = sum / (float)total;
        float ss = 0.f;
        for (int i = 0; i < n; ++i)
            if (labels[i] == (int64_t)c) { const float t = emb[(int64_t)i * lde + d] - mean; ss = fmaf(t, t, ss); }
        // var(unbiased=False) * n/(n-1) for n > 1 (fasa_bbox_head.py:133-137)
        const float var = total > 1 ? (ss / (float)total) * ((float)total / (float)(total - 1)) : ss / (float)total;
        const int64_t o = (int64_t)c * D + d;
        if (seen) {
            fmean[o] = decay * mean + (1.f - decay) * fmean[o];
            fvar[o] = decay * var + (1.f - decay) * fvar[o];
        } else {
            fmean[o] = mean;
            fvar[o] = var;
        }
    }
    if (threadIdx.x == 0 && !seen) fused[c] += 1.f;
}

// one block: classes with rand[c] < prob[c] and used[c] > 0, in ascending order, get slot k
__global__ void __launch_bounds__(256) fasa_select_kernel(const float* rnd, const float* prob, const float* fused, int C,
                                                          int* slot_class, int* count) {
    __shared__ int base_s;
    __shared__ int wsum[4];
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + threadIdx.x;
        const int take = (c < C && rnd[c] < prob[c] && fused[c] > 0.f) ? 1 : 0;
        // exclusive prefix inside the block (wave scan + wave totals)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int incl = take;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int woff = 0;
        for (int i = 0; i < wv; ++i) woff += wsum[i];
        const int pos = base_s + woff + incl - take;
        if (take) slot_class[pos] = c;
        __syncthreads();
        if (threadIdx.x == 255) base_s = pos + take;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

// block k (< *count): out[k] = mean[c] + sqrt(var[c]) * normal[c], labels[k] = c
__global__ void __launch_bounds__(256) fasa_generate_kernel(const int* slot_class, const int* count, const float* fmean,
                                                            const float* fvar, const float* normal, int D, float* out,
                                                            int64_t* out_labels) {
    const int k = blockIdx.x;
    if (k >= *count) return;
    const int c = slot_class[k];
    for (int d = threadIdx.x; d < D; d += 256) {
        const int64_t o = (int64_t)c * D + d;
        out[(int64_t)k * D + d] = fmean[o] + sqrtf(fvar[o]) * normal[o];
    }
    if (threadIdx.x == 0) out_labels[k] = c;
}

}  // namespace

extern "C" {

int iif_class_accumulate(const float* rows, const int64_t* labels, int n, int c, float* cum_losses, float* cum_labels,
                         void* stream) {
    if (n < 0 || c <= 0) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!rows || !labels || !cum_losses || !cum_labels) return IIF_EINVAL;
    hipLaunchKernelGGL(class_accumulate_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), rows, labels, n, c,
                       cum_losses, cum_labels);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_fasa_update(const float* embedding, const int64_t* labels, int n, int d, int64_t ld, int c, float decay,
                    float* feature_mean, float* feature_var, float* feature_used, void* stream) {
    if (n < 0 || d <= 0 || c <= 0 || ld < d) return IIF_EINVAL;
    if (n == 0) return IIF_OK;
    if (!embedding || !labels || !feature_mean || !feature_var || !feature_used) return IIF_EINVAL;
    hipLaunchKernelGGL(fasa_update_kernel, dim3(c), dim3(256), 0, as_stream(stream), embedding, labels, n, d, ld, decay,
                       feature_mean, feature_var, feature_used);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

int iif_fasa_generate(const float* rnd, const float* prob, const float* feature_used, const float* feature_mean,
                      const float* feature_var, const float* normal, int c, int d, int* slot_class, int* count,
                      float* out, int64_t* out_labels, void* stream) {
    if (c <= 0 || d <= 0) return IIF_EINVAL;
    if (!rnd || !prob || !feature_used || !feature_mean || !feature_var || !normal || !slot_class || !count || !out || !out_labels)
        return IIF_EINVAL;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(fasa_select_kernel, dim3(1), dim3(256), 0, st, rnd, prob, feature_used, c, slot_class, count);
    IIF_LAUNCH_CHECK();
    hipLaunchKernelGGL(fasa_generate_kernel, dim3(c), dim3(256), 0, st, slot_class, count, feature_mean, feature_var, normal, d,
                       out, out_labels);
    IIF_LAUNCH_CHECK();
    return IIF_OK;
}

}  // extern "C"
