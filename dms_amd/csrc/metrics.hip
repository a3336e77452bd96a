// dms_amd/csrc/metrics.hip -- threshold sweep of the evaluation loop in one pass (gfx950).
//
// Replaces, per sample, the reference's 39 x { (out > thresh).float(); compute_mask_IU }
// (dmn_pytorch/train.py:333-338, 366-374): 39 element-wise passes, 78 reductions and a
// host sync per threshold.  Here each pixel is read ONCE: with ascending thresholds
// (out > th_j) holds exactly for j < c, c = #{j : th_j < out}, so the kernel histograms c
// (weighted by 1 and by the mask value) in LDS and a suffix sum gives, for every threshold,
//   inter_j = sum_{c > j} hist_m[c]        pred_j = sum_{c > j} hist_1[c]
//   union_j = pred_j + sum(mask) - inter_j          (train.py:336-337)
// HBM-bound (8 B per pixel); sums are integers for binary masks, so fp32 atomics are exact
// and order-independent up to 2^24 pixels per sample.
#include "common.h"

namespace dmn {

constexpr int MAX_THRESH = 64;

__global__ void __launch_bounds__(256)
threshold_iou_kernel(const float *__restrict__ prob, const float *__restrict__ mask,
                     const float *__restrict__ thresholds, float *__restrict__ inter,
                     float *__restrict__ uni, int N, int NT)
{
    __shared__ float th[MAX_THRESH];
    __shared__ float h1[MAX_THRESH + 1], hm[MAX_THRESH + 1];
    const int b = blockIdx.y;
    if (threadIdx.x < NT) th[threadIdx.x] = thresholds[threadIdx.x];
    if (threadIdx.x <= NT) { h1[threadIdx.x] = 0.0f; hm[threadIdx.x] = 0.0f; }
    __syncthreads();
    const float *p = prob + (size_t)b * N, *m = mask + (size_t)b * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float v = p[i], w = m[i];
        int c = 0;
        for (int j = 0; j < NT; ++j) c += (v > th[j]) ? 1 : 0;
        atomicAdd(&h1[c], 1.0f);
        if (w != 0.0f) atomicAdd(&hm[c], w);
    }
    __syncthreads();
    if (threadIdx.x < NT) {
        const int j = threadIdx.x;
        float pred = 0.0f, in = 0.0f, msum = 0.0f;
        for (int c = 0; c <= NT; ++c) {
            msum += hm[c];
            if (c > j) { pred += h1[c]; in += hm[c]; }
        }
        atomicAdd(&inter[(size_t)b * NT + j], in);
        atomicAdd(&uni[(size_t)b * NT + j], pred + msum - in);
    }
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_threshold_iou(const float *prob, const float *mask, const float *thresholds,
                                 float *inter, float *uni, int B, int N, int NT, void *stream)
{
    if (!prob || !mask || !thresholds || !inter || !uni) return fail(-1, "dmn_threshold_iou: NULL pointer");
    if (B <= 0 || N <= 0 || NT <= 0) return fail(-4, "dmn_threshold_iou: bad sizes B=%d N=%d NT=%d", B, N, NT);
    if (NT > MAX_THRESH) return fail(-5, "dmn_threshold_iou: at most %d thresholds (got %d)", MAX_THRESH, NT);
    if (B > 65535) return fail(-5, "dmn_threshold_iou: B=%d exceeds the grid y limit", B);
    int bx = (N + 256 * 8 - 1) / (256 * 8);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(threshold_iou_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, prob,
                       mask, thresholds, inter, uni, N, NT);
    return launch_status("dmn_threshold_iou");
}
