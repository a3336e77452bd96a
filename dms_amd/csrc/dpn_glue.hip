// dms_amd/csrc/dpn_glue.hip -- inference-time element-wise glue of the DPN backbone (gfx950).
//
// The convolutions of the backbone stay on MIOpen (north_star); what is fused here is the
// memory-bound glue BETWEEN them, which at B=32 / 512 px costs ~17 ms of a 113 ms step when
// run as separate torch kernels (profiles/r1_bench_steady_state_per_step.txt: BatchNorm 4.9,
// ReLU 4.3, cat 4.8, add 3.4 ms):
//   dmn_bn_relu   : eval-mode BatchNorm2d + ReLU in one pass
//                   (dmn_pytorch/models/dpn/dpn_model.py:156-177, CatBnAct / BnActConv2d)
//   dmn_dpn_merge : the dual-path merge  resid = x_s1 + out1 ; dense = cat([x_s2, out2])
//                   (dpn_model.py:265-269) AND the next block's cat(resid, dense)
//                   (dpn_model.py:246) in one pass: the block output is ONE contiguous tensor
//                   [resid | dense] whose channel slices are the two paths.
#include "common.h"

namespace dmn {

// y = max((x - mean[c]) * scale[c] + bias[c], 0),  scale = weight / sqrt(var + eps)
// grid: x over the HWv items of one (n, c) plane, y = c, z = n: no integer division at all
template <int VEC>
__global__ void __launch_bounds__(256)
bn_relu_kernel(const float *__restrict__ x, const float *__restrict__ mean,
               const float *__restrict__ scale, const float *__restrict__ bias,
               float *__restrict__ y, long long total, int C, int HWv)
{
    const int hw = blockIdx.x * blockDim.x + threadIdx.x;
    if (hw >= HWv) return;
    const int c = blockIdx.y;
    const long long i = ((long long)blockIdx.z * C + c) * HWv + hw;
    (void)total;
    const float m = mean[c], s = scale[c], b = bias[c];
    if (VEC == 4) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        o.x = fmaxf(__builtin_fmaf(v.x - m, s, b), 0.0f);
        o.y = fmaxf(__builtin_fmaf(v.y - m, s, b), 0.0f);
        o.z = fmaxf(__builtin_fmaf(v.z - m, s, b), 0.0f);
        o.w = fmaxf(__builtin_fmaf(v.w - m, s, b), 0.0f);
        reinterpret_cast<float4 *>(y)[i] = o;
    } else {
        y[i] = fmaxf(__builtin_fmaf(x[i] - m, s, b), 0.0f);
    }
}

// out (N, Cb + Cy - c1c, HW):  [ base[:, :c1c] + yv[:, :c1c] | base[:, c1c:] | yv[:, c1c:] ]
// BN: additionally act = relu(bn(out)) -- the NEXT consumer's pre-activation -- while the
// merged values are still in registers (saves re-reading `out`).
template <int VEC, bool BN>
__global__ void __launch_bounds__(256)
dpn_merge_kernel(const float *__restrict__ base, const float *__restrict__ yv,
                 float *__restrict__ out, float *__restrict__ act,
                 const float *__restrict__ mean, const float *__restrict__ scale,
                 const float *__restrict__ bias,
                 long long total, int Cb, int Cy, int c1c, int HWv)
{
    const int Co = Cb + Cy - c1c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HWv);
        const long long pl = i / HWv;
        const int c = (int)(pl % Co);
        const long long n = pl / Co;
        const long long ib = (n * Cb + c) * HWv + hw;                  // valid when c < Cb
        const long long iy = (n * Cy + (c < c1c ? c : c - Cb + c1c)) * HWv + hw;
        float m = 0.0f, s = 0.0f, b = 0.0f;
        if (BN) { m = mean[c]; s = scale[c]; b = bias[c]; }
        if (VEC == 4) {
            float4 o;
            if (c < c1c) {
                const float4 a = reinterpret_cast<const float4 *>(base)[ib];
                const float4 y = reinterpret_cast<const float4 *>(yv)[iy];
                o = make_float4(a.x + y.x, a.y + y.y, a.z + y.z, a.w + y.w);
            } else if (c < Cb) {
                o = reinterpret_cast<const float4 *>(base)[ib];
            } else {
                o = reinterpret_cast<const float4 *>(yv)[iy];
            }
            reinterpret_cast<float4 *>(out)[i] = o;          // (non-temporal: measured, no difference)
            if (BN) {
                float4 r;
                r.x = fmaxf(__builtin_fmaf(o.x - m, s, b), 0.0f);
                r.y = fmaxf(__builtin_fmaf(o.y - m, s, b), 0.0f);
                r.z = fmaxf(__builtin_fmaf(o.z - m, s, b), 0.0f);
                r.w = fmaxf(__builtin_fmaf(o.w - m, s, b), 0.0f);
                reinterpret_cast<float4 *>(act)[i] = r;
            }
        } else {
            const float o = c < c1c ? base[ib] + yv[iy] : (c < Cb ? base[ib] : yv[iy]);
            out[i] = o;
            if (BN) act[i] = fmaxf(__builtin_fmaf(o - m, s, b), 0.0f);
        }
    }
}

// In-place variant of the merge for a whole DPN stage.  The stage keeps ONE buffer per sample,
//   rows: [ dense0 (2*inc) | res (c1c) | slot (inc) | o2_1 (inc) | o2_2 (inc) | ... ]
// The block's last 1x1 convolution is a library GEMM with beta = 1 whose output rows are
// [res | slot]: the residual add `resid = x_s1 + out1` (dpn_model.py:265-266) happens in the GEMM's
// epilogue and the new dense channels `out2` land in the (zeroed) slot.  This pass then produces
// the next consumer's pre-activation in the REFERENCE channel order
//   [ res | dense0 | o2_1 .. o2_k | slot ]         (cat(resid, cat(x_s2, out2)), dpn_model.py:246,267-269)
// moves the slot's channels to their final rows, clears the slot for the next block and, on request
// (pyramid taps), writes the block output itself in reference order.  Compared with dmn_dpn_merge
// the residual and old dense channels are read once and never copied.
template <int VEC>
__global__ void __launch_bounds__(256)
dpn_stage_pass_kernel(float *__restrict__ buf, float *__restrict__ act, float *__restrict__ out,
                      const float *__restrict__ mean, const float *__restrict__ scale,
                      const float *__restrict__ bias, long long total, int rows, int c1c, int inc,
                      int Co, int HWv, long long out_bsv)
{
    // grid: x over the Co*HWv items of one sample (32-bit index math, one item per lane), y = sample
    const unsigned per = (unsigned)Co * (unsigned)HWv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < per) {
        const long long n = blockIdx.y;
        const int c = (int)(idx / (unsigned)HWv);
        const int hw = (int)(idx - (unsigned)c * (unsigned)HWv);
        const long long i = n * per + idx;
        (void)total;
        const bool fresh = c >= Co - inc;                               // this block's out2: in the slot
        const int row = c < c1c ? 2 * inc + c : (c < c1c + 2 * inc ? c - c1c : (fresh ? c1c + 2 * inc + c - (Co - inc) : c + inc));
        const long long src = ((n * rows + row) * HWv + hw);
        const float m = mean[c], s = scale[c], b = bias[c];
        if (VEC == 4) {
            const float4 v = reinterpret_cast<const float4 *>(buf)[src];
            if (fresh) {
                reinterpret_cast<float4 *>(buf)[(n * rows + c + inc) * HWv + hw] = v;
                reinterpret_cast<float4 *>(buf)[src] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            float4 r;
            r.x = fmaxf(__builtin_fmaf(v.x - m, s, b), 0.0f);
            r.y = fmaxf(__builtin_fmaf(v.y - m, s, b), 0.0f);
            r.z = fmaxf(__builtin_fmaf(v.z - m, s, b), 0.0f);
            r.w = fmaxf(__builtin_fmaf(v.w - m, s, b), 0.0f);
            reinterpret_cast<float4 *>(act)[i] = r;
            if (out) reinterpret_cast<float4 *>(out)[n * out_bsv + (long long)c * HWv + hw] = v;
        } else {
            const float v = buf[src];
            if (fresh) {
                buf[(n * rows + c + inc) * HWv + hw] = v;
                buf[src] = 0.0f;
            }
            act[i] = fmaxf(__builtin_fmaf(v - m, s, b), 0.0f);
            if (out) out[n * out_bsv + (long long)c * HWv + hw] = v;
        }
    }
}

// y[n,c,oy,ox] = relu(bn(x[n,c,2oy,2ox])): the pre-activation of a stride-2 1x1 projection only
// needs every other row and column.  One lane = 2 adjacent outputs (one float4 of input).
__global__ void __launch_bounds__(256)
bn_relu_s2_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                  const float *__restrict__ scale, const float *__restrict__ bias,
                  float *__restrict__ y, long long total, int C, int Ho, int Wo2, int W, long long x_bs)
{
    // grid: x over the Ho*Wo2 items of one (n, c) plane (32-bit math), y = c, z = n
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)Ho * (unsigned)Wo2) return;
    (void)total;
    const int oy = (int)(idx / (unsigned)Wo2), xo = (int)(idx - (unsigned)oy * (unsigned)Wo2);
    const int c = blockIdx.y;
    const long long n = blockIdx.z;
    const float m = mean[c], s = scale[c], b = bias[c];
    const f32x4_t v = __builtin_nontemporal_load(
        reinterpret_cast<const f32x4_t *>(x + n * x_bs + ((long long)c * 2 * Ho + 2 * oy) * W + 4 * xo));
    float2 o;
    o.x = fmaxf(__builtin_fmaf(v.x - m, s, b), 0.0f);
    o.y = fmaxf(__builtin_fmaf(v.z - m, s, b), 0.0f);
    *reinterpret_cast<float2 *>(y + ((n * C + c) * Ho + oy) * (2 * Wo2) + 2 * xo) = o;
}

// y = maxpool3x3/stride2/pad1( relu(bn(x)) ): the stem's bn + act + pool (dpn_model.py:186-197) in
// one pass over the raw conv output (which has to stay in memory as pyramid level 0).  After the
// ReLU every value is >= 0, so the pool's implicit -inf padding can be 0.  One lane = 2 adjacent
// output columns (one aligned float4 of input + its left neighbour) x RO output rows; the
// horizontal maxima of the row shared by two consecutive windows are carried in registers.
__global__ void __launch_bounds__(256)
bn_relu_maxpool_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                       const float *__restrict__ scale, const float *__restrict__ bias,
                       float *__restrict__ y, long long total, int C, int H, int W, int Ho, int RO)
{
    const int Wq = W / 4, Wo = W / 2;
    const int nchunk = (Ho + RO - 1) / RO;
    // grid: x over the nchunk*Wq items of one (n, c) plane (32-bit math), y = c, z = n
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    (void)total;
    if (idx < (unsigned)nchunk * (unsigned)Wq) {
        const int ch = (int)(idx / (unsigned)Wq);
        const int l = (int)(idx - (unsigned)ch * (unsigned)Wq);
        const int oy0 = ch * RO;
        const int c = blockIdx.y;
        const long long pl = (long long)blockIdx.z * C + c;        // n * C + c
        const float m = mean[c], s = scale[c], b = bias[c];
        const float *xp = x + pl * H * W + 4 * l;
        auto hrow = [&](int r, float &h0, float &h1) {
            const float4 v = *reinterpret_cast<const float4 *>(xp + (long long)r * W);   // (non-temporal: slower, rows are shared)
            const float left = l > 0 ? fmaxf(__builtin_fmaf(xp[(long long)r * W - 1] - m, s, b), 0.0f) : 0.0f;
            const float a0 = fmaxf(__builtin_fmaf(v.x - m, s, b), 0.0f);
            const float a1 = fmaxf(__builtin_fmaf(v.y - m, s, b), 0.0f);
            const float a2 = fmaxf(__builtin_fmaf(v.z - m, s, b), 0.0f);
            const float a3 = fmaxf(__builtin_fmaf(v.w - m, s, b), 0.0f);
            h0 = fmaxf(fmaxf(left, a0), a1);
            h1 = fmaxf(fmaxf(a1, a2), a3);
        };
        float p0 = 0.0f, p1 = 0.0f;
        if (oy0 > 0) hrow(2 * oy0 - 1, p0, p1);
        const int oy1 = oy0 + RO < Ho ? oy0 + RO : Ho;
        for (int oy = oy0; oy < oy1; ++oy) {
            float c0, c1, n0 = 0.0f, n1 = 0.0f;
            hrow(2 * oy, c0, c1);
            if (2 * oy + 1 < H) hrow(2 * oy + 1, n0, n1);
            *reinterpret_cast<float2 *>(y + (pl * Ho + oy) * Wo + 2 * l) =
                make_float2(fmaxf(fmaxf(p0, c0), n0), fmaxf(fmaxf(p1, c1), n1));
            p0 = n0; p1 = n1;
        }
    }
}

// out[n,p] = sum_c w[c] * x[n,c,p] + bias0: a 1x1 convolution to ONE channel (the head's out_layer,
// dmn.py:327,341, and output_collapse, dmn.py:213) is a channel reduction that reads x once.
template <int VEC>
__global__ void __launch_bounds__(256)
channel_dot_kernel(const float *__restrict__ x, const float *__restrict__ w,
                   const float *__restrict__ bias0, float *__restrict__ out,
                   long long total, int C, int HWv)
{
    const float b0 = bias0 ? bias0[0] : 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HWv);
        const long long n = i / HWv;
        float a0 = b0, a1 = b0, a2 = b0, a3 = b0;
        const float *xp = x + (n * C * HWv + p) * VEC;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const float wc = w[c];
            if (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(xp + (long long)c * HWv * 4);
                a0 = __builtin_fmaf(wc, v.x, a0); a1 = __builtin_fmaf(wc, v.y, a1);
                a2 = __builtin_fmaf(wc, v.z, a2); a3 = __builtin_fmaf(wc, v.w, a3);
            } else {
                a0 = __builtin_fmaf(wc, xp[(long long)c * HWv], a0);
            }
        }
        if (VEC == 4) reinterpret_cast<float4 *>(out)[i] = make_float4(a0, a1, a2, a3);
        else out[i] = a0;
    }
}

static int grid_for(long long items)
{
    long long blocks = (items + 255) / 256;
    const long long cap = (long long)NUM_CU * 16;
    return (int)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_bn_relu(const float *x, const float *mean, const float *scale, const float *bias,
                           float *y, int N, int C, int HW, void *stream)
{
    if (!x || !mean || !scale || !bias || !y) return fail(-1, "dmn_bn_relu: NULL pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return fail(-4, "dmn_bn_relu: bad sizes N=%d C=%d HW=%d", N, C, HW);
    hipStream_t s = (hipStream_t)stream;
    if (N > 65535 || C > 65535) return fail(-5, "dmn_bn_relu: N or C exceed the grid limits");
    const bool vec = HW % 4 == 0 && !(((uintptr_t)x | (uintptr_t)y) & 15);
    const int hwv = vec ? HW / 4 : HW;
    const int bt = hwv >= 256 ? 256 : ((hwv + 63) / 64) * 64;          // small planes: fewer idle lanes
    const dim3 grid((hwv + bt - 1) / bt, C, N);
    const long long total = (long long)N * C * hwv;
    if (vec) hipLaunchKernelGGL(bn_relu_kernel<4>, grid, dim3(bt), 0, s, x, mean, scale, bias, y, total, C, hwv);
    else hipLaunchKernelGGL(bn_relu_kernel<1>, grid, dim3(bt), 0, s, x, mean, scale, bias, y, total, C, hwv);
    return launch_status("dmn_bn_relu");
}

extern "C" int dmn_dpn_merge(const float *base, const float *yv, float *out, float *act,
                             const float *mean, const float *scale, const float *bias,
                             int N, int Cb, int Cy, int c1c, int HW, void *stream)
{
    if (!base || !yv || !out) return fail(-1, "dmn_dpn_merge: NULL pointer");
    if (act && (!mean || !scale || !bias))
        return fail(-1, "dmn_dpn_merge: act requested without mean/scale/bias");
    if (N <= 0 || HW <= 0 || c1c <= 0 || Cb < c1c || Cy < c1c)
        return fail(-4, "dmn_dpn_merge: bad sizes N=%d Cb=%d Cy=%d c1c=%d HW=%d", N, Cb, Cy, c1c, HW);
    hipStream_t s = (hipStream_t)stream;
    const int Co = Cb + Cy - c1c;
    const bool vec = HW % 4 == 0 &&
                     !(((uintptr_t)base | (uintptr_t)yv | (uintptr_t)out | (uintptr_t)act) & 15);
    const int hwv = vec ? HW / 4 : HW;
    const long long total = (long long)N * Co * hwv;
    const dim3 grid(grid_for(total)), block(256);
#define DMN_MERGE(V, B) hipLaunchKernelGGL((dpn_merge_kernel<V, B>), grid, block, 0, s, base, yv, out, \
                                           act, mean, scale, bias, total, Cb, Cy, c1c, hwv)
    if (vec) { if (act) DMN_MERGE(4, true); else DMN_MERGE(4, false); }
    else     { if (act) DMN_MERGE(1, true); else DMN_MERGE(1, false); }
#undef DMN_MERGE
    return launch_status("dmn_dpn_merge");
}

extern "C" int dmn_dpn_stage_pass(float *buf, float *act, float *out,
                                  const float *mean, const float *scale, const float *bias,
                                  int N, int rows, int c1c, int inc, int Co, int HW, long long out_bs,
                                  void *stream)
{
    if (out && out_bs < (long long)Co * HW)
        return fail(-4, "dmn_dpn_stage_pass: out batch stride %lld < Co*HW", out_bs);
    if (!buf || !act || !mean || !scale || !bias) return fail(-1, "dmn_dpn_stage_pass: NULL pointer");
    if (N <= 0 || HW <= 0 || c1c <= 0 || inc <= 0 || Co < c1c + 3 * inc || (Co - c1c - 2 * inc) % inc ||
        rows < Co + inc)
        return fail(-4, "dmn_dpn_stage_pass: bad sizes N=%d rows=%d c1c=%d inc=%d Co=%d HW=%d", N, rows, c1c, inc, Co, HW);
    const bool vec = HW % 4 == 0 && out_bs % 4 == 0 && !(((uintptr_t)buf | (uintptr_t)act | (uintptr_t)out) & 15);
    const int hwv = vec ? HW / 4 : HW;
    const long long obs = vec ? out_bs / 4 : out_bs;
    const long long total = (long long)N * Co * hwv;
    hipStream_t s = (hipStream_t)stream;
    const long long per = (long long)Co * hwv;
    if (per >= (1LL << 31) || N > 65535) return fail(-5, "dmn_dpn_stage_pass: Co*HW or N too large for the grid");
    const dim3 grid((unsigned)((per + 255) / 256), N);
    if (vec) hipLaunchKernelGGL(dpn_stage_pass_kernel<4>, grid, dim3(256), 0, s, buf, act, out, mean, scale, bias, total, rows, c1c, inc, Co, hwv, obs);
    else hipLaunchKernelGGL(dpn_stage_pass_kernel<1>, grid, dim3(256), 0, s, buf, act, out, mean, scale, bias, total, rows, c1c, inc, Co, hwv, obs);
    return launch_status("dmn_dpn_stage_pass");
}

extern "C" int dmn_bn_relu_s2(const float *x, const float *mean, const float *scale, const float *bias,
                              float *y, int N, int C, int H, int W, long long x_bs, void *stream)
{
    if (x_bs < (long long)C * H * W || x_bs % 4)
        return fail(-4, "dmn_bn_relu_s2: x batch stride %lld must be >= C*H*W and a multiple of 4", x_bs);
    if (!x || !mean || !scale || !bias || !y) return fail(-1, "dmn_bn_relu_s2: NULL pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || H % 2 || W % 4)
        return fail(-4, "dmn_bn_relu_s2: needs H %% 2 == 0 and W %% 4 == 0 (N=%d C=%d H=%d W=%d)", N, C, H, W);
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return fail(-7, "dmn_bn_relu_s2: x must be 16-byte, y 8-byte aligned");
    if (N > 65535 || C > 65535) return fail(-5, "dmn_bn_relu_s2: N or C exceed the grid limits");
    const long long total = (long long)N * C * (H / 2) * (W / 4);
    const int per = (H / 2) * (W / 4);
    const int bt = per >= 256 ? 256 : ((per + 63) / 64) * 64;
    hipLaunchKernelGGL(bn_relu_s2_kernel, dim3((per + bt - 1) / bt, C, N), dim3(bt), 0, (hipStream_t)stream,
                       x, mean, scale, bias, y, total, C, H / 2, W / 4, W, x_bs);
    return launch_status("dmn_bn_relu_s2");
}

extern "C" int dmn_bn_relu_maxpool(const float *x, const float *mean, const float *scale, const float *bias,
                                   float *y, int N, int C, int H, int W, void *stream)
{
    if (!x || !mean || !scale || !bias || !y) return fail(-1, "dmn_bn_relu_maxpool: NULL pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || W % 4)
        return fail(-4, "dmn_bn_relu_maxpool: needs W %% 4 == 0 (N=%d C=%d H=%d W=%d)", N, C, H, W);
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return fail(-7, "dmn_bn_relu_maxpool: x must be 16-byte, y 8-byte aligned");
    const int Ho = (H - 1) / 2 + 1, RO = 4;
    if (N > 65535 || C > 65535) return fail(-5, "dmn_bn_relu_maxpool: N or C exceed the grid limits");
    const long long total = (long long)N * C * ((Ho + RO - 1) / RO) * (W / 4);
    const int per = ((Ho + RO - 1) / RO) * (W / 4);
    const int bt = per >= 256 ? 256 : ((per + 63) / 64) * 64;
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3((per + bt - 1) / bt, C, N), dim3(bt), 0, (hipStream_t)stream,
                       x, mean, scale, bias, y, total, C, H, W, Ho, RO);
    return launch_status("dmn_bn_relu_maxpool");
}

extern "C" int dmn_channel_dot(const float *x, const float *w, const float *bias0, float *out,
                               int N, int C, int HW, void *stream)
{
    if (!x || !w || !out) return fail(-1, "dmn_channel_dot: NULL pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return fail(-4, "dmn_channel_dot: bad sizes N=%d C=%d HW=%d", N, C, HW);
    hipStream_t s = (hipStream_t)stream;
    if (HW % 4 == 0 && !(((uintptr_t)x | (uintptr_t)out) & 15)) {
        const long long total = (long long)N * (HW / 4);
        hipLaunchKernelGGL(channel_dot_kernel<4>, dim3(grid_for(total)), dim3(256), 0, s, x, w, bias0, out, total, C, HW / 4);
    } else {
        const long long total = (long long)N * HW;
        hipLaunchKernelGGL(channel_dot_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, x, w, bias0, out, total, C, HW);
    }
    return launch_status("dmn_channel_dot");
}
