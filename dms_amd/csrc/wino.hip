// dms_amd/csrc/wino.hip -- Winograd F(2x2, 3x3) halves of the high-res head's convolutions (gfx950).
//
// UpsamplingModule level (dmn_pytorch/models/dmn.py:324,337-338):
//     x = conv3x3(Upsample(x2, bilinear)(cat([x, feat], 1)))
// MIOpen runs that 3x3 convolution at 111-122 TFLOP/s (conv-equivalent; its fp32 Winograd kernel
// is a gfx9 assembly kernel).  Split the Winograd algorithm instead so that its multiply part is
// a plain batched fp32 GEMM, which rocBLAS runs at ~133 TFLOP/s ACTUAL = ~300 conv-equivalent
// (profiles/r1_wino_probe.txt):
//     V[xi]  = B^T d B      input transform, per 4x4 input tile d (tiles overlap by 2)   <- here
//     M[xi]  = U[xi] . V[xi]   16 GEMMs (Cout x C) . (C x tiles), U = G g G^T           <- rocBLAS
//     Y      = A^T M A + bias  output transform, one 2x2 output tile per input tile      <- here
// The input transform is FUSED with the cat and the bilinear x2 upsample: tile (ty, tx) of the
// upsampled image is a function of the 3x3 low-res neighbourhood of pixel (ty, tx) alone, so the
// upsampled tensor (up to 2 GB per level) is never written or read.
//
// Bilinear x2, align_corners=False:  up[2i] = .25 lo[i-1] + .75 lo[i],  up[2i+1] = .75 lo[i] + .25 lo[i+1]
// (indices clamped to the image); the conv zero-pads `up` by one pixel.
// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  A^T = [1 1 1 0; 0 1 -1 -1].
#include "common.h"

namespace dmn {

// rows/columns -1 .. +2 around up-pixel 2t from the three low-res samples (lm, l0, lp) =
// lo[t-1], lo[t], lo[t+1] (already clamped by the caller), then B^T applied:
//   h0 = up[2t-1] (zero when t == 0: conv padding), h1 = up[2t], h2 = up[2t+1],
//   h3 = up[2t+2] (zero when t == n-1)
__device__ __forceinline__ void up_then_bt(float lm, float l0, float lp, bool first, bool last, float o[4])
{
    const float h0 = first ? 0.0f : __builtin_fmaf(0.75f, lm, 0.25f * l0);
    const float h1 = __builtin_fmaf(0.25f, lm, 0.75f * l0);
    const float h2 = __builtin_fmaf(0.75f, l0, 0.25f * lp);
    const float h3 = last ? 0.0f : __builtin_fmaf(0.25f, l0, 0.75f * lp);
    o[0] = h0 - h2;
    o[1] = h1 + h2;
    o[2] = h2 - h1;
    o[3] = h1 - h3;
}

// V (16, C, NT), NT = B*H*W tiles, tile index = (b*H + ty)*W + tx.  One thread = VEC adjacent tiles
// of one channel; grid (ceil(NT/VEC / 256), C).  16 stores per thread, each a dense 16 B x 64 run.
template <int VEC>
__global__ void __launch_bounds__(256)
wino_in_kernel(const float *__restrict__ x, const float *__restrict__ feat, float *__restrict__ V,
               int B, int Cx, int Cf, int H, int W)
{
    const int Wv = W / VEC;
    const unsigned per_img = (unsigned)H * Wv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * per_img) return;
    const int c = blockIdx.y;
    const int b = idx / per_img;
    const unsigned r = idx - b * per_img;
    const int ty = r / Wv, tx0 = (r - ty * Wv) * VEC;
    const float *plane = c < Cx ? x + ((size_t)b * Cx + c) * H * W
                                : feat + ((size_t)b * Cf + (c - Cx)) * H * W;
    const int rows[3] = {ty > 0 ? ty - 1 : 0, ty, ty < H - 1 ? ty + 1 : H - 1};
    // column transform of the three low-res rows: g[j][tile][q]
    float g[3][VEC][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *row = plane + (size_t)rows[j] * W;
        float l[VEC + 2];
        l[0] = row[tx0 > 0 ? tx0 - 1 : 0];
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(row + tx0);
            l[1] = v.x; l[2] = v.y; l[3] = v.z; l[VEC] = v.w;
        } else {
            l[1] = row[tx0];
        }
        l[VEC + 1] = row[tx0 + VEC < W ? tx0 + VEC : W - 1];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            up_then_bt(l[e], l[e + 1], l[e + 2], tx0 + e == 0, tx0 + e == W - 1, g[j][e]);
    }
    const size_t NT = (size_t)B * H * W;
    const size_t tile0 = ((size_t)b * H + ty) * W + tx0;
    float *out = V + (size_t)c * NT + tile0;
    const size_t plane_stride = (size_t)(Cx + Cf) * NT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[VEC][4];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            up_then_bt(g[0][e][q], g[1][e][q], g[2][e][q], ty == 0, ty == H - 1, v[e]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float *dst = out + (size_t)(p * 4 + q) * plane_stride;
            if (VEC == 4) store_stream4(dst, v[0][p], v[1][p], v[2][p], v[VEC - 1][p]);
            else dst[0] = v[0][p];
        }
    }
}

// out (B, Cout, 2H, 2W) from M (16, Cout, NT) + bias.  One thread = VEC adjacent tiles of one
// output channel; grid (ceil(NT/VEC / 256), Cout).
template <int VEC>
__global__ void __launch_bounds__(256)
wino_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                int B, int Cout, int H, int W)
{
    const int Wv = W / VEC;
    const unsigned per_img = (unsigned)H * Wv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * per_img) return;
    const int o = blockIdx.y;
    const int b = idx / per_img;
    const unsigned r = idx - b * per_img;
    const int ty = r / Wv, tx0 = (r - ty * Wv) * VEC;
    const size_t NT = (size_t)B * H * W;
    const size_t tile0 = ((size_t)b * H + ty) * W + tx0;
    const float *src = M + (size_t)o * NT + tile0;
    const size_t plane_stride = (size_t)Cout * NT;
    const float bv = bias ? bias[o] : 0.0f;
    // t[p][e][0..1] = row p of M A for tile e
    float y0[VEC][2], y1[VEC][2];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { y0[e][0] = y0[e][1] = bv; y1[e][0] = y1[e][1] = bv; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float m[4][VEC];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *s = src + (size_t)(p * 4 + q) * plane_stride;
            if (VEC == 4) {
                const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(s));
                m[q][0] = v.x; m[q][1] = v.y; m[q][2] = v.z; m[q][VEC - 1] = v.w;
            } else {
                m[q][0] = s[0];
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float t0 = m[0][e] + m[1][e] + m[2][e];
            const float t1 = m[1][e] - m[2][e] - m[3][e];
            // A^T rows: y0 += t (p = 0, 1, 2);  y1 += t (p = 1), -= t (p = 2, 3)
            if (p < 3) { y0[e][0] += t0; y0[e][1] += t1; }
            if (p == 1) { y1[e][0] += t0; y1[e][1] += t1; }
            if (p >= 2) { y1[e][0] -= t0; y1[e][1] -= t1; }
        }
    }
    float *dst = out + (((size_t)b * Cout + o) * 2 * H + 2 * ty) * 2 * W + 2 * tx0;
    if (VEC == 4) {
        float4 *d0 = reinterpret_cast<float4 *>(dst);
        float4 *d1 = reinterpret_cast<float4 *>(dst + 2 * W);
        d0[0] = make_float4(y0[0][0], y0[0][1], y0[1][0], y0[1][1]);
        d0[1] = make_float4(y0[2][0], y0[2][1], y0[VEC - 1][0], y0[VEC - 1][1]);
        d1[0] = make_float4(y1[0][0], y1[0][1], y1[1][0], y1[1][1]);
        d1[1] = make_float4(y1[2][0], y1[2][1], y1[VEC - 1][0], y1[VEC - 1][1]);
    } else {
        dst[0] = y0[0][0]; dst[1] = y0[0][1];
        dst[2 * W] = y1[0][0]; dst[2 * W + 1] = y1[0][1];
    }
}

static int check_wino(const char *fn, int B, int C, int H, int W)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(-4, "%s: bad sizes B=%d C=%d H=%d W=%d", fn, B, C, H, W);
    if (C > 65535) return fail(-5, "%s: %d channels exceed the grid y limit", fn, C);
    if ((long long)B * H * W >= 0x7fffffffLL) return fail(-5, "%s: too many tiles", fn);
    return 0;
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_wino_cat_upsample2x_transform(const float *x, const float *feat, float *V,
                                                 int B, int Cx, int Cf, int H, int W, void *stream)
{
    if (Cx < 0 || Cf < 0) return fail(-4, "dmn_wino_cat_upsample2x_transform: negative channel count");
    if (int rc = check_wino("dmn_wino_cat_upsample2x_transform", B, Cx + Cf, H, W)) return rc;
    if ((!x && Cx > 0) || (!feat && Cf > 0) || !V) return fail(-1, "dmn_wino_cat_upsample2x_transform: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = W % 4 == 0 && !(((uintptr_t)x | (uintptr_t)feat | (uintptr_t)V) & 15);
    const long long items = (long long)B * H * (vec ? W / 4 : W);
    dim3 grid((unsigned)((items + 255) / 256), Cx + Cf);
    if (vec) hipLaunchKernelGGL(wino_in_kernel<4>, grid, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
    else     hipLaunchKernelGGL(wino_in_kernel<1>, grid, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
    return launch_status("dmn_wino_cat_upsample2x_transform");
}

extern "C" int dmn_wino_output_transform(const float *M, const float *bias, float *out,
                                         int B, int Cout, int H, int W, void *stream)
{
    if (int rc = check_wino("dmn_wino_output_transform", B, Cout, H, W)) return rc;
    if (!M || !out) return fail(-1, "dmn_wino_output_transform: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = W % 4 == 0 && !(((uintptr_t)M | (uintptr_t)out) & 15);
    const long long items = (long long)B * H * (vec ? W / 4 : W);
    dim3 grid((unsigned)((items + 255) / 256), Cout);
    if (vec) hipLaunchKernelGGL(wino_out_kernel<4>, grid, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
    else     hipLaunchKernelGGL(wino_out_kernel<1>, grid, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
    return launch_status("dmn_wino_output_transform");
}
