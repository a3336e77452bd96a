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
        if constexpr (VEC == 4) {
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
            if constexpr (VEC == 4) store_stream4(dst, v[0][p], v[1][p], v[2][p], v[VEC - 1][p]);
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
            if constexpr (VEC == 4) {
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
    if constexpr (VEC == 4) {
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

// ------------------------------------------------------------------------------------
// F(4x4, 3x3): 6x6 input tiles (36 GEMMs, 4x fewer multiplies than direct, V is 2.25x the
// upsampled tensor instead of 4x).  Tile (ty, tx) = up rows 4ty-1 .. 4ty+4 = low-res rows
// 2ty-1 .. 2ty+2, so H and W must be even.  fp32 error ~1e-5 relative (F(2x2) ~7e-7).
// B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
// A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// ------------------------------------------------------------------------------------
// four low-res samples la..ld = lo[2t-1 .. 2t+2] (clamped by the caller) -> six up samples
// up[4t-1 .. 4t+4] (first / last: conv padding) -> B^T
__device__ __forceinline__ void up_then_bt6(float la, float lb, float lc, float ld, bool first, bool last,
                                            float o[6])
{
    const float u0 = first ? 0.0f : __builtin_fmaf(0.75f, la, 0.25f * lb);
    const float u1 = __builtin_fmaf(0.25f, la, 0.75f * lb);
    const float u2 = __builtin_fmaf(0.75f, lb, 0.25f * lc);
    const float u3 = __builtin_fmaf(0.25f, lb, 0.75f * lc);
    const float u4 = __builtin_fmaf(0.75f, lc, 0.25f * ld);
    const float u5 = last ? 0.0f : __builtin_fmaf(0.25f, lc, 0.75f * ld);
    o[0] = __builtin_fmaf(4.0f, u0, __builtin_fmaf(-5.0f, u2, u4));
    o[1] = __builtin_fmaf(-4.0f, u1 + u2, u3 + u4);
    o[2] = __builtin_fmaf(4.0f, u1 - u2, u4 - u3);
    o[3] = __builtin_fmaf(2.0f, u3 - u1, u4 - u2);
    o[4] = __builtin_fmaf(2.0f, u1 - u3, u4 - u2);
    o[5] = __builtin_fmaf(4.0f, u1, __builtin_fmaf(-5.0f, u3, u5));
}

// V (36, C, NT), NT = B*(H/2)*(W/2), tile index = (b*H/2 + ty)*W/2 + tx; one thread = VEC tiles
template <int VEC>
__global__ void __launch_bounds__(256)
wino4_in_kernel(const float *__restrict__ x, const float *__restrict__ feat, float *__restrict__ V,
                int B, int Cx, int Cf, int H, int W)
{
    const int Ht = H / 2, Wt = W / 2, Wv = Wt / VEC;
    const unsigned per_img = (unsigned)Ht * Wv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * per_img) return;
    const int c = blockIdx.y;
    const int b = idx / per_img;
    const unsigned r = idx - b * per_img;
    const int ty = r / Wv, tx0 = (r - ty * Wv) * VEC;
    const float *plane = c < Cx ? x + ((size_t)b * Cx + c) * H * W
                                : feat + ((size_t)b * Cf + (c - Cx)) * H * W;
    const int rows[4] = {ty > 0 ? 2 * ty - 1 : 0, 2 * ty, 2 * ty + 1, 2 * ty + 2 < H ? 2 * ty + 2 : H - 1};
    const int c0 = 2 * tx0;                                    // first low-res column of the thread
    float g[4][VEC][6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float *row = plane + (size_t)rows[j] * W;
        float l[2 * VEC + 2];
        l[0] = row[c0 > 0 ? c0 - 1 : 0];
        if constexpr (VEC == 4) {
            const float4 v0 = *reinterpret_cast<const float4 *>(row + c0);
            const float4 v1 = *reinterpret_cast<const float4 *>(row + c0 + 4);
            l[1] = v0.x; l[2] = v0.y; l[3] = v0.z; l[4] = v0.w;
            l[5] = v1.x; l[6] = v1.y; l[7] = v1.z; l[2 * VEC] = v1.w;
        } else {
            l[1] = row[c0]; l[2] = row[c0 + 1];
        }
        l[2 * VEC + 1] = row[c0 + 2 * VEC < W ? c0 + 2 * VEC : W - 1];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            up_then_bt6(l[2 * e], l[2 * e + 1], l[2 * e + 2], l[2 * e + 3], tx0 + e == 0, tx0 + e == Wt - 1, g[j][e]);
    }
    const size_t NT = (size_t)B * Ht * Wt;
    const size_t tile0 = ((size_t)b * Ht + ty) * Wt + tx0;
    float *out = V + (size_t)c * NT + tile0;
    const size_t plane_stride = (size_t)(Cx + Cf) * NT;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float v[VEC][6];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            up_then_bt6(g[0][e][q], g[1][e][q], g[2][e][q], g[3][e][q], ty == 0, ty == Ht - 1, v[e]);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            float *dst = out + (size_t)(p * 6 + q) * plane_stride;
            if constexpr (VEC == 4) store_stream4(dst, v[0][p], v[1][p], v[2][p], v[VEC - 1][p]);
            else dst[0] = v[0][p];
        }
    }
}

__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float y[4])
{
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y[0] = m0 + s12 + s34;
    y[1] = __builtin_fmaf(2.0f, d34, d12);
    y[2] = __builtin_fmaf(4.0f, s34, s12);
    y[3] = __builtin_fmaf(8.0f, d34, d12) + m5;
}

// out (B, Cout, 2H, 2W) from M (36, Cout, NT) + bias; one thread = VEC tiles of one channel
template <int VEC>
__global__ void __launch_bounds__(256)
wino4_out_kernel(const float *__restrict__ M, const float *__restrict__ bias, float *__restrict__ out,
                 int B, int Cout, int H, int W)
{
    const int Ht = H / 2, Wt = W / 2, Wv = Wt / VEC;
    const unsigned per_img = (unsigned)Ht * Wv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * per_img) return;
    const int o = blockIdx.y;
    const int b = idx / per_img;
    const unsigned r = idx - b * per_img;
    const int ty = r / Wv, tx0 = (r - ty * Wv) * VEC;
    const size_t NT = (size_t)B * Ht * Wt;
    const float *src = M + (size_t)o * NT + ((size_t)b * Ht + ty) * Wt + tx0;
    const size_t plane_stride = (size_t)Cout * NT;
    const float bv = bias ? bias[o] : 0.0f;
    // t[p][e][0..3] = (M A)[p] for tile e
    float t[6][VEC][4];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        float m[6][VEC];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float *sp = src + (size_t)(p * 6 + q) * plane_stride;
            if constexpr (VEC == 4) {
                const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(sp));
                m[q][0] = v.x; m[q][1] = v.y; m[q][2] = v.z; m[q][VEC - 1] = v.w;
            } else {
                m[q][0] = sp[0];
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) at6(m[0][e], m[1][e], m[2][e], m[3][e], m[4][e], m[5][e], t[p][e]);
    }
    const int W2 = 2 * W;
    float *dst = out + (((size_t)b * Cout + o) * 2 * H + 4 * ty) * W2 + 4 * tx0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            float y[4];
            at6(t[0][e][cc], t[1][e][cc], t[2][e][cc], t[3][e][cc], t[4][e][cc], t[5][e][cc], y);
#pragma unroll
            for (int a = 0; a < 4; ++a) t[a][e][cc] = y[a] + bv;       // reuse: t[a] = output row a
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float *d = dst + (size_t)a * W2 + 4 * e;
            *reinterpret_cast<float4 *>(d) = make_float4(t[a][e][0], t[a][e][1], t[a][e][2], t[a][e][3]);
        }
    }
}

// ------------------------------------------------------------------------------------
// Last head level + out_layer as ONE single-output-channel convolution.
// UpsamplingModule ends with conv3x3(C -> Co) (dmn.py:324) followed directly by out_layer, a 1x1
// conv Co -> 1 (dmn.py:327,341), with no non-linearity between them (non_linearity=False, the
// reference default): their composition is a 3x3 conv C -> 1 with
//     w_eff[c, tap] = sum_o w_out[o] * w[o, c, tap],  b_eff = b_out + sum_o w_out[o] * b[o],
// i.e. Co (= 32) times fewer multiplies, and the (B, 32, 512, 512) tensor (1 GB at B = 32) is
// never produced.  Fused with cat + bilinear x2 like the Winograd input transform: a lane owns
// VEC low-res pixels = VEC 2x2 output tiles and walks the channels; for each channel it rebuilds
// the 4x4 up-res patch of each tile from the 3x3 low-res neighbourhood.  HBM-bound on reading
// x and feat once.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void up4(float lm, float l0, float lp, bool first, bool last, float o[4])
{
    o[0] = first ? 0.0f : __builtin_fmaf(0.75f, lm, 0.25f * l0);
    o[1] = __builtin_fmaf(0.25f, lm, 0.75f * l0);
    o[2] = __builtin_fmaf(0.75f, l0, 0.25f * lp);
    o[3] = last ? 0.0f : __builtin_fmaf(0.25f, l0, 0.75f * lp);
}

template <int VEC>
__global__ void __launch_bounds__(256)
head_final_kernel(const float *__restrict__ x, const float *__restrict__ feat,
                  const float *__restrict__ weff, const float *__restrict__ bias0,
                  float *__restrict__ out, int B, int Cx, int Cf, int H, int W)
{
    const int Wv = W / VEC;
    const unsigned per_img = (unsigned)H * Wv;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)B * per_img) return;
    const int b = idx / per_img;
    const unsigned r = idx - b * per_img;
    const int ty = r / Wv, tx0 = (r - ty * Wv) * VEC;
    const int rows[3] = {(ty > 0 ? ty - 1 : 0) * W, ty * W, (ty < H - 1 ? ty + 1 : H - 1) * W};
    const int cl = tx0 > 0 ? tx0 - 1 : 0, cr = tx0 + VEC < W ? tx0 + VEC : W - 1;
    const bool top = ty == 0, bottom = ty == H - 1;
    const float b0 = bias0 ? bias0[0] : 0.0f;
    float y[VEC][2][2];
#pragma unroll
    for (int e = 0; e < VEC; ++e) y[e][0][0] = y[e][0][1] = y[e][1][0] = y[e][1][1] = b0;

    const int C = Cx + Cf;
    const size_t HW = (size_t)H * W;
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        const float *plane = c < Cx ? x + ((size_t)b * Cx + c) * HW : feat + ((size_t)b * Cf + (c - Cx)) * HW;
        const float *wc = weff + c * 9;                      // uniform: scalar loads
        float g[3][VEC][4];                                  // column-upsampled low-res rows
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float *row = plane + rows[j];
            float l[VEC + 2];
            l[0] = row[cl];
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + tx0);
                l[1] = v.x; l[2] = v.y; l[3] = v.z; l[4] = v.w;
            } else {
                l[1] = row[tx0];
            }
            l[VEC + 1] = row[cr];
#pragma unroll
            for (int e = 0; e < VEC; ++e) up4(l[e], l[e + 1], l[e + 2], tx0 + e == 0, tx0 + e == W - 1, g[j][e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float d[4][4];                                   // up-res patch rows 2ty-1 .. 2ty+2
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float col[4];
                up4(g[0][e][q], g[1][e][q], g[2][e][q], top, bottom, col);
#pragma unroll
                for (int p = 0; p < 4; ++p) d[p][q] = col[p];
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = wc[dy * 3 + dx];
                    y[e][0][0] = __builtin_fmaf(wv, d[dy][dx], y[e][0][0]);
                    y[e][0][1] = __builtin_fmaf(wv, d[dy][dx + 1], y[e][0][1]);
                    y[e][1][0] = __builtin_fmaf(wv, d[dy + 1][dx], y[e][1][0]);
                    y[e][1][1] = __builtin_fmaf(wv, d[dy + 1][dx + 1], y[e][1][1]);
                }
        }
    }
    float *dst = out + ((size_t)b * 2 * H + 2 * ty) * 2 * W + 2 * tx0;
    if constexpr (VEC == 4) {
        float4 *d0 = reinterpret_cast<float4 *>(dst), *d1 = reinterpret_cast<float4 *>(dst + 2 * W);
        d0[0] = make_float4(y[0][0][0], y[0][0][1], y[1][0][0], y[1][0][1]);
        d0[1] = make_float4(y[2][0][0], y[2][0][1], y[3][0][0], y[3][0][1]);
        d1[0] = make_float4(y[0][1][0], y[0][1][1], y[1][1][0], y[1][1][1]);
        d1[1] = make_float4(y[2][1][0], y[2][1][1], y[3][1][0], y[3][1][1]);
    } else {
        dst[0] = y[0][0][0]; dst[1] = y[0][0][1];
        dst[2 * W] = y[0][1][0]; dst[2 * W + 1] = y[0][1][1];
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
                                                 int B, int Cx, int Cf, int H, int W, int m, void *stream)
{
    if (m != 2 && m != 4) return fail(-2, "dmn_wino_cat_upsample2x_transform: m must be 2 or 4 (got %d)", m);
    if (m == 4 && ((H | W) & 1)) return fail(-4, "dmn_wino_cat_upsample2x_transform: m == 4 needs even H, W (got %dx%d)", H, W);
    if (Cx < 0 || Cf < 0) return fail(-4, "dmn_wino_cat_upsample2x_transform: negative channel count");
    if (int rc = check_wino("dmn_wino_cat_upsample2x_transform", B, Cx + Cf, H, W)) return rc;
    if ((!x && Cx > 0) || (!feat && Cf > 0) || !V) return fail(-1, "dmn_wino_cat_upsample2x_transform: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = W % 4 == 0 && !(((uintptr_t)x | (uintptr_t)feat | (uintptr_t)V) & 15);
    const long long items = (long long)B * H * (vec ? W / 4 : W);
    dim3 grid((unsigned)((items + 255) / 256), Cx + Cf);
    if (m == 4) {
        const bool vec4 = W % 8 == 0 && !(((uintptr_t)x | (uintptr_t)feat | (uintptr_t)V) & 15);
        const long long it4 = (long long)B * (H / 2) * (vec4 ? W / 8 : W / 2);
        dim3 grid4((unsigned)((it4 + 255) / 256), Cx + Cf);
        if (vec4) hipLaunchKernelGGL(wino4_in_kernel<4>, grid4, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
        else      hipLaunchKernelGGL(wino4_in_kernel<1>, grid4, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
        return launch_status("dmn_wino_cat_upsample2x_transform");
    }
    if (vec) hipLaunchKernelGGL(wino_in_kernel<4>, grid, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
    else     hipLaunchKernelGGL(wino_in_kernel<1>, grid, dim3(256), 0, s, x, feat, V, B, Cx, Cf, H, W);
    return launch_status("dmn_wino_cat_upsample2x_transform");
}

extern "C" int dmn_wino_output_transform(const float *M, const float *bias, float *out,
                                         int B, int Cout, int H, int W, int m, void *stream)
{
    if (m != 2 && m != 4) return fail(-2, "dmn_wino_output_transform: m must be 2 or 4 (got %d)", m);
    if (m == 4 && ((H | W) & 1)) return fail(-4, "dmn_wino_output_transform: m == 4 needs even H, W (got %dx%d)", H, W);
    if (int rc = check_wino("dmn_wino_output_transform", B, Cout, H, W)) return rc;
    if (!M || !out) return fail(-1, "dmn_wino_output_transform: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = W % 4 == 0 && !(((uintptr_t)M | (uintptr_t)out) & 15);
    const long long items = (long long)B * H * (vec ? W / 4 : W);
    dim3 grid((unsigned)((items + 255) / 256), Cout);
    if (m == 4) {
        if ((uintptr_t)out & 15) return fail(-7, "dmn_wino_output_transform: out must be 16-byte aligned for m == 4");
        const bool vec4 = W % 8 == 0 && !(((uintptr_t)M | (uintptr_t)out) & 15);
        const long long it4 = (long long)B * (H / 2) * (vec4 ? W / 8 : W / 2);
        dim3 grid4((unsigned)((it4 + 255) / 256), Cout);
        if (vec4) hipLaunchKernelGGL(wino4_out_kernel<4>, grid4, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
        else      hipLaunchKernelGGL(wino4_out_kernel<1>, grid4, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
        return launch_status("dmn_wino_output_transform");
    }
    if (vec) hipLaunchKernelGGL(wino_out_kernel<4>, grid, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
    else     hipLaunchKernelGGL(wino_out_kernel<1>, grid, dim3(256), 0, s, M, bias, out, B, Cout, H, W);
    return launch_status("dmn_wino_output_transform");
}

extern "C" int dmn_head_final(const float *x, const float *feat, const float *weff, const float *bias0,
                              float *out, int B, int Cx, int Cf, int H, int W, void *stream)
{
    if (Cx < 0 || Cf < 0) return fail(-4, "dmn_head_final: negative channel count");
    if (int rc = check_wino("dmn_head_final", B, Cx + Cf, H, W)) return rc;
    if ((!x && Cx > 0) || (!feat && Cf > 0) || !weff || !out) return fail(-1, "dmn_head_final: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = W % 4 == 0 && !(((uintptr_t)x | (uintptr_t)feat | (uintptr_t)out) & 15);
    const long long items = (long long)B * H * (vec ? W / 4 : W);
    dim3 grid((unsigned)((items + 255) / 256));
    if (vec) hipLaunchKernelGGL(head_final_kernel<4>, grid, dim3(256), 0, s, x, feat, weff, bias0, out, B, Cx, Cf, H, W);
    else     hipLaunchKernelGGL(head_final_kernel<1>, grid, dim3(256), 0, s, x, feat, weff, bias0, out, B, Cx, Cf, H, W);
    return launch_status("dmn_head_final");
}
