// dms_amd/csrc/glue.hip -- concat / coordinate / bilinear-x2 glue kernels (gfx950).
//
//  dmn_cat_coords        : dmn.py:142 + 225-246   (coordinate channels written on device)
//  dmn_cat_upsample2x    : dmn.py:324,337-338     (cat + nn.Upsample(x2, bilinear) in one pass)
//  dmn_bias_sigmoid      : dmn.py:173-174         (the filter Linear's bias add + F.sigmoid, in place)
//
// All of these are pure HBM streaming: one 16-byte store per lane, inputs re-read from
// L1/L2 (each input element feeds 4 outputs of the x2 upsample).
#include "common.h"

namespace dmn {

// coordinate channel `ch` at (h, w): python-float (fp64) arithmetic in the reference's order
__device__ __forceinline__ float coord_val(int ch, int h, int w, int H, int W)
{
    const double xmin = (double)w / W * 2 - 1, xmax = (double)(w + 1) / W * 2 - 1;
    const double ymin = (double)h / H * 2 - 1, ymax = (double)(h + 1) / H * 2 - 1;
    switch (ch) {
        case 0: return (float)xmin;
        case 1: return (float)ymin;
        case 2: return (float)xmax;
        case 3: return (float)ymax;
        case 4: return (float)((xmin + xmax) / 2);
        case 5: return (float)((ymin + ymax) / 2);
        case 6: return (float)(1.0 / W);
        default: return (float)(1.0 / H);
    }
}

template <int VEC>
__global__ void __launch_bounds__(256)
cat_coords_kernel(const float *__restrict__ vis, float *__restrict__ V, int B, int C, int H, int W)
{
    const int S = H * W, Sv = S / VEC, C8 = C + 8;
    const long long total = (long long)B * C8 * Sv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int sv = (int)(i % Sv);
        const long long plane = i / Sv;
        const int c = (int)(plane % C8);
        const int b = (int)(plane / C8);
        float *dst = V + plane * S + (long long)sv * VEC;
        if (c < C) {
            const float *src = vis + ((long long)b * C + c) * S + (long long)sv * VEC;
            if (VEC == 4) *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
            else *dst = *src;
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int s = sv * VEC + e;
                dst[e] = coord_val(c - C, s / W, s % W, H, W);
            }
        }
    }
}

// PyTorch's area_pixel_compute_source_index for align_corners=False, scale 1/2
__device__ __forceinline__ void src_index(int o, int n_in, int &i0, int &i1, float &l0, float &l1)
{
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

template <int VEC>
__global__ void __launch_bounds__(256)
cat_upsample2x_kernel(const float *__restrict__ x, const float *__restrict__ feat,
                      float *__restrict__ out, int B, int Cx, int Cf, int H, int W)
{
    const int Ct = Cx + Cf, OH = 2 * H, OW = 2 * W, OWv = OW / VEC;
    const long long total = (long long)B * Ct * OH * OWv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int oxv = (int)(i % OWv);
        long long r = i / OWv;
        const int oy = (int)(r % OH);
        r /= OH;
        const int c = (int)(r % Ct);
        const int b = (int)(r / Ct);
        const float *src = (c < Cx) ? x + ((long long)b * Cx + c) * H * W
                                    : feat + ((long long)b * Cf + (c - Cx)) * H * W;
        int y0, y1; float ly0, ly1;
        src_index(oy, H, y0, y1, ly0, ly1);
        const float *r0 = src + (long long)y0 * W, *r1 = src + (long long)y1 * W;
        float res[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            int x0, x1; float lx0, lx1;
            src_index(oxv * VEC + e, W, x0, x1, lx0, lx1);
            res[e] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
        }
        float *dst = out + (((long long)b * Ct + c) * OH + oy) * OW + (long long)oxv * VEC;
        if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(res[0], res[1], res[2], res[3]);
        else dst[0] = res[0];
    }
}

// Fast path (W even): one lane = R consecutive input rows x CW input columns -> a
// 2R x 2CW output block.  Per input row one aligned load; the two neighbour columns come
// from the adjacent lanes by wave shuffle (loaded only at a row edge or a wave edge); the
// vertical pass slides a 3-row window, so R rows cost R+2 row loads instead of 3R.  Flat work
// index (plane, row group, q) so that small planes still fill 256-thread workgroups.
// Measured (B=32, 512 px levels, MI355X): CW = 2 -- every 16-byte store instruction of a wave
// writes a DENSE 1 KiB run -- reaches 3.9-5.1 TB/s; CW = 4 (16-byte loads, but each store
// instruction covers 2 KiB at 50 % density) only 3.5-4.3 TB/s.  Weights are PyTorch's
// (align_corners=False, scale 1/2): 0.25/0.75 inside, (1, 0) on the first row/column,
// clamped neighbour on the last.
template <int R, int CW>     // CW input columns per lane: 4 (16-byte loads) or 2 (dense stores)
__global__ void __launch_bounds__(256)
cat_upsample2x_rows(const float *__restrict__ x, const float *__restrict__ feat,
                    float *__restrict__ out, long long total, int Cx, int Cf, int H, int W)
{
    constexpr int NO = 2 * CW;                        // outputs per lane per output row
    const int Ct = Cx + Cf, Wq = W / CW, G = (H + R - 1) / R;
    const int lane = threadIdx.x & 63;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = id < total;
    const long long idc = active ? id : total - 1;    // inactive lanes mirror the last one
    const int q = (int)(idc % Wq);
    const long long pg = idc / Wq;
    const int g = (int)(pg % G);
    const int plane = (int)(pg / G);
    const int b = plane / Ct, c = plane - b * Ct;
    const float *src = (c < Cx) ? x + ((size_t)b * Cx + c) * H * W
                                : feat + ((size_t)b * Cf + (c - Cx)) * H * W;
    const bool own_a = (lane == 0) || (q == 0), own_d = (lane == 63) || (q == Wq - 1);
    const int ca = q > 0 ? CW * q - 1 : 0, cd = q < Wq - 1 ? CW * q + CW : W - 1;
    const float w0lo = q > 0 ? 0.25f : 0.0f, w0hi = 1.0f - w0lo;
    const int m0 = g * R;

    // horizontally interpolated row `row` (clamped into the plane): NO outputs of this lane
    auto hrow = [&](int row, float (&hr)[NO]) {
        row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
        const float *rp = src + (size_t)row * W;
        float v[CW];
        if (CW == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(rp + 4 * q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[CW - 1] = t.w;
        } else {
            const float2 t = *reinterpret_cast<const float2 *>(rp + 2 * q);
            v[0] = t.x; v[CW - 1] = t.y;
        }
        float a = __shfl_up(v[CW - 1], 1, 64), d = __shfl_down(v[0], 1, 64);
        if (own_a) a = rp[ca];
        if (own_d) d = rp[cd];
        hr[0] = w0lo * a + w0hi * v[0];
#pragma unroll
        for (int j = 0; j + 1 < CW; ++j) {
            hr[2 * j + 1] = 0.75f * v[j] + 0.25f * v[j + 1];
            hr[2 * j + 2] = 0.25f * v[j] + 0.75f * v[j + 1];
        }
        hr[NO - 1] = 0.75f * v[CW - 1] + 0.25f * d;
    };
    float win[3][NO];                                 // rows m-1, m, m+1 (rotating)
    hrow(m0 - 1, win[0]);
    hrow(m0, win[1]);
    float *oplane = out + (size_t)plane * (2 * H) * (2 * W) + NO * q;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = m0 + r;
        float (&prev)[NO] = win[r % 3], (&cur)[NO] = win[(r + 1) % 3], (&next)[NO] = win[(r + 2) % 3];
        hrow(m + 1, next);
        if (active && m < H) {
            const float v0lo = m > 0 ? 0.25f : 0.0f, v0hi = 1.0f - v0lo;
            float o0[NO], o1[NO];
#pragma unroll
            for (int e = 0; e < NO; ++e) {
                o0[e] = v0lo * prev[e] + v0hi * cur[e];
                o1[e] = 0.75f * cur[e] + 0.25f * next[e];
            }
            float *op = oplane + (size_t)(2 * m) * (2 * W);
#pragma unroll
            for (int e = 0; e < NO; e += 4) {
                store_stream4(op + e, o0[e], o0[e + 1], o0[e + 2], o0[e + 3]);
                store_stream4(op + 2 * W + e, o1[e], o1[e + 1], o1[e + 2], o1[e + 3]);
            }
        }
    }
}

// gather form of the transpose: every input pixel sums the <= 4x4 outputs that read it
__global__ void __launch_bounds__(256)
cat_upsample2x_bwd_kernel(const float *__restrict__ g, float *__restrict__ gx,
                          float *__restrict__ gfeat, int B, int Cx, int Cf, int H, int W)
{
    const int Ct = Cx + Cf, OH = 2 * H, OW = 2 * W;
    const long long total = (long long)B * Ct * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        long long r = i / W;
        const int yy = (int)(r % H);
        r /= H;
        const int c = (int)(r % Ct);
        const int b = (int)(r / Ct);
        float *dst = (c < Cx) ? gx : gfeat;
        if (!dst) continue;
        const float *gp = g + ((long long)b * Ct + c) * OH * OW;
        float acc = 0.0f;
        const int oy_lo = max(2 * yy - 1, 0), oy_hi = min(2 * yy + 2, OH - 1);
        const int ox_lo = max(2 * xx - 1, 0), ox_hi = min(2 * xx + 2, OW - 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly0, ly1;
            src_index(oy, H, y0, y1, ly0, ly1);
            const float wy = (y0 == yy ? ly0 : 0.0f) + (y1 == yy ? ly1 : 0.0f);
            if (wy == 0.0f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx0, lx1;
                src_index(ox, W, x0, x1, lx0, lx1);
                const float wx = (x0 == xx ? lx0 : 0.0f) + (x1 == xx ? lx1 : 0.0f);
                acc += wy * wx * gp[(long long)oy * OW + ox];
            }
        }
        const long long o = (c < Cx) ? (((long long)b * Cx + c) * H + yy) * W + xx
                                     : (((long long)b * Cf + (c - Cx)) * H + yy) * W + xx;
        dst[o] = acc;
    }
}

static int stream_grid(long long work_items)
{
    long long blocks = (work_items + 255) / 256;
    const long long cap = (long long)NUM_CU * 16;   // grid-stride beyond 16 blocks per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// x[r, c] = sigmoid(x[r, c] + bias[c]) in place, 4 columns per lane (cols % 4 == 0) or 1.
// One streaming pass: 8 B per element.  The GEMM in front of it runs bias-free (beta = 0).
template <int VEC>
__global__ void __launch_bounds__(256)
bias_sigmoid_kernel(float *__restrict__ x, const float *__restrict__ bias, long long rows, int cols)
{
    const int cv = cols / VEC;
    const int c = (blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= cv) return;
    float b[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) b[e] = bias[c * VEC + e];
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
        float *p = x + r * cols + (long long)c * VEC;
        if (VEC == 4) {
            const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
            *reinterpret_cast<float4 *>(p) = make_float4(sigmoid_f(v.x + b[0]), sigmoid_f(v.y + b[1]),
                                                         sigmoid_f(v.z + b[2]), sigmoid_f(v.w + b[VEC - 1]));
        } else {
            *p = sigmoid_f(*p + b[0]);
        }
    }
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_bias_sigmoid(float *x, const float *bias, long long rows, int cols, void *stream)
{
    if (!x || !bias) return fail(-1, "dmn_bias_sigmoid: NULL pointer");
    if (rows < 0 || cols <= 0) return fail(-4, "dmn_bias_sigmoid: bad sizes rows=%lld cols=%d", rows, cols);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (cols % 4 == 0) && !(((uintptr_t)x | (uintptr_t)bias) & 15);
    const int cv = vec ? cols / 4 : cols;
    const unsigned gy = (unsigned)(rows < 2048 ? rows : 2048);
    dim3 grid((cv + 255) / 256, gy);
    if (vec) hipLaunchKernelGGL(bias_sigmoid_kernel<4>, grid, dim3(256), 0, s, x, bias, rows, cols);
    else     hipLaunchKernelGGL(bias_sigmoid_kernel<1>, grid, dim3(256), 0, s, x, bias, rows, cols);
    return launch_status("dmn_bias_sigmoid");
}

extern "C" int dmn_cat_coords(const float *vis, float *V, int B, int C, int H, int W, void *stream)
{
    if ((!vis && C > 0) || !V) return fail(-1, "dmn_cat_coords: NULL pointer");
    if (B <= 0 || C < 0 || H <= 0 || W <= 0) return fail(-4, "dmn_cat_coords: bad sizes B=%d C=%d H=%d W=%d", B, C, H, W);
    const int S = H * W;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (S % 4 == 0) && !(((uintptr_t)vis | (uintptr_t)V) & 15);
    if (vec) {
        const long long n = (long long)B * (C + 8) * (S / 4);
        hipLaunchKernelGGL(cat_coords_kernel<4>, dim3(stream_grid(n)), dim3(256), 0, s, vis, V, B, C, H, W);
    } else {
        const long long n = (long long)B * (C + 8) * S;
        hipLaunchKernelGGL(cat_coords_kernel<1>, dim3(stream_grid(n)), dim3(256), 0, s, vis, V, B, C, H, W);
    }
    return launch_status("dmn_cat_coords");
}

extern "C" int dmn_cat_upsample2x(const float *x, const float *feat, float *out, int B, int Cx,
                                  int Cf, int H, int W, void *stream)
{
    if (!out || (Cx > 0 && !x) || (Cf > 0 && !feat)) return fail(-1, "dmn_cat_upsample2x: NULL pointer");
    if (B <= 0 || Cx < 0 || Cf < 0 || Cx + Cf <= 0 || H <= 0 || W <= 0)
        return fail(-4, "dmn_cat_upsample2x: bad sizes B=%d Cx=%d Cf=%d H=%d W=%d", B, Cx, Cf, H, W);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = ((2 * W) % 4 == 0) && !((uintptr_t)out & 15);
    const long long planes = (long long)B * (Cx + Cf);
    const bool fast = vec && !(((uintptr_t)x | (uintptr_t)feat) & 7) && planes < 0x7fffffffLL;
    if (fast) {
        constexpr int R = 8, CW = 2;
        const long long total = planes * ((H + R - 1) / R) * (W / CW);
        const long long blocks = (total + 255) / 256;
        if (blocks > 0x7fffffffLL) return fail(-5, "dmn_cat_upsample2x: tensor too large");
        hipLaunchKernelGGL((cat_upsample2x_rows<R, CW>), dim3((unsigned)blocks), dim3(256), 0, s, x, feat, out, total, Cx, Cf, H, W);
    } else if (vec) {
        const long long n = (long long)B * (Cx + Cf) * (2 * H) * (2 * W / 4);
        hipLaunchKernelGGL(cat_upsample2x_kernel<4>, dim3(stream_grid(n)), dim3(256), 0, s, x, feat, out, B, Cx, Cf, H, W);
    } else {
        const long long n = (long long)B * (Cx + Cf) * (2 * H) * (2 * W);
        hipLaunchKernelGGL(cat_upsample2x_kernel<1>, dim3(stream_grid(n)), dim3(256), 0, s, x, feat, out, B, Cx, Cf, H, W);
    }
    return launch_status("dmn_cat_upsample2x");
}

extern "C" int dmn_cat_upsample2x_bwd(const float *g, float *gx, float *gfeat, int B, int Cx,
                                      int Cf, int H, int W, void *stream)
{
    if (!g) return fail(-1, "dmn_cat_upsample2x_bwd: NULL gradient");
    if (B <= 0 || Cx < 0 || Cf < 0 || Cx + Cf <= 0 || H <= 0 || W <= 0)
        return fail(-4, "dmn_cat_upsample2x_bwd: bad sizes");
    const long long n = (long long)B * (Cx + Cf) * H * W;
    hipLaunchKernelGGL(cat_upsample2x_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, g, gx, gfeat, B, Cx, Cf, H, W);
    return launch_status("dmn_cat_upsample2x_bwd");
}
