// dms_amd/csrc/gconv.hip -- grouped 3x3 convolution of the DPN blocks with its two BatchNorm+ReLU
// neighbours fused in (inference, gfx950).
//
// DualPathBlock (dmn_pytorch/models/dpn/dpn_model.py:246-263) runs, between its two 1x1 convs,
//     y = c3x3_b.conv( relu(c3x3_b.bn(x)) )          grouped, 32 groups of 3 / 6 / 12 / 24 channels
//     z = relu(c1x1_c.bn(y))                         pre-activation of the next 1x1 conv
// With 3..24 channels per group this convolution is a few hundred flops per output element:
// it is HBM-bound (read x once, write z once), yet MIOpen's Winograd assembly kernel needs
// 1.0 / 0.30 / 0.11 / 0.05 ms for the four DPN92 stages at B=32 where the traffic is worth
// 0.08 / 0.04 / 0.02 / 0.01 ms (profiles/r1_miopen_fused_conv_probe.txt), and the two
// BatchNorm+ReLU passes around it cost two more launches and two more round trips.
//
// One workgroup = one (image, group, spatial tile).  The tile's input channels are staged into
// LDS a few at a time WITH the first BN+ReLU applied (zero outside the image: the conv pads the
// activated tensor), the group's weights sit in LDS as [ci][tap][co] so that the CG output
// channels of one (ci, tap) are one broadcast read; a lane owns 4 adjacent pixels x all CG
// output channels in registers; the second BN+ReLU is applied to the accumulators.
#include "common.h"

namespace dmn {

constexpr int GC_PX = 4;        // pixels per lane (along x)
// input channels staged per LDS pass: a divisor of CG (6 for 6/12/24, 4 for 4/8/16/32, else CG)
constexpr int gc_chunk(int cg) { return cg % 6 == 0 ? 6 : (cg % 4 == 0 ? 4 : cg); }

template <int CG>
__global__ void __launch_bounds__(256)
gconv3x3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                const float *__restrict__ m1, const float *__restrict__ s1, const float *__restrict__ b1,
                const float *__restrict__ m2, const float *__restrict__ s2, const float *__restrict__ b2,
                float *__restrict__ y, int C, int H, int W, int txn)
{
    constexpr int CH = gc_chunk(CG);
    extern __shared__ float lds[];
    const int tw = txn * GC_PX, th = 256 / txn;              // tile size in pixels
    const int lw = tw + 2, lh = th + 2;                      // staged tile incl. halo
    float *Wl = lds;                                         // [CG ci][9][CG co]
    float *Xl = lds + CG * 9 * CG;                           // [CH][lh][lw]
    const int tiles_x = (W + tw - 1) / tw;
    const int x0 = (blockIdx.x % tiles_x) * tw, y0 = (blockIdx.x / tiles_x) * th;
    const int g = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, tx = tid % txn, ty = tid / txn;
    const int c0 = g * CG;

    // weights of the group, already [ci][tap][co]
    for (int i = tid; i < CG * CG * 9; i += 256) Wl[i] = w[(size_t)g * CG * 9 * CG + i];

    float acc[CG][GC_PX];
#pragma unroll
    for (int co = 0; co < CG; ++co)
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) acc[co][p] = 0.0f;

    const float *xin = x + ((size_t)n * C + c0) * H * W;
    for (int cb = 0; cb < CG; cb += CH) {
        __syncthreads();                                     // previous chunk fully consumed
        // 4 tile rows per pass, 64 lanes along x: coalesced row segments, no index divisions
        for (int ci = 0; ci < CH; ++ci) {
            const int c = c0 + cb + ci;
            const float mm = m1 ? m1[c] : 0.0f, ss = m1 ? s1[c] : 1.0f, bb = m1 ? b1[c] : 0.0f;
            const float *plane = xin + (size_t)(cb + ci) * H * W;
            for (int ly = tid >> 6; ly < lh; ly += 4) {
                const int gy = y0 + ly - 1;
                const bool row_ok = gy >= 0 && gy < H;
                for (int lx = tid & 63; lx < lw; lx += 64) {
                    const int gx = x0 + lx - 1;
                    float v = 0.0f;
                    if (row_ok && gx >= 0 && gx < W) {
                        v = plane[(size_t)gy * W + gx];
                        if (m1) v = fmaxf(__builtin_fmaf(v - mm, ss, bb), 0.0f);
                    }
                    Xl[(ci * lh + ly) * lw + lx] = v;
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ci = 0; ci < CH; ++ci) {
            float in[3][GC_PX + 2];
            const float *xp = Xl + ((size_t)ci * lh + ty) * lw + tx * GC_PX;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < GC_PX + 2; ++q) in[r][q] = xp[r * lw + q];
            const float *wp = Wl + (size_t)(cb + ci) * 9 * CG;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int co = 0; co < CG; ++co) {
                    const float wv = wp[tap * CG + co];
#pragma unroll
                    for (int p = 0; p < GC_PX; ++p)
                        acc[co][p] = __builtin_fmaf(wv, in[dy][p + dx], acc[co][p]);
                }
            }
        }
    }

    const int oy = y0 + ty, ox = x0 + tx * GC_PX;
    if (oy >= H || ox >= W) return;
    float *yout = y + (((size_t)n * C + c0) * H + oy) * W + ox;
    const bool vec = (W % 4 == 0);                           // ox % 4 == 0 by construction
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        float o[GC_PX];
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) o[p] = acc[co][p];
        if (m2) {
            const float mm = m2[c0 + co], ss = s2[c0 + co], bb = b2[c0 + co];
#pragma unroll
            for (int p = 0; p < GC_PX; ++p) o[p] = fmaxf(__builtin_fmaf(o[p] - mm, ss, bb), 0.0f);
        }
        float *dst = yout + (size_t)co * H * W;
        if (vec) {
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int p = 0; p < GC_PX; ++p)
                if (ox + p < W) dst[p] = o[p];
        }
    }
}

// Fast path (W % 4 == 0): no LDS staging of the input and no barrier in the channel loop, so the
// loads of successive channels overlap.  A lane reads its 4 pixels of the three rows as float4
// (the rows overlap between vertical neighbours: L1 hits) and gets the two halo pixels from
// its horizontal neighbours by wave shuffle; only tile-edge lanes load them.  SPLIT > 1: small
// images (16 x 16 at the last DPN stage) would fill 64 of the 256 lanes, so the CG output
// channels are split over SPLIT lane groups that share the same pixels.
// Register budget: the kernel is HBM/latency-bound, so occupancy matters more than keeping every
// weight and prefetched row in registers (unconstrained, hipcc used 188 VGPRs for CG = 3, i.e.
// 2 waves per SIMD, and the kernel ran at 20 % of HBM bandwidth).
// (Measured and rejected: capping the CG = 12 instantiation at 128 VGPRs so that the 4 workgroups
// per CU of DPN92's conv4 stage run in one round -- the 37 spilled dwords cost more, 59 vs 56 us.)
template <int CG, int SPLIT>
__global__ void __launch_bounds__(256)
gconv3x3_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                     const float *__restrict__ m1, const float *__restrict__ s1, const float *__restrict__ b1,
                     const float *__restrict__ m2, const float *__restrict__ s2, const float *__restrict__ b2,
                     float *__restrict__ y, int C, int H, int W, int txn)
{
    constexpr int COS = CG / SPLIT;                          // output channels per lane
    constexpr int LANES = 256 / SPLIT;                       // pixel lanes (a multiple of 64)
    const int tw = txn * GC_PX, th = LANES / txn;
    const int tiles_x = (W + tw - 1) / tw;
    const int x0 = (blockIdx.x % tiles_x) * tw, y0 = (blockIdx.x / tiles_x) * th;
    const int g = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid % LANES;
    // LANES is a multiple of 64, so `part` is wave-uniform: say so, the weight addresses below
    // are then scalar and the weights arrive through the scalar cache straight into SGPR
    // operands of the FMAs (no LDS stage, no VGPRs spent on weights)
    const int part = __builtin_amdgcn_readfirstlane(tid / LANES);
    const int tx = lane % txn, ty = lane / txn;
    const int c0 = g * CG;
    const float *wg = w + (size_t)g * CG * 9 * CG + part * COS;     // [ci][tap][co] of this group

    const int oy = y0 + ty, ox = x0 + tx * GC_PX;
    const bool col_ok = ox < W;                              // W % 4 == 0: whole float4 in or out
    // Every load is UNCONDITIONAL on a clamped (always valid) address and invalid values are
    // zeroed by a select afterwards: per-lane `if (ok) load` turns into one exec-masked block
    // and one s_waitcnt per load, which serialises the loads' latencies.
    const bool halo = tiles_x > 1;                           // uniform: tiles narrower than the image
    const int oxc = col_ok ? ox : 0;
    const int oxl = ox > 0 ? (ox - 1 < W ? ox - 1 : 0) : 0, oxr = ox + GC_PX < W ? ox + GC_PX : 0;
    const bool l_ok = col_ok && ox > 0, r_ok = ox + GC_PX < W;
    bool row_ok[3];
    int row_off[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int gy = oy + r - 1;
        row_ok[r] = gy >= 0 && gy < H;
        row_off[r] = (gy < 0 ? 0 : (gy < H ? gy : H - 1)) * W;
    }

    // even widths: two adjacent output channels per v_pk_fma_f32 (weight pair = one 64-bit scalar
    // operand, the pixel broadcast through op_sel): twice the fp32 FMA rate of v_fma_f32
    constexpr bool PK = (CG % 2 == 0) && (COS % 2 == 0);
    float acc[PK ? 1 : COS][GC_PX];
    f32x2_t accp[PK ? COS / 2 : 1][GC_PX];
#pragma unroll
    for (int co = 0; co < (PK ? 1 : COS); ++co)
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) acc[co][p] = 0.0f;
#pragma unroll
    for (int co = 0; co < (PK ? COS / 2 : 1); ++co)
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) accp[co][p] = (f32x2_t){0.0f, 0.0f};

    const float *xin = x + ((size_t)n * C + c0) * H * W;
    // software pipeline over the input channels: the raw rows of channel ci + 1 are requested
    // before channel ci is activated, shuffled and multiplied
    struct Raw { float4 v[3]; float l[3], r[3]; };
    auto load_raw = [&](int ci, Raw &R) {
        const float *plane = xin + (size_t)ci * H * W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R.v[r] = *reinterpret_cast<const float4 *>(plane + row_off[r] + oxc);
            if (halo) {
                R.l[r] = plane[row_off[r] + oxl];
                R.r[r] = plane[row_off[r] + oxr];
            }
        }
    };
    // one input row at a time: 6 activated pixels live, 3 * COS weights per row
    auto consume = [&](const Raw &R, int ci) {
        const int c = c0 + ci;
        const float mm = m1 ? m1[c] : 0.0f, ss = m1 ? s1[c] : 1.0f, bb = m1 ? b1[c] : 0.0f;
        const float *wp = wg + (size_t)ci * 9 * CG;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float4 v = R.v[r];
            float l = halo ? R.l[r] : 0.0f, rr = halo ? R.r[r] : 0.0f;
            if (m1) {
                v.x = fmaxf(__builtin_fmaf(v.x - mm, ss, bb), 0.0f);
                v.y = fmaxf(__builtin_fmaf(v.y - mm, ss, bb), 0.0f);
                v.z = fmaxf(__builtin_fmaf(v.z - mm, ss, bb), 0.0f);
                v.w = fmaxf(__builtin_fmaf(v.w - mm, ss, bb), 0.0f);
                l = fmaxf(__builtin_fmaf(l - mm, ss, bb), 0.0f);
                rr = fmaxf(__builtin_fmaf(rr - mm, ss, bb), 0.0f);
            }
            const bool ok = row_ok[r] && col_ok;             // the conv pads the ACTIVATED tensor with zeros
            v.x = ok ? v.x : 0.0f; v.y = ok ? v.y : 0.0f; v.z = ok ? v.z : 0.0f; v.w = ok ? v.w : 0.0f;
            l = (halo && row_ok[r] && l_ok) ? l : 0.0f;
            rr = (halo && row_ok[r] && r_ok) ? rr : 0.0f;
            // halo from the horizontal neighbours (same wave: txn divides 64)
            const float from_l = __shfl_up(v.w, 1), from_r = __shfl_down(v.x, 1);
            const float in[GC_PX + 2] = {tx == 0 ? l : from_l, v.x, v.y, v.z, v.w,
                                         tx == txn - 1 ? rr : from_r};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if constexpr (PK) {
#pragma unroll
                    for (int cp = 0; cp < COS / 2; ++cp) {
                        const f32x2_t wv = *reinterpret_cast<const f32x2_t *>(wp + (r * 3 + dx) * CG + 2 * cp);
#pragma unroll
                        for (int p = 0; p < GC_PX; ++p)
                            accp[cp][p] = __builtin_elementwise_fma(wv, (f32x2_t){in[p + dx], in[p + dx]}, accp[cp][p]);
                    }
                } else {
#pragma unroll
                    for (int co = 0; co < COS; ++co) {
                        const float wv = wp[(r * 3 + dx) * CG + co];
#pragma unroll
                        for (int p = 0; p < GC_PX; ++p)
                            acc[co][p] = __builtin_fmaf(wv, in[p + dx], acc[co][p]);
                    }
                }
            }
        }
    };
    Raw ra, rb;
    load_raw(0, ra);
#pragma unroll 1
    for (int ci = 0; ci + 1 < CG; ci += 2) {
        load_raw(ci + 1, rb);
        consume(ra, ci);
        if (ci + 2 < CG) load_raw(ci + 2, ra);
        consume(rb, ci + 1);
    }
    if constexpr (CG % 2 == 1) consume(ra, CG - 1);

    if (oy >= H || !col_ok) return;
    float *yout = y + (((size_t)n * C + c0 + part * COS) * H + oy) * W + ox;
#pragma unroll
    for (int co = 0; co < COS; ++co) {
        float o[GC_PX];
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) o[p] = PK ? accp[PK ? co / 2 : 0][p][co % 2] : acc[PK ? 0 : co][p];
        if (m2) {
            const int c = c0 + part * COS + co;
            const float mm = m2[c], ss = s2[c], bb = b2[c];
#pragma unroll
            for (int p = 0; p < GC_PX; ++p) o[p] = fmaxf(__builtin_fmaf(o[p] - mm, ss, bb), 0.0f);
        }
        *reinterpret_cast<float4 *>(yout + (size_t)co * H * W) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Stride 2 (the first block of DPN stages 3-5): same structure in OUTPUT coordinates.  A lane's 4
// output pixels ox .. ox+3 read input columns 2ox-1 .. 2ox+7: two float4 plus one halo pixel
// from the left neighbour; input rows 2oy-1 .. 2oy+1.  Needs H % 2 == 0 and W % 8 == 0.
template <int CG, int SPLIT>
__global__ void __launch_bounds__(256)
gconv3x3_s2_kernel(const float *__restrict__ x, const float *__restrict__ w,
                   const float *__restrict__ m1, const float *__restrict__ s1, const float *__restrict__ b1,
                   const float *__restrict__ m2, const float *__restrict__ s2, const float *__restrict__ b2,
                   float *__restrict__ y, int C, int H, int W, int txn)
{
    constexpr int COS = CG / SPLIT;
    constexpr int LANES = 256 / SPLIT;
    const int Ho = H / 2, Wo = W / 2;
    const int tw = txn * GC_PX, th = LANES / txn;            // tile in output pixels
    const int tiles_x = (Wo + tw - 1) / tw;
    const int x0 = (blockIdx.x % tiles_x) * tw, y0 = (blockIdx.x / tiles_x) * th;
    const int g = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid % LANES;
    const int part = __builtin_amdgcn_readfirstlane(tid / LANES);
    const int tx = lane % txn, ty = lane / txn;
    const int c0 = g * CG;
    const float *wg = w + (size_t)g * CG * 9 * CG + part * COS;

    const int oy = y0 + ty, ox = x0 + tx * GC_PX;
    const bool col_ok = ox < Wo;
    const bool halo = tiles_x > 1;
    const int ixc = col_ok ? 2 * ox : 0;                     // first input column of the lane
    const int ixl = (col_ok && ox > 0) ? 2 * ox - 1 : 0;
    const bool l_ok = col_ok && ox > 0;
    bool row_ok[3];
    int row_off[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int gy = 2 * oy + r - 1;
        row_ok[r] = gy >= 0 && gy < H;
        row_off[r] = (gy < 0 ? 0 : (gy < H ? gy : H - 1)) * W;
    }

    // even widths: two adjacent output channels per v_pk_fma_f32 (weight pair = one 64-bit scalar
    // operand, the pixel broadcast through op_sel): twice the fp32 FMA rate of v_fma_f32
    constexpr bool PK = (CG % 2 == 0) && (COS % 2 == 0);
    float acc[PK ? 1 : COS][GC_PX];
    f32x2_t accp[PK ? COS / 2 : 1][GC_PX];
#pragma unroll
    for (int co = 0; co < (PK ? 1 : COS); ++co)
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) acc[co][p] = 0.0f;
#pragma unroll
    for (int co = 0; co < (PK ? COS / 2 : 1); ++co)
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) accp[co][p] = (f32x2_t){0.0f, 0.0f};

    const float *xin = x + ((size_t)n * C + c0) * H * W;
    struct Raw { float4 a[3], b[3]; float l[3]; };
    auto load_raw = [&](int ci, Raw &R) {
        const float *plane = xin + (size_t)ci * H * W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R.a[r] = *reinterpret_cast<const float4 *>(plane + row_off[r] + ixc);
            R.b[r] = *reinterpret_cast<const float4 *>(plane + row_off[r] + ixc + 4);
            if (halo) R.l[r] = plane[row_off[r] + ixl];
        }
    };
    auto consume = [&](const Raw &R, int ci) {
        const int c = c0 + ci;
        const float mm = m1 ? m1[c] : 0.0f, ss = m1 ? s1[c] : 1.0f, bb = m1 ? b1[c] : 0.0f;
        const float *wp = wg + (size_t)ci * 9 * CG;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float v[8] = {R.a[r].x, R.a[r].y, R.a[r].z, R.a[r].w, R.b[r].x, R.b[r].y, R.b[r].z, R.b[r].w};
            float l = halo ? R.l[r] : 0.0f;
            const bool ok = row_ok[r] && col_ok;
            if (m1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(__builtin_fmaf(v[q] - mm, ss, bb), 0.0f);
                l = fmaxf(__builtin_fmaf(l - mm, ss, bb), 0.0f);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ok ? v[q] : 0.0f;
            l = (halo && row_ok[r] && l_ok) ? l : 0.0f;
            const float from_l = __shfl_up(v[7], 1);
            const float in[9] = {tx == 0 ? l : from_l, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if constexpr (PK) {
#pragma unroll
                    for (int cp = 0; cp < COS / 2; ++cp) {
                        const f32x2_t wv = *reinterpret_cast<const f32x2_t *>(wp + (r * 3 + dx) * CG + 2 * cp);
#pragma unroll
                        for (int p = 0; p < GC_PX; ++p)
                            accp[cp][p] = __builtin_elementwise_fma(wv, (f32x2_t){in[2 * p + dx], in[2 * p + dx]}, accp[cp][p]);
                    }
                } else {
#pragma unroll
                    for (int co = 0; co < COS; ++co) {
                        const float wv = wp[(r * 3 + dx) * CG + co];
#pragma unroll
                        for (int p = 0; p < GC_PX; ++p)
                            acc[co][p] = __builtin_fmaf(wv, in[2 * p + dx], acc[co][p]);
                    }
                }
            }
        }
    };
    Raw ra, rb;
    load_raw(0, ra);
#pragma unroll 1
    for (int ci = 0; ci + 1 < CG; ci += 2) {
        load_raw(ci + 1, rb);
        consume(ra, ci);
        if (ci + 2 < CG) load_raw(ci + 2, ra);
        consume(rb, ci + 1);
    }
    if constexpr (CG % 2 == 1) consume(ra, CG - 1);

    if (oy >= Ho || !col_ok) return;
    float *yout = y + (((size_t)n * C + c0 + part * COS) * Ho + oy) * Wo + ox;
#pragma unroll
    for (int co = 0; co < COS; ++co) {
        float o[GC_PX];
#pragma unroll
        for (int p = 0; p < GC_PX; ++p) o[p] = PK ? accp[PK ? co / 2 : 0][p][co % 2] : acc[PK ? 0 : co][p];
        if (m2) {
            const int c = c0 + part * COS + co;
            const float mm = m2[c], ss = s2[c], bb = b2[c];
#pragma unroll
            for (int p = 0; p < GC_PX; ++p) o[p] = fmaxf(__builtin_fmaf(o[p] - mm, ss, bb), 0.0f);
        }
        *reinterpret_cast<float4 *>(yout + (size_t)co * Ho * Wo) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_gconv3x3_bn_relu(const float *x, const float *w,
                                    const float *mean1, const float *scale1, const float *bias1,
                                    const float *mean2, const float *scale2, const float *bias2,
                                    float *y, int N, int C, int H, int W, int groups, int stride,
                                    void *stream)
{
    if (!x || !w || !y) return fail(-1, "dmn_gconv3x3_bn_relu: NULL pointer");
    if (stride != 1 && stride != 2) return fail(-2, "dmn_gconv3x3_bn_relu: stride must be 1 or 2 (got %d)", stride);
    if ((mean1 && (!scale1 || !bias1)) || (mean2 && (!scale2 || !bias2)))
        return fail(-1, "dmn_gconv3x3_bn_relu: mean given without scale / bias");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups)
        return fail(-4, "dmn_gconv3x3_bn_relu: bad sizes N=%d C=%d H=%d W=%d groups=%d", N, C, H, W, groups);
    if (N > 65535 || groups > 65535) return fail(-5, "dmn_gconv3x3_bn_relu: N or groups exceed the grid limits");
    if ((((uintptr_t)x | (uintptr_t)y) & 15) || ((uintptr_t)w & 7))
        return fail(-7, "dmn_gconv3x3_bn_relu: x and y must be 16-byte, w 8-byte aligned");
    const int cg = C / groups;
    if (stride == 2) {
        if (H % 2 || W % 8) return fail(-4, "dmn_gconv3x3_bn_relu: stride 2 needs H %% 2 == 0 and W %% 8 == 0 (got %dx%d)", H, W);
        const int Ho = H / 2, Wo = W / 2;
        int txo = 1;
        while (txo < 32 && txo * GC_PX < Wo) txo *= 2;
        const int two = txo * GC_PX;
        const long long wgs1 = (long long)((Wo + two - 1) / two) * ((Ho + 256 / txo - 1) / (256 / txo)) * groups * N;
        const bool split2 = cg % 4 == 0 && cg >= 12 && (txo * Ho <= 64 || wgs1 < 2 * NUM_CU);
        const int tho = (split2 ? 64 : 256) / txo;
        dim3 grid3(((Wo + two - 1) / two) * ((Ho + tho - 1) / tho), groups, N);
        hipStream_t s2 = (hipStream_t)stream;
#define DMN_G2(CG, SP) hipLaunchKernelGGL((gconv3x3_s2_kernel<CG, SP>), grid3, dim3(256), 0, s2, x, w, mean1, \
                                          scale1, bias1, mean2, scale2, bias2, y, C, H, W, txo)
#define DMN_G2S(CG) do { if (split2) DMN_G2(CG, 4); else DMN_G2(CG, 1); } while (0)
        switch (cg) {
        case 3: DMN_G2(3, 1); break;
        case 4: DMN_G2(4, 1); break;
        case 6: DMN_G2(6, 1); break;
        case 8: DMN_G2(8, 1); break;
        case 12: DMN_G2S(12); break;
        case 16: DMN_G2S(16); break;
        case 24: DMN_G2S(24); break;
        case 32: DMN_G2S(32); break;
        default:
            return fail(-5, "dmn_gconv3x3_bn_relu: %d channels per group not instantiated "
                            "(3, 4, 6, 8, 12, 16, 24, 32)", cg);
        }
#undef DMN_G2S
#undef DMN_G2
        return launch_status("dmn_gconv3x3_bn_relu");
    }
    // lanes across a tile row: 4 pixels each, power of two, at most 32 lanes (128 pixels: whole
    // rows for every DPN stage, so a tile is one contiguous run of memory per channel)
    int txn = 1;
    while (txn < 32 && txn * GC_PX < W) txn *= 2;
    const int tw = txn * GC_PX, th = 256 / txn;
    const int ch = gc_chunk(cg);
    const size_t lds = ((size_t)cg * 9 * cg + (size_t)ch * (th + 2) * (tw + 2)) * sizeof(float);
    dim3 grid(((W + tw - 1) / tw) * ((H + th - 1) / th), groups, N);
    hipStream_t s = (hipStream_t)stream;
    if (W % 4 == 0) {
        // fast path; output channels split over 4 lane groups when the image fills <= 64 lanes
        // ... or when there are too few whole-group workgroups to fill the chip (small batches)
        const long long wgs = (long long)((W + tw - 1) / tw) * ((H + th - 1) / th) * groups * N;
        const bool split = cg % 4 == 0 && cg >= 12 && (txn * H <= 64 || wgs < 2 * NUM_CU);
        const int lanes = split ? 64 : 256;
        const int th2 = lanes / txn;
        const size_t ldsw = 0;
        dim3 grid2(((W + tw - 1) / tw) * ((H + th2 - 1) / th2), groups, N);
#define DMN_GR(CG, SP) hipLaunchKernelGGL((gconv3x3_rows_kernel<CG, SP>), grid2, dim3(256), ldsw, s, x, w, mean1, \
                                          scale1, bias1, mean2, scale2, bias2, y, C, H, W, txn)
#define DMN_GRS(CG) do { if (split) DMN_GR(CG, 4); else DMN_GR(CG, 1); } while (0)
        switch (cg) {
        case 3: DMN_GR(3, 1); break;
        case 4: DMN_GR(4, 1); break;
        case 6: DMN_GR(6, 1); break;
        case 8: DMN_GR(8, 1); break;
        case 12: DMN_GRS(12); break;
        case 16: DMN_GRS(16); break;
        case 24: DMN_GRS(24); break;
        case 32: DMN_GRS(32); break;
        default:
            return fail(-5, "dmn_gconv3x3_bn_relu: %d channels per group not instantiated "
                            "(3, 4, 6, 8, 12, 16, 24, 32)", cg);
        }
#undef DMN_GRS
#undef DMN_GR
        return launch_status("dmn_gconv3x3_bn_relu");
    }
#define DMN_GC(CG) hipLaunchKernelGGL((gconv3x3_kernel<CG>), grid, dim3(256), lds, s, x, w, mean1, scale1, bias1, \
                                      mean2, scale2, bias2, y, C, H, W, txn)
    switch (cg) {
    case 3: DMN_GC(3); break;
    case 4: DMN_GC(4); break;
    case 6: DMN_GC(6); break;
    case 8: DMN_GC(8); break;
    case 12: DMN_GC(12); break;
    case 16: DMN_GC(16); break;
    case 24: DMN_GC(24); break;
    case 32: DMN_GC(32); break;
    default:
        return fail(-5, "dmn_gconv3x3_bn_relu: %d channels per group not instantiated "
                        "(3, 4, 6, 8, 12, 16, 24, 32)", cg);
    }
#undef DMN_GC
    return launch_status("dmn_gconv3x3_bn_relu");
}
