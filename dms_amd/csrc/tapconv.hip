// dms_amd/csrc/tapconv.hip -- high-res head level as "project at low res, gather at high res" (gfx950).
//
// One level of UpsamplingModule.forward (dmn_pytorch/models/dmn.py:324,337-338) computes
//     y = conv3x3_pad1( Upsample(x2, bilinear)( cat([x, feat], 1) ) ) + bias.
// Bilinear upsampling acts per channel and the channel mixing of a conv tap acts per pixel, so
// the two commute:  W_tap . up(X) = up(W_tap . X).  The nine tap matrices are therefore applied
// at LOW resolution by ONE plain library GEMM
//     Z[b] (9*Cout, H*W) = Wt (9*Cout, Cx+Cf) . cat(x, feat)[b] (Cx+Cf, H*W)
// (2.25 MACs per high-res output and input channel, the same as Winograd F(4x4,3x3), but exact
// fp32 products, no input transform and no 2.25x-sized V tensor), and what is left is the
// HBM-bound gather below: upsample each of the nine Z planes, shift it by its tap offset with
// zero padding at the HIGH-res border, sum, add the bias.  cat() costs nothing either way: inside
// DMN the backbone writes the tap behind room for x and the previous gather writes x there (out_bs),
// so [x | feat] is one operand; a stand-alone caller splits the GEMM over K into an x and a feat part.
//
// dmn_tap_project is the same projection for the case Cout == 1 (last level composed with
// out_layer), where a 9-row GEMM would be a poor fit for the BLAS libraries: a lane owns 4
// pixels and 9 accumulators each, the weights arrive as scalar operands.
#include "common.h"

namespace dmn {

// One lane = CPL adjacent low-res columns of one (b, co) plane; it sweeps `RB` low-res rows and
// keeps the horizontally finished rows r-1, r, r+1 of the three tap rows in registers.
//   hrow[ky][c] = sum_kx upx(Z[ky*3+kx][r])[2*j0 + c + kx - 1]      c = 0 .. 2*CPL-1
template <int CPL>
struct TapRow {
    float v[3][2 * CPL];
};

template <int CPL>
__device__ __forceinline__ void tap_hrow(const float *__restrict__ zplane, long long tapstride,
                                         int r, int W, int j0, TapRow<CPL> &R)
{
    const int jm = j0 > 0 ? j0 - 1 : 0;
    const int jp = j0 + CPL < W ? j0 + CPL : W - 1;
    const float ml = j0 > 0 ? 1.0f : 0.0f;              // high-res column 2*j0-1 exists
    const float mr = j0 + CPL < W ? 1.0f : 0.0f;        // high-res column 2*(j0+CPL) exists
    const long long rowoff = (long long)r * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int c = 0; c < 2 * CPL; ++c) R.v[ky][c] = 0.0f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float *z = zplane + (ky * 3 + kx) * tapstride + rowoff;
            float zl[CPL + 2];
            zl[0] = z[jm];
            if (CPL == 2) {
                const float2 m = *reinterpret_cast<const float2 *>(z + j0);
                zl[1] = m.x; zl[2] = m.y;
            } else {
                zl[1] = z[j0];
            }
            zl[CPL + 1] = z[jp];
            // u[q] = upx(z)[2*j0 - 1 + q], q = 0 .. 2*CPL+1
            float u[2 * CPL + 2];
            u[0] = ml * (0.75f * zl[0] + 0.25f * zl[1]);
#pragma unroll
            for (int m = 0; m < CPL; ++m) {
                u[1 + 2 * m] = 0.25f * zl[m] + 0.75f * zl[m + 1];
                u[2 + 2 * m] = 0.75f * zl[m + 1] + 0.25f * zl[m + 2];
            }
            u[2 * CPL + 1] = mr * (0.25f * zl[CPL] + 0.75f * zl[CPL + 1]);
#pragma unroll
            for (int c = 0; c < 2 * CPL; ++c) R.v[ky][c] += u[c + kx];
        }
    }
}

template <int CPL>
__global__ void __launch_bounds__(256)
tap_gather_kernel(const float *__restrict__ Z, const float *__restrict__ bias, float *__restrict__ out,
                  int B, int Cout, int H, int W, int RB, long long out_bs)
{
    const int Wl = (W + CPL - 1) / CPL;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Cout * Wl;
    if (g >= total) return;
    const int plane = (int)(g / Wl), jl = (int)(g - (long long)plane * Wl);
    const int b = plane / Cout, co = plane - b * Cout;
    const int j0 = jl * CPL;
    const int r0 = blockIdx.y * RB;
    const int r1 = r0 + RB < H ? r0 + RB : H;
    const long long HW = (long long)H * W;
    const long long tapstride = (long long)Cout * HW;
    const float *zplane = Z + ((long long)b * 9 * Cout + co) * HW;
    const float bv = bias ? bias[co] : 0.0f;
    float *oplane = out + (long long)b * out_bs + (long long)co * 4 * HW;
    const int W2 = 2 * W;

    TapRow<CPL> Rm, Rc, Rp;
    tap_hrow<CPL>(zplane, tapstride, r0 > 0 ? r0 - 1 : 0, W, j0, Rm);
    tap_hrow<CPL>(zplane, tapstride, r0, W, j0, Rc);
    for (int i = r0; i < r1; ++i) {
        tap_hrow<CPL>(zplane, tapstride, i + 1 < H ? i + 1 : H - 1, W, j0, Rp);
        const float mt = i > 0 ? 1.0f : 0.0f;          // high-res row 2i-1 exists
        const float mb = i + 1 < H ? 1.0f : 0.0f;      // high-res row 2i+2 exists
        float o0[2 * CPL], o1[2 * CPL];
#pragma unroll
        for (int c = 0; c < 2 * CPL; ++c) {
            o0[c] = bv + mt * (0.75f * Rm.v[0][c] + 0.25f * Rc.v[0][c])
                       + (0.25f * Rm.v[1][c] + 0.75f * Rc.v[1][c])
                       + (0.75f * Rc.v[2][c] + 0.25f * Rp.v[2][c]);
            o1[c] = bv + (0.25f * Rm.v[0][c] + 0.75f * Rc.v[0][c])
                       + (0.75f * Rc.v[1][c] + 0.25f * Rp.v[1][c])
                       + mb * (0.25f * Rc.v[2][c] + 0.75f * Rp.v[2][c]);
        }
        float *p0 = oplane + (long long)(2 * i) * W2 + 2 * j0;
        float *p1 = p0 + W2;
        if (CPL == 2) {
            *reinterpret_cast<float4 *>(p0) = make_float4(o0[0], o0[1], o0[2 * CPL - 2], o0[2 * CPL - 1]);
            *reinterpret_cast<float4 *>(p1) = make_float4(o1[0], o1[1], o1[2 * CPL - 2], o1[2 * CPL - 1]);
        } else {
            *reinterpret_cast<float2 *>(p0) = make_float2(o0[0], o0[1]);
            *reinterpret_cast<float2 *>(p1) = make_float2(o1[0], o1[1]);
        }
        Rm = Rc;
        Rc = Rp;
    }
}

// ------------------------------------------------------------------------------------------------
// Tiled variant for power-of-two widths W = 16 .. 256 (all five head levels at 512 px), round 2.
// The sweep kernel above spends 71 % of its wave cycles in s_waitcnt (rocprofv3 SQ_WAIT_ANY) with the
// VALU 5 % busy: every lane walks its rows one memory round trip at a time, three loads per tap (8-byte
// + two neighbour dwords), 94 VGPRs.  Here a 256-thread workgroup takes a BAND of 1024 consecutive
// floats of every tap plane -- NR = 1024 / W low-res rows; for fixed (b, tap) the (co, row) planes are
// one contiguous slab, so a band is 4 KiB of contiguous memory per tap, whole planes when H <= NR --
// with ONE 16-byte load per lane and tap (nine independent loads in flight per lane), does the
// horizontal pass in registers (the two neighbour columns come from the adjacent lanes), hands the
// three horizontally finished rows to the lanes of the rows above and below through LDS, and writes
// the 2 x 8 outputs of its four columns.  When a plane is taller than a band, lanes 0 .. W/2-1 also
// fetch the band's two halo rows; consecutive bands are mapped to the SAME XCD (workgroups are dealt to
// the 8 XCDs round-robin), so that those re-reads are L2 hits.
template <int W, int LM>
__global__ void __launch_bounds__(256)
tap_gather_tile_kernel(const float *__restrict__ Z, const float *__restrict__ bias, float *__restrict__ out,
                       int Cout, int H, long long out_bs, int nbands, int total, int xcd_chunk)
{
    constexpr int NR = 1024 / W, LPR = W / 4, W2 = 2 * W;
    __shared__ float hl[3][NR + 2][W2];                       // horizontally finished rows, band + 2 halo rows
    const int t = threadIdx.x;
    const int vb = (int)(blockIdx.x % 8) * xcd_chunk + (int)(blockIdx.x / 8);
    if (vb >= total) return;                                  // whole workgroup: uniform
    const int b = vb / nbands, band = vb - b * nbands;
    const long long slabrows = (long long)Cout * H;
    const long long tapstride = slabrows * W;
    const float *zb = Z + (long long)b * 9 * tapstride;
    const int R0 = band * NR;

    // a lane's nine 16-byte tap loads of one low-res row segment
    auto fetch = [&](long long R, int c0, f32x4_t (&z)[9], bool keep) {
        const float *zr = zb + R * W + c0;
        if (LM == 1) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                z[tap] = *reinterpret_cast<const f32x4_t *>(zr + tap * tapstride);
        } else if (LM == 2 && keep) {
            // rows that a neighbouring band reads again as halo stay in L2 (plain loads); the empty asm keeps
            // hipcc from merging this branch with the streamed one below (the merge drops the nt hint)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                z[tap] = *reinterpret_cast<const f32x4_t *>(zr + tap * tapstride);
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                z[tap] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(zr + tap * tapstride));
        }
    };
    // horizontal pass: h[ky][c] = sum_kx upx(z[ky*3+kx])[2*c0 + c + kx - 1]
    auto hpass = [&](const f32x4_t (&z)[9], int c0, float (&h)[3][8]) {
        const bool el = c0 == 0, er = c0 + 4 == W;           // row edges: clamp (bilinear) and zero (conv pad)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int c = 0; c < 8; ++c) h[ky][c] = 0.0f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4_t v = z[ky * 3 + kx];
                const float lft = __shfl_up(v.w, 1), rgt = __shfl_down(v.x, 1);
                float zl[6] = {el ? v.x : lft, v.x, v.y, v.z, v.w, er ? v.w : rgt};
                float u[10];
                u[0] = el ? 0.0f : (0.75f * zl[0] + 0.25f * zl[1]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    u[1 + 2 * m] = 0.25f * zl[m] + 0.75f * zl[m + 1];
                    u[2 + 2 * m] = 0.75f * zl[m + 1] + 0.25f * zl[m + 2];
                }
                u[9] = er ? 0.0f : (0.25f * zl[4] + 0.75f * zl[5]);
#pragma unroll
                for (int c = 0; c < 8; ++c) h[ky][c] += u[c + kx];
            }
        }
    };
    auto put = [&](int row, int c0, const float (&h)[3][8]) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float4 *dst = reinterpret_cast<float4 *>(&hl[ky][row][2 * c0]);
            dst[0] = make_float4(h[ky][0], h[ky][1], h[ky][2], h[ky][3]);
            dst[1] = make_float4(h[ky][4], h[ky][5], h[ky][6], h[ky][7]);
        }
    };

    const int rr = t / LPR, c0 = (t % LPR) * 4;
    const long long R = (long long)R0 + rr;
    const int co = (int)(R / H), r = (int)(R - (long long)co * H);
    // halo rows (same plane only) are fetched by lanes 0 .. 2*LPR-1 TOGETHER with their own row: both sets of
    // nine loads are in flight before either is consumed (one memory round trip for those waves, not two)
    bool halo = false;
    int bottom = 0;
    f32x4_t zh[9];
    if (H > NR && t < 2 * LPR) {
        bottom = t / LPR;
        const int rfirst = (int)((long long)R0 % H);          // row of the band's first line inside its plane
        halo = bottom ? (rfirst + NR < H) : (rfirst > 0);
        if (halo) fetch(bottom ? (long long)R0 + NR : (long long)R0 - 1, c0, zh, true);
    }
    f32x4_t zc[9];
    fetch(R, c0, zc, H > NR && (rr == 0 || rr == NR - 1));
    float Rc[3][8];
    hpass(zc, c0, Rc);
    put(rr + 1, c0, Rc);
    if (halo) {
        float Rh3[3][8];
        hpass(zh, c0, Rh3);
        put(bottom ? NR + 1 : 0, c0, Rh3);
    }
    __syncthreads();

    // vertical pass; rows outside the plane: clamp for the bilinear terms, zero for the conv padding
    const bool top = r == 0, bot = r + 1 == H;
    const float bv = bias ? bias[co] : 0.0f;
    float o0[8], o1[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float4 m0 = *reinterpret_cast<const float4 *>(&hl[0][rr][2 * c0 + 4 * half]);
        const float4 m1 = *reinterpret_cast<const float4 *>(&hl[1][rr][2 * c0 + 4 * half]);
        const float4 p1 = *reinterpret_cast<const float4 *>(&hl[1][rr + 2][2 * c0 + 4 * half]);
        const float4 p2 = *reinterpret_cast<const float4 *>(&hl[2][rr + 2][2 * c0 + 4 * half]);
        const float am0[4] = {m0.x, m0.y, m0.z, m0.w}, am1[4] = {m1.x, m1.y, m1.z, m1.w};
        const float ap1[4] = {p1.x, p1.y, p1.z, p1.w}, ap2[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * half + e;
            const float Rm0 = top ? Rc[0][c] : am0[e], Rm1 = top ? Rc[1][c] : am1[e];
            const float Rp1 = bot ? Rc[1][c] : ap1[e], Rp2 = bot ? Rc[2][c] : ap2[e];
            o0[c] = bv + (top ? 0.0f : (0.75f * Rm0 + 0.25f * Rc[0][c]))
                       + (0.25f * Rm1 + 0.75f * Rc[1][c])
                       + (0.75f * Rc[2][c] + 0.25f * Rp2);
            o1[c] = bv + (0.25f * Rm0 + 0.75f * Rc[0][c])
                       + (0.75f * Rc[1][c] + 0.25f * Rp1)
                       + (bot ? 0.0f : (0.25f * Rc[2][c] + 0.75f * Rp2));
        }
    }
    float *p0 = out + (long long)b * out_bs + ((long long)co * 2 * H + 2 * r) * W2 + 2 * c0;   // plane = (2H) x (2W)
    float *p1 = p0 + W2;
    // plain (write-back) stores: a lane owns 32 contiguous bytes of each output row and writes them as two
    // 16-byte pieces, so one store instruction covers only half of every 32-byte sector.  As streamed
    // (non-temporal) stores those half-written lines left L2 in pieces: WRITE_SIZE +13 .. +17 % over the
    // output bytes and 5.2-5.4 TB/s; left to merge in L2 the traffic is exact (+0.0 %) and the three
    // smaller levels run at 5.9-6.1 TB/s (tools/tap_store_modes.sh, profiles/r2_tap_store_modes.txt)
    *reinterpret_cast<float4 *>(p0) = make_float4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<float4 *>(p0 + 4) = make_float4(o0[4], o0[5], o0[6], o0[7]);
    *reinterpret_cast<float4 *>(p1) = make_float4(o1[0], o1[1], o1[2], o1[3]);
    *reinterpret_cast<float4 *>(p1 + 4) = make_float4(o1[4], o1[5], o1[6], o1[7]);
}

// Z[b, r, p] = sum_c w[r, c] * in[b, c, p], r = 0..8, in = cat(x, feat) over channels (never built).
// VEC pixels per lane; the weight reads are uniform (scalar loads).
template <int VEC>
__global__ void __launch_bounds__(256)
tap_project_kernel(const float *__restrict__ x, const float *__restrict__ feat,
                   const float *__restrict__ w, float *__restrict__ Z, int Cx, int Cf, int HW)
{
    const int b = blockIdx.y;
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (p >= HW) return;
    const int C = Cx + Cf;
    float acc[9][VEC];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[r][e] = 0.0f;
    for (int half = 0; half < 2; ++half) {
        const int n = half ? Cf : Cx, c0 = half ? Cx : 0;
        const float *src = (half ? feat : x) + (long long)b * n * HW + p;
#pragma unroll 4
        for (int c = 0; c < n; ++c) {
            float v[VEC];
            if (VEC == 4) {
                // read once: non-temporal, so the stream does not displace the 9 output rows
                const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + (long long)c * HW));
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[VEC - 1] = t.w;
            } else {
                v[0] = src[(long long)c * HW];
            }
            if (VEC == 4) {
                // pixel pairs on v_pk_fma_f32 (the weight is one scalar, broadcast to both halves): 18 instead of 36
                // VALU instructions per 16 bytes loaded; the same fmaf per element
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const float wv = w[r * C + c0 + c];
                    const f32x2_t w2 = {wv, wv};
                    const f32x2_t a01 = __builtin_elementwise_fma(w2, (f32x2_t){v[0], v[1]}, (f32x2_t){acc[r][0], acc[r][1]});
                    const f32x2_t a23 = __builtin_elementwise_fma(w2, (f32x2_t){v[2], v[VEC - 1]}, (f32x2_t){acc[r][2], acc[r][VEC - 1]});
                    acc[r][0] = a01.x; acc[r][1] = a01.y; acc[r][2] = a23.x; acc[r][VEC - 1] = a23.y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const float wv = w[r * C + c0 + c];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[r][e] = __builtin_fmaf(wv, v[e], acc[r][e]);
                }
            }
        }
    }
    float *zp = Z + (long long)b * 9 * HW + p;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (VEC == 4) *reinterpret_cast<float4 *>(zp + (long long)r * HW) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][VEC - 1]);
        else zp[(long long)r * HW] = acc[r][0];
    }
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_tap_upsample_gather(const float *Z, const float *bias, float *out,
                                       int B, int Cout, int H, int W, long long out_bs, void *stream)
{
    if (out_bs < (long long)Cout * 4 * H * W || out_bs % 4)
        return fail(-4, "dmn_tap_upsample_gather: out batch stride %lld must be >= Cout*4*H*W and a multiple of 4", out_bs);
    if (!Z || !out) return fail(-1, "dmn_tap_upsample_gather: NULL pointer");
    if (B <= 0 || Cout <= 0 || H <= 0 || W <= 0)
        return fail(-4, "dmn_tap_upsample_gather: bad sizes B=%d Cout=%d H=%d W=%d", B, Cout, H, W);
    if ((long long)B * Cout >= (1LL << 31) / 4)
        return fail(-5, "dmn_tap_upsample_gather: B*Cout=%lld too large", (long long)B * Cout);
    hipStream_t s = (hipStream_t)stream;
    // tiled kernel: power-of-two widths 16 .. 256, bands of 1024 / W rows that tile the (co, row) slab
    // and either hold whole planes or divide one (DMN_TAP_TILE=0: the sweep kernel, for A/B runs)
    {
        const char *te = getenv("DMN_TAP_TILE");
        const bool wok = W == 16 || W == 32 || W == 64 || W == 128 || W == 256;
        const int NR = wok ? 1024 / W : 1;
        if (wok && !(te && atoi(te) == 0) && ((long long)Cout * H) % NR == 0 && (H % NR == 0 || NR % H == 0)
            && !(((uintptr_t)Z | (uintptr_t)out) & 15) && (long long)B * Cout * H / NR < (1LL << 30)) {
            const int nbands = (int)((long long)Cout * H / NR);
            const int total = B * nbands;
            const int xcd_chunk = (total + 7) / 8;
            dim3 tgrid((unsigned)xcd_chunk * 8);
            // load policy (tools/tap_store_modes.sh, profiles/r2_tap_store_modes.txt): streamed loads, except the
            // rows a neighbouring band reads again as halo (2); W = 16 (whole planes per band, 36 KiB per tap
            // set) runs 13 % faster with plain loads (1).  DMN_TAP_LOAD=0|1|2 forces one for A/B runs.
            const char *se = getenv("DMN_TAP_LOAD");
            const int sm = se ? atoi(se) : (W == 16 ? 1 : 2);
#define DMN_TILE2(WV, SMV) hipLaunchKernelGGL((tap_gather_tile_kernel<WV, SMV>), tgrid, dim3(256), 0, s, Z, bias, out, Cout, H, out_bs, nbands, total, xcd_chunk)
#define DMN_TILE(WV) do { if (sm == 0) DMN_TILE2(WV, 0); else if (sm == 1) DMN_TILE2(WV, 1); else DMN_TILE2(WV, 2); } while (0)
            if (W == 16) DMN_TILE(16);
            else if (W == 32) DMN_TILE(32);
            else if (W == 64) DMN_TILE(64);
            else if (W == 128) DMN_TILE(128);
            else DMN_TILE(256);
#undef DMN_TILE2
#undef DMN_TILE
            return launch_status("dmn_tap_upsample_gather");
        }
    }
    const bool two = (W % 2 == 0) && !(((uintptr_t)Z & 7) | ((uintptr_t)out & 15));
    if (!two && ((uintptr_t)out & 7))
        return fail(-5, "dmn_tap_upsample_gather: out must be 8-byte aligned");
    const int Wl = two ? W / 2 : W;
    const long long total = (long long)B * Cout * Wl;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(-5, "dmn_tap_upsample_gather: grid too large");
    // rows per lane: as long as possible (2 halo rows are recomputed per chunk) while the launch
    // still has ~8 wavefronts per SIMD
    int RB = H;
    while (RB > 16 && blocks * 4 * ((H + RB - 1) / RB) < 8192) RB = (RB + 1) / 2;   // (2048 .. 32768: no difference)
    const int chunks = (H + RB - 1) / RB;
    if (chunks > 65535) return fail(-5, "dmn_tap_upsample_gather: H=%d too large", H);
    dim3 grid((unsigned)blocks, chunks);
    if (two) hipLaunchKernelGGL(tap_gather_kernel<2>, grid, dim3(256), 0, s, Z, bias, out, B, Cout, H, W, RB, out_bs);
    else hipLaunchKernelGGL(tap_gather_kernel<1>, grid, dim3(256), 0, s, Z, bias, out, B, Cout, H, W, RB, out_bs);
    return launch_status("dmn_tap_upsample_gather");
}

extern "C" int dmn_tap_project(const float *x, const float *feat, const float *w, float *Z,
                               int B, int Cx, int Cf, int HW, void *stream)
{
    if (!w || !Z || (Cx > 0 && !x) || (Cf > 0 && !feat)) return fail(-1, "dmn_tap_project: NULL pointer");
    if (B <= 0 || Cx < 0 || Cf < 0 || Cx + Cf <= 0 || HW <= 0)
        return fail(-4, "dmn_tap_project: bad sizes B=%d Cx=%d Cf=%d HW=%d", B, Cx, Cf, HW);
    if (B > 65535) return fail(-5, "dmn_tap_project: B=%d exceeds the grid y limit", B);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (HW % 4 == 0) && !(((uintptr_t)x | (uintptr_t)feat | (uintptr_t)Z) & 15);
    if (vec) {
        dim3 grid((HW / 4 + 255) / 256, B);
        hipLaunchKernelGGL(tap_project_kernel<4>, grid, dim3(256), 0, s, x, feat, w, Z, Cx, Cf, HW);
    } else {
        dim3 grid((HW + 255) / 256, B);
        hipLaunchKernelGGL(tap_project_kernel<1>, grid, dim3(256), 0, s, x, feat, w, Z, Cx, Cf, HW);
    }
    return launch_status("dmn_tap_project");
}
