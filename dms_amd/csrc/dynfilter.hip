// dms_amd/csrc/dynfilter.hip -- dynamic-filter application as ONE batched thin GEMM (gfx950).
//
// Replaces the reference's python loop of L separate `F.conv2d(vis, filters[t])` calls and
// the torch.cat of their results (dmn_pytorch/models/dmn.py:178-184).  Per sample b:
//     P_b (L*K x S) = F_b (L*K x C) . V_b (C x S),   L*K = 200, C = 2696, S = 256 @512px
// 276 MFLOP and 5.1 MB per sample -> arithmetic intensity ~54 flop/B: fp32-FMA bound, tiny
// next to the projection GEMMs.  Plain FMA + LDS tiling (north_star: "kept on plain FMA/LDS
// unless rocprof shows MFMA pays off"): 64x64 output tile per 256-thread workgroup, 4x4
// register tile per lane, K-step 16, next tile prefetched into registers during the FMAs.
// Rows r = (t, k) of F / P are addressed in place in the reference's (L, B, K, .) layout,
// so no permuted copy of the filters is ever made.
#include "common.h"

namespace dmn {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;

// element offset of row r = t*K + k of an (L, B, K, ld) tensor, batch entry b
__device__ __forceinline__ long long lbk_row(int r, int b, int B, int K, int ld)
{
    const int t = r / K, k = r - t * K;
    return ((long long)(t * B + b) * K + k) * ld;
}

// ---- forward, vector path: C % 4 == 0, S % 4 == 0, 16-byte aligned bases -------------
__global__ void __launch_bounds__(256)
dynfilter_fwd_vec(const float *__restrict__ V, const float *__restrict__ F, float *__restrict__ P,
                  int B, int R, int K, int C, int S)
{
    __shared__ float As[BK][BM + PAD];   // As[kk][row]  (transposed on store)
    __shared__ float Bs[BK][BN + PAD];   // Bs[kk][col]
    const int b = blockIdx.z, r0 = blockIdx.y * BM, s0 = blockIdx.x * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float *Vb = V + (long long)b * C * S;

    // this lane's slice of the two tiles
    const int ar = tid >> 2, ac = (tid & 3) * 4;     // A: row ar, k-offset ac..ac+3
    const int bk = tid >> 4, bc = (tid & 15) * 4;    // B: k-row bk, col bc..bc+3
    const bool a_ok = (r0 + ar) < R;
    const float *Ap = F + (a_ok ? lbk_row(r0 + ar, b, B, K, C) : 0);
    const bool b_ok = (s0 + bc) < S;                 // S % 4 == 0 -> whole float4 in or out

    // column pairs ride v_pk_fma_f32 (the filter value is broadcast by op_sel): twice the FMA rate
    f32x2_t acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i][0] = acc2[i][1] = (f32x2_t){0.0f, 0.0f};
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    auto fetch = [&](int c0) {
        av = (a_ok && c0 + ac < C) ? *reinterpret_cast<const float4 *>(Ap + c0 + ac) : make_float4(0, 0, 0, 0);
        bv = (b_ok && c0 + bk < C) ? *reinterpret_cast<const float4 *>(Vb + (long long)(c0 + bk) * S + s0 + bc)
                                   : make_float4(0, 0, 0, 0);
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += BK) {
        As[ac + 0][ar] = av.x; As[ac + 1][ar] = av.y; As[ac + 2][ar] = av.z; As[ac + 3][ar] = av.w;
        *reinterpret_cast<float4 *>(&Bs[bk][bc]) = bv;
        __syncthreads();
        if (c0 + BK < C) fetch(c0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[kk][ty * 4]);
            const float4 w = *reinterpret_cast<const float4 *>(&Bs[kk][tx * 4]);
            const float aa[4] = {a.x, a.y, a.z, a.w};
            const f32x2_t w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2_t ai = {aa[i], aa[i]};
                acc2[i][0] = __builtin_elementwise_fma(ai, w01, acc2[i][0]);
                acc2[i][1] = __builtin_elementwise_fma(ai, w23, acc2[i][1]);
            }
        }
        __syncthreads();
    }
    const int sc = s0 + tx * 4;
    if (sc < S) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty * 4 + i;
            if (r < R)
                *reinterpret_cast<float4 *>(P + lbk_row(r, b, B, K, S) + sc) =
                    make_float4(acc2[i][0].x, acc2[i][0].y, acc2[i][1].x, acc2[i][1].y);
        }
    }
}

// ---- generic tiled GEMM with element-wise guarded loads -------------------------------
// MODE 0: P[r,s]  = sum_c F[r,c]  V[c,s]      (forward, any C / S)
// MODE 1: gF[r,c] = sum_s gP[r,s] V[c,s]      (m = r, n = c, k = s)
// MODE 2: gV[c,s] = sum_r F[r,c]  gP[r,s]     (m = c, n = s, k = r)
template <int MODE>
__global__ void __launch_bounds__(256)
dynfilter_generic(const float *__restrict__ V, const float *__restrict__ X /*F or gP*/,
                  const float *__restrict__ Y /*gP for MODE 2*/, float *__restrict__ O,
                  int B, int R, int K, int C, int S)
{
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];
    const int b = blockIdx.z, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float *Vb = V + (long long)b * C * S;
    const int Md = (MODE == 2) ? C : R;
    const int Nd = (MODE == 0) ? S : (MODE == 1 ? C : S);
    const int Kd = (MODE == 0) ? C : (MODE == 1 ? S : R);

    auto a_at = [&](int m, int k) -> float {
        if (m >= Md || k >= Kd) return 0.0f;
        if (MODE == 0) return X[lbk_row(m, b, B, K, C) + k];
        if (MODE == 1) return X[lbk_row(m, b, B, K, S) + k];
        return X[lbk_row(k, b, B, K, C) + m];
    };
    auto b_at = [&](int k, int n) -> float {
        if (k >= Kd || n >= Nd) return 0.0f;
        if (MODE == 0) return Vb[(long long)k * S + n];
        if (MODE == 1) return Vb[(long long)n * S + k];
        return Y[lbk_row(k, b, B, K, S) + n];
    };

    float acc[4][4] = {};
    for (int k0 = 0; k0 < Kd; k0 += BK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            if (MODE == 2) {                 // A is m-contiguous
                const int kk = idx >> 6, mm = idx & 63;
                As[kk][mm] = a_at(m0 + mm, k0 + kk);
            } else {                         // A is k-contiguous
                const int mm = idx >> 4, kk = idx & 15;
                As[kk][mm] = a_at(m0 + mm, k0 + kk);
            }
            if (MODE == 1) {                 // B is k-contiguous
                const int nn = idx >> 4, kk = idx & 15;
                Bs[kk][nn] = b_at(k0 + kk, n0 + nn);
            } else {                         // B is n-contiguous
                const int kk = idx >> 6, nn = idx & 63;
                Bs[kk][nn] = b_at(k0 + kk, n0 + nn);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[kk][ty * 4]);
            const float4 w = *reinterpret_cast<const float4 *>(&Bs[kk][tx * 4]);
            const float aa[4] = {a.x, a.y, a.z, a.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(aa[i], ww[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= Md) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= Nd) continue;
            if (MODE == 0) O[lbk_row(m, b, B, K, S) + n] = acc[i][j];
            else if (MODE == 1) O[lbk_row(m, b, B, K, C) + n] = acc[i][j];
            else O[((long long)b * C + m) * S + n] = acc[i][j];
        }
    }
}

static int check_sizes(const char *fn, int B, int L, int K, int C, int S)
{
    if (B <= 0 || L <= 0 || K <= 0 || C <= 0 || S <= 0)
        return fail(-4, "%s: bad sizes B=%d L=%d K=%d C=%d S=%d", fn, B, L, K, C, S);
    if (B > 65535) return fail(-5, "%s: B=%d exceeds the grid z limit", fn, B);
    return 0;
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_dynfilter_fwd(const float *V, const float *F, float *P, int B, int L, int K,
                                 int C, int S, void *stream)
{
    if (!V || !F || !P) return fail(-1, "dmn_dynfilter_fwd: NULL pointer");
    if (int rc = check_sizes("dmn_dynfilter_fwd", B, L, K, C, S)) return rc;
    const int R = L * K;
    dim3 grid((S + BN - 1) / BN, (R + BM - 1) / BM, B);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && (S % 4 == 0) &&
                     !(((uintptr_t)V | (uintptr_t)F | (uintptr_t)P) & 15);
    if (vec) hipLaunchKernelGGL(dynfilter_fwd_vec, grid, dim3(256), 0, s, V, F, P, B, R, K, C, S);
    else hipLaunchKernelGGL(dynfilter_generic<0>, grid, dim3(256), 0, s, V, F, (const float *)nullptr, P, B, R, K, C, S);
    return launch_status("dmn_dynfilter_fwd");
}

extern "C" int dmn_dynfilter_bwd(const float *V, const float *F, const float *gP, float *gV,
                                 float *gF, int B, int L, int K, int C, int S, void *stream)
{
    if (!V || !F || !gP) return fail(-1, "dmn_dynfilter_bwd: NULL pointer");
    if (int rc = check_sizes("dmn_dynfilter_bwd", B, L, K, C, S)) return rc;
    const int R = L * K;
    hipStream_t s = (hipStream_t)stream;
    if (gF) {
        dim3 grid((C + BN - 1) / BN, (R + BM - 1) / BM, B);
        hipLaunchKernelGGL(dynfilter_generic<1>, grid, dim3(256), 0, s, V, gP, (const float *)nullptr, gF, B, R, K, C, S);
    }
    if (gV) {
        dim3 grid((S + BN - 1) / BN, (C + BM - 1) / BM, B);
        hipLaunchKernelGGL(dynfilter_generic<2>, grid, dim3(256), 0, s, V, F, gP, gV, B, R, K, C, S);
    }
    return launch_status("dmn_dynfilter_bwd");
}
