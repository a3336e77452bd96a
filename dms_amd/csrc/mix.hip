// dms_amd/csrc/mix.hip -- concat-free 1x1 mix + reorder into the mSRU sequence (gfx950).
//
// Replaces dmn_pytorch/models/dmn.py:187-201: the (L, 4706, H, W) concat (96 MB/sample at
// 512 px), the 4706->1000 comb_conv over it (24 GMAC) and the permute/contiguous into
// sequence order n = s*L + t.  comb_conv is linear, so with its weight split by input
// block  [vis+coords | WE | h | P]  the caller computes two small GEMMs (A: once per pixel,
// Bt: once per word) and this kernel does the rank-K update and the broadcast add while
// writing Q directly in (S*L, B, M) order:
//     Q[(s*L + t), b, m] = A[b,s,m] + Bt[t,b,m] + sum_k Wp[m,k] * P[t,b,k,s]
// Pure HBM write stream (Q = 20.5 MB/sample); A/Bt/P/Wp are re-read from L2.  One lane
// owns 4 adjacent m (one 16-byte store per (s,t)); a wave writes 1 KiB contiguous.
#include "common.h"
#include "sru_ring.h"

namespace dmn {

constexpr int MIX_PT = 8;      // pixels per workgroup
constexpr int MIX_KMAX = 16;   // filters kept in registers

__device__ __forceinline__ long long lbk_off(int t, int b, int k, int B, int K, int ld)
{
    return ((long long)(t * B + b) * K + k) * ld;
}

// KR > 0: the K <= KR filter weights of a lane stay in registers; PT pixels per workgroup
template <int VEC, int KR, int PT>
__global__ void __launch_bounds__(256)
mix_fwd_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
               const float *__restrict__ P, const float *__restrict__ Wp, int ldw,
               float *__restrict__ Q, int B, int L, int K, int S, int M)
{
    extern __shared__ float Ps[];            // [L*K][PT]
    // launch order = (piece, b) fastest, pixel tile slowest: workgroups that run together
    // write the same contiguous (s,t) slabs of Q (B*M*4 bytes each) -> DRAM page locality
    const int b = blockIdx.y, s0 = blockIdx.z * PT;
    const int LK = L * K;
    for (int idx = threadIdx.x; idx < LK * PT; idx += blockDim.x) {
        const int r = idx / PT, pp = idx - r * PT;
        const int t = r / K, k = r - t * K;
        const int s = s0 + pp;
        Ps[idx] = s < S ? P[lbk_off(t, b, k, B, K, S) + s] : 0.0f;
    }
    __syncthreads();

    const int Mv = M / VEC;
    // blockIdx.x walks the row in pieces of 256 lanes (M = 1000: one piece; the folded
    // layer-0 projection has M = 4020: four), so the grid grows instead of a per-lane loop
    for (int mv = blockIdx.x * blockDim.x + threadIdx.x; mv < Mv; mv += gridDim.x * blockDim.x) {
        const int m = mv * VEC;
        constexpr bool KREG = KR > 0;
        float w[VEC][KREG ? KR : 1];
        if (KREG) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int k = 0; k < KR; ++k)
                    w[e][k] = k < K ? Wp[(long long)(m + e) * ldw + k] : 0.0f;
        }
        // A rows of the tile's pixels stay in registers; Bt[t] is loaded once per word and
        // serves all PT pixels (PT independent 16-byte stores per load)
        float a[PT][VEC];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) {
            const int s = s0 + pp < S ? s0 + pp : S - 1;
            const float *Ap = A + ((long long)b * S + s) * M + m;
            if (VEC == 4) { const float4 v = *reinterpret_cast<const float4 *>(Ap); a[pp][0] = v.x; a[pp][1] = v.y; a[pp][2] = v.z; a[pp][3] = v.w; }
            else a[pp][0] = Ap[0];
        }
#pragma unroll 2
        for (int t = 0; t < L; ++t) {
            float bt[VEC];
            const float *Bp = Bt + ((long long)t * B + b) * M + m;
            if (VEC == 4) { const float4 v = *reinterpret_cast<const float4 *>(Bp); bt[0] = v.x; bt[1] = v.y; bt[2] = v.z; bt[3] = v.w; }
            else bt[0] = Bp[0];
            if (KREG) {
                // all PT pixels of a word advance together: one filter's PT responses are adjacent in LDS
                // (16-byte broadcast reads instead of PT 4-byte ones); per output the k order is unchanged
                float q[PT][VEC];
#pragma unroll
                for (int pp = 0; pp < PT; ++pp)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) q[pp][e] = a[pp][e] + bt[e];
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    if (k < K) {
                        float pv[PT];
                        const float *pr = Ps + (t * K + k) * PT;
                        if (PT % 4 == 0) {
#pragma unroll
                            for (int p4 = 0; p4 < PT / 4; ++p4) {
                                const float4 v = *reinterpret_cast<const float4 *>(pr + 4 * p4);
                                pv[4 * p4] = v.x; pv[4 * p4 + 1] = v.y; pv[4 * p4 + 2] = v.z; pv[4 * p4 + 3] = v.w;
                            }
                        } else {
#pragma unroll
                            for (int pp = 0; pp < PT; ++pp) pv[pp] = pr[pp];
                        }
#pragma unroll
                        for (int pp = 0; pp < PT; ++pp)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) q[pp][e] = __builtin_fmaf(w[e][k], pv[pp], q[pp][e]);
                    }
                }
#pragma unroll
                for (int pp = 0; pp < PT; ++pp) {
                    const int s = s0 + pp;
                    if (s < S) {
                        float *Qp = Q + (((long long)s * L + t) * B + b) * M + m;
                        if (VEC == 4) *reinterpret_cast<float4 *>(Qp) = make_float4(q[pp][0], q[pp][1], q[pp][2], q[pp][VEC - 1]);
                        else Qp[0] = q[pp][0];
                    }
                }
            } else {
#pragma unroll
                for (int pp = 0; pp < PT; ++pp) {
                    const int s = s0 + pp;
                    float q[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) q[e] = a[pp][e] + bt[e];
                    for (int k = 0; k < K; ++k) {
                        const float pv = Ps[(t * K + k) * PT + pp];
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            q[e] = __builtin_fmaf(Wp[(long long)(m + e) * ldw + k], pv, q[e]);
                    }
                    if (s < S) {
                        float *Qp = Q + (((long long)s * L + t) * B + b) * M + m;
                        if (VEC == 4) *reinterpret_cast<float4 *>(Qp) = make_float4(q[0], q[1], q[2], q[VEC - 1]);
                        else Qp[0] = q[0];
                    }
                }
            }
        }
    }
}

// The K <= KR, 16-byte case again (the model's shapes), two changes, same arithmetic per element:
//  * column windows start on 64-byte lines: sample b's rows begin (b*M mod 16) floats into a line when
//    B*M is a multiple of 16 (the folded width 4020: 0, 4, 8 or 12 floats), so lanes are shifted by that
//    many quarters (`shift4`) and every wave's 1 KiB store covers whole lines -- a 4020-float row written
//    as 16-byte pieces from an unaligned start left two half-written lines per wave and store;
//  * the rank-K update runs on v_pk_fma_f32 (column pairs), halving the VALU issue per stored byte.
template <int KR, int PT, bool NT>
__global__ void __launch_bounds__(256)
mix_fwd_pk_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                  const float *__restrict__ P, const float *__restrict__ Wp, int ldw,
                  float *__restrict__ Q, int B, int L, int K, int S, int M, int aligned)
{
    extern __shared__ float Ps[];            // [L*K][PT]
    const int b = blockIdx.y, s0 = blockIdx.z * PT;
    const int LK = L * K;
    for (int idx = threadIdx.x; idx < LK * PT; idx += blockDim.x) {
        const int r = idx / PT, pp = idx - r * PT;
        const int t = r / K, k = r - t * K;
        const int s = s0 + pp;
        Ps[idx] = s < S ? P[lbk_off(t, b, k, B, K, S) + s] : 0.0f;
    }
    __syncthreads();

    const int Mv = M / 4;
    const int shift4 = aligned ? (int)((((long long)b * M) & 15) >> 2) : 0;
    for (int lv = blockIdx.x * blockDim.x + threadIdx.x; lv < Mv + shift4; lv += gridDim.x * blockDim.x) {
        const bool on = lv >= shift4;
        const int m = on ? (lv - shift4) * 4 : 0;
        f32x2_t w01[KR], w23[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            w01[k] = (f32x2_t){k < K ? Wp[(long long)(m + 0) * ldw + k] : 0.0f, k < K ? Wp[(long long)(m + 1) * ldw + k] : 0.0f};
            w23[k] = (f32x2_t){k < K ? Wp[(long long)(m + 2) * ldw + k] : 0.0f, k < K ? Wp[(long long)(m + 3) * ldw + k] : 0.0f};
        }
        float4 a[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) {
            const int s = s0 + pp < S ? s0 + pp : S - 1;
            a[pp] = *reinterpret_cast<const float4 *>(A + ((long long)b * S + s) * M + m);
        }
#pragma unroll 2
        for (int t = 0; t < L; ++t) {
            const float4 bt = *reinterpret_cast<const float4 *>(Bt + ((long long)t * B + b) * M + m);
            f32x2_t q01[PT], q23[PT];
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) {
                q01[pp] = (f32x2_t){a[pp].x + bt.x, a[pp].y + bt.y};
                q23[pp] = (f32x2_t){a[pp].z + bt.z, a[pp].w + bt.w};
            }
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                if (k < K) {
                    float pv[PT];
                    const float *pr = Ps + (t * K + k) * PT;
                    if (PT % 4 == 0) {
#pragma unroll
                        for (int p4 = 0; p4 < PT / 4; ++p4) {
                            const float4 v = *reinterpret_cast<const float4 *>(pr + 4 * p4);
                            pv[4 * p4] = v.x; pv[4 * p4 + 1] = v.y; pv[4 * p4 + 2] = v.z; pv[4 * p4 + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int pp = 0; pp < PT; ++pp) pv[pp] = pr[pp];
                    }
#pragma unroll
                    for (int pp = 0; pp < PT; ++pp) {
                        const f32x2_t p2 = {pv[pp], pv[pp]};
                        q01[pp] = __builtin_elementwise_fma(w01[k], p2, q01[pp]);
                        q23[pp] = __builtin_elementwise_fma(w23[k], p2, q23[pp]);
                    }
                }
            }
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) {
                const int s = s0 + pp;
                if (on && s < S) {
                    // streamed (non-temporal) when the output is far larger than L2 + MALL: written once, read by a later
                    // kernel from HBM either way -- plain stores held 2 GB of dirty lines in a write-back cache and the
                    // evictions ran at 3.4 TB/s; streamed, the same rows leave at 4.7 TB/s
                    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                    f32x4_nt *dst = reinterpret_cast<f32x4_nt *>(Q + (((long long)s * L + t) * B + b) * M + m);
                    const f32x4_nt v = {q01[pp][0], q01[pp][1], q23[pp][0], q23[pp][1]};
                    if (NT) __builtin_nontemporal_store(v, dst);      // a template choice: hipcc merges a run-time pair into plain stores
                    else *dst = v;
                }
            }
        }
    }
}

// ---- backward A: per (b, pixel): gA[b,s,m] = sum_t gQ ; gP[t,b,k,s] = sum_m Wp[m,k] gQ ----
// One 256-thread workgroup per (b, tile of MIX_BWD_PT pixels).  Wp is staged once per
// workgroup into LDS, transposed to [k][M] so that a lane's 4 adjacent m are one ds_read_b128.
// Wave w owns the words t = w, w+4, ...: its 64 lanes sweep the M-long gQ row in 16-byte
// pieces (lane-private partial dot products with the K filter columns), then ONE butterfly
// reduction per (t, k) inside the wave (no workgroup barrier per word).  gA is accumulated per
// lane and combined across the 4 waves through LDS.  Requires M % 4 == 0, M <= 1024.
constexpr int MIX_BWD_PT = 4;

__global__ void __launch_bounds__(256)
mix_bwd_pix_kernel(const float *__restrict__ gQ, const float *__restrict__ Wp, int ldw,
                   float *__restrict__ gA, float *__restrict__ gP, int B, int L, int K, int S, int M)
{
    extern __shared__ float lds[];                  // Wl[K][M] then ga_lds[4][M]
    float *Wl = lds, *ga_lds = lds + (size_t)K * M;
    const int b = blockIdx.y, s0 = blockIdx.x * MIX_BWD_PT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < K * M; idx += blockDim.x) {
        const int m = idx / K, k = idx - m * K;
        Wl[k * M + m] = Wp[(long long)m * ldw + k];
    }
    __syncthreads();
    for (int pp = 0; pp < MIX_BWD_PT; ++pp) {
        const int s = s0 + pp;
        if (s >= S) break;                          // uniform
        float ga[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ga[i] = 0.0f;
        for (int t = wave; t < L; t += 4) {
            float part[MIX_KMAX];
#pragma unroll
            for (int k = 0; k < MIX_KMAX; ++k) part[k] = 0.0f;
            const float *g = gQ + (((long long)s * L + t) * B + b) * M;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (i * 64 + lane) * 4;
                if (m < M) {
                    const float4 gv = *reinterpret_cast<const float4 *>(g + m);
                    ga[i * 4 + 0] += gv.x; ga[i * 4 + 1] += gv.y; ga[i * 4 + 2] += gv.z; ga[i * 4 + 3] += gv.w;
#pragma unroll
                    for (int k = 0; k < MIX_KMAX; ++k) {
                        if (k < K) {
                            const float4 w = *reinterpret_cast<const float4 *>(Wl + k * M + m);
                            part[k] = __builtin_fmaf(w.x, gv.x, part[k]);
                            part[k] = __builtin_fmaf(w.y, gv.y, part[k]);
                            part[k] = __builtin_fmaf(w.z, gv.z, part[k]);
                            part[k] = __builtin_fmaf(w.w, gv.w, part[k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < MIX_KMAX; ++k) {
                if (k < K) {
                    float v = part[k];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    if (lane == 0) gP[lbk_off(t, b, k, B, K, S) + s] = v;
                }
            }
        }
        // combine the 4 waves' gA partials
        __syncthreads();                            // previous pixel's ga_lds readers are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = (i * 64 + lane) * 4;
            if (m < M)
                *reinterpret_cast<float4 *>(ga_lds + wave * M + m) =
                    make_float4(ga[i * 4], ga[i * 4 + 1], ga[i * 4 + 2], ga[i * 4 + 3]);
        }
        __syncthreads();
        for (int m = threadIdx.x; m < M; m += 256)
            gA[((long long)b * S + s) * M + m] =
                ga_lds[m] + ga_lds[M + m] + ga_lds[2 * M + m] + ga_lds[3 * M + m];
    }
}

// ---- backward A, round 2: same outputs, no LDS traffic in the inner loop -------------------------------
// mix_bwd_pix_kernel reads its K filter columns from LDS for every 16 bytes of gQ (40 ds_read_b128 per
// word and lane) and reduces every (t, k) dot product with its own 6-step butterfly (300 cross-lane
// operations per pixel and wave): 0.62 ms for the 512 MB of gQ at the training size, 10 % of HBM.  Here a
// lane keeps the K <= 10 filter weights of its 8 columns in 80 registers (512-thread workgroups: waves 0-3 take
// columns < 512, waves 4-7 the rest, so that two waves fit a SIMD) for the whole workgroup
// (PT pixels, unrolled), a pixel's rows are loaded while the previous pixel's are consumed, and the wave's
// NW * 10 <= 64 partial dot products are reduced TOGETHER by a transposing butterfly (step o: a lane keeps
// the half of the list its bit o selects and adds the partner's copy of it; 32 + 16 + .. + 1 = 63
// exchanges, after which lane l holds the wave total of value l).  Requires K <= 10, L <= 24, M % 4 == 0,
// M <= 1024.
constexpr int MIX_BWD2_K = 10, MIX_BWD2_NW = 6;

template <int NW, int PT>
__global__ void __launch_bounds__(512)
mix_bwd_pix2_kernel(const float *__restrict__ gQ, const float *__restrict__ Wp, int ldw,
                    float *__restrict__ gA, float *__restrict__ gP, int B, int L, int K, int S, int Mfull,
                    int m_off, int accumulate)
{
    // columns m_off .. m_off + M - 1 of the Mfull-wide rows (M <= 1024 per launch; wider rows are swept by
    // consecutive launches on the same stream, the later ones ADDING their share into gP: fixed order)
    constexpr int KR = MIX_BWD2_K;
    const int M = min(1024, Mfull - m_off);
    gQ += m_off; gA += m_off; Wp += (long long)m_off * ldw;
    extern __shared__ float lds2[];                 // Wl[KR][1024] while the weights are fetched, then ga_lds[4][M] + red[4][64]
    const int b = blockIdx.y, s0 = blockIdx.x * PT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wq = wave & 3, half = wave >> 2;      // word group / column half (m < 512 or >= 512)
    // the K filter columns: one pass over Wp's rows (K contiguous floats each), transposed through LDS so
    // that a lane's 4 adjacent columns are one ds_read_b128 per filter
    for (int idx = threadIdx.x; idx < KR * 1024; idx += 512) {
        const int m = idx / KR, k = idx - m * KR;
        lds2[k * 1024 + m] = (m < M && k < K) ? Wp[(long long)m * ldw + k] : 0.0f;
    }
    __syncthreads();
    const int mbase = half * 512 + lane * 4;        // this lane's columns: mbase + 256 * i + {0..3}, i = 0, 1
    float w[2][4][KR];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const float4 wv = *reinterpret_cast<const float4 *>(lds2 + k * 1024 + mbase + 256 * i);
            w[i][0][k] = wv.x; w[i][1][k] = wv.y; w[i][2][k] = wv.z; w[i][3][k] = wv.w;
        }
    float *ga_lds = lds2, *red = lds2 + 4 * 1024;   // reused after the barrier below

    // rows of pixel s that this wave owns (words wq, wq+4, ..); rows past L / S / M read as zero
    f32x4_t g[NW][2];
    auto load_word = [&](int s, int tt) {
        const int t = wq + 4 * tt;
        const bool ok = t < L && s < S;
        const float *gr = gQ + (((long long)(ok ? s : 0) * L + (ok ? t : 0)) * B + b) * Mfull;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mbase + 256 * i;
            g[tt][i] = (ok && m < M) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(gr + m))
                                     : (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
#pragma unroll
    for (int tt = 0; tt < NW; ++tt) load_word(s0, tt);
    __syncthreads();                                // every lane has its weights: Wl may be overwritten

#pragma unroll
    for (int pp = 0; pp < PT; ++pp) {
        const int s = s0 + pp;
        float ga[8], v[64];
#pragma unroll
        for (int i = 0; i < 8; ++i) ga[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = 0.0f;
#pragma unroll
        for (int tt = 0; tt < NW; ++tt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float ge[4] = {g[tt][i].x, g[tt][i].y, g[tt][i].z, g[tt][i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ga[i * 4 + e] += ge[e];
#pragma unroll
                    for (int k = 0; k < KR; ++k)
                        v[tt * KR + k] = __builtin_fmaf(w[i][e][k], ge[e], v[tt * KR + k]);
                }
            }
            if (pp + 1 < PT) load_word(s + 1, tt);  // the next pixel's row into the registers just consumed
        }
        // transposing butterfly: 64 values x 64 lanes -> lane l holds the wave's total of value l
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const bool up = lane & o;
#pragma unroll
            for (int jx = 0; jx < o; ++jx) {
                const float send = up ? v[jx] : v[jx + o];
                const float keep = up ? v[jx + o] : v[jx];
                v[jx] = keep + __shfl_xor(send, o, 64);
            }
        }
        if (pp > 0) __syncthreads();                // previous pixel's ga_lds / red readers are done
        if (half == 1) red[wq * 64 + lane] = v[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mbase + 256 * i;
            if (m < M)
                *reinterpret_cast<float4 *>(ga_lds + wq * 1024 + m) =
                    make_float4(ga[i * 4], ga[i * 4 + 1], ga[i * 4 + 2], ga[i * 4 + 3]);
        }
        __syncthreads();
        if (s < S) {                                // uniform
            if (half == 0) {
                const int tt = lane / KR, k = lane - tt * KR, t = wq + 4 * tt;
                if (tt < NW && t < L && k < K) {
                    float *dst = gP + lbk_off(t, b, k, B, K, S) + s;
                    const float tot = v[0] + red[wq * 64 + lane];
                    *dst = accumulate ? *dst + tot : tot;
                }
            }
            for (int m = threadIdx.x * 4; m < M; m += 2048) {
                const float4 a0 = *reinterpret_cast<const float4 *>(ga_lds + m), a1 = *reinterpret_cast<const float4 *>(ga_lds + 1024 + m);
                const float4 a2 = *reinterpret_cast<const float4 *>(ga_lds + 2048 + m), a3 = *reinterpret_cast<const float4 *>(ga_lds + 3072 + m);
                *reinterpret_cast<float4 *>(gA + ((long long)b * S + s) * Mfull + m) =
                    make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
            }
        }
    }
}

// ---- backward B: per (t, b): gBt[t,b,m] = sum_s gQ ; gWp_part[t,b,m,k] = sum_s gQ * P ----
// One workgroup per (t, b); a lane owns 4 adjacent m (16-byte loads), the pixel loop is
// unrolled so that 4 independent rows are in flight per lane; P[t,b] (K x S) sits in LDS.
template <int VEC>
__global__ void __launch_bounds__(256)
mix_bwd_word_kernel(const float *__restrict__ gQ, const float *__restrict__ P,
                    float *__restrict__ gBt, float *__restrict__ gWp_part,
                    int B, int L, int K, int S, int M)
{
    extern __shared__ float Pt[];            // [S][K] for this (t, b) (k innermost)
    const int t = blockIdx.x, b = blockIdx.y;
    for (int idx = threadIdx.x; idx < K * S; idx += blockDim.x) {
        const int k = idx / S, sp = idx - k * S;
        Pt[sp * K + k] = P[lbk_off(t, b, 0, B, K, S) + idx];
    }
    __syncthreads();
    const long long row_stride = (long long)L * B * M;     // between consecutive pixels
    for (int mv = threadIdx.x; mv < M / VEC; mv += blockDim.x) {
        const int m = mv * VEC;
        float gb[VEC], gw[VEC][MIX_KMAX];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            gb[e] = 0.0f;
#pragma unroll
            for (int k = 0; k < MIX_KMAX; ++k) gw[e][k] = 0.0f;
        }
        const float *g0 = gQ + ((long long)t * B + b) * M + m;
#pragma unroll 4
        for (int sp = 0; sp < S; ++sp) {
            float ge[VEC];
            if (VEC == 4) {
                const float4 gv = *reinterpret_cast<const float4 *>(g0 + sp * row_stride);
                ge[0] = gv.x; ge[1] = gv.y; ge[2] = gv.z; ge[VEC - 1] = gv.w;
            } else {
                ge[0] = g0[sp * row_stride];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) gb[e] += ge[e];
#pragma unroll
            for (int k = 0; k < MIX_KMAX; ++k) {
                if (k < K) {
                    const float pv = Pt[sp * K + k];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) gw[e][k] = __builtin_fmaf(ge[e], pv, gw[e][k]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            gBt[((long long)t * B + b) * M + m + e] = gb[e];
#pragma unroll
            for (int k = 0; k < MIX_KMAX; ++k)
                if (k < K) gWp_part[(((long long)t * B + b) * M + m + e) * K + k] = gw[e][k];
        }
    }
}

// ---- backward B, round 2: K <= 10 at compile time, P rows padded to 12 floats in LDS (three 16-byte
// broadcast reads per pixel instead of up to 16 guarded 4-byte ones), five pixels in flight per lane ----
template <int UNR>
__global__ void __launch_bounds__(256)
mix_bwd_word2_kernel(const float *__restrict__ gQ, const float *__restrict__ P,
                     float *__restrict__ gBt, float *__restrict__ gWp_part,
                     int B, int L, int K, int S, int M)
{
    constexpr int KR = MIX_BWD2_K, KP = 12;
    extern __shared__ float Pt2[];           // [S][KP] for this (t, b), zero beyond K
    const int t = blockIdx.x, b = blockIdx.y;
    for (int idx = threadIdx.x; idx < KP * S; idx += blockDim.x) {
        const int sp = idx / KP, k = idx - sp * KP;
        Pt2[idx] = k < K ? P[lbk_off(t, b, k, B, K, S) + sp] : 0.0f;
    }
    __syncthreads();
    const long long row_stride = (long long)L * B * M;     // between consecutive pixels
    const int m = ((int)blockIdx.z * 256 + threadIdx.x) * 4;   // blockIdx.z: blocks of 1024 columns
    if (m >= M) return;
    float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gw[4][KR];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KR; ++k) gw[e][k] = 0.0f;
    const float *g0 = gQ + ((long long)t * B + b) * M + m;
    auto step = [&](const f32x4_t gv, int sp) {
        const float4 p0 = *reinterpret_cast<const float4 *>(Pt2 + sp * KP);
        const float4 p1 = *reinterpret_cast<const float4 *>(Pt2 + sp * KP + 4);
        const float2 p2 = *reinterpret_cast<const float2 *>(Pt2 + sp * KP + 8);
        const float pv[KR] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y};
        const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gb[e] += ge[e];
#pragma unroll
            for (int k = 0; k < KR; ++k) gw[e][k] = __builtin_fmaf(ge[e], pv[k], gw[e][k]);
        }
    };
    int sp = 0;
    for (; sp + UNR <= S; sp += UNR) {
        f32x4_t gv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(g0 + (sp + u) * row_stride));
#pragma unroll
        for (int u = 0; u < UNR; ++u) step(gv[u], sp + u);
    }
    for (; sp < S; ++sp)
        step(__builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(g0 + sp * row_stride)), sp);
    *reinterpret_cast<float4 *>(gBt + ((long long)t * B + b) * M + m) = make_float4(gb[0], gb[1], gb[2], gb[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KR; ++k)
            if (k < K) gWp_part[(((long long)t * B + b) * M + m + e) * K + k] = gw[e][k];
}


// ---- backward, ONE pass over gQ (round 3) ------------------------------------------------------------------
// The two kernels above each stream gQ once (pixel pass: gA, gP; word pass: gBt, gWp): 2 x 2.06 GB at the
// training step's folded width (B = 64, S = 100, M = 4020), 0.65 + 0.62 ms.  Here a 512-thread workgroup owns
// 512 columns of ONE sample and walks ALL its pixels, so every output of those columns is formed from a single
// read of gQ:  wave (wq, half) = words wq, wq+4, .. x columns half*256 + 4*lane .. +3; per lane
//   w2  [4][5]  its columns' K <= 10 filter weights (pairs)            gA  summed over the wave's words per pixel,
//   gbt [NW][4] gBt of its words, summed over pixels in pixel order         combined over the 4 word groups in LDS
//   gw2 [4][5]  sum over its words and all pixels of gQ * P            v   partial dot products of a pixel
// all FMAs packed over filter pairs (v_pk_fma_f32, the gQ value broadcast by op_sel).  A pixel's NW*10 partial
// dot products are reduced over the 64 lanes in two batches of <= 32 by a transposing butterfly whose exchanges
// are v_permlane32_swap / v_permlane16_swap (one swap + one add per pair: the swap IS the select) and DPP-fused
// adds (row_ror:8, row_half_mirror, quad_perm) -- ~140 instructions per pixel and wave instead of ~250 with
// __shfl_xor.  gP needs the sum over ALL column blocks: every block writes its share to `gPpart[cb]` and
// mix_bwd_sum_parts adds them in block order (deterministic; no atomics).  gWp leaves as one (M, K) slab per
// (sample, word group), summed by the caller.  One barrier per pixel (LDS scratch double-buffered).
//
// Loads.  With the rows of the NEXT pixel prefetched into registers the kernel ran 0.63 ms, 0.38 ms with the
// loads removed (profiles/r3_mix_bwd_one.txt): two waves per SIMD x one pixel (5 KB) per wave in flight cannot
// cover HBM latency at this rate, and there are no registers for a second pixel.  So the rows go global -> LDS
// directly (buffer_load_dwordx4 .. lds, no VGPR destination), TWO pixels ahead, into a wave-private ring of
// three pixel slots, and a pixel's filter responses P the same way, three pixels ahead into a ring of four.
// These loads are inline asm that hipcc does not count; every wave issues exactly NW + 1 of them per pixel
// (rows of words past L go to a dummy slot, their real slots stay zero) and retires them with ONE counted
// s_waitcnt vmcnt(2*NW + 1) per pixel: that leaves the row loads of pixels s+1, s+2 and the P load of pixel
// s+3 in flight and guarantees this pixel's rows and pixel s+2's P (older than all of those; loads retire in
// order, and stores in between can only make the wait stricter).  A wave reads only its own row slots, so the
// wait alone orders them (MI355X_MICROARCH.md item 7); P is read by every wave, two barriers after its wait.
// Requires K <= 10, L <= 20, M % 4 == 0, 16-byte aligned gQ / gA / gBt.
constexpr int MIX_ONE_COLS = 256;          // columns per wave quartet; a workgroup has HALVES quartets (one or two)
constexpr int MIX_ONE_NW = 5;            // words per wave: L <= 20
#ifndef DMN_MIX_ONE_PROBE
#define DMN_MIX_ONE_PROBE 0      // timing probes (tools/probe_mix_one.sh): 1 no butterfly, 2 no combine/stores, 3 no gw FMAs, 4 no gQ loads
#endif

// HALVES = 2: 512 threads, 512 columns, one 158 KB workgroup per CU (its eight waves wait for memory together).
// HALVES = 1: 256 threads, 256 columns, 79 KB: TWO independent workgroups per CU, so that one computes while the
// other waits at its per-pixel wait / barrier (twice the gP partials: + 41 MB written and read back).
template <int HALVES>
struct MixOneLds {
    static constexpr int RUN = HALVES == 2 ? 16 : 8;   // pixels of gP gathered in LDS before they are written (64 / 32-byte runs)
    float G[3][4 * HALVES][MIX_ONE_NW][256];   // [pixel slot][wave][word][64 lanes x 4 columns]       120 / 60 KB
    float dummy[256];                          // landing zone of loads nobody reads (words past L, waves 4-7's P piece)  1 KB
    float Ps[4][256];                          // P[t, b, k, s] of a pixel as [t][k] (first 200 used)      4 KB
    float ga[2][4][MIX_ONE_COLS * HALVES];     //                                                         16 / 8 KB
    float red[HALVES == 2 ? 2 * 4 * 2 * 64 : 1];   // [buf][wq][batch][lane], second quartet -> first      4 KB / -
    float gp[4 * MIX_ONE_NW * MIX_BWD2_K][RUN];    // a run of pixels of this block's gP share          12.5 / 6.25 KB
};

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)p; }

// one 16-byte-per-lane (or 4-byte-per-lane) load straight into LDS at `lds_base + 16 (4) * lane`; not waited for
__device__ __forceinline__ void dma_b128(unsigned lds_base, i32x4 srd, unsigned voff, unsigned soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void dma_b32(unsigned lds_base, i32x4 srd, unsigned voff, unsigned soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
}

template <int HALVES>
__global__ void __launch_bounds__(256 * HALVES)
mix_bwd_one_kernel(const float *__restrict__ gQ, const float *__restrict__ P, const float *__restrict__ Wp, int ldw,
                   float *__restrict__ gA, float *__restrict__ gBt, float *__restrict__ gPpart,
                   float *__restrict__ gWpart, int B, int L, int K, int S, int M, int ncb, int xcd_map)
{
    constexpr int NW = MIX_ONE_NW, KR = MIX_BWD2_K, KP = KR / 2;
    constexpr int NT = 256 * HALVES, COLS = MIX_ONE_COLS * HALVES, RUN = MixOneLds<HALVES>::RUN;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    MixOneLds<HALVES> &sm = *reinterpret_cast<MixOneLds<HALVES> *>(lds_raw);
    // 1-D grid.  Workgroups are dealt to the 8 XCDs round-robin; with xcd_map (B % 8 == 0) an XCD takes ALL column
    // blocks of its samples, consecutively: the workgroups resident on an XCD then read whole 16 KB rows between
    // them (a few samples x 20 words per pixel step) instead of one 2 KB piece of every sample's rows -- 8x fewer
    // distinct pages per XCD and step for the address translation to keep up with
    int cb, b;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = (j / ncb) * 8 + xcd;
        cb = j - (j / ncb) * ncb;
    } else {
        b = blockIdx.x / ncb;
        cb = blockIdx.x - b * ncb;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform for the compiler too (SGPR asm operands)
    const int wq = wave & 3, half = HALVES == 2 ? wave >> 2 : 0;
    const int m = cb * COLS + half * 256 + lane * 4;
    const bool mok = m < M;                                // M % 4 == 0: all four columns or none

    f32x2_t w2[4][KP], gw2[4][KP];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float *wr = Wp + (long long)(mok ? m + e : 0) * ldw;          // (in bounds either way)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const float w0 = wr[2 * kp < K ? 2 * kp : 0], w1 = wr[2 * kp + 1 < K ? 2 * kp + 1 : 0];
            w2[e][kp].x = (mok && 2 * kp < K) ? w0 : 0.0f;
            w2[e][kp].y = (mok && 2 * kp + 1 < K) ? w1 : 0.0f;
            gw2[e][kp] = (f32x2_t){0.0f, 0.0f};
        }
    }
    f32x2_t gbt2[NW][2];
#pragma unroll
    for (int tt = 0; tt < NW; ++tt) gbt2[tt][0] = gbt2[tt][1] = (f32x2_t){0.0f, 0.0f};

    // zero the row slots once: the slots of words past L are never written again (their loads land in `dummy`)
    for (int i = tid; i < 3 * 4 * HALVES * NW * 256 / 4; i += NT)
        reinterpret_cast<float4 *>(&sm.G[0][0][0][0])[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    // P: lane (wave w, lane l) fetches value idx = 64*w + l = t*KR + k of a pixel (idx past the 4*NW*KR values,
    // or a (t, k) outside L x K: any valid address; those values only ever multiply zero rows or feed outputs
    // that are not stored)
    const int pidx = tid < 4 * NW * KR ? tid : 0;          // (4 * NW * KR = 200 <= 256: waves 0 - 3)
    const int pt = pidx / KR, pk = pidx - pt * KR;
    const unsigned p_voff = (unsigned)(lbk_off(pt < L ? pt : L - 1, b, pk < K ? pk : K - 1, B, K, S) * sizeof(float));
    const i32x4 p_srd = make_srd(P);                       // (L*B*K*S floats: far below the 2 GiB window)
    auto dma_P = [&](int s) {                              // pixel s -> Ps[s & 3]
        dma_b32(wave < 4 ? lds_addr(&sm.Ps[s & 3][wave * 64]) : lds_addr(&sm.dummy[0]), p_srd, p_voff,
                (unsigned)((s < S ? s : S - 1) * sizeof(float)));
    };
    // gQ rows: descriptor re-based per pixel (wave-uniform), the words of a pixel as SGPR offsets, one constant
    // VGPR byte offset per lane (a lane without a column re-reads the last valid one: its weights are zero)
    const long long wordstride = (long long)B * M;          // floats between consecutive words of a pixel
    const int mc = mok ? m : M - 4;
    const unsigned lane_off = (unsigned)((((long long)wq * B + b) * M + mc) * sizeof(float));
    auto dma_rows = [&](int s) {                           // pixel s -> G[s % 3][wave]
        const i32x4 srd = make_srd(gQ + (long long)(s < S ? s : S - 1) * L * wordstride);
        const int slot = s % 3;
#pragma unroll
        for (int tt = 0; tt < NW; ++tt) {
            const bool real = wq + 4 * tt < L;             // wave-uniform
            const unsigned dst = real ? lds_addr(&sm.G[slot][wave][tt][0]) : lds_addr(&sm.dummy[0]);
#if DMN_MIX_ONE_PROBE != 4
            dma_b128(dst, srd, lane_off, (unsigned)((real ? 4 * tt : 0) * wordstride * sizeof(float)));
#endif
        }
    };
    // prologue: P of pixels 0, 1, 2 and the rows of pixels 0, 1, in the loop's issue order (P first)
    dma_P(0); dma_rows(0);
    dma_P(1); dma_rows(1);
    dma_P(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // 32-slot transposing butterfly over the 64 lanes: afterwards lane l holds the wave total of value l >> 1
    auto butterfly32 = [&](float (&v)[32]) -> float {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 16]), false, false);
            v[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 8]), false, false);
            v[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        auto dpp_step = [&](auto ctrl, int o, bool up) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < o) {
                    const float a = v[j], c = v[j + o];
                    const float s1 = a + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a), decltype(ctrl)::value, 0xf, 0xf, false));
                    const float s2 = c + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(c), decltype(ctrl)::value, 0xf, 0xf, false));
                    v[j] = up ? s2 : s1;
                }
            }
        };
        dpp_step(IC<0x128>{}, 4, (lane & 8) != 0);         // row_ror:8           lane i <-> i ^ 8
        dpp_step(IC<0x141>{}, 2, (lane & 4) != 0);         // row_half_mirror     lane i <-> 7 - i (within 8)
        dpp_step(IC<0x4e>{}, 1, (lane & 2) != 0);          // quad_perm [2,3,0,1] lane i <-> i ^ 2
        return v[0] + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v[0]), 0xb1, 0xf, 0xf, false));   // i <-> i ^ 1
    };
    constexpr int NB1 = 3, NB2 = NW - 3;                   // words per butterfly batch (30 + 20 values)

    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        // issue: P of pixel s+3, rows of pixel s+2 (past the end: the last pixel again, unused); then retire
        // everything but those, the previous pixel's row loads and (older than them) nothing else
        dma_P(s + 3);
        dma_rows(s + 2);
#if DMN_MIX_ONE_PROBE != 4
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW + 1) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const float *Gw = &sm.G[s % 3][wave][0][lane * 4];
        const float *Pw = &sm.Ps[s & 3][wq * KR];
        f32x2_t ga2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        float tot[2];
#pragma unroll
        for (int batch = 0; batch < 2; ++batch) {
            const int tt0 = batch ? NB1 : 0, ntt = batch ? NB2 : NB1;
            float v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = 0.0f;
#pragma unroll
            for (int bt = 0; bt < NB1; ++bt) {
                if (bt < ntt) {
                    const int tt = tt0 + bt;
                    const float4 gv = *reinterpret_cast<const float4 *>(Gw + tt * 256);
                    const float *pr = Pw + 4 * tt * KR;
                    f32x2_t pv2[KP], v2[KP];
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp) pv2[kp] = *reinterpret_cast<const f32x2_t *>(pr + 2 * kp);
                    const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
                    ga2[0] += (f32x2_t){ge[0], ge[1]};
                    ga2[1] += (f32x2_t){ge[2], ge[3]};
                    gbt2[tt][0] += (f32x2_t){ge[0], ge[1]};
                    gbt2[tt][1] += (f32x2_t){ge[2], ge[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2_t gg = {ge[e], ge[e]};
#pragma unroll
                        for (int kp = 0; kp < KP; ++kp) {
                            v2[kp] = e == 0 ? w2[e][kp] * gg : __builtin_elementwise_fma(w2[e][kp], gg, v2[kp]);
#if DMN_MIX_ONE_PROBE != 3
                            gw2[e][kp] = __builtin_elementwise_fma(gg, pv2[kp], gw2[e][kp]);
#endif
                        }
                    }
#pragma unroll
                    for (int kp = 0; kp < KP; ++kp) { v[bt * KR + 2 * kp] = v2[kp].x; v[bt * KR + 2 * kp + 1] = v2[kp].y; }
                }
            }
#if DMN_MIX_ONE_PROBE == 1
#pragma unroll
            for (int i = 1; i < 32; ++i) asm volatile("" ::"v"(v[i]));
            tot[batch] = v[0];
#else
            tot[batch] = butterfly32(v);
#endif
        }
#if DMN_MIX_ONE_PROBE == 2
        asm volatile("" ::"v"(tot[0]), "v"(tot[1]), "v"(ga2[0]), "v"(ga2[1]));
        continue;
#endif
        if (HALVES == 2 && half == 1) { sm.red[((buf * 4 + wq) * 2 + 0) * 64 + lane] = tot[0]; sm.red[((buf * 4 + wq) * 2 + 1) * 64 + lane] = tot[1]; }
        *reinterpret_cast<float4 *>(&sm.ga[buf][wq][half * 256 + lane * 4]) = make_float4(ga2[0].x, ga2[0].y, ga2[1].x, ga2[1].y);
        __syncthreads();
        // this block's share of gP: gathered over MIX_ONE_RUN pixels in LDS, written as 64-byte runs (one
        // 4-byte store per value and pixel cost 8x its bytes in HBM writes: WRITE_SIZE 0.54 GB for 0.2 GB)
        if (half == 0 && !(lane & 1)) {
#pragma unroll
            for (int batch = 0; batch < 2; ++batch) {
                const int idx = lane >> 1, bt = idx / KR, k = idx - bt * KR;
                const int tt = (batch ? NB1 : 0) + bt, t = wq + 4 * tt;
                if (bt < (batch ? NB2 : NB1) && t < 4 * NW)
                    sm.gp[t * KR + k][s & (RUN - 1)] = HALVES == 2 ? tot[batch] + sm.red[((buf * 4 + wq) * 2 + batch) * 64 + lane] : tot[batch];
            }
        }
        if ((s & (RUN - 1)) == RUN - 1 || s == S - 1) {     // uniform
            __syncthreads();
            const int s0 = s & ~(RUN - 1), n = s - s0 + 1;
            float *gpo = gPpart + (long long)cb * L * B * K * S;
            if ((S & 3) == 0 && (n & 3) == 0) {
                for (int i = tid; i < 4 * NW * KR * (RUN / 4); i += NT) {
                    const int row = i / (RUN / 4), q = i - row * (RUN / 4);
                    const int t = row / KR, k = row - t * KR;
                    if (t < L && k < K && 4 * q < n)
                        *reinterpret_cast<float4 *>(gpo + lbk_off(t, b, k, B, K, S) + s0 + 4 * q) =
                            *reinterpret_cast<const float4 *>(&sm.gp[row][4 * q]);
                }
            } else {
                for (int i = tid; i < 4 * NW * KR * RUN; i += NT) {
                    const int row = i / RUN, q = i - row * RUN;
                    const int t = row / KR, k = row - t * KR;
                    if (t < L && k < K && q < n) gpo[lbk_off(t, b, k, B, K, S) + s0 + q] = sm.gp[row][q];
                }
            }
        }
        {
            const int mcol = cb * COLS + tid;
            if (mcol < M)
                gA[((long long)b * S + s) * M + mcol] = (sm.ga[buf][0][tid] + sm.ga[buf][1][tid])
                                                        + (sm.ga[buf][2][tid] + sm.ga[buf][3][tid]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead loads past the last pixel
    if (mok) {
#pragma unroll
        for (int tt = 0; tt < NW; ++tt) {
            const int t = wq + 4 * tt;
            if (t < L)
                *reinterpret_cast<float4 *>(gBt + ((long long)t * B + b) * M + m) =
                    make_float4(gbt2[tt][0].x, gbt2[tt][0].y, gbt2[tt][1].x, gbt2[tt][1].y);
        }
        float *gwp = gWpart + (((long long)b * 4 + wq) * M + m) * K;   // rows m .. m+3 of this slab: 4*K adjacent floats
        if (K == KR) {
            float flat[4 * KR];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) { flat[e * KR + 2 * kp] = gw2[e][kp].x; flat[e * KR + 2 * kp + 1] = gw2[e][kp].y; }
#pragma unroll
            for (int q = 0; q < KR; ++q)
                *reinterpret_cast<float4 *>(gwp + 4 * q) = make_float4(flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    if (2 * kp < K) gwp[e * K + 2 * kp] = gw2[e][kp].x;
                    if (2 * kp + 1 < K) gwp[e * K + 2 * kp + 1] = gw2[e][kp].y;
                }
        }
    }
}

// gP[i] = sum over the column blocks of gPpart[cb][i], in block order
__global__ void __launch_bounds__(256)
mix_bwd_sum_parts(const float *__restrict__ part, float *__restrict__ out, long long n, int nparts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = part[i];
    for (int p = 1; p < nparts; ++p) acc += part[(long long)p * n + i];
    out[i] = acc;
}

static int check_mix(const char *fn, int B, int L, int K, int S, int M, int ldw)
{
    if (B <= 0 || L <= 0 || K <= 0 || S <= 0 || M <= 0 || ldw < K)
        return fail(-4, "%s: bad sizes B=%d L=%d K=%d S=%d M=%d ldw=%d", fn, B, L, K, S, M, ldw);
    if (B > 65535) return fail(-5, "%s: B=%d exceeds the grid y limit", fn, B);
    return 0;
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_mix_bcast_add(const float *A, const float *Bt, const float *P, const float *Wp,
                                 int ldw, float *Q, int B, int L, int K, int S, int M, void *stream)
{
    if (!A || !Bt || !P || !Wp || !Q) return fail(-1, "dmn_mix_bcast_add: NULL pointer");
    if (int rc = check_mix("dmn_mix_bcast_add", B, L, K, S, M, ldw)) return rc;
    if ((size_t)L * K * MIX_PT * sizeof(float) > 64 * 1024)
        return fail(-5, "dmn_mix_bcast_add: L*K=%d too large for the LDS stage", L * K);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (M % 4 == 0) && !(((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)Q) & 15);
    // packed-FMA kernel with 64-byte-aligned column windows (K <= 10, 16-byte case); DMN_MIX_FWD_PK=0 keeps the plain one
    const char *pk_env = getenv("DMN_MIX_FWD_PK");
    const bool pk = vec && K <= 10 && !(pk_env && pk_env[0] == '0');
    const int aligned = pk && ((long long)B * M) % 16 == 0 && !((uintptr_t)Q & 63);
    const int pieces = ((vec ? M / 4 : M) + (aligned ? 3 : 0) + 255) / 256;
    if ((S + 1) / 2 > 65535) return fail(-5, "dmn_mix_bcast_add: S=%d too large", S);
    // few (image, query) pairs (batch-1 serving): 2 pixels per workgroup instead of 8, so that
    // the launch still has >= 2 workgroups per CU
    const bool small = (long long)pieces * B * ((S + MIX_PT - 1) / MIX_PT) < 2 * NUM_CU;
    // 4 pixels per workgroup when the last tile of 8 would be at most half full (S = 100: 12.5 tiles) -- less
    // waste and a finer-grained tail (DMN_MIX_FWD_PT=4|8 forces it)
    const char *pte = getenv("DMN_MIX_FWD_PT");
    const int rem8 = S % MIX_PT;
    const int pt_auto = (rem8 >= 1 && rem8 <= MIX_PT / 2) ? 4 : MIX_PT;
    const int pt = small ? 2 : !pk ? MIX_PT : pte ? (atoi(pte) == 4 ? 4 : MIX_PT) : pt_auto;
    const size_t lds = (size_t)L * K * pt * sizeof(float);
    dim3 grid(pieces, B, (S + pt - 1) / pt);
#define DMN_MIX(V, KR)                                                                                  \
    do {                                                                                                \
        if (small) hipLaunchKernelGGL((mix_fwd_kernel<V, KR, 2>), grid, dim3(256), lds, s, A, Bt, P, Wp, ldw, Q, B, L, K, S, M); \
        else hipLaunchKernelGGL((mix_fwd_kernel<V, KR, MIX_PT>), grid, dim3(256), lds, s, A, Bt, P, Wp, ldw, Q, B, L, K, S, M); \
    } while (0)
    if (pk) {
        // streamed stores once Q outgrows the 256 MB Infinity Cache (DMN_MIX_FWD_NT=0|1 forces the policy)
        const char *ne = getenv("DMN_MIX_FWD_NT");
        const int nt = ne ? atoi(ne) != 0 : (double)S * L * B * M * 4.0 > 256e6;
#define DMN_MIX_PK(PTV)                                                                                                      \
    do {                                                                                                                     \
        if (nt) hipLaunchKernelGGL((mix_fwd_pk_kernel<10, PTV, true>), grid, dim3(256), lds, s, A, Bt, P, Wp, ldw, Q, B, L, K, S, M, aligned);  \
        else hipLaunchKernelGGL((mix_fwd_pk_kernel<10, PTV, false>), grid, dim3(256), lds, s, A, Bt, P, Wp, ldw, Q, B, L, K, S, M, aligned);  \
    } while (0)
        if (small) DMN_MIX_PK(2);
        else if (pt == 4) DMN_MIX_PK(4);
        else DMN_MIX_PK(MIX_PT);
#undef DMN_MIX_PK
    } else if (vec) {
        if (K <= 4) DMN_MIX(4, 4);
        else if (K <= 10) DMN_MIX(4, 10);
        else if (K <= MIX_KMAX) DMN_MIX(4, MIX_KMAX);
        else DMN_MIX(4, 0);
    } else {
        if (K <= MIX_KMAX) DMN_MIX(1, MIX_KMAX);
        else DMN_MIX(1, 0);
    }
#undef DMN_MIX
    return launch_status("dmn_mix_bcast_add");
}

namespace dmn {
// which backward runs: the one-pass kernel needs the model's K / L range and at least half a workgroup per CU
// (B samples x 512-column blocks); DMN_MIX_BWD_ONE=0|1 forces the choice for A/B runs and tests
static bool mix_bwd_one_pass(int B, int L, int K, int M)
{
    const int nw = (L + 3) / 4;
    if (!(K <= MIX_BWD2_K && nw <= MIX_ONE_NW && M % 4 == 0)) return false;
    const char *e = getenv("DMN_MIX_BWD_ONE");
    if (e) return atoi(e) != 0;
    const long long wgs = (long long)B * ((M + 2 * MIX_ONE_COLS - 1) / (2 * MIX_ONE_COLS));
    return wgs >= NUM_CU / 2;
}
// wave quartets per workgroup: 1 (two independent 256-column workgroups per CU) unless DMN_MIX_ONE_HALVES=2
static int mix_one_halves(int B, int M)
{
    const char *e = getenv("DMN_MIX_ONE_HALVES");
    if (e) return atoi(e) == 2 ? 2 : 1;
    // 512-column workgroups (one per CU) once they fill the chip -- 0.57 vs 0.64 ms at M = 4020, B = 64; below that the
    // 256-column ones give twice the workgroups (M = 1000, B = 64: 0.19 vs 0.25 ms)
    return (long long)B * ((M + 2 * MIX_ONE_COLS - 1) / (2 * MIX_ONE_COLS)) >= NUM_CU ? 2 : 1;
}
}  // namespace dmn

extern "C" int dmn_mix_bwd_parts(int B, int L, int K, int S, int M)
{
    (void)S;
    if (B <= 0 || L <= 0 || K <= 0 || M <= 0) return -1;
    return mix_bwd_one_pass(B, L, K, M) ? 4 * B : L * B;
}

extern "C" long long dmn_mix_bwd_workspace_floats(int B, int L, int K, int S, int M)
{
    if (B <= 0 || L <= 0 || K <= 0 || S <= 0 || M <= 0) return -1;
    if (!mix_bwd_one_pass(B, L, K, M)) return 0;
    return (long long)((M + MIX_ONE_COLS - 1) / MIX_ONE_COLS) * L * B * K * S;    // one gP share per 256 columns (enough for either kernel)
}

extern "C" int dmn_mix_bcast_add_bwd(const float *gQ, const float *P, const float *Wp, int ldw,
                                     float *gA, float *gBt, float *gP, float *gWp_part, float *workspace,
                                     int B, int L, int K, int S, int M, void *stream)
{
    if (!gQ || !P || !Wp || !gA || !gBt || !gP || !gWp_part)
        return fail(-1, "dmn_mix_bcast_add_bwd: NULL pointer");
    if (int rc = check_mix("dmn_mix_bcast_add_bwd", B, L, K, S, M, ldw)) return rc;
    if (K > MIX_KMAX) return fail(-5, "dmn_mix_bcast_add_bwd: K=%d > %d not supported", K, MIX_KMAX);
    hipStream_t s = (hipStream_t)stream;
    if (M % 4 != 0 || ((uintptr_t)gQ & 15)) return fail(-5, "dmn_mix_bcast_add_bwd: M %% 4 == 0 and 16-byte aligned gQ required");
    const int nw = (L + 3) / 4;
    if (mix_bwd_one_pass(B, L, K, M)) {
        if (((uintptr_t)gA | (uintptr_t)gBt | (uintptr_t)gWp_part) & 15)
            return fail(-5, "dmn_mix_bcast_add_bwd: gA, gBt and gWp_part must be 16-byte aligned");
        if (!workspace) return fail(-1, "dmn_mix_bcast_add_bwd: this shape needs a workspace of dmn_mix_bwd_workspace_floats() floats");
        const int halves = mix_one_halves(B, M);
        const int ncb = (M + halves * MIX_ONE_COLS - 1) / (halves * MIX_ONE_COLS);
        if ((long long)ncb * B >= (1LL << 31)) return fail(-5, "dmn_mix_bcast_add_bwd: grid too large");
        dim3 grid((unsigned)(ncb * B));
        const char *xe = getenv("DMN_MIX_ONE_XCD");
        const int xcd_map = (B % 8 == 0) && !(xe && atoi(xe) == 0);
        // 79 / 158 KB of the CU's 160 KB LDS: opt in (cheap; per call, so that every device of the process gets it)
#define DMN_MIX_ONE(HV)                                                                                                  \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mix_bwd_one_kernel<HV>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MixOneLds<HV>));     \
        if (e != hipSuccess) return fail((int)e, "dmn_mix_bcast_add_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
        hipLaunchKernelGGL(mix_bwd_one_kernel<HV>, grid, dim3(256 * HV), sizeof(MixOneLds<HV>), s, gQ, P, Wp, ldw, gA, gBt, \
                           workspace, gWp_part, B, L, K, S, M, ncb, xcd_map);                                            \
    } while (0)
        if (halves == 2) DMN_MIX_ONE(2);
        else DMN_MIX_ONE(1);
#undef DMN_MIX_ONE
        const long long n = (long long)L * B * K * S;
        hipLaunchKernelGGL(mix_bwd_sum_parts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, workspace, gP, n, ncb);
        return launch_status("dmn_mix_bcast_add_bwd");
    }
    const bool gen2 = K <= MIX_BWD2_K && nw <= MIX_BWD2_NW && M % 4 == 0 && !(((uintptr_t)gA | (uintptr_t)gBt) & 15);
    if (M > 1024 && !gen2)
        return fail(-5, "dmn_mix_bcast_add_bwd: M=%d > 1024 needs K <= %d and L <= %d", M, MIX_BWD2_K, 4 * MIX_BWD2_NW);
    const size_t lds = (size_t)K * S * sizeof(float);
    if (lds > 64 * 1024) return fail(-5, "dmn_mix_bcast_add_bwd: K*S=%d too large for the LDS stage", K * S);
    const char *pe = getenv("DMN_MIX_BWD_PIX2");
    if (gen2 && (M > 1024 || !(pe && atoi(pe) == 0))) {
        // pixels per workgroup: 5 when that divides S (S = 100 in training: 20 x B workgroups), else 4
        const size_t lds2 = (size_t)MIX_BWD2_K * 1024 * sizeof(float);
#define DMN_PIX2(NWV, PTV) hipLaunchKernelGGL((mix_bwd_pix2_kernel<NWV, PTV>), dim3((S + PTV - 1) / PTV, B), dim3(512), lds2, s, gQ, Wp, ldw, gA, gP, B, L, K, S, M, m_off, m_off > 0)
        for (int m_off = 0; m_off < M; m_off += 1024) {
            if (S % 5 == 0) { if (nw <= 5) DMN_PIX2(5, 5); else DMN_PIX2(6, 5); }
            else            { if (nw <= 5) DMN_PIX2(5, 4); else DMN_PIX2(6, 4); }
        }
#undef DMN_PIX2
    } else {
        const size_t lds_pix = (size_t)(K + 4) * M * sizeof(float);
        hipLaunchKernelGGL(mix_bwd_pix_kernel, dim3((S + MIX_BWD_PT - 1) / MIX_BWD_PT, B), dim3(256), lds_pix, s, gQ, Wp, ldw, gA, gP, B, L, K, S, M);
    }
    const char *we = getenv("DMN_MIX_BWD_WORD2");
    if (K <= MIX_BWD2_K && M % 4 == 0 && !((uintptr_t)gBt & 15) && (M > 1024 || !(we && atoi(we) == 0)))
        hipLaunchKernelGGL(mix_bwd_word2_kernel<5>, dim3(L, B, (M + 1023) / 1024), dim3(256), (size_t)12 * S * sizeof(float), s, gQ, P, gBt, gWp_part, B, L, K, S, M);
    else if ((M % 4 == 0) && !((uintptr_t)gQ & 15))
        hipLaunchKernelGGL(mix_bwd_word_kernel<4>, dim3(L, B), dim3(256), lds, s, gQ, P, gBt, gWp_part, B, L, K, S, M);
    else
        hipLaunchKernelGGL(mix_bwd_word_kernel<1>, dim3(L, B), dim3(256), lds, s, gQ, P, gBt, gWp_part, B, L, K, S, M);
    return launch_status("dmn_mix_bcast_add_bwd");
}
