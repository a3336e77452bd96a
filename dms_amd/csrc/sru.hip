// dms_amd/csrc/sru.hip -- SRU forward / backward scans for gfx950 (MI355X).
//
// Replaces the third-party `sru` package's cupy/pynvrtc kernels `sru_fwd` / `sru_bwd`
// reached from dmn_pytorch/models/dmn.py:156,204 (see include/dmn_hip.h).  Written
// from the SRU equations (arXiv:1709.02755), not from that CUDA source.
//
// Design (HBM-bound element-wise recurrence, 20 B of traffic per (t, b, j)):
//  * one lane per column (b, j); a wave covers 64 adjacent columns, so for every time
//    step the wave reads ONE contiguous 768 B (k=3) / 1 KiB (k=4) segment of U with a
//    single global_load_dwordx3 / dwordx4 per lane (the k gate pre-activations of a
//    column are adjacent in the reference (T,B,d,k) layout) -- fully coalesced, no
//    layout change and no LDS transpose needed;
//  * the loads do not depend on the recurrence, so each lane keeps a ring of DEPTH time
//    steps in flight in registers (software prefetch DEPTH steps ahead): with only
//    B*d/64 waves on 256 CUs (2 per CU at B=32, d=1005) the bytes in flight have to
//    come from depth, not from occupancy;
//  * 64-thread workgroups so that small column counts still spread over all 256 CUs;
//  * gates use v_exp_f32 / v_rcp_f32 directly (sigmoid_f / tanh_f in common.h).
#include "common.h"

namespace dmn {

struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

// forward scan kernels live in sru_fwd.inl (compiled per k in sru_fwd_k3.hip / sru_fwd_k4.hip)
void launch_sru_fwd_k3(int act, bool store_c, int depth, dim3 grid, hipStream_t s, const float *U,
                       const float *x, const float *bias, const float *c0, float *h, float *c_all,
                       float *c_last, int T, int ncols, int d, float scale_x, int h_t0,
                       int chunk_len, const float2 *carries);
void launch_sru_fwd_k4(int act, bool store_c, int depth, dim3 grid, hipStream_t s, const float *U,
                       const float *x, const float *bias, const float *c0, float *h, float *c_all,
                       float *c_last, int T, int ncols, int d, float scale_x, int h_t0,
                       int chunk_len, const float2 *carries);
void launch_sru_fwd_k2(int depth, dim3 grid, hipStream_t s, const float *U, const float *bias,
                       const float *c0, float *c_last, int T, int ncols, int d);
constexpr int WHOLE_SEQUENCE = 0x7fffffff / 2;   // chunk_len of the un-chunked scan


// Chunk aggregates for the time-parallel variant: the cell update is the affine map
// c -> f*c + (1-f)*u0, so a chunk [t0, t1) composes to c_out = A*c_in + Bv with
// A = prod f_t and Bv = the chunk run from c_in = 0.  Only u0 and u1 are needed.
template <int K, int DEPTH>
__global__ void __launch_bounds__(WAVE)
sru_chunk_aggregate(const float *__restrict__ U, const float *__restrict__ bias,
                    float2 *__restrict__ carries, int T_total, int ncols, int d, int chunk_len)
{
    const int col = blockIdx.x * WAVE + threadIdx.x;
    if (col >= ncols) return;
    const float bf = bias[col % d];
    const size_t nc = (size_t)ncols;
    const int t_begin = blockIdx.y * chunk_len;
    const int T = min(T_total, t_begin + chunk_len);
    float a = 1.0f, b = 0.0f;
    auto load2 = [&](int t) -> float2 {
        const float *p = U + ((size_t)t * nc + col) * K;
        if (K == 4 || K == 2) return *reinterpret_cast<const float2 *>(p);
        return make_float2(p[0], p[1]);
    };
    if (T > t_begin) {
        float2 ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = load2(t_begin + i < T ? t_begin + i : T - 1);
        auto step = [&](const float2 v) {
            const float f = sigmoid_f(v.y + bf);
            b = __builtin_fmaf(b - v.x, f, v.x);
            a *= f;
        };
        int t0 = t_begin;
        for (; t0 + DEPTH <= T; t0 += DEPTH) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const float2 v = ring[i];
                int tn = t0 + i + DEPTH;
                tn = tn < T ? tn : T - 1;
                ring[i] = load2(tn);
                step(v);
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (t0 + i < T) step(ring[i]);
    }
    carries[(size_t)blockIdx.y * nc + col] = make_float2(a, b);
}

// ------------------------------------------------------------------------------------
// backward: reverse-time scan, same mapping; ring slot i of a block holds time tb - i
// ------------------------------------------------------------------------------------
struct BwdSlot { float u0, u1, u2, xv, c, gh; };

template <int K>
__device__ __forceinline__ BwdSlot load_bwd(const float *__restrict__ U, const float *__restrict__ x,
                                            const float *__restrict__ c, const float *__restrict__ gh,
                                            size_t ncols, int col, int t)
{
    const size_t e = (size_t)t * ncols + col;
    BwdSlot s;
    if (K == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(U + e * 4);
        s.u0 = v.x; s.u1 = v.y; s.u2 = v.z; s.xv = v.w;
    } else {
        const F3 v = *reinterpret_cast<const F3 *>(U + e * 3);
        s.u0 = v.a; s.u1 = v.b; s.u2 = v.c; s.xv = x[e];
    }
    s.c = c[e];
    s.gh = gh[e];
    return s;
}

template <int K, int ACT, int DEPTH>
__global__ void __launch_bounds__(WAVE)
sru_bwd_cols(const float *__restrict__ U, const float *__restrict__ x,
             const float *__restrict__ bias, const float *__restrict__ c0,
             const float *__restrict__ c_all, const float *__restrict__ grad_h,
             const float *__restrict__ grad_last,
             float *__restrict__ grad_U, float *__restrict__ grad_x,
             float *__restrict__ grad_bias_part, float *__restrict__ grad_c0,
             int T, int ncols, int d, float scale_x)
{
    const int col = blockIdx.x * WAVE + threadIdx.x;
    if (col >= ncols) return;
    const int j = col % d;
    const float bf = bias[j], br = bias[j + d];
    const float c0v = c0 ? c0[col] : 0.0f;
    const size_t nc = (size_t)ncols;
    const float sx = (K == 3) ? scale_x : 1.0f;
    float cur = grad_last ? grad_last[col] : 0.0f;   // dL/dc flowing back in time
    float gb1 = 0.0f, gb2 = 0.0f;

    if (T > 0) {
        BwdSlot ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const int t = T - 1 - i;
            ring[i] = load_bwd<K>(U, x, c_all, grad_h, nc, col, t > 0 ? t : 0);
        }

        auto step = [&](const BwdSlot s, int t, float c_prev_ring) {
            const float g1 = sigmoid_f(s.u1 + bf);
            const float g2 = sigmoid_f(s.u2 + br);
            const float cval = act_fwd<ACT>(s.c);
            const float xv = s.xv * sx;
            const float cprev = t > 0 ? c_prev_ring : c0v;
            const size_t e = (size_t)t * nc + col;
            // h = (act(c) - x') * g2 + x'
            const float gxp = s.gh * (1.0f - g2);
            const float gg2 = s.gh * (cval - xv) * (g2 * (1.0f - g2));
            gb2 += gg2;
            // c = (c' - u0) * g1 + u0
            const float gc = __builtin_fmaf(s.gh * g2, act_bwd<ACT>(cval), cur);
            const float gu0 = gc * (1.0f - g1);
            const float gg1 = gc * (cprev - s.u0) * (g1 * (1.0f - g1));
            gb1 += gg1;
            cur = gc * g1;
            if (K == 4) {
                *reinterpret_cast<float4 *>(grad_U + e * 4) = make_float4(gu0, gg1, gg2, gxp);
            } else {
                F3 o; o.a = gu0; o.b = gg1; o.c = gg2;
                *reinterpret_cast<F3 *>(grad_U + e * 3) = o;
                grad_x[e] = gxp * sx;
            }
        };

        int tb = T - 1;
        for (; tb - DEPTH + 1 >= 0; tb -= DEPTH) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const BwdSlot s = ring[i];
                int tn = tb - i - DEPTH;
                tn = tn > 0 ? tn : 0;              // clamped, in bounds, unused when < 0
                ring[i] = load_bwd<K>(U, x, c_all, grad_h, nc, col, tn);
                // c_{t-1}: slot i+1 still holds time t-1; for the last slot it is the
                // freshly re-loaded slot 0 of the next block
                const float cp = (i + 1 < DEPTH) ? ring[i + 1].c : ring[0].c;
                step(s, tb - i, cp);
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (tb - i >= 0) step(ring[i], tb - i, (i + 1 < DEPTH) ? ring[i + 1].c : 0.0f);
    }
    grad_bias_part[col] = gb1;
    grad_bias_part[nc + col] = gb2;
    grad_c0[col] = cur;
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
// tuning / test hook: DMN_SRU_DEPTH=<n> forces the prefetch depth (read on every call)
static int depth_override()
{
    const char *e = getenv("DMN_SRU_DEPTH");
    return e ? atoi(e) : -1;
}

static int check_common(const char *fn, const void *U, const void *x, const void *bias, int T,
                        int B, int d, int k, int act)
{
    if ((!U && T > 0) || !bias) return fail(-1, "%s: U and bias must not be NULL", fn);
    if (k != 3 && k != 4) return fail(-2, "%s: k must be 3 or 4 (got %d)", fn, k);
    if (k == 3 && !x && T > 0) return fail(-3, "%s: k == 3 needs the highway input x", fn);
    if (k == 4 && x) return fail(-3, "%s: k == 4 takes the highway term from U[..,3]; x must be NULL", fn);
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "%s: bad sizes T=%d B=%d d=%d", fn, T, B, d);
    // 32-bit row offsets inside a 2 GiB descriptor window: DEPTH (<= 32) rows of U must fit
    if ((long long)B * d * k * 4 * 33 >= 0x7fffffffLL) return fail(-5, "%s: B*d=%lld too large", fn, (long long)B * d);
    if (act < 0 || act > 2) return fail(-6, "%s: unknown activation %d", fn, act);
    if (k == 4 && ((uintptr_t)U & 15)) return fail(-7, "%s: U must be 16-byte aligned for k == 4", fn);
    return 0;
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_sru_fwd(const float *U, const float *x, const float *bias, const float *c0,
                           float *h, float *c_all, float *c_last, int T, int B, int d, int k,
                           int act, float scale_x, int h_t0, void *stream)
{
    if (int rc = check_common("dmn_sru_fwd", U, x, bias, T, B, d, k, act)) return rc;
    if (h_t0 < 0 || h_t0 > T) return fail(-8, "dmn_sru_fwd: h_t0=%d outside [0, T=%d]", h_t0, T);
    if (!h && h_t0 < T) return fail(-1, "dmn_sru_fwd: h must not be NULL");
    const int ncols = B * d;
    const int nwaves = (ncols + WAVE - 1) / WAVE;
    // bytes in flight must come from depth when there are few waves per CU
    int depth = depth_override();
    if (depth <= 0) depth = (nwaves <= 3 * NUM_CU) ? 32 : (nwaves <= 6 * NUM_CU ? 16 : 8);
    dim3 grid(nwaves);
    hipStream_t s = (hipStream_t)stream;
    if (k == 3) launch_sru_fwd_k3(act, c_all != nullptr, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, WHOLE_SEQUENCE, nullptr);
    else        launch_sru_fwd_k4(act, c_all != nullptr, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, WHOLE_SEQUENCE, nullptr);
    return launch_status("dmn_sru_fwd");
}

extern "C" int dmn_sru_state_fwd(const float *U2, const float *bias, const float *c0,
                                 float *c_last, int T, int B, int d, void *stream)
{
    if ((!U2 && T > 0) || !bias || !c_last) return fail(-1, "dmn_sru_state_fwd: U2, bias and c_last must not be NULL");
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "dmn_sru_state_fwd: bad sizes T=%d B=%d d=%d", T, B, d);
    if ((long long)B * d * 2 * 4 * 33 >= 0x7fffffffLL) return fail(-5, "dmn_sru_state_fwd: B*d=%lld too large", (long long)B * d);
    if ((uintptr_t)U2 & 7) return fail(-7, "dmn_sru_state_fwd: U2 must be 8-byte aligned");
    const int ncols = B * d;
    const int nwaves = (ncols + WAVE - 1) / WAVE;
    int depth = depth_override();
    if (depth <= 0) depth = (nwaves <= 3 * NUM_CU) ? 32 : (nwaves <= 6 * NUM_CU ? 16 : 8);
    launch_sru_fwd_k2(depth, dim3(nwaves), (hipStream_t)stream, U2, bias, c0, c_last, T, ncols, d);
    return launch_status("dmn_sru_state_fwd");
}

namespace dmn {
// c_last = composition of the chunk aggregates applied to c0 (state-only time-parallel scan)
__global__ void __launch_bounds__(256)
sru_fold_carries(const float2 *__restrict__ carries, const float *__restrict__ c0,
                 float *__restrict__ c_last, int ncols, int nchunks)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    float cur = c0 ? c0[col] : 0.0f;
    for (int cc = 0; cc < nchunks; ++cc) {
        const float2 ab = carries[(size_t)cc * ncols + col];
        cur = __builtin_fmaf(ab.x, cur, ab.y);
    }
    c_last[col] = cur;
}
}  // namespace dmn

extern "C" int dmn_sru_state_fwd_scan(const float *U2, const float *bias, const float *c0,
                                      float *c_last, float *workspace, int T, int B, int d,
                                      int nchunks, void *stream)
{
    if ((!U2 && T > 0) || !bias || !c_last) return fail(-1, "dmn_sru_state_fwd_scan: U2, bias and c_last must not be NULL");
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "dmn_sru_state_fwd_scan: bad sizes T=%d B=%d d=%d", T, B, d);
    if ((uintptr_t)U2 & 7) return fail(-7, "dmn_sru_state_fwd_scan: U2 must be 8-byte aligned");
    if (nchunks < 1 || nchunks > 65535) return fail(-9, "dmn_sru_state_fwd_scan: nchunks=%d outside [1, 65535]", nchunks);
    if (!workspace || ((uintptr_t)workspace & 7)) return fail(-1, "dmn_sru_state_fwd_scan: workspace must be non-NULL, 8-byte aligned");
    const int ncols = B * d;
    const int chunk_len = T > 0 ? (T + nchunks - 1) / nchunks : 1;
    const int used = T > 0 ? (T + chunk_len - 1) / chunk_len : 0;     // chunks that hold steps
    hipStream_t s = (hipStream_t)stream;
    float2 *carries = reinterpret_cast<float2 *>(workspace);
    if (used > 0)
        hipLaunchKernelGGL((sru_chunk_aggregate<2, 8>), dim3((ncols + WAVE - 1) / WAVE, used), dim3(WAVE), 0, s,
                           U2, bias, carries, T, ncols, d, chunk_len);
    hipLaunchKernelGGL(sru_fold_carries, dim3((ncols + 255) / 256), dim3(256), 0, s, carries, c0, c_last, ncols, used);
    return launch_status("dmn_sru_state_fwd_scan");
}

namespace dmn {
template <int K>
static void launch_bwd(int act, dim3 grid, hipStream_t s, const float *U, const float *x,
                       const float *bias, const float *c0, const float *c_all,
                       const float *grad_h, const float *grad_last, float *grad_U, float *grad_x,
                       float *gbp, float *grad_c0, int T, int ncols, int d, float scale_x)
{
    constexpr int DEP = 8;
#define DMN_BWD(A)                                                                               \
    hipLaunchKernelGGL((sru_bwd_cols<K, A, DEP>), grid, dim3(WAVE), 0, s, U, x, bias, c0, c_all, \
                       grad_h, grad_last, grad_U, grad_x, gbp, grad_c0, T, ncols, d, scale_x)
    if (act == DMN_ACT_TANH) DMN_BWD(DMN_ACT_TANH);
    else if (act == DMN_ACT_RELU) DMN_BWD(DMN_ACT_RELU);
    else DMN_BWD(DMN_ACT_IDENTITY);
#undef DMN_BWD
}
}  // namespace dmn

extern "C" int dmn_sru_bwd(const float *U, const float *x, const float *bias, const float *c0,
                           const float *c_all, const float *grad_h, const float *grad_last,
                           float *grad_U, float *grad_x, float *grad_bias_part, float *grad_c0,
                           int T, int B, int d, int k, int act, float scale_x, void *stream)
{
    if (int rc = check_common("dmn_sru_bwd", U, x, bias, T, B, d, k, act)) return rc;
    if (((!c_all || !grad_h || !grad_U) && T > 0) || !grad_bias_part || !grad_c0)
        return fail(-1, "dmn_sru_bwd: c_all, grad_h, grad_U, grad_bias_part, grad_c0 must not be NULL");
    if (k == 3 && !grad_x && T > 0) return fail(-3, "dmn_sru_bwd: k == 3 needs grad_x");
    if (k == 4 && ((uintptr_t)grad_U & 15)) return fail(-7, "dmn_sru_bwd: grad_U must be 16-byte aligned");
    const int ncols = B * d;
    dim3 grid((ncols + WAVE - 1) / WAVE);
    hipStream_t s = (hipStream_t)stream;
    if (k == 3) launch_bwd<3>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias_part, grad_c0, T, ncols, d, scale_x);
    else        launch_bwd<4>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias_part, grad_c0, T, ncols, d, scale_x);
    return launch_status("dmn_sru_bwd");
}

// ------------------------------------------------------------------------------------
// time-parallel variant for FEW columns (e.g. B = 1: 1005 columns = 16 waves on 256 CUs):
// chunk aggregates -> per-lane carry composition -> chunk apply.  U is read twice, which
// at these sizes (<= 82 MB) is served by the 256 MiB Infinity Cache the second time.
// ------------------------------------------------------------------------------------
extern "C" int dmn_sru_scan_workspace_floats(int T, int B, int d, int nchunks)
{
    (void)T;
    if (B <= 0 || d <= 0 || nchunks <= 0) return -1;
    const long long n = 2LL * nchunks * B * d;
    return n > 0x7fffffffLL ? -1 : (int)n;
}

extern "C" int dmn_sru_fwd_scan(const float *U, const float *x, const float *bias,
                                const float *c0, float *h, float *c_last, float *workspace,
                                int T, int B, int d, int k, int act, float scale_x, int h_t0,
                                int nchunks, void *stream)
{
    if (int rc = check_common("dmn_sru_fwd_scan", U, x, bias, T, B, d, k, act)) return rc;
    if (h_t0 < 0 || h_t0 > T) return fail(-8, "dmn_sru_fwd_scan: h_t0=%d outside [0, T=%d]", h_t0, T);
    if (!h && h_t0 < T) return fail(-1, "dmn_sru_fwd_scan: h must not be NULL");
    if (nchunks < 1 || nchunks > 65535) return fail(-9, "dmn_sru_fwd_scan: nchunks=%d outside [1, 65535]", nchunks);
    if (!workspace || ((uintptr_t)workspace & 7)) return fail(-1, "dmn_sru_fwd_scan: workspace must be non-NULL, 8-byte aligned");
    const int ncols = B * d;
    const int chunk_len = T > 0 ? (T + nchunks - 1) / nchunks : 1;
    const int used = T > 0 ? (T + chunk_len - 1) / chunk_len : 1;     // chunks that hold steps
    dim3 grid((ncols + WAVE - 1) / WAVE, used);
    hipStream_t s = (hipStream_t)stream;
    float2 *carries = reinterpret_cast<float2 *>(workspace);
    if (k == 3) hipLaunchKernelGGL((sru_chunk_aggregate<3, 8>), grid, dim3(WAVE), 0, s, U, bias, carries, T, ncols, d, chunk_len);
    else        hipLaunchKernelGGL((sru_chunk_aggregate<4, 8>), grid, dim3(WAVE), 0, s, U, bias, carries, T, ncols, d, chunk_len);
    if (k == 3) launch_sru_fwd_k3(act, false, 8, grid, s, U, x, bias, c0, h, nullptr, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries);
    else        launch_sru_fwd_k4(act, false, 8, grid, s, U, x, bias, c0, h, nullptr, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries);
    return launch_status("dmn_sru_fwd_scan");
}
