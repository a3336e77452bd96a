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
#include "sru_ring.h"
#include <cmath>
#include <mutex>

namespace dmn {

// forward scan kernels live in sru_fwd.inl (compiled per k in sru_fwd_k3.hip / sru_fwd_k4.hip)
void launch_sru_fwd_k3(int act, bool store_c, int depth, dim3 grid, hipStream_t s, const float *U,
                       const float *x, const float *bias, const float *c0, float *h, float *c_all,
                       float *c_last, int T, int ncols, int d, float scale_x, int h_t0,
                       int chunk_len, const float2 *carries, int blk0 = 0);
void launch_sru_fwd_k4(int act, bool store_c, int depth, dim3 grid, hipStream_t s, const float *U,
                       const float *x, const float *bias, const float *c0, float *h, float *c_all,
                       float *c_last, int T, int ncols, int d, float scale_x, int h_t0,
                       int chunk_len, const float2 *carries, int blk0 = 0);
void launch_sru_fwd_img_k3(int act, int depth, dim3 grid, hipStream_t s, const float *U, const float *x,
                           const float *bias, const float *c0, float *out, float *c_last, int T, int ncols,
                           int d, float scale_x, int h_t0, int chunk_len, const float2 *carries, long long out_bs);
void launch_sru_fwd_img_k4(int act, int depth, dim3 grid, hipStream_t s, const float *U, const float *x,
                           const float *bias, const float *c0, float *out, float *c_last, int T, int ncols,
                           int d, float scale_x, int h_t0, int chunk_len, const float2 *carries, long long out_bs);
void launch_sru_fwd_k2(int depth, dim3 grid, hipStream_t s, const float *U, const float *bias,
                       const float *c0, float *c_all, float *c_last, int T, int ncols, int d, int blk0 = 0);
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
// host side
// ------------------------------------------------------------------------------------
// tuning / test hook: DMN_SRU_DEPTH=<n> forces the prefetch depth (read on every call)
static int depth_override()
{
    const char *e = getenv("DMN_SRU_DEPTH");
    return e ? atoi(e) : -1;
}

// Ring depth of the forward scan.  Measured on MI355X (tools/scan_bench.py depth / depth2, round 2, with
// the stores left out of the counted waits -- sru_ring.h -- so a wait also covers the stores issued
// since the slot's load and a ring is worth about half its slots on a launch that stores every step):
//  * launches that STORE every step (h, and c in training) are HBM-bound on a read/write mix and do
//    best with about 8 MB of ring device-wide, whatever the wave count: 503 waves (B = 32, d = 1005)
//    depth 16 = 74.9 % of 8 TB/s (k = 3) / 72.9 % (k = 4) vs 58 % at depth 8 and 68-70 % at 24-32;
//    1005 waves depth 8 = 73.3 % vs 69.5 % at 16 (training forward with c: 75.8 % k = 3, 71.7 % k = 4);
//    252 waves depth 24-32 = 63.5 %; 2000 waves (configs[4]) depth 3 = 67.0 % vs 65.7 % at 4 and
//    64.4 % at 6 -- more rows in flight only compete with the stores in the memory system;
//  * read-mostly launches (state-only scan, the last layer's tail call) want all the depth they can
//    get: 89 % at depth 32 vs 77 % at 16 and 48 % at 8 on 503 waves.
//  * a wait also covers the stores issued since the slot's load; the k = 4 training scan (ONE load, TWO
//    stores per step: h and c) needs twice the slots for the same loads in flight: 1005 waves depth 16 =
//    73.2 % vs 71.7 % at 8 (the other variants measured best at the unscaled depth).
static int pick_depth(int nwaves, bool stores_every_step, int loads = 2, int stores = 1, double load_bytes = 16.0)
{
    if (!stores_every_step)
        return (nwaves <= 3 * NUM_CU) ? 32 : (nwaves <= 6 * NUM_CU ? 16 : 8);
    if (nwaves >= 6 * NUM_CU) return 3;
    const double per_load = (loads == 1 && stores == 2) ? 2.0 : 1.0;
    const double raw = per_load * 8.0e6 / ((double)nwaves * WAVE * load_bytes);   // bytes loaded per column and step
    static const int cand[] = {3, 4, 6, 8, 16, 24, 32};
    int best = cand[0];
    for (int c : cand)
        if (fabs(c - raw) < fabs(best - raw)) best = c;
    return best;
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
    hipStream_t s = (hipStream_t)stream;
    const bool stores = c_all != nullptr || 2 * (long long)h_t0 < T;
    // A scan that stores every step streams best with about one wave per SIMD and a deep ring (1005 waves:
    // 73 % of 8 TB/s); two waves per SIMD were 67 % at the best depth (3).  So a wide scan (configs[4]: 2000
    // waves) goes out as consecutive column ranges of <= 4 * NUM_CU waves each (DMN_SRU_SPLIT=0: one launch).
    int parts = 1;
    const char *sp = getenv("DMN_SRU_SPLIT");
    if (stores && nwaves >= 6 * NUM_CU && !(sp && atoi(sp) == 0)) parts = (nwaves + 4 * NUM_CU - 1) / (4 * NUM_CU);
    if (sp && atoi(sp) > 1) parts = atoi(sp);                  // (A/B runs: force the number of ranges)
    const int per = (nwaves + parts - 1) / parts;
    for (int blk0 = 0; blk0 < nwaves; blk0 += per) {
        const int nw = nwaves - blk0 < per ? nwaves - blk0 : per;
        int depth = depth_override();
        if (depth <= 0) depth = pick_depth(nw, stores, k == 3 ? 2 : 1, c_all ? 2 : 1);
        dim3 grid(nw);
        if (k == 3) launch_sru_fwd_k3(act, c_all != nullptr, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, WHOLE_SEQUENCE, nullptr, blk0);
        else        launch_sru_fwd_k4(act, c_all != nullptr, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, WHOLE_SEQUENCE, nullptr, blk0);
    }
    return launch_status("dmn_sru_fwd");
}

static int state_fwd_impl(const char *fn, const float *U2, const float *bias, const float *c0, float *c_all,
                          float *c_last, int T, int B, int d, void *stream)
{
    if ((!U2 && T > 0) || !bias || !c_last) return fail(-1, "%s: U2, bias and c_last must not be NULL", fn);
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "%s: bad sizes T=%d B=%d d=%d", fn, T, B, d);
    if ((long long)B * d * 2 * 4 * 33 >= 0x7fffffffLL) return fail(-5, "%s: B*d=%lld too large", fn, (long long)B * d);
    if ((uintptr_t)U2 & 7) return fail(-7, "%s: U2 must be 8-byte aligned", fn);
    const int ncols = B * d;
    const int nwaves = (ncols + WAVE - 1) / WAVE;
    int depth = depth_override();
    // k = 2 loads 8 B per step: twice the slots for the same bytes in flight (1005 waves, c stored: depth 16 =
    // 78.7 % of 8 TB/s vs 60 % at 8; tools/state_train_bench.py)
    if (depth <= 0) depth = pick_depth(nwaves, c_all != nullptr, 1, 1, 8.0);
    launch_sru_fwd_k2(depth, dim3(nwaves), (hipStream_t)stream, U2, bias, c0, c_all, c_last, T, ncols, d);
    return launch_status(fn);
}

extern "C" int dmn_sru_state_fwd(const float *U2, const float *bias, const float *c0,
                                 float *c_last, int T, int B, int d, void *stream)
{
    return state_fwd_impl("dmn_sru_state_fwd", U2, bias, c0, nullptr, c_last, T, B, d, stream);
}

extern "C" int dmn_sru_state_fwd_train(const float *U2, const float *bias, const float *c0, float *c_all,
                                       float *c_last, int T, int B, int d, void *stream)
{
    if (!c_all && T > 0) return fail(-1, "dmn_sru_state_fwd_train: c_all must not be NULL");
    return state_fwd_impl("dmn_sru_state_fwd_train", U2, bias, c0, c_all, c_last, T, B, d, stream);
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

namespace dmn {
// shared by dmn_sru_fwd_scan (caller's workspace) and dmn_sru_fwd_scan_noalloc (library scratch)
static int fwd_scan_impl(const char *fn, const float *U, const float *x, const float *bias,
                         const float *c0, float *h, float *c_last, float2 *carries,
                         int T, int B, int d, int k, int act, float scale_x, int h_t0,
                         int nchunks, hipStream_t s)
{
    const int ncols = B * d;
    const int chunk_len = T > 0 ? (T + nchunks - 1) / nchunks : 1;
    const int used = T > 0 ? (T + chunk_len - 1) / chunk_len : 1;     // chunks that hold steps
    dim3 grid((ncols + WAVE - 1) / WAVE, used);
    if (k == 3) hipLaunchKernelGGL((sru_chunk_aggregate<3, 8>), grid, dim3(WAVE), 0, s, U, bias, carries, T, ncols, d, chunk_len);
    else        hipLaunchKernelGGL((sru_chunk_aggregate<4, 8>), grid, dim3(WAVE), 0, s, U, bias, carries, T, ncols, d, chunk_len);
    if (k == 3) launch_sru_fwd_k3(act, false, 8, grid, s, U, x, bias, c0, h, nullptr, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries);
    else        launch_sru_fwd_k4(act, false, 8, grid, s, U, x, bias, c0, h, nullptr, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries);
    return launch_status(fn);
}

static int check_scan(const char *fn, const float *U, const float *x, const float *bias, float *h,
                      int T, int B, int d, int k, int act, int h_t0, int nchunks)
{
    if (int rc = check_common(fn, U, x, bias, T, B, d, k, act)) return rc;
    if (h_t0 < 0 || h_t0 > T) return fail(-8, "%s: h_t0=%d outside [0, T=%d]", fn, h_t0, T);
    if (!h && h_t0 < T) return fail(-1, "%s: h must not be NULL", fn);
    if (nchunks < 1 || nchunks > 65535) return fail(-9, "%s: nchunks=%d outside [1, 65535]", fn, nchunks);
    return 0;
}
}  // namespace dmn

extern "C" int dmn_sru_fwd_scan(const float *U, const float *x, const float *bias,
                                const float *c0, float *h, float *c_last, float *workspace,
                                int T, int B, int d, int k, int act, float scale_x, int h_t0,
                                int nchunks, void *stream)
{
    if (int rc = check_scan("dmn_sru_fwd_scan", U, x, bias, h, T, B, d, k, act, h_t0, nchunks)) return rc;
    if (!workspace || ((uintptr_t)workspace & 7)) return fail(-1, "dmn_sru_fwd_scan: workspace must be non-NULL, 8-byte aligned");
    return fwd_scan_impl("dmn_sru_fwd_scan", U, x, bias, c0, h, c_last, reinterpret_cast<float2 *>(workspace),
                         T, B, d, k, act, scale_x, h_t0, nchunks, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------
// The scan with its hidden states stored as an image: out[b*out_bs + j*T + t] = h[t, b, j], i.e. the
// (B, d, H, W) map of dmn.py:209-212 when the call scans the last H*W steps of the last mSRU layer -- no
// (T, B, d) tensor, no permute / contiguous copy; out_bs lets the map be the first d channels of the head's
// [x | feat] buffer (dmn.py:337).  nchunks > 1: the time-parallel variant for few columns (workspace as for
// dmn_sru_fwd_scan); nchunks == 1: workspace may be NULL.
// ------------------------------------------------------------------------------------
extern "C" int dmn_sru_fwd_image(const float *U, const float *x, const float *bias, const float *c0,
                                 float *out, float *c_last, float *workspace, int T, int B, int d, int k,
                                 int act, float scale_x, long long out_bs, int nchunks, void *stream)
{
    const char *fn = "dmn_sru_fwd_image";
    if (int rc = check_common(fn, U, x, bias, T, B, d, k, act)) return rc;
    if (!out && T > 0) return fail(-1, "%s: out must not be NULL", fn);
    if (out_bs < (long long)d * T) return fail(-4, "%s: out batch stride %lld < d*T = %lld", fn, out_bs, (long long)d * T);
    if (nchunks < 1 || nchunks > 65535) return fail(-9, "%s: nchunks=%d outside [1, 65535]", fn, nchunks);
    if (nchunks > 1 && (!workspace || ((uintptr_t)workspace & 7)))
        return fail(-1, "%s: nchunks > 1 needs a non-NULL, 8-byte aligned workspace", fn);
    const int ncols = B * d;
    const int nwaves = (ncols + WAVE - 1) / WAVE;
    hipStream_t s = (hipStream_t)stream;
    if (T == 0) {                                               // nothing to scan: c_last = c0
        if (c_last) {
            hipError_t e = c0 ? hipMemcpyAsync(c_last, c0, sizeof(float) * ncols, hipMemcpyDeviceToDevice, s)
                              : hipMemsetAsync(c_last, 0, sizeof(float) * ncols, s);
            if (e != hipSuccess) return fail((int)e, "%s: %s", fn, hipGetErrorString(e));
        }
        return 0;
    }
    int chunk_len = WHOLE_SEQUENCE, used = 1;
    const float2 *carries = nullptr;
    if (nchunks > 1) {
        chunk_len = (T + nchunks - 1) / nchunks;
        used = (T + chunk_len - 1) / chunk_len;
        float2 *cw = reinterpret_cast<float2 *>(workspace);
        dim3 g(nwaves, used);
        if (k == 3) hipLaunchKernelGGL((sru_chunk_aggregate<3, 8>), g, dim3(WAVE), 0, s, U, bias, cw, T, ncols, d, chunk_len);
        else        hipLaunchKernelGGL((sru_chunk_aggregate<4, 8>), g, dim3(WAVE), 0, s, U, bias, cw, T, ncols, d, chunk_len);
        carries = cw;
    }
    int depth = depth_override();
    if (depth <= 0) depth = nchunks > 1 ? 8 : pick_depth(nwaves, true, k == 3 ? 2 : 1, 1);
    dim3 grid(nwaves, used);
    if (k == 3) launch_sru_fwd_img_k3(act, depth, grid, s, U, x, bias, c0, out, c_last, T, ncols, d, scale_x, 0, chunk_len, carries, out_bs);
    else        launch_sru_fwd_img_k4(act, depth, grid, s, U, x, bias, c0, out, c_last, T, ncols, d, scale_x, 0, chunk_len, carries, out_bs);
    return launch_status(fn);
}

// ------------------------------------------------------------------------------------
// dmn_sru_fwd_scan for callers that cannot allocate (HIP graph capture, tight serving loops):
// the chunk-carry workspace lives in a file-scope device array instead of caller memory.
// `scan_scratch` is a HOST-SIDE STAND-IN for the symbol, not a device address: the per-device
// address comes from hipGetSymbolAddress (once per device, that device current) and only that
// pointer is ever handed to a kernel.
// ------------------------------------------------------------------------------------
namespace dmn {
constexpr int SCAN_SCRATCH_FLOATS = DMN_SCAN_SCRATCH_FLOATS;       // 4 MiB on every GPU the library runs on
static __device__ __align__(256) float scan_scratch[SCAN_SCRATCH_FLOATS];
constexpr int MAX_DEVICES = 64;
static std::mutex scratch_mutex;
static float *scratch_dev_ptr[MAX_DEVICES];                        // per device: each GPU holds its own copy

// device address of the CURRENT device's copy of scan_scratch (looked up on the first call per device)
static int scan_scratch_ptr(const char *fn, float **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "%s: hipGetDevice: %s", fn, hipGetErrorString(e));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(-10, "%s: device index %d outside [0, %d)", fn, dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_dev_ptr[dev]) {
        float *p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scan_scratch));
        if (e != hipSuccess || !p)
            return fail(e != hipSuccess ? (int)e : -10, "%s: hipGetSymbolAddress(scan_scratch): %s", fn, hipGetErrorString(e));
        scratch_dev_ptr[dev] = p;
    }
    *out = scratch_dev_ptr[dev];
    return 0;
}
}  // namespace dmn

extern "C" int dmn_sru_fwd_scan_noalloc(const float *U, const float *x, const float *bias,
                                        const float *c0, float *h, float *c_last,
                                        int T, int B, int d, int k, int act, float scale_x, int h_t0,
                                        int nchunks, void *stream)
{
    const char *fn = "dmn_sru_fwd_scan_noalloc";
    if (int rc = check_scan(fn, U, x, bias, h, T, B, d, k, act, h_t0, nchunks)) return rc;
    const long long need = 2LL * nchunks * B * d;
    if (need > SCAN_SCRATCH_FLOATS)
        return fail(-11, "%s: 2*nchunks*B*d = %lld floats exceed the built-in scratch of %d floats "
                         "(use dmn_sru_fwd_scan with a caller workspace, or fewer chunks)", fn, need, SCAN_SCRATCH_FLOATS);
    float *ws = nullptr;
    if (int rc = scan_scratch_ptr(fn, &ws)) return rc;
    // nothing in the scratch has to start at zero: the aggregate kernel writes every (chunk, column)
    // pair before the scan kernel (same stream) reads it
    return fwd_scan_impl(fn, U, x, bias, c0, h, c_last, reinterpret_cast<float2 *>(ws),
                         T, B, d, k, act, scale_x, h_t0, nchunks, (hipStream_t)stream);
}
