// dms_amd/csrc/sru_fwd.inl -- SRU forward scan kernel template + launcher for ONE k.
// Included by sru_fwd_k3.hip / sru_fwd_k4.hip (#define DMN_SRU_K first) so that the two
// sets of instantiations compile in parallel.  See sru.hip for the C ABI.
#include "common.h"
#include "sru_ring.h"

namespace dmn {

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
// With B*d/64 <= 2 waves per CU each wave sits ALONE on a SIMD, where the in-order issue
// of one wave (~4-5 cycles per instruction) bounds the scan as much as HBM does: the first
// version of this kernel spent ~48 instructions and a branch per step (290 cycles/step).
// Hence:
//  (i)   gates folded to fma -> v_exp -> add -> v_rcp; tanh = 2*sigmoid(2c) - 1: 17 VALU/step;
//  (ii)  buffer addressing: SRD base = wave-uniform row pointer re-based every DEPTH steps
//        (2 GiB window, so tensors of any size work), lane byte offset in ONE constant VGPR,
//        time step as an SGPR offset -> no per-step VALU address math;
//  (iii) a ROLLING register ring: hipcc's own s_waitcnt placement drains the ring with
//        vmcnt(0) at every loop header (measured: loads then only overlap one block of
//        compute), so the loads are inline asm that hipcc does not count, and each slot is
//        waited for with an exact counted vmcnt(N) naming the slot's registers "+v"
//        (guide section 5.7 form ii).  N = VMEM ops issued after that slot's load; stores
//        are compiler builtins, ordered against the asm by the "memory" clobber and counted
//        here by hand.
template <int K> struct Slot;
template <> struct Slot<4> { f32x4 u; };
template <> struct Slot<3> { f32x3 u; float x; };
// K = 2: state-only scan over [u0 | forget] pairs -- the steps of the LAST layer whose hidden
// state nobody reads (dmn.py:209 keeps the last H*W steps) need neither the reset gate nor
// the highway term, so their projection is computed for 2 of the k columns only.
template <> struct Slot<2> { f32x2 u; };

// issue the loads of one time step into a ring slot (NOT waited for; see wait_slot).
// The destination is "+v" (read-write), not "=v": tied to the register the slot already
// lives in, so hipcc has no reason to load into a fresh register and v_mov it into the
// loop-carried one BEFORE the data has landed (it did exactly that with "=v" for k = 3).
__device__ __forceinline__ void issue_load(Slot<4> &s, i32x4 srdU, i32x4, unsigned offU, unsigned,
                                           unsigned sU, unsigned)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(srdU), "s"(sU) : "memory");
}
__device__ __forceinline__ void issue_load(Slot<3> &s, i32x4 srdU, i32x4 srdX, unsigned offU,
                                           unsigned offX, unsigned sU, unsigned sX)
{
    asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(srdU), "s"(sU) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.x) : "v"(offX), "s"(srdX), "s"(sX) : "memory");
}

__device__ __forceinline__ void issue_load(Slot<2> &s, i32x4 srdU, i32x4, unsigned offU, unsigned,
                                           unsigned sU, unsigned)
{
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(srdU), "s"(sU) : "memory");
}

// wait until at most N younger VMEM operations are outstanding; naming the slot "+v" makes
// every consumer of the slot depend on this statement
template <int N>
__device__ __forceinline__ void wait_slot(Slot<4> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(s.u) : "n"(n) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_slot(Slot<3> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.u), "+v"(s.x) : "n"(n) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_slot(Slot<2> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(s.u) : "n"(n) : "memory");
}

__device__ __forceinline__ float4 slot_values(const Slot<2> &s) { return make_float4(s.u.x, s.u.y, 0.0f, 0.0f); }
__device__ __forceinline__ float4 slot_values(const Slot<4> &s) { return make_float4(s.u.x, s.u.y, s.u.z, s.u.w); }
__device__ __forceinline__ float4 slot_values(const Slot<3> &s) { return make_float4(s.u.x, s.u.y, s.u.z, s.x); }

// h stored as an IMAGE (dmn.py:209-212: output[-(H*W):] -> permute(1, 2, 0) -> view(B, Hm, H, W)): column
// (b, j) owns the row out[b*out_bs + j*T_img ..+ T_img) and time runs along it.  A lane per column would write
// 4 bytes into 64 different rows per step; instead the wave parks 64 steps x 64 columns in LDS (rows padded to
// 65 floats: both the per-step write and the transposed read are bank-conflict free) and writes them out as
// 64 runs of 64 consecutive floats, lane l carrying time step l of the group.
struct HtOut {
    float *tile;             // LDS, 64 x 65 floats, this wave's
    float *out;              // image base
    unsigned row_lo, row_hi; // this lane's row start b*out_bs + j*T_img (elements), as two 32-bit halves for v_readlane
    int t0;                  // image time index of local step 0 of the range being scanned
    int ncols;               // valid columns of this wave (1 .. 64)
};

// write local steps [tl0, tl0 + cnt) of every valid column of the wave (cnt <= 64)
__device__ __forceinline__ void ht_flush(const HtOut &ht, int tl0, int cnt)
{
    const int lane = threadIdx.x;
    auto put = [&](int r, float v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)ht.row_lo, r);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)ht.row_hi, r);
        float *row = ht.out + (long long)(((unsigned long long)hi << 32) | lo);
        if (lane < cnt) row[ht.t0 + tl0 + lane] = v;
    };
    int r = 0;
    for (; r + 8 <= ht.ncols; r += 8) {                         // eight LDS reads in flight, then eight stores
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ht.tile[lane * 65 + r + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) put(r + i, v[i]);
    }
    for (; r < ht.ncols; ++r) put(r, ht.tile[lane * 65 + r]);
}

struct ScanPtrs {
    const char *ldU, *ldX;   // row of the first step of the range (loads)
    char *stH, *stC;         // row of the first step of the range (stores)
};

// n >= 1 steps starting at the rows in `p`; returns the carried cell state and leaves `p`
// at row n.
template <int K, int ACT, int DEPTH, bool STORE_C, bool STORE_H, bool HT = false>
__device__ __forceinline__ float scan_range(ScanPtrs &p, unsigned strideU, unsigned strideX,
                                            unsigned offU, unsigned offX, int n, float cur,
                                            float bf2, float br2, float sx, const HtOut *ht = nullptr)
{
    constexpr int NLD = (K == 3) ? 2 : 1;                       // VMEM loads per step
    // VMEM stores per step that a wait may leave outstanding: none (sru_ring.h, DMN_VMCNT_COUNT_STORES)
    constexpr int NST = DMN_VMCNT_COUNT_STORES ? ((STORE_H ? 1 : 0) + (STORE_C ? 1 : 0)) : 0;
    Slot<K> ring[DEPTH] = {};
    const ScanPtrs start = p;

    // i: row inside the block (compile-time at every call site); tl: local step inside the range (HT only)
    auto step = [&](const float4 v, rsrc_t rH, rsrc_t rC, int i, int tl) {
        // sigmoid(u + b) = 1 / (1 + 2^(-(u + b) log2 e)), bias pre-multiplied into b?2
        const float f = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(v.y, -LOG2E, bf2)));
        const float r = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(v.z, -LOG2E, br2)));
        cur = __builtin_fmaf(cur - v.x, f, v.x);
        if (STORE_C)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cur), rC, offX, i * strideX, DMN_ST_AUX);
        if (STORE_H) {
            const float xv = (K == 3) ? v.w * sx : v.w;
            const float hv = __builtin_fmaf(act_fwd<ACT>(cur) - xv, r, xv);
            if constexpr (HT) {
                ht->tile[(tl & 63) * 65 + threadIdx.x] = hv;
                if ((tl & 63) == 63) ht_flush(*ht, tl - 63, 64);        // wave-uniform branch
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), rH, offX, i * strideX, DMN_ST_AUX);
            }
        }
    };

    // Loads of one block: SRD base = first row of the block (re-based every block, uniform
    // SALU arithmetic), row i of the block = SGPR offset i*stride.  CLAMP: rows past the end
    // of the range re-read row n-1 (in bounds, value unused) so that the VMEM op count per
    // step never changes; only the first and the last full block need it.
    auto block_srds = [&](int row0, i32x4 &srdU, i32x4 &srdX, unsigned &limU, unsigned &limX) {
        const int base = row0 < n - 1 ? row0 : n - 1;
        const int lim = n - 1 - base;                            // >= 0
        srdU = make_srd(start.ldU + (size_t)base * strideU);
        srdX = make_srd(start.ldX + (size_t)base * strideX);
        limU = (unsigned)lim * strideU;
        limX = (unsigned)lim * strideX;
    };
    auto load_clamped = [&](Slot<K> &slot, i32x4 srdU, i32x4 srdX, unsigned limU, unsigned limX, int i) {
        const unsigned sU = (unsigned)i * strideU, sX = (unsigned)i * strideX;
        issue_load(slot, srdU, srdX, offU, offX, sU < limU ? sU : limU, sX < limX ? sX : limX);
    };

    // prologue: DEPTH steps in flight
    {
        i32x4 srdU, srdX; unsigned limU, limX;
        block_srds(0, srdU, srdX, limU, limX);
        static_for<0, DEPTH>([&](auto ic) {
            load_clamped(ring[decltype(ic)::value], srdU, srdX, limU, limX, decltype(ic)::value);
        });
    }

    const int nfull = n / DEPTH;
    int done = 0;
    if (nfull > 0) {
        // first full block: slot i is followed by the rest of the prologue and by i steps
        {
            const rsrc_t rH = make_rsrc(start.stH), rC = make_rsrc(start.stC);
            i32x4 srdU, srdX; unsigned limU, limX;
            block_srds(DEPTH, srdU, srdX, limU, limX);
            static_for<0, DEPTH>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wait_slot<(DEPTH - 1 - i) * NLD + i * (NLD + NST)>(ring[i]);
                step(slot_values(ring[i]), rH, rC, i, i);
                load_clamped(ring[i], srdU, srdX, limU, limX, i);
            });
        }
        // steady state (every row loaded is inside the range): slot i is followed by
        // DEPTH-1 full steps
        for (done = DEPTH; done + 2 * DEPTH <= n; done += DEPTH) {
            const rsrc_t rH = make_rsrc(start.stH + (size_t)done * strideX);
            const rsrc_t rC = make_rsrc(start.stC + (size_t)done * strideX);
            const i32x4 srdU = make_srd(start.ldU + (size_t)(done + DEPTH) * strideU);
            const i32x4 srdX = make_srd(start.ldX + (size_t)(done + DEPTH) * strideX);
            static_for<0, DEPTH>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wait_slot<(DEPTH - 1) * (NLD + NST)>(ring[i]);
                step(slot_values(ring[i]), rH, rC, i, done + i);
                issue_load(ring[i], srdU, srdX, offU, offX, (unsigned)i * strideU, (unsigned)i * strideX);
            });
        }
        // last full block (if any is left): same counts, clamped rows
        for (; done + DEPTH <= n; done += DEPTH) {
            const rsrc_t rH = make_rsrc(start.stH + (size_t)done * strideX);
            const rsrc_t rC = make_rsrc(start.stC + (size_t)done * strideX);
            i32x4 srdU, srdX; unsigned limU, limX;
            block_srds(done + DEPTH, srdU, srdX, limU, limX);
            static_for<0, DEPTH>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wait_slot<(DEPTH - 1) * (NLD + NST)>(ring[i]);
                step(slot_values(ring[i]), rH, rC, i, done + i);
                load_clamped(ring[i], srdU, srdX, limU, limX, i);
            });
        }
    }
    // last partial block (no loads): at least the rest of the previous block's (or the
    // prologue's) loads follow slot i
    {
        const rsrc_t rH = make_rsrc(start.stH + (size_t)done * strideX);
        const rsrc_t rC = make_rsrc(start.stC + (size_t)done * strideX);
        static_for<0, DEPTH>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            wait_slot<(DEPTH - 1 - i) * NLD>(ring[i]);
            if (done + i < n) step(slot_values(ring[i]), rH, rC, i, done + i);
        });
    }
    if constexpr (HT)
        if (n & 63) ht_flush(*ht, n - (n & 63), n & 63);                // the last, partial group of steps
    p.ldU = start.ldU + (size_t)n * strideU;
    p.ldX = start.ldX + (size_t)n * strideX;
    if (STORE_H && !HT) p.stH = start.stH + (size_t)n * strideX;
    if (STORE_C) p.stC = start.stC + (size_t)n * strideX;
    return cur;
}

template <int K, int ACT, int DEPTH, bool STORE_C>
__global__ void __launch_bounds__(WAVE)
sru_fwd_cols(const float *__restrict__ U, const float *__restrict__ x,
             const float *__restrict__ bias, const float *__restrict__ c0,
             float *__restrict__ h, float *__restrict__ c_all, float *__restrict__ c_last,
             int T_total, int ncols, int d, float scale_x, int h_t0,
             int chunk_len, const float2 *__restrict__ carries, int blk0)
{
    // blk0: first 64-column block of this launch (a wide scan is launched in column ranges, see dmn_sru_fwd)
    const int col = (blk0 + (int)blockIdx.x) * WAVE + threadIdx.x;
    if (col >= ncols) return;
    const int j = col % d;
    const float bf2 = -LOG2E * bias[j], br2 = -LOG2E * bias[j + d];
    float cur = c0 ? c0[col] : 0.0f;
    const size_t nc = (size_t)ncols;
    // time range of this workgroup: the whole sequence (gridDim.y == 1) or one chunk, whose
    // incoming state is the composition of the preceding chunks' (A, B) aggregates
    const int t_begin = blockIdx.y * chunk_len;
    const int T = min(T_total, t_begin + chunk_len);
    if (carries)
        for (int cc = 0; cc < (int)blockIdx.y; ++cc) {
            const float2 ab = carries[(size_t)cc * nc + col];
            cur = __builtin_fmaf(ab.x, cur, ab.y);
        }

    if (T > t_begin) {
        const unsigned strideU = (unsigned)(nc * K * sizeof(float)), strideX = (unsigned)(nc * sizeof(float));
        const unsigned offU = (unsigned)col * K * sizeof(float), offX = (unsigned)col * sizeof(float);
        ScanPtrs p;
        p.ldU = reinterpret_cast<const char *>(U) + (size_t)t_begin * strideU;
        p.ldX = reinterpret_cast<const char *>(x) + (size_t)t_begin * strideX;
        p.stC = reinterpret_cast<char *>(c_all) + (size_t)t_begin * strideX;
        // phase 1: steps before h_t0 produce no h (dmn.py:209 keeps the last H*W steps only)
        const int t_mid = max(t_begin, min(h_t0, T));
        p.stH = reinterpret_cast<char *>(h) + (size_t)(t_mid - h_t0) * strideX;
        if (t_mid > t_begin)
            cur = scan_range<K, ACT, DEPTH, STORE_C, false>(p, strideU, strideX, offU, offX,
                                                            t_mid - t_begin, cur, bf2, br2, scale_x);
        if constexpr (K != 2) {
            if (T > t_mid)
                cur = scan_range<K, ACT, DEPTH, STORE_C, true>(p, strideU, strideX, offU, offX,
                                                               T - t_mid, cur, bf2, br2, scale_x);
        }
    }
    if (c_last && blockIdx.y == gridDim.y - 1) c_last[col] = cur;
}


// The same scan with h stored as an image (HtOut above): the last mSRU layer's tail, whose S hidden states ARE
// the (B, Hm, H, W) map the head reads (dmn.py:209-212).  All 64 lanes stay alive (a lane past the last column
// re-computes column ncols-1: in the write-out every lane carries a time step, not its own column).
template <int K, int ACT, int DEPTH>
__global__ void __launch_bounds__(WAVE)
sru_fwd_cols_img(const float *__restrict__ U, const float *__restrict__ x,
                 const float *__restrict__ bias, const float *__restrict__ c0,
                 float *__restrict__ out, float *__restrict__ c_last,
                 int T_total, int ncols, int d, float scale_x, int h_t0,
                 int chunk_len, const float2 *__restrict__ carries, long long out_bs)
{
    __shared__ float tile[64 * 65];
    const int col0 = (int)blockIdx.x * WAVE;
    const int live = ncols - col0 < WAVE ? ncols - col0 : WAVE;          // >= 1 by the grid size
    const int col = col0 + ((int)threadIdx.x < live ? (int)threadIdx.x : live - 1);
    const int j = col % d;
    const float bf2 = -LOG2E * bias[j], br2 = -LOG2E * bias[j + d];
    float cur = c0 ? c0[col] : 0.0f;
    const size_t nc = (size_t)ncols;
    const int t_begin = blockIdx.y * chunk_len;
    const int T = min(T_total, t_begin + chunk_len);
    if (carries)
        for (int cc = 0; cc < (int)blockIdx.y; ++cc) {
            const float2 ab = carries[(size_t)cc * nc + col];
            cur = __builtin_fmaf(ab.x, cur, ab.y);
        }
    if (T > t_begin) {
        const unsigned strideU = (unsigned)(nc * K * sizeof(float)), strideX = (unsigned)(nc * sizeof(float));
        const unsigned offU = (unsigned)col * K * sizeof(float), offX = (unsigned)col * sizeof(float);
        ScanPtrs p;
        p.ldU = reinterpret_cast<const char *>(U) + (size_t)t_begin * strideU;
        p.ldX = reinterpret_cast<const char *>(x) + (size_t)t_begin * strideX;
        p.stC = nullptr;
        p.stH = nullptr;
        const int t_mid = max(t_begin, min(h_t0, T));
        if (t_mid > t_begin)
            cur = scan_range<K, ACT, DEPTH, false, false>(p, strideU, strideX, offU, offX,
                                                          t_mid - t_begin, cur, bf2, br2, scale_x);
        if (T > t_mid) {
            const unsigned long long row = (unsigned long long)((long long)(col / d) * out_bs
                                                                + (long long)j * (T_total - h_t0));
            HtOut ht;
            ht.tile = tile;
            ht.out = out;
            ht.row_lo = (unsigned)(row & 0xffffffffull);
            ht.row_hi = (unsigned)(row >> 32);
            ht.t0 = t_mid - h_t0;
            ht.ncols = live;
            cur = scan_range<K, ACT, DEPTH, false, true, true>(p, strideU, strideX, offU, offX,
                                                               T - t_mid, cur, bf2, br2, scale_x, &ht);
        }
    }
    if (c_last && blockIdx.y == gridDim.y - 1 && (int)threadIdx.x < live) c_last[col] = cur;
}

template <int K, int ACT>
static void launch_fwd_img(int depth, dim3 grid, hipStream_t s, const float *U, const float *x,
                           const float *bias, const float *c0, float *out, float *c_last, int T, int ncols,
                           int d, float scale_x, int h_t0, int chunk_len, const float2 *carries, long long out_bs)
{
    if (depth <= 8)
        hipLaunchKernelGGL((sru_fwd_cols_img<K, ACT, 8>), grid, dim3(WAVE), 0, s, U, x, bias, c0, out, c_last,
                           T, ncols, d, scale_x, h_t0, chunk_len, carries, out_bs);
    else
        hipLaunchKernelGGL((sru_fwd_cols_img<K, ACT, 16>), grid, dim3(WAVE), 0, s, U, x, bias, c0, out, c_last,
                           T, ncols, d, scale_x, h_t0, chunk_len, carries, out_bs);
}

template <int K, int ACT, bool STORE_C>
static void launch_fwd(int depth, dim3 grid, hipStream_t s, const float *U, const float *x,
                       const float *bias, const float *c0, float *h, float *c_all, float *c_last,
                       int T, int ncols, int d, float scale_x, int h_t0, int chunk_len,
                       const float2 *carries, int blk0)
{
#define DMN_FWD(DEP)                                                                              \
    hipLaunchKernelGGL((sru_fwd_cols<K, ACT, DEP, STORE_C>), grid, dim3(WAVE), 0, s, U, x, bias, \
                       c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0)
    if (depth <= 3) DMN_FWD(3);
    else if (depth <= 4) DMN_FWD(4);
    else if (depth <= 6) DMN_FWD(6);
    else if (depth <= 8) DMN_FWD(8);
    else if (depth <= 16) DMN_FWD(16);
    else if (depth <= 24) DMN_FWD(24);
    else DMN_FWD(32);
#undef DMN_FWD
}

template <int K, bool STORE_C>
static void launch_fwd_act(int act, int depth, dim3 grid, hipStream_t s, const float *U,
                           const float *x, const float *bias, const float *c0, float *h,
                           float *c_all, float *c_last, int T, int ncols, int d, float scale_x,
                           int h_t0, int chunk_len, const float2 *carries, int blk0)
{
    if (act == DMN_ACT_TANH)
        launch_fwd<K, DMN_ACT_TANH, STORE_C>(depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0);
    else if (act == DMN_ACT_RELU)
        launch_fwd<K, DMN_ACT_RELU, STORE_C>(depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0);
    else
        launch_fwd<K, DMN_ACT_IDENTITY, STORE_C>(depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0);
}

#if DMN_SRU_K == 2
// state-only: no h, no c_all, the activation is never applied; the caller passes h_t0 = T
void DMN_SRU_LAUNCH_NAME(int depth, dim3 grid, hipStream_t s, const float *U, const float *bias,
                         const float *c0, float *c_all, float *c_last, int T, int ncols, int d, int blk0)
{
    // c_all != NULL (training): every step's cell state is kept for the backward scan
    if (c_all)
        launch_fwd<2, DMN_ACT_IDENTITY, true>(depth, grid, s, U, nullptr, bias, c0, nullptr, c_all,
                                              c_last, T, ncols, d, 1.0f, T, 0x7fffffff / 2, nullptr, blk0);
    else
        launch_fwd<2, DMN_ACT_IDENTITY, false>(depth, grid, s, U, nullptr, bias, c0, nullptr, nullptr,
                                               c_last, T, ncols, d, 1.0f, T, 0x7fffffff / 2, nullptr, blk0);
}
#else
void DMN_SRU_LAUNCH_IMG_NAME(int act, int depth, dim3 grid, hipStream_t s, const float *U, const float *x,
                             const float *bias, const float *c0, float *out, float *c_last, int T, int ncols,
                             int d, float scale_x, int h_t0, int chunk_len, const float2 *carries, long long out_bs)
{
    if (act == DMN_ACT_TANH)
        launch_fwd_img<DMN_SRU_K, DMN_ACT_TANH>(depth, grid, s, U, x, bias, c0, out, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, out_bs);
    else if (act == DMN_ACT_RELU)
        launch_fwd_img<DMN_SRU_K, DMN_ACT_RELU>(depth, grid, s, U, x, bias, c0, out, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, out_bs);
    else
        launch_fwd_img<DMN_SRU_K, DMN_ACT_IDENTITY>(depth, grid, s, U, x, bias, c0, out, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, out_bs);
}

void DMN_SRU_LAUNCH_NAME(int act, bool store_c, int depth, dim3 grid, hipStream_t s, const float *U,
                         const float *x, const float *bias, const float *c0, float *h,
                         float *c_all, float *c_last, int T, int ncols, int d, float scale_x,
                         int h_t0, int chunk_len, const float2 *carries, int blk0)
{
    if (store_c) launch_fwd_act<DMN_SRU_K, true>(act, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0);
    else         launch_fwd_act<DMN_SRU_K, false>(act, depth, grid, s, U, x, bias, c0, h, c_all, c_last, T, ncols, d, scale_x, h_t0, chunk_len, carries, blk0);
}
#endif

}  // namespace dmn
