// dms_amd/csrc/sru_bwd.hip -- SRU backward scan (reverse time) for gfx950 (MI355X).
//
// Replaces the third-party `sru` package's `SRU_Compute.backward` -> cupy/pynvrtc kernel
// `sru_bwd` (autograd of dmn_pytorch/models/dmn.py:156,204; see include/dmn_hip.h).  Written
// from the SRU equations (arXiv:1709.02755), not from that CUDA source.
//
// Same mapping and the same treatment as the forward scan (sru_fwd.inl): one lane per column
// (b, j), a wave reads one contiguous segment of U / x / c / grad_h per time step, buffer
// addressing with a wave-uniform SRD re-based every DEPTH steps, and a ROLLING register ring of
// DEPTH time steps whose loads are inline asm that hipcc does not count, each slot waited for
// with an exact counted `s_waitcnt vmcnt(N)` (hipcc's own placement drains a ring with vmcnt(0)
// at every loop header).  Per (t, b, j): read 4k (U) + 4 (x, k = 3) + 4 (c) + 4 (grad_h), write
// 4k (grad_U) + 4 (grad_x, k = 3) = 40 B for k = 3 and for k = 4.
//
// k = 2 is the state-only variant (steps whose hidden state nothing reads: [u0 | forget] columns, no grad_h,
// 8 + 4 B read and 8 B written per element; dmn_sru_state_bwd).
//
// Two things differ from the forward:
//  * the step for time t needs c_{t-1}, which arrives with the NEXT ring slot.  The loop is
//    therefore skewed by one step: slot(t) is waited for, its c finishes the PENDING step t+1
//    (grad of the forget gate, the stores of time t+1), then step t is computed up to that point
//    and becomes the pending one.  No slot is ever read before its own counted wait.
//  * the bias gradient sums over time AND batch.  Every lane accumulates its column over time in
//    registers and adds the two sums into grad_bias (2*d) with ONE hardware float atomic each at
//    the end of the kernel (B adders per address, 2*B*d atomics per launch: noise next to the
//    T*B*d*40 bytes streamed), so no (2, B, d) partials leave the kernel.  The order of those B
//    additions is not fixed: grad_bias may differ in the last bit between two launches.
#include "common.h"
#include "sru_ring.h"

namespace dmn {

template <int K> struct BSlot;
template <> struct BSlot<4> { f32x4 u; float c, gh; };
template <> struct BSlot<3> { f32x3 u; float x, c, gh; };
// K = 2: the state-only steps of the last layer in training (no hidden state is read from them, so only
// the cell-state chain carries gradient): U2 = [u0 | forget], no grad_h
template <> struct BSlot<2> { f32x2 u; float c; };

struct BwdSrds { i32x4 U, X, C, G; };

// issue the loads of one time step into a ring slot (NOT waited for; see wait_bslot); "+v"
// destinations for the reason given in sru_fwd.inl (issue_load)
__device__ __forceinline__ void issue_bload(BSlot<4> &s, const BwdSrds r, unsigned offU, unsigned offX,
                                            unsigned sU, unsigned sX)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(r.U), "s"(sU) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.c) : "v"(offX), "s"(r.C), "s"(sX) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.gh) : "v"(offX), "s"(r.G), "s"(sX) : "memory");
}
__device__ __forceinline__ void issue_bload(BSlot<3> &s, const BwdSrds r, unsigned offU, unsigned offX,
                                            unsigned sU, unsigned sX)
{
    asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(r.U), "s"(sU) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.x) : "v"(offX), "s"(r.X), "s"(sX) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.c) : "v"(offX), "s"(r.C), "s"(sX) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.gh) : "v"(offX), "s"(r.G), "s"(sX) : "memory");
}

__device__ __forceinline__ void issue_bload(BSlot<2> &s, const BwdSrds r, unsigned offU, unsigned offX,
                                            unsigned sU, unsigned sX)
{
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.u) : "v"(offU), "s"(r.U), "s"(sU) : "memory");
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" DMN_LD_POLICY : "+v"(s.c) : "v"(offX), "s"(r.C), "s"(sX) : "memory");
}

// wait until at most N younger VMEM operations are outstanding; naming the slot "+v" makes every
// consumer of the slot depend on this statement
template <int N>
__device__ __forceinline__ void wait_bslot(BSlot<4> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s.u), "+v"(s.c), "+v"(s.gh) : "n"(n) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_bslot(BSlot<3> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s.u), "+v"(s.x), "+v"(s.c), "+v"(s.gh) : "n"(n) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_bslot(BSlot<2> &s)
{
    constexpr int n = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.u), "+v"(s.c) : "n"(n) : "memory");
}

// The wide gradient stores go out as inline asm WITH two wait states attached.  gfx950 reads the data
// registers of a store of more than 64 bits late: a VALU instruction issued directly behind the store
// that writes one of them can win.  hipcc pads global_store_dwordx3/x4 for this but assumes a buffer
// store with an SGPR soffset is safe, and its scheduler likes to recycle the data registers in the very
// next instruction.  Seen in round 2 at T = 2000 x 64 320 columns: ~100 wavefronts per launch (the
// fourth wave of a CU, last quad of each 16-lane row) stored the NEXT step's `1 + exp2(..)` in exactly
// the components whose register the next instruction rewrote (tools/bwd_repeat.py); components
// rewritten one instruction later were never hit.  The s_nop inside the statement cannot be moved.
#ifndef DMN_BWD_STORE_NOP
#define DMN_BWD_STORE_NOP "\n\ts_nop 1"
#endif
__device__ __forceinline__ void store_wide(u32x4 o, i32x4 srd, unsigned off, unsigned soff)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" DMN_BWD_STORE_NOP :: "v"(o), "v"(off), "s"(srd), "s"(soff) : "memory");
}
// 64-bit store (K = 2): its data is read with the address, no wait states needed
__device__ __forceinline__ void store_pair(u32x2 o, i32x4 srd, unsigned off, unsigned soff)
{
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen nt" :: "v"(o), "v"(off), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void store_wide(u32x3 o, i32x4 srd, unsigned off, unsigned soff)
{
    asm volatile("buffer_store_dwordx3 %0, %1, %2, %3 offen nt" DMN_BWD_STORE_NOP :: "v"(o), "v"(off), "s"(srd), "s"(soff) : "memory");
}

// what a computed step leaves open until c_{t-1} is known
struct Pending { float gc, u0, dg1, gu0, gg2, gxp; };

template <int K, int ACT, int DEPTH>
__global__ void __launch_bounds__(WAVE)
sru_bwd_cols(const float *__restrict__ U, const float *__restrict__ x,
             const float *__restrict__ bias, const float *__restrict__ c0,
             const float *__restrict__ c_all, const float *__restrict__ grad_h,
             const float *__restrict__ grad_last,
             float *__restrict__ grad_U, float *__restrict__ grad_x,
             float *__restrict__ grad_bias, float *__restrict__ grad_c0,
             int T, int ncols, int d, float scale_x, float *__restrict__ gb_part)
{
    constexpr int NLD = (K == 3) ? 4 : (K == 4 ? 3 : 2);      // VMEM loads per step
    constexpr int NST = DMN_VMCNT_COUNT_STORES ? ((K == 3) ? 2 : 1) : 0;   // VMEM stores per step that a wait may leave outstanding (sru_ring.h)
    const int col = blockIdx.x * WAVE + threadIdx.x;
    if (col >= ncols) return;
    const int j = col % d;
    const float bf2 = -LOG2E * bias[j], br2 = -LOG2E * bias[j + d];
    const float c0v = c0 ? c0[col] : 0.0f;
    const size_t nc = (size_t)ncols;
    const float sx = (K == 3) ? scale_x : 1.0f;
    float cur = grad_last ? grad_last[col] : 0.0f;   // dL/dc flowing back in time
    float gb1 = 0.0f, gb2 = 0.0f;

    if (T > 0) {
        const unsigned strideU = (unsigned)(nc * K * sizeof(float)), strideX = (unsigned)(nc * sizeof(float));
        const unsigned offU = (unsigned)col * K * sizeof(float), offX = (unsigned)col * sizeof(float);
        const char *pU = reinterpret_cast<const char *>(U), *pX = reinterpret_cast<const char *>(x);
        const char *pC = reinterpret_cast<const char *>(c_all), *pG = reinterpret_cast<const char *>(grad_h);
        char *pGU = reinterpret_cast<char *>(grad_U), *pGX = reinterpret_cast<char *>(grad_x);
        BSlot<K> ring[DEPTH] = {};
        Pending pend = {};

        // reverse index r = T-1-t.  Blocks of DEPTH steps; slot i of the block at `done` holds time
        // T-1-done-i.  SRDs are based at the LOWEST time a block touches, slot i sits (DEPTH-1-i)
        // rows above it.  Times below 0 (first and last blocks only) are clamped to row 0: in
        // bounds, value unused, VMEM op count per step unchanged.
        // (the address words go through v_readfirstlane: for k = 3 hipcc folds `base + time * stride`
        // into a v_mad_u64_u32 and then hands the asm's "s" operand a VGPR quad -- "invalid operand")
        auto srd_uniform = [](const char *p) {
            i32x4 r = make_srd(p);
            r.x = __builtin_amdgcn_readfirstlane(r.x);
            r.y = __builtin_amdgcn_readfirstlane(r.y);
            return r;
        };
        auto srds_at = [&](int time) {
            BwdSrds r;
            r.U = srd_uniform(pU + (size_t)time * strideU);
            r.X = srd_uniform(pX + (size_t)time * strideX);
            r.C = srd_uniform(pC + (size_t)time * strideX);
            r.G = srd_uniform(pG + (size_t)time * strideX);
            return r;
        };
        // loads of the DEPTH times hi, hi-1, .., hi-DEPTH+1 (slot i = time hi-i), clamped at 0
        auto load_clamped = [&](BSlot<K> &slot, const BwdSrds r, int base, int hi, int i) {
            int t = hi - i;
            t = t > 0 ? t : 0;
            const unsigned rows = (unsigned)(t - base);
            issue_bload(slot, r, offU, offX, rows * strideU, rows * strideX);
        };

        // finish the pending step (time tp) now that c_{tp-1} = cprev is known, and store its
        // gradients at row offsets sU / sX from the store descriptors
        auto finish = [&](float cprev, i32x4 rGU, rsrc_t rGX, unsigned sU, unsigned sX) {
            const float gg1 = pend.gc * (cprev - pend.u0) * pend.dg1;
            gb1 += gg1;
            if constexpr (K == 2) {
                const u32x2 o = {__float_as_uint(pend.gu0), __float_as_uint(gg1)};
                store_pair(o, rGU, offU, sU);
            } else if (K == 4) {
                const u32x4 o = {__float_as_uint(pend.gu0), __float_as_uint(gg1), __float_as_uint(pend.gg2),
                                 __float_as_uint(pend.gxp)};
                store_wide(o, rGU, offU, sU);
            } else {
                const u32x3 o = {__float_as_uint(pend.gu0), __float_as_uint(gg1), __float_as_uint(pend.gg2)};
                store_wide(o, rGU, offU, sU);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pend.gxp * sx), rGX, offX, sX, DMN_ST_AUX);
            }
        };
        // step t up to the point where c_{t-1} is needed
        auto compute = [&](const BSlot<K> &s) {
            const float g1 = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(s.u.y, -LOG2E, bf2)));
            if constexpr (K == 2) {
                // state-only steps: dL/dc arrives from the later steps alone
                pend.gc = cur;
            } else {
                const float g2 = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(s.u.z, -LOG2E, br2)));
                const float cval = act_fwd<ACT>(s.c);
                float xv;
                if constexpr (K == 3) xv = s.x * sx; else xv = s.u.w;
                // h = (act(c) - x') * g2 + x'
                pend.gxp = s.gh * (1.0f - g2);
                pend.gg2 = s.gh * (cval - xv) * (g2 * (1.0f - g2));
                gb2 += pend.gg2;
                pend.gc = __builtin_fmaf(s.gh * g2, act_bwd<ACT>(cval), cur);
            }
            // c = (c' - u0) * g1 + u0
            pend.gu0 = pend.gc * (1.0f - g1);
            pend.dg1 = g1 * (1.0f - g1);
            // u0 outlives the slot's reload: move it OUT of the slot's register tuple with a real
            // v_mov.  As a plain copy hipcc keeps the old tuple alive instead, reloads the slot into a
            // fresh tuple and rotates the ring through v_movs at the loop's back edge -- copies of
            // registers whose loads are still in flight (caught by tools/check_asm_waits.py).
            asm volatile("v_mov_b32 %0, %1" : "=v"(pend.u0) : "v"(s.u.x));
            cur = pend.gc * g1;
        };

        // prologue: DEPTH steps in flight (times T-1 .. T-DEPTH)
        {
            const int base = T - DEPTH > 0 ? T - DEPTH : 0;
            const BwdSrds r = srds_at(base);
            static_for<0, DEPTH>([&](auto ic) {
                load_clamped(ring[decltype(ic)::value], r, base, T - 1, decltype(ic)::value);
            });
        }

        const int nfull = T / DEPTH;
        int done = 0;
        if (nfull > 0) {
            // first full block: slot i is followed by the rest of the prologue, i reloads and the
            // stores of steps 1 .. i-1 (step 0 has nothing pending)
            {
                const int tl = T - DEPTH;                                   // lowest time of this block (>= 0)
                const i32x4 rGU = srd_uniform(pGU + (size_t)(tl + 1) * strideU);
                const rsrc_t rGX = make_rsrc(pGX + (size_t)(tl + 1) * strideX);
                const int base = tl - DEPTH > 0 ? tl - DEPTH : 0;
                const BwdSrds r = srds_at(base);
                static_for<0, DEPTH>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    wait_bslot<(DEPTH - 1) * NLD + (i > 1 ? i - 1 : 0) * NST>(ring[i]);
                    if constexpr (i > 0)
                        finish(ring[i].c, rGU, rGX, (unsigned)(DEPTH - 1 - i) * strideU, (unsigned)(DEPTH - 1 - i) * strideX);
                    compute(ring[i]);
                    load_clamped(ring[i], r, base, tl - 1, i);
                });
            }
            // steady state (every time loaded is >= 0): slot i is followed by DEPTH-1 full steps
            for (done = DEPTH; done + 2 * DEPTH <= T; done += DEPTH) {
                const int tl = T - done - DEPTH;
                const i32x4 rGU = srd_uniform(pGU + (size_t)(tl + 1) * strideU);
                const rsrc_t rGX = make_rsrc(pGX + (size_t)(tl + 1) * strideX);
                const BwdSrds r = srds_at(tl - DEPTH);
                static_for<0, DEPTH>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    wait_bslot<(DEPTH - 1) * (NLD + NST)>(ring[i]);
                    finish(ring[i].c, rGU, rGX, (unsigned)(DEPTH - 1 - i) * strideU, (unsigned)(DEPTH - 1 - i) * strideX);
                    compute(ring[i]);
                    issue_bload(ring[i], r, offU, offX, (unsigned)(DEPTH - 1 - i) * strideU, (unsigned)(DEPTH - 1 - i) * strideX);
                });
            }
            // last full block (if any is left): same counts, clamped loads
            for (; done + DEPTH <= T; done += DEPTH) {
                const int tl = T - done - DEPTH;
                const i32x4 rGU = srd_uniform(pGU + (size_t)(tl + 1) * strideU);
                const rsrc_t rGX = make_rsrc(pGX + (size_t)(tl + 1) * strideX);
                const int base = tl - DEPTH > 0 ? tl - DEPTH : 0;
                const BwdSrds r = srds_at(base);
                static_for<0, DEPTH>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    wait_bslot<(DEPTH - 1) * (NLD + NST)>(ring[i]);
                    finish(ring[i].c, rGU, rGX, (unsigned)(DEPTH - 1 - i) * strideU, (unsigned)(DEPTH - 1 - i) * strideX);
                    compute(ring[i]);
                    load_clamped(ring[i], r, base, tl - 1, i);
                });
            }
        }
        // last partial block (no loads): at least the rest of the previous block's (or the
        // prologue's) loads follow slot i.  Stores are addressed from row 0 (times < DEPTH + 1).
        {
            const i32x4 rGU = srd_uniform(pGU);
            const rsrc_t rGX = make_rsrc(pGX);
            static_for<0, DEPTH>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                wait_bslot<(DEPTH - 1 - i) * NLD>(ring[i]);
                if (done + i < T) {
                    if (done + i > 0) {
                        const unsigned tp = (unsigned)(T - done - i);       // the pending step's time
                        finish(ring[i].c, rGU, rGX, tp * strideU, tp * strideX);
                    }
                    compute(ring[i]);
                }
            });
            finish(c0v, rGU, rGX, 0u, 0u);                                  // time 0: c_{-1} = c0
        }
    }
    // batch reduction of the bias gradient.  gb_part given: the column's sums go to gb_part (2, ncols) and
    // sru_bwd_bias_reduce adds them over the batch in sample order (what the reference's Python does with its
    // (2, B, d) partials): the same bits on every launch.  NULL: one hardware float atomic per lane and gate
    // (the order of the B additions per address is not fixed).
    if (gb_part) {
        gb_part[col] = gb1;
        if (K != 2) gb_part[(size_t)ncols + col] = gb2;
    } else {
        unsafeAtomicAdd(grad_bias + j, gb1);
        if (K != 2) unsafeAtomicAdd(grad_bias + d + j, gb2);
    }
    grad_c0[col] = cur;
}

// grad_bias[g*d + j] = sum over b (ascending) of part[g*ncols + b*d + j], g < ngates
__global__ void __launch_bounds__(64)
sru_bwd_bias_reduce(const float *__restrict__ part, float *__restrict__ grad_bias, int B, int d, int ngates)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngates * d) return;
    const int g = i / d, j = i - g * d;
    const float *p = part + (size_t)g * B * d + j;
    float acc = 0.0f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {                     // eight loads in flight, added in sample order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(b + q) * d];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; b < B; ++b) acc += p[(size_t)b * d];
    grad_bias[i] = acc;
}

template <int K, int DEPTH>
static void launch_bwd_act(int act, dim3 grid, hipStream_t s, const float *U, const float *x,
                           const float *bias, const float *c0, const float *c_all,
                           const float *grad_h, const float *grad_last, float *grad_U, float *grad_x,
                           float *gb, float *grad_c0, int T, int ncols, int d, float scale_x, float *gb_part)
{
#define DMN_BWD(A)                                                                                 \
    hipLaunchKernelGGL((sru_bwd_cols<K, A, DEPTH>), grid, dim3(WAVE), 0, s, U, x, bias, c0, c_all, \
                       grad_h, grad_last, grad_U, grad_x, gb, grad_c0, T, ncols, d, scale_x, gb_part)
    if (act == DMN_ACT_TANH) DMN_BWD(DMN_ACT_TANH);
    else if (act == DMN_ACT_RELU) DMN_BWD(DMN_ACT_RELU);
    else DMN_BWD(DMN_ACT_IDENTITY);
#undef DMN_BWD
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_sru_state_bwd(const float *U2, const float *bias, const float *c0, const float *c_all,
                                 const float *grad_last, float *grad_U2, float *grad_bias_f, float *grad_c0,
                                 float *bias_partials, int T, int B, int d, void *stream)
{
    const char *fn = "dmn_sru_state_bwd";
    if ((!U2 && T > 0) || !bias || !grad_bias_f || !grad_c0) return fail(-1, "%s: NULL pointer", fn);
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "%s: bad sizes T=%d B=%d d=%d", fn, T, B, d);
    if ((long long)B * d * 2 * 4 * 34 >= 0x7fffffffLL) return fail(-5, "%s: B*d=%lld too large", fn, (long long)B * d);
    if ((!c_all || !grad_U2) && T > 0) return fail(-1, "%s: c_all and grad_U2 must not be NULL", fn);
    if (((uintptr_t)U2 | (uintptr_t)grad_U2) & 7) return fail(-7, "%s: U2 and grad_U2 must be 8-byte aligned", fn);
    const int ncols = B * d;
    hipStream_t s = (hipStream_t)stream;
    if (!bias_partials) {
        hipError_t e = hipMemsetAsync(grad_bias_f, 0, sizeof(float) * (size_t)d, s);
        if (e != hipSuccess) return fail((int)e, "%s: hipMemsetAsync(grad_bias_f): %s", fn, hipGetErrorString(e));
    }
    const char *env = getenv("DMN_SRU_BWD_DEPTH");
    int depth = env ? atoi(env) : 0;
    if (depth <= 0) depth = 8;   // tools/state_train_bench.py, 1005 waves: 76.2 % of 8 TB/s at 8, 71 % at 6, 74 % at 16
    dim3 grid((ncols + WAVE - 1) / WAVE);
    // K = 2 reads bias[j] (forget) only; grad_bias_f has d entries (the reset half has no gradient here)
#define DMN_SB(DV) hipLaunchKernelGGL((sru_bwd_cols<2, DMN_ACT_IDENTITY, DV>), grid, dim3(WAVE), 0, s, U2, (const float *)nullptr, bias, c0, c_all, \
                                      (const float *)nullptr, grad_last, grad_U2, (float *)nullptr, grad_bias_f, grad_c0, T, ncols, d, 1.0f, bias_partials)
    if (depth <= 3) DMN_SB(3);
    else if (depth <= 4) DMN_SB(4);
    else if (depth <= 6) DMN_SB(6);
    else if (depth <= 8) DMN_SB(8);
    else DMN_SB(16);
#undef DMN_SB
    if (bias_partials)
        hipLaunchKernelGGL(sru_bwd_bias_reduce, dim3((d + 63) / 64), dim3(64), 0, s, bias_partials, grad_bias_f, B, d, 1);
    return launch_status(fn);
}

extern "C" int dmn_sru_bwd(const float *U, const float *x, const float *bias, const float *c0,
                           const float *c_all, const float *grad_h, const float *grad_last,
                           float *grad_U, float *grad_x, float *grad_bias, float *grad_c0,
                           float *bias_partials, int T, int B, int d, int k, int act, float scale_x, void *stream)
{
    const char *fn = "dmn_sru_bwd";
    if ((!U && T > 0) || !bias) return fail(-1, "%s: U and bias must not be NULL", fn);
    if (k != 3 && k != 4) return fail(-2, "%s: k must be 3 or 4 (got %d)", fn, k);
    if (k == 3 && !x && T > 0) return fail(-3, "%s: k == 3 needs the highway input x", fn);
    if (k == 4 && x) return fail(-3, "%s: k == 4 takes the highway term from U[..,3]; x must be NULL", fn);
    if (T < 0 || B <= 0 || d <= 0) return fail(-4, "%s: bad sizes T=%d B=%d d=%d", fn, T, B, d);
    // 32-bit row offsets inside a 2 GiB descriptor window: DEPTH + 1 (<= 33) rows of U must fit
    if ((long long)B * d * k * 4 * 34 >= 0x7fffffffLL) return fail(-5, "%s: B*d=%lld too large", fn, (long long)B * d);
    if (act < 0 || act > 2) return fail(-6, "%s: unknown activation %d", fn, act);
    if (((!c_all || !grad_h || !grad_U) && T > 0) || !grad_bias || !grad_c0)
        return fail(-1, "%s: c_all, grad_h, grad_U, grad_bias, grad_c0 must not be NULL", fn);
    if (k == 3 && !grad_x && T > 0) return fail(-3, "%s: k == 3 needs grad_x", fn);
    if (k == 4 && (((uintptr_t)U | (uintptr_t)grad_U) & 15)) return fail(-7, "%s: U and grad_U must be 16-byte aligned for k == 4", fn);
    const int ncols = B * d;
    dim3 grid((ncols + WAVE - 1) / WAVE);
    hipStream_t s = (hipStream_t)stream;
    if (!bias_partials) {
        // the kernel ADDS its per-column sums into grad_bias: start from zero (no allocation, same stream)
        hipError_t e = hipMemsetAsync(grad_bias, 0, sizeof(float) * 2 * (size_t)d, s);
        if (e != hipSuccess) return fail((int)e, "%s: hipMemsetAsync(grad_bias): %s", fn, hipGetErrorString(e));
    }
    // tuning / test hook: DMN_SRU_BWD_DEPTH=<n> forces the ring depth (read on every call)
    const char *env = getenv("DMN_SRU_BWD_DEPTH");
    int depth = env ? atoi(env) : 0;
    // measured (tools/sru_bwd_bench.py, T = 2000, 64 320 columns): depth 4 runs 73.1 % (k = 3) / 72.6 % (k = 4)
    // of 8 TB/s, depth 3 68-69 %, depth 2 54-59 %, depths 6-16 68-69 % / 71 %: the launch is HBM-bound and a
    // deeper ring only adds rows that compete in the memory system
    if (depth <= 0) depth = 4;
#define DMN_BWD_K(KV)                                                                                                                             \
    do {                                                                                                                                          \
        if (depth <= 2) launch_bwd_act<KV, 2>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);        \
        else if (depth <= 3) launch_bwd_act<KV, 3>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);        \
        else if (depth <= 4) launch_bwd_act<KV, 4>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);        \
        else if (depth <= 6) launch_bwd_act<KV, 6>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);   \
        else if (depth <= 8) launch_bwd_act<KV, 8>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);   \
        else if (depth <= 12) launch_bwd_act<KV, 12>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials); \
        else launch_bwd_act<KV, 16>(act, grid, s, U, x, bias, c0, c_all, grad_h, grad_last, grad_U, grad_x, grad_bias, grad_c0, T, ncols, d, scale_x, bias_partials);                  \
    } while (0)
    if (k == 3) DMN_BWD_K(3);
    else        DMN_BWD_K(4);
#undef DMN_BWD_K
    if (bias_partials)
        hipLaunchKernelGGL(sru_bwd_bias_reduce, dim3((2 * d + 63) / 64), dim3(64), 0, s, bias_partials, grad_bias, B, d, 2);
    return launch_status(fn);
}
