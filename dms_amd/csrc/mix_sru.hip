// dms_amd/csrc/mix_sru.hip -- concat-free mix FUSED into the mSRU layer-0 scan (gfx950, inference).
//
// With comb_conv folded into the layer-0 projection (dmn.py:192 followed by the `x.mm(weight)` of
// the first SRUCell under dmn.py:204) the layer-0 pre-activations are
//     U0[(s*L + t), b, j, g] = A[b,s,4j+g] + Bt[t,b,4j+g] + sum_k Wp[4j+g,k] * P[t,b,k,s]
// dmn_mix_bcast_add writes that tensor (16 B per element, 2.6 GB at B=32 / 512 px) and dmn_sru_fwd
// reads it back once.  Here the scan lane that owns column (b, j) builds its four gate values of a
// step in registers and consumes them on the spot: U0 is never stored.  What is left in memory
// traffic is A (read once), the per-word rows Bt, P (wave-uniform) and the h store.  The kernel
// is bound by one wave's in-order issue, not by HBM (generic variant: Bt rows and P staged in
// LDS, ~380 cycles per step; K = 10 / L = 20 variant: Bt rows in registers, P through
// v_readlane, ~280 cycles per step = 10 readlane + 22 packed FMA/add + the scan's 17 operations),
// and beats the two-kernel path once there are enough columns to occupy the SIMDs (the caller
// decides, see MIX_SCAN_MIN_COLS in dms_amd/models/dmn.py).
//
// The arithmetic repeats dmn_mix_bcast_add's and dmn_sru_fwd's operation for operation (same
// order of the additions and FMAs, same gate formulas), so h is bit-identical to the unfused path.
#include "sru_ring.h"

namespace dmn {

constexpr float MS_LOG2E = 1.4426950408889634f;

template <int I> struct TC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for_t(F &&f)
{
    if constexpr (I < N) {
        f(TC<I>{});
        static_for_t<I + 1, N>(f);
    }
}

// One 64-lane workgroup = 64 adjacent columns j of ONE sample b (so P is wave-uniform).
// Pt: P transposed to (B, S, L, K) so that the L*K values of a pixel are adjacent.
// KX > 0: K == KX exactly (unrolled); KX == 0: runtime K <= 16.
// LX > 0: L == LX exactly -- the word loop is fully unrolled and P never goes through LDS: the
// pixel's L*K values stay spread over the lanes' prefetch registers and each step pulls its K
// values into scalar registers with v_readlane (constant lane index), no memory wait at all.
// WB: how many of the step loop's h stores the per-pixel wait may leave outstanding (a compile-time
// bucket <= L chosen by the host: 20, 8 or 0).  It used to be picked by `if (L >= 20) .. else if (L >= 8)`
// INSIDE the pixel loop; hipcc then copied the prefetch registers into the branch's own registers
// right before the asm wait, i.e. while their loads could still be in flight (check_asm_waits.py).
template <int ACT, int KX, int LX, int WB>
__global__ void __launch_bounds__(WAVE)
mix_sru_fwd_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                   const float *__restrict__ Pt, const float *__restrict__ Wp, int ldw,
                   const float *__restrict__ bias, const float *__restrict__ c0,
                   float *__restrict__ h, float *__restrict__ c_last,
                   int B, int L, int K, int S, int d)
{
    extern __shared__ float4 bt_lds[];                        // [L][64]: this lane's Bt rows
    constexpr int KR = KX > 0 ? KX : 16;
    const int b = blockIdx.y;
    // lanes past the last column stay alive (they carry their share of the P staging below) on the
    // CLAMPED column d-1: same inputs, same arithmetic, so they store the same bits to the same
    // addresses as the lane that owns it -- benign, and no exec masking in the loop
    const int jr = blockIdx.x * WAVE + threadIdx.x;
    const int j = jr < d ? jr : d - 1;
    const int M4 = 4 * d;
    const int col = b * d + j;

    // gate pairs (u0, forget) and (reset, highway) ride one v_pk_fma_f32 each per filter
    f32x2_t w01[KR], w23[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const bool on = KX > 0 || k < K;
        w01[k] = (f32x2_t){on ? Wp[(long long)(4 * j + 0) * ldw + k] : 0.0f, on ? Wp[(long long)(4 * j + 1) * ldw + k] : 0.0f};
        w23[k] = (f32x2_t){on ? Wp[(long long)(4 * j + 2) * ldw + k] : 0.0f, on ? Wp[(long long)(4 * j + 3) * ldw + k] : 0.0f};
    }
    for (int t = 0; t < L; ++t)
        bt_lds[t * WAVE + threadIdx.x] =
            *reinterpret_cast<const float4 *>(Bt + ((long long)t * B + b) * M4 + 4 * j);

    const float bf2 = -MS_LOG2E * bias[j], br2 = -MS_LOG2E * bias[j + d];
    float cur = c0 ? c0[col] : 0.0f;
    const float *pb = Pt + (long long)b * S * L * K;          // uniform
    float *hp = h + col;
    const long long hstride = (long long)B * d;
    // Per pixel the lane needs A[b,s,4j..] (16 B) and the wave needs P[b,s] (L*K values, uniform).
    // Both are fetched one PIXEL ahead.  Two measured dead ends shape how:
    //  * P through scalar loads: SMEM returns out of order, every wait is lgkmcnt(0), the
    //    look-ahead collapses to one step (580 cycles per step);
    //  * compiler-visible vector loads: vmcnt also counts the h stores on gfx9 and hipcc's waits
    //    for the prefetched registers land in the per-step path (`s_waitcnt vmcnt(1)` = wait for
    //    the previous step's store to be acknowledged; again ~580 cycles per step).
    // So, as in sru_fwd.inl, the prefetch loads are inline asm that hipcc does not count and are
    // waited for once per pixel with an exact counted vmcnt(N), N = the h stores issued since.
    // P goes registers -> LDS (padded, unconditional) and is read back as broadcasts.
    const int LK = L * K;
    constexpr int PREG = 4, PBUF = PREG * WAVE;                // LK <= 256 (checked by the host)
    float *p_lds = reinterpret_cast<float *>(bt_lds + (size_t)L * WAVE);     // [2][PBUF]
    unsigned poff[PREG];
#pragma unroll
    for (int i = 0; i < PREG; ++i) {
        const int idx = i * WAVE + (int)threadIdx.x;
        poff[i] = (unsigned)(idx < LK ? idx : LK - 1) * 4u;
    }
    const unsigned aoff = (unsigned)(4 * j) * 4u;
    const float *Ab = A + (long long)b * S * M4;               // uniform bases
    f32x4_t a_nx = {0.0f, 0.0f, 0.0f, 0.0f};
    float pre[PREG] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto fetch_pixel = [&](int sp) {
        const int sc = sp < S ? sp : S - 1;
        const float *asrc = Ab + (long long)sc * M4;
        const float *psrc = pb + (long long)sc * LK;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(a_nx) : "v"(aoff), "s"(asrc) : "memory");
#pragma unroll
        for (int i = 0; i < PREG; ++i)
            asm volatile("global_load_dword %0, %1, %2" : "+v"(pre[i]) : "v"(poff[i]), "s"(psrc) : "memory");
    };
    // wait until at most N younger VMEM operations (the h stores since the fetch) are outstanding
#define DMN_MS_WAIT(N) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a_nx), "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]) : "n"(N) : "memory")
    // LX path: with at most one wave per SIMD occupancy is moot, so the lane's L rows of Bt live in
    // registers (4*LX VGPRs) and the step loop touches no memory but the h store
    float4 btr[LX > 0 ? LX : 1];
    if constexpr (LX > 0) {
#pragma unroll
        for (int t = 0; t < LX; ++t) btr[t] = bt_lds[t * WAVE + threadIdx.x];
    }
    fetch_pixel(0);
    DMN_MS_WAIT(0);
    int buf = 0;
    for (int s = 0; s < S; ++s) {
        // This pixel's A leaves the prefetch registers through asm moves: as a plain copy hipcc kept `ac` in the old
        // registers, gave the next fetch new ones and moved them back at the loop's end -- IN FRONT of the counted wait
        // below, i.e. while that load could still be in flight (K = 10, L != 20 variants; found by
        // tools/check_asm_waits.py once it followed this kernel's control flow).  volatile asm statements keep their
        // order, so a_nx now lives in one register set for the whole loop.
        float4 ac;
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(ac.x), "=&v"(ac.y), "=&v"(ac.z), "=&v"(ac.w)
                     : "v"(a_nx.x), "v"(a_nx.y), "v"(a_nx.z), "v"(a_nx.w));
        if constexpr (LX > 0 && KX > 0) {
            float hold[PREG];                                 // this pixel's P while the next one loads
#pragma unroll
            for (int i = 0; i < PREG; ++i) hold[i] = pre[i];
            fetch_pixel(s + 1);
            float *hs = hp + (long long)s * LX * hstride;
            static_for_t<0, LX>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const float4 bt = btr[t];
                f32x2_t q01 = {ac.x + bt.x, ac.y + bt.y}, q23 = {ac.z + bt.z, ac.w + bt.w};
#pragma unroll
                for (int k = 0; k < KX; ++k) {
                    const int idx = t * KX + k;
                    const float pv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                        __builtin_bit_cast(int, hold[idx / WAVE]), idx % WAVE));
                    q01 = __builtin_elementwise_fma(w01[k], (f32x2_t){pv, pv}, q01);
                    q23 = __builtin_elementwise_fma(w23[k], (f32x2_t){pv, pv}, q23);
                }
                const float f = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(q01.y, -MS_LOG2E, bf2)));
                const float r = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(q23.x, -MS_LOG2E, br2)));
                cur = __builtin_fmaf(cur - q01.x, f, q01.x);
                const float hv = __builtin_fmaf(act_fwd<ACT>(cur) - q23.y, r, q23.y);
                __builtin_nontemporal_store(hv, hs + (long long)t * hstride);
            });
        } else {
#pragma unroll
        for (int i = 0; i < PREG; ++i) p_lds[buf * PBUF + i * WAVE + (int)threadIdx.x] = pre[i];
        fetch_pixel(s + 1);
        const float *pl = p_lds + buf * PBUF;
        float pn[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) pn[k] = (KX > 0 || k < K) ? pl[k] : 0.0f;
        float4 btn = bt_lds[threadIdx.x];
        float *hs = hp + (long long)s * L * hstride;
        auto word = [&](int t) {
            float pv[KR];
#pragma unroll
            for (int k = 0; k < KR; ++k) pv[k] = pn[k];
            const float4 bt = btn;
            const int tn = t + 1 < L ? t + 1 : t;             // next step's operands (last: re-read)
#pragma unroll
            for (int k = 0; k < KR; ++k) pn[k] = (KX > 0 || k < K) ? pl[tn * K + k] : 0.0f;
            btn = bt_lds[tn * WAVE + threadIdx.x];
            // (a sched_barrier here to pin the LDS reads a full step ahead: measured slower, 0.85 vs 0.78 ms)
            f32x2_t q01 = {ac.x + bt.x, ac.y + bt.y}, q23 = {ac.z + bt.z, ac.w + bt.w};
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                if (KX > 0 || k < K) {
                    q01 = __builtin_elementwise_fma(w01[k], (f32x2_t){pv[k], pv[k]}, q01);
                    q23 = __builtin_elementwise_fma(w23[k], (f32x2_t){pv[k], pv[k]}, q23);
                }
            }
            const float q[4] = {q01.x, q01.y, q23.x, q23.y};
            // SRU step (k = 4): q = [u0, forget, reset, highway]
            const float f = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(q[1], -MS_LOG2E, bf2)));
            const float r = fast_rcp(1.0f + fast_exp2(__builtin_fmaf(q[2], -MS_LOG2E, br2)));
            cur = __builtin_fmaf(cur - q[0], f, q[0]);
            const float hv = __builtin_fmaf(act_fwd<ACT>(cur) - q[3], r, q[3]);
            __builtin_nontemporal_store(hv, hs + (long long)t * hstride);
        };
        // the first WB words as a loop of their own with a compile-time trip count: the wait below counts on WB
        // stores behind the fetch, and this way the code itself says so (L >= WB is the host's choice of WB;
        // tools/check_asm_waits.py follows counted loops, not a promise about a run-time L)
        if constexpr (WB > 0) {
#pragma unroll 2
            for (int t = 0; t < WB; ++t) word(t);
        }
#pragma unroll 2
        for (int t = WB; t < L; ++t) word(t);
        }
        // the L stores above are younger than the fetch: leave (at most) WB <= L of them outstanding
        DMN_MS_WAIT(WB);
        buf ^= 1;
    }
#undef DMN_MS_WAIT
    if (c_last) c_last[col] = cur;
}

// ------------------------------------------------------------------------------------------------
// K = 10 filters, L = 20 words (the model's own shape): TWO cooperating waves per 64 columns.
//
// The single-wave kernel above is bound by one wave's in-order issue (~280 cycles per step) while
// half of the 1024 SIMDs idle (512 waves at B = 32).  The cell state needs only the gates (u0,
// forget); the hidden state needs (reset, highway) and act(c_t).  So a 128-thread workgroup runs
//   wave 0 ("state"):  q01 = A.xy + Bt.xy + sum_k Wp01[k] P[t,k]  ->  f, c_t, front half of act(c_t) -> LDS
//                      (tanh = 2 / (1 + 2^(-2c log2 e)) - 1: the exponential here, the reciprocal over there,
//                      three quarter-rate instructions per step on either side instead of four and two)
//   wave 1 ("output"): q23 = A.zw + Bt.zw + sum_k Wp23[k] P[t,k]  ->  r, h_t = (act(c_t) - hw) r + hw -> HBM
// each with half of the packed FMAs (10 instead of 20 per step) and half of the transcendentals.
// The hand-over is one pixel (20 steps) deep: the state wave writes pixel s into LDS buffer s & 1 while
// the output wave consumes pixel s - 1 from the other buffer; ONE s_barrier per pixel.  The two waves
// sit on different SIMDs of one CU.  Every operation is the single-wave kernel's, in the same
// order per gate, so h stays bit-identical to dmn_mix_bcast_add + dmn_sru_fwd.
//
// Store alignment: h rows are d = 1005 floats (4020 B), so a 64-lane store that starts at column
// 64 m of sample b starts at byte (b d + 64 m) * 4 of its time step's row -- on no particular boundary;
// the PMC counters showed +11 % HBM write traffic from the partial 64-byte segments at both ends of
// every store.  When B d is a multiple of 16 floats (every row of h starts 64-byte aligned) the column
// windows of sample b are shifted left by (b d) mod 16 columns, so that every store covers whole
// 64-byte segments (the first window's lanes below column 0 ride along on the clamped column 0).
//
// Workgroup = 4 waves = two independent (state, output) pairs on two adjacent column windows: a CU
// then receives whole workgroups whose four waves go to its four SIMDs, one each.  With 2-wave
// workgroups (two per CU) the second workgroup's waves often landed on the SIMDs the first one already
// occupied: 0.655 ms instead of 0.396 ms at B = 32 (tools/mix_sru_bench.py).
//
// Measured and rejected (round 2): the step's 10 filter responses through LDS broadcast reads (state wave
// publishes the pixel's 200 values, ten ds_read_b128 per group of 4 steps issued one group ahead as uncounted
// inline asm) instead of v_readlane -> SGPR: 30 % fewer VALU issue slots, same 0.43 ms -- the wave is bound
// by its dependent chains, not by issue slots; left to hipcc's own placement the reads cost 0.51 ms.
constexpr int MP_PAIRS = 2;
template <int ACT>
__global__ void __launch_bounds__(2 * MP_PAIRS * WAVE)
mix_sru_pair_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                    const float *__restrict__ Pt, const float *__restrict__ Wp, int ldw,
                    const float *__restrict__ bias, const float *__restrict__ c0,
                    float *__restrict__ h, float *__restrict__ c_last, int B, int S, int d, int shift_on)
{
    constexpr int KX = 10, LX = 20, LK = KX * LX, PREG = 4;
    __shared__ float hand_all[MP_PAIRS][2][LX][WAVE];          // act_front(c_t) of one pixel, double-buffered, per pair
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int role = wave & 1;                                // 0 = state wave, 1 = output wave
    float (*hand)[LX][WAVE] = hand_all[wave >> 1];
    const int win = (int)blockIdx.x * MP_PAIRS + (wave >> 1); // 64-column window of this pair
    const int b = blockIdx.y;
    const int sh = shift_on ? (int)(((long long)b * d) & 15) : 0;
    // lanes (and whole windows) outside [0, d) stay alive on the clamped column (same inputs, same
    // arithmetic, same bits to the same address as the lane that owns it): they carry their share of
    // the P prefetch and every wave meets every barrier
    const int jraw = win * WAVE + lane - sh;
    const int j = jraw < 0 ? 0 : (jraw < d ? jraw : d - 1);
    const int M4 = 4 * d;
    const int col = b * d + j;
    const int g0 = 2 * role;                                  // this wave's gate pair: (0,1) or (2,3)

    f32x2_t w[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k)
        w[k] = (f32x2_t){Wp[(long long)(4 * j + g0) * ldw + k], Wp[(long long)(4 * j + g0 + 1) * ldw + k]};
    f32x2_t btr[LX];
#pragma unroll
    for (int t = 0; t < LX; ++t) {
        const float2 v = *reinterpret_cast<const float2 *>(Bt + ((long long)t * B + b) * M4 + 4 * j + g0);
        btr[t] = (f32x2_t){v.x, v.y};
    }
    const float bg2 = -MS_LOG2E * bias[j + role * d];         // forget bias (state) / reset bias (output)

    // per-pixel prefetch, one PIXEL ahead, inline asm that hipcc does not count (see the kernel above)
    unsigned poff[PREG];
#pragma unroll
    for (int i = 0; i < PREG; ++i) {
        const int idx = i * WAVE + lane;
        poff[i] = (unsigned)(idx < LK ? idx : LK - 1) * 4u;
    }
    const unsigned aoff = (unsigned)(4 * j + g0) * 4u;
    const float *Ab = A + (long long)b * S * M4;
    const float *pb = Pt + (long long)b * S * LK;
    f32x2_t a_nx = {0.0f, 0.0f};
    float pre[PREG] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto fetch_pixel = [&](int sp) {
        const int sc = sp < S ? sp : S - 1;
        const float *asrc = Ab + (long long)sc * M4;
        const float *psrc = pb + (long long)sc * LK;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(a_nx) : "v"(aoff), "s"(asrc) : "memory");
#pragma unroll
        for (int i = 0; i < PREG; ++i)
            asm volatile("global_load_dword %0, %1, %2" : "+v"(pre[i]) : "v"(poff[i]), "s"(psrc) : "memory");
    };
#define DMN_MP_WAIT(N) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a_nx), "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]) : "n"(N) : "memory")
    // LDS traffic of this wave done, then meet the partner (the h stores are NOT waited for)
#define DMN_MP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    // this wave's gate pairs of G consecutive steps, q[i] = A + Bt[t0+i] + sum_k w[k] * P[t0+i, k].
    // A lone wave issues in order, and one step's 10 packed FMAs are ONE dependent chain (the order of
    // the sum over k is part of the bit-identity contract): G independent chains are interleaved so
    // that the chain's latency, not just its issue slots, is covered.  Same for the transcendentals.
#ifndef DMN_MP_G
#define DMN_MP_G 5   // measured at B = 32: G = 2 0.449 ms, 4 0.426, 5 0.412, 10 0.415
#endif
    constexpr int G = DMN_MP_G;
    static_assert(LX % G == 0, "steps are processed in groups of G");
    auto gates = [&](auto gc, const f32x2_t ac, const float (&hold)[PREG], f32x2_t (&q)[G]) {
        constexpr int t0 = decltype(gc)::value * G;
#pragma unroll
        for (int i = 0; i < G; ++i) q[i] = (f32x2_t){ac.x + btr[t0 + i].x, ac.y + btr[t0 + i].y};
        // v_readlane -> SGPR -> v_pk_fma back to back costs an s_nop per FMA (VALU-writes-SGPR hazard):
        // the P values of TWO filters x G steps are pulled into 2G distinct SGPRs first, then consumed
#pragma unroll
        for (int k = 0; k < KX; k += 2) {
            float pv[2][G];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int idx = (t0 + i) * KX + k + kk;
                    pv[kk][i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                        __builtin_bit_cast(int, hold[idx / WAVE]), idx % WAVE));
                }
            __builtin_amdgcn_sched_barrier(0);                // keep the readlanes ahead of their FMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < G; ++i)
                    q[i] = __builtin_elementwise_fma(w[k + kk], (f32x2_t){pv[kk][i], pv[kk][i]}, q[i]);
        }
    };

    fetch_pixel(0);
    DMN_MP_WAIT(0);
    if (role == 0) {
        float cur = c0 ? c0[col] : 0.0f;
        for (int s = 0; s < S; ++s) {
            const f32x2_t ac = a_nx;
            float hold[PREG];
#pragma unroll
            for (int i = 0; i < PREG; ++i) hold[i] = pre[i];
            fetch_pixel(s + 1);
            float *hb = &hand[s & 1][0][lane];
            static_for_t<0, LX / G>([&](auto gc) {
                constexpr int t0 = decltype(gc)::value * G;
                f32x2_t q[G];                                              // (u0, forget)
                gates(gc, ac, hold, q);
                float f[G], cs[G];
#pragma unroll
                for (int i = 0; i < G; ++i) f[i] = fast_exp2(__builtin_fmaf(q[i].y, -MS_LOG2E, bg2));
#pragma unroll
                for (int i = 0; i < G; ++i) f[i] = fast_rcp(1.0f + f[i]);
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    cur = __builtin_fmaf(cur - q[i].x, f[i], q[i].x);
                    cs[i] = cur;
                }
#pragma unroll
                for (int i = 0; i < G; ++i) hb[(t0 + i) * WAVE] = act_front<ACT>(cs[i]);   // tanh: 1 + 2^(-2c log2 e)
            });
            DMN_MP_WAIT(0);                                   // this wave stores nothing to global memory
            DMN_MP_BARRIER();                                 // pixel s is in LDS; the partner has left pixel s - 1
        }
        DMN_MP_BARRIER();                                     // pairs with the partner's last pixel
        if (c_last && jraw == j) c_last[col] = cur;
    } else {
        float *hp = h + col;
        const long long hstride = (long long)B * d;
        DMN_MP_BARRIER();                                     // pixel 0 is in LDS
        for (int s = 0; s < S; ++s) {
            const f32x2_t ac = a_nx;
            float hold[PREG];
#pragma unroll
            for (int i = 0; i < PREG; ++i) hold[i] = pre[i];
            fetch_pixel(s + 1);
            const float *hb = &hand[s & 1][0][lane];
            float av[LX];
#pragma unroll
            for (int t = 0; t < LX; ++t) av[t] = hb[t * WAVE];
            float *hs = hp + (long long)s * LX * hstride;
            static_for_t<0, LX / G>([&](auto gc) {
                constexpr int t0 = decltype(gc)::value * G;
                f32x2_t q[G];                                              // (reset, highway)
                gates(gc, ac, hold, q);
                float r[G];
#pragma unroll
                for (int i = 0; i < G; ++i) r[i] = fast_exp2(__builtin_fmaf(q[i].x, -MS_LOG2E, bg2));
#pragma unroll
                for (int i = 0; i < G; ++i) r[i] = fast_rcp(1.0f + r[i]);
                float ac_t[G];                                             // act(c_t): the partner did the front half
#pragma unroll
                for (int i = 0; i < G; ++i) ac_t[i] = act_back<ACT>(av[t0 + i]);
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const float hv = __builtin_fmaf(ac_t[i] - q[i].y, r[i], q[i].y);
                    __builtin_nontemporal_store(hv, hs + (long long)(t0 + i) * hstride);
                }
            });
            DMN_MP_WAIT(20);                                  // the 20 h stores above are younger than the fetch
            DMN_MP_BARRIER();                                 // done reading pixel s; pixel s + 1 is in LDS
        }
    }
#undef DMN_MP_WAIT
#undef DMN_MP_BARRIER
}

// ------------------------------------------------------------------------------------------------
// K = 10, L = 20: the rank-10 update on the matrix cores (round 2).
//
// In the pair kernel above the 40 FMAs of a step and the 2 x 10 v_readlane that feed them are 60 % of the
// VALU issue slots, and at B = 32 every SIMD already holds a wave: only another pipe can take work away.
// For a block of 32 consecutive steps the pre-activations of one gate of 32 columns are a 32 x 32 tile
//     D[t, c] = (A[s(t), c] + Bt[l(t), c])  +  sum_k P[t, k] * Wp[c, k]
// i.e. five v_mfma_f32_32x32x2_f32 (k = 0..9 in pairs) on top of an accumulator initialised with A + Bt.
// fp32 MFMA adds its two products in k order with a rounding after each (measured: the dynamic-filter
// kernels agree bit for bit), so this continues the SAME fmaf chain as dmn_mix_bcast_add and h stays
// bit-identical to dmn_mix_bcast_add + dmn_sru_fwd.
//
// Roles as in the pair kernel (state wave: u0, forget -> c_t, front half of the activation -> LDS; output
// wave: reset, highway -> h), one block of 32 steps apart, one s_barrier per block.  A wave owns 64 columns
// as two sets X (lanes' columns 0..31 of the window) and Y (32..63) x its 2 gates = four accumulators.
// The MFMA result layout gives lane (n, hf = lane / 32) rows 8 (r / 4) + 4 hf + r % 4 of column n in register
// r: half of the time steps of a column sit in the other half-wave.  One v_permlane32_swap per register
// pair (X_r, Y_r) fixes that: afterwards lane n < 32 owns column X_n, lane n >= 32 column Y_(n-32) -- the
// window's natural column order -- with steps 8 q .. 8 q + 3 in X_(4q ..) and 8 q + 4 .. 8 q + 7 in Y_(4q ..).
// Time is walked in super-blocks of 160 steps (8 pixels = 5 blocks) so that a block's pixel / word of every
// accumulator register is a compile-time constant; the half-wave's +4 step offset becomes a per-lane
// rotation of the word-group index under which the column's Bt rows are read from LDS.
typedef float f32x16m_t __attribute__((ext_vector_type(16)));
template <int ACT>
__global__ void __launch_bounds__(2 * MP_PAIRS * WAVE) __attribute__((amdgpu_waves_per_eu(2, 2)))
mix_sru_mfma_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                    const float *__restrict__ Pt, const float *__restrict__ Wp, int ldw,
                    const float *__restrict__ bias, const float *__restrict__ c0,
                    float *__restrict__ h, float *__restrict__ c_last, int B, int S, int d, int shift_on)
{
    constexpr int KX = 10, LX = 20, BLK = 32, NBLK = 5, PIX = 8;
    __shared__ float hand_all[MP_PAIRS][2][BLK][WAVE];        // act_front(c_t) of one block, double-buffered, per pair
    // The wave's Bt rows (64 columns x 2 gates x 20 words) live in LDS, four words to a 16-byte entry, instead
    // of 80 registers per lane: with them in registers the kernel needs 272 VGPRs, hipcc parks the MFMA tiles
    // in AGPRs and every block pays 128 v_accvgpr moves.  Under 256 (waves_per_eu above) the tiles stay in VGPRs.
    __shared__ float4 bt_all[2 * MP_PAIRS][2][LX / 4][WAVE];  // [wave][gate][word group][window column]
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int role = wave & 1;                                // 0 = state wave, 1 = output wave
    float (*hand)[BLK][WAVE] = hand_all[wave >> 1];
    const int win = (int)blockIdx.x * MP_PAIRS + (wave >> 1);
    const int b = blockIdx.y;
    const int sh = shift_on ? (int)(((long long)b * d) & 15) : 0;
    const int n32 = lane & 31, hf = lane >> 5;
    auto clampj = [&](int jr) { return jr < 0 ? 0 : (jr < d ? jr : d - 1); };
    const int jset[2] = {clampj(win * WAVE + n32 - sh), clampj(win * WAVE + 32 + n32 - sh)};   // X / Y column of lane n32
    const int jraw = win * WAVE + lane - sh;
    const int j = clampj(jraw);                               // the column this lane owns after the swap
    const int M4 = 4 * d;
    const int col = b * d + j;
    const int g0 = 2 * role;

    // B operands: Wp[c, 2 jj + hf] of the lane's X / Y column, both gates
    float wB[2][2][KX / 2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const long long c = 4 * jset[st] + g0 + gg;
#pragma unroll
            for (int jj = 0; jj < KX / 2; ++jj) wB[st][gg][jj] = Wp[c * ldw + 2 * jj + hf];
        }
    // lane l stages window column l (its own column j); read back per set as column st * 32 + n32.  The
    // half-wave hf = 1 holds the tile rows 4 steps later = the NEXT word group (20 words: groups wrap mod 5).
    float4 (*btw)[LX / 4][WAVE] = bt_all[wave];
#pragma unroll
    for (int grp = 0; grp < LX / 4; ++grp) {
        float2 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = *reinterpret_cast<const float2 *>(Bt + ((long long)(4 * grp + e) * B + b) * M4 + 4 * j + g0);
        btw[0][grp][lane] = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
        btw[1][grp][lane] = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
    }
    int gsel[LX / 4];                                         // entry of word group g for this half-wave, set X
#pragma unroll
    for (int g = 0; g < LX / 4; ++g) gsel[g] = ((g + hf) % (LX / 4)) * WAVE + n32;
    __syncthreads();
    const float bg2 = -MS_LOG2E * bias[j + role * d];
    const float *Ab = A + (long long)b * S * M4;
    float cur = (role == 0 && c0) ? c0[col] : 0.0f;
    const long long hstride = (long long)B * d;
    const int hvoff = col * 4, hrowB = B * d * 4;             // bytes (the host checks 32 rows < 2 GiB)
#define DMN_MM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    const long long Ttot = (long long)S * LX;
    if (role == 1) DMN_MM_BARRIER();                          // block 0 of the state wave is in LDS
    int parity = 0;                                           // hand-over buffer of the block being processed
    // One super-block.  TAIL: the last, partial one when S is not a multiple of 8 -- its own copy of the code,
    // run once: pixels / P rows past the end are clamped (their tile rows are never read out), blocks and
    // 4-step groups past the end are skipped by both roles alike (20 steps per pixel: the end is on a group).
    // Both operand streams are fetched one stage ahead -- the block's P rows during the previous block, the
    // super-block's A during the previous super-block -- so their latency passes under the scan instead of in
    // front of the MFMAs.  Rows / pixels past the end are clamped (fetched, never used).
    float pa_nx[KX / 2];
    float2 av_nx[2][PIX];
    const float *prow = Pt + (long long)b * Ttot * KX + hf;
    auto fetch_p = [&](long long T0n) {
        long long tr = T0n + n32;
        tr = tr < Ttot ? tr : Ttot - 1;
#pragma unroll
        for (int jj = 0; jj < KX / 2; ++jj) pa_nx[jj] = prow[tr * KX + 2 * jj];
    };
    auto fetch_a = [&](int sbn) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int px = 0; px < PIX; ++px) {
                int sp = sbn * PIX + px;
                sp = sp < S ? sp : S - 1;
                av_nx[st][px] = *reinterpret_cast<const float2 *>(Ab + (long long)sp * M4 + 4 * jset[st] + g0);
            }
    };
    fetch_a(0);
    fetch_p(0);
    auto super_block = [&](int sb, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value != 0;
        const int nst = TAIL ? (S - sb * PIX) * LX : PIX * LX;                 // steps of this super-block
        // A of the super-block's 8 pixels: the lane's X / Y column, both gates
        float av[2][2][PIX];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int px = 0; px < PIX; ++px) {
                av[st][0][px] = av_nx[st][px].x;
                av[st][1][px] = av_nx[st][px].y;
            }
        if constexpr (!TAIL) fetch_a(sb + 1);
        static_for_t<0, NBLK>([&](auto bc) {
            constexpr int blk = decltype(bc)::value;
            if constexpr (TAIL) {
                if (blk * BLK >= nst) return;
            }
            const long long T0 = (long long)sb * (PIX * LX) + blk * BLK;       // first step of the block
            float pa[KX / 2];
#pragma unroll
            for (int jj = 0; jj < KX / 2; ++jj) pa[jj] = pa_nx[jj];
            fetch_p(T0 + BLK);
            f32x16m_t acc[2][2];
            static_for_t<0, 4>([&](auto qc) {
                constexpr int q4 = decltype(qc)::value;
                constexpr int tq = blk * BLK + 8 * q4;                          // hf = 0 step of register 4 q4 (hf = 1: + 4)
                constexpr int G0 = (tq % LX) / 4;                               // its word group (a multiple of 4 steps in)
                const float4 *bsrc = &btw[0][0][0] + gsel[G0];
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const float4 bt4 = bsrc[gg * (LX / 4) * WAVE + st * 32];
                        const float bte[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int t0r = tq + e, p0 = t0r / LX, p1 = (t0r + 4) / LX;   // p1 <= 7: 159 is the last step
                            const float a = (p0 == p1 || !hf) ? av[st][gg][p0] : av[st][gg][p1];
                            acc[st][gg][4 * q4 + e] = a + bte[e];
                        }
                    }
            });
#pragma unroll
            for (int jj = 0; jj < KX / 2; ++jj)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg)
                        acc[st][gg] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[jj], wB[st][gg][jj], acc[st][gg], 0, 0, 0);
            // half-wave exchange: lanes < 32 keep column X, lanes >= 32 take column Y
            float q[2][BLK];                                  // [gate][step of the block]
            static_for_t<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int i0 = 8 * (r / 4) + r % 4;
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    // (element -> float temporaries first: __builtin_bit_cast applied to a vector-element
                    //  lvalue reads element 0 whatever the index with this compiler)
                    const float xr = acc[0][gg][r], yr = acc[1][gg][r];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xr), __float_as_uint(yr), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    q[gg][i0] = __uint_as_float(s0);
                    q[gg][i0 + 4] = __uint_as_float(s1);
                }
            });
            if (role == 0) {
                float *hb = &hand[parity][0][lane];
                static_for_t<0, BLK / 4>([&](auto gc) {
                    constexpr int i0 = decltype(gc)::value * 4;
                    if constexpr (TAIL) {
                        if (blk * BLK + i0 >= nst) return;
                    }
                    float f[4], cs[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = fast_exp2(__builtin_fmaf(q[1][i0 + i], -MS_LOG2E, bg2));
#pragma unroll
                    for (int i = 0; i < 4; ++i) f[i] = fast_rcp(1.0f + f[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cur = __builtin_fmaf(cur - q[0][i0 + i], f[i], q[0][i0 + i]);
                        cs[i] = cur;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) hb[(i0 + i) * WAVE] = act_front<ACT>(cs[i]);
                });
            } else {
                const float *hb = &hand[parity][0][lane];
                // h rows through a buffer descriptor re-based on the block's first row (uniform): per-lane byte
                // offset of the column + a uniform row offset in an SGPR -- no 64-bit address arithmetic per store
                const rsrc_t rH = make_rsrc(h + T0 * hstride);
                static_for_t<0, BLK / 4>([&](auto gc) {
                    constexpr int i0 = decltype(gc)::value * 4;
                    if constexpr (TAIL) {
                        if (blk * BLK + i0 >= nst) return;
                    }
                    float r[4], ac_t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = fast_exp2(__builtin_fmaf(q[0][i0 + i], -MS_LOG2E, bg2));
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = fast_rcp(1.0f + r[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ac_t[i] = act_back<ACT>(hb[(i0 + i) * WAVE]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float hv = __builtin_fmaf(ac_t[i] - q[1][i0 + i], r[i], q[1][i0 + i]);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), rH, hvoff, (i0 + i) * hrowB, DMN_ST_AUX);
                    }
                });
            }
            parity ^= 1;
            DMN_MM_BARRIER();
        });
    };
    for (int sb = 0; sb < S / PIX; ++sb) super_block(sb, TC<0>{});
    if (S % PIX) super_block(S / PIX, TC<1>{});
    if (role == 0) {
        DMN_MM_BARRIER();                                     // pairs with the partner's last block
        if (c_last && jraw == j) c_last[col] = cur;
    }
#undef DMN_MM_BARRIER
}

}  // namespace dmn

using namespace dmn;

extern "C" int dmn_mix_sru_fwd(const float *A, const float *Bt, const float *Pt, const float *Wp,
                               int ldw, const float *bias, const float *c0, float *h, float *c_last,
                               int B, int L, int K, int S, int d, int act, void *stream)
{
    if (!A || !Bt || !Pt || !Wp || !bias || !h) return fail(-1, "dmn_mix_sru_fwd: NULL pointer");
    if (B <= 0 || L <= 0 || K <= 0 || S <= 0 || d <= 0 || ldw < K)
        return fail(-4, "dmn_mix_sru_fwd: bad sizes B=%d L=%d K=%d S=%d d=%d ldw=%d", B, L, K, S, d, ldw);
    if (K > 16) return fail(-5, "dmn_mix_sru_fwd: K=%d > 16 not supported (use dmn_mix_bcast_add + dmn_sru_fwd)", K);
    if (B > 65535) return fail(-5, "dmn_mix_sru_fwd: B=%d exceeds the grid y limit", B);
    if (((uintptr_t)A | (uintptr_t)Bt) & 15) return fail(-7, "dmn_mix_sru_fwd: A and Bt must be 16-byte aligned");
    if (L * K > 256) return fail(-5, "dmn_mix_sru_fwd: L*K=%d > 256 not supported", L * K);
    const size_t lds = (size_t)L * WAVE * sizeof(float4) + 2 * 4 * WAVE * sizeof(float);
    if (lds > 64 * 1024) return fail(-5, "dmn_mix_sru_fwd: L=%d too large for the LDS stage", L);
    if (act != DMN_ACT_TANH && act != DMN_ACT_RELU && act != DMN_ACT_IDENTITY)
        return fail(-2, "dmn_mix_sru_fwd: unknown activation %d", act);
    hipStream_t s = (hipStream_t)stream;
    // the model's own shape: two cooperating waves per 64 columns (DMN_MIX_SRU_PAIR=0: the single-wave kernel)
    const char *pe = getenv("DMN_MIX_SRU_PAIR");
    if (K == 10 && L == 20 && !(pe && atoi(pe) == 0)) {
        const int shift_on = ((long long)B * d) % 16 == 0 && (d + 15 + WAVE - 1) / WAVE == (d + WAVE - 1) / WAVE;
        const int nwin = (d + (shift_on ? 15 : 0) + WAVE - 1) / WAVE;
        dim3 pgrid((nwin + MP_PAIRS - 1) / MP_PAIRS, B);
        // the rank-10 update on the matrix cores (DMN_MIX_SRU_MFMA=0: the VALU pair
        // kernel).  147 VGPRs and 72 KB of LDS: two workgroups share a CU from B = 64 on (measured at S = 256:
        // 0.354 vs 0.411 ms at B = 32, 0.630 vs 0.760 at B = 64).
        const char *me = getenv("DMN_MIX_SRU_MFMA");
        if (!(((uintptr_t)A | (uintptr_t)Bt) & 7) && (long long)B * d * 4 * 33 < (1LL << 31) && !(me && atoi(me) == 0)) {
#define DMN_MM(ACTV) hipLaunchKernelGGL((mix_sru_mfma_kernel<ACTV>), pgrid, dim3(2 * MP_PAIRS * WAVE), 0, s, A, Bt, Pt, Wp, ldw, \
                                        bias, c0, h, c_last, B, S, d, shift_on)
            if (act == DMN_ACT_TANH) DMN_MM(DMN_ACT_TANH);
            else if (act == DMN_ACT_RELU) DMN_MM(DMN_ACT_RELU);
            else DMN_MM(DMN_ACT_IDENTITY);
#undef DMN_MM
            return launch_status("dmn_mix_sru_fwd");
        }
#define DMN_MP(ACTV) hipLaunchKernelGGL((mix_sru_pair_kernel<ACTV>), pgrid, dim3(2 * MP_PAIRS * WAVE), 0, s, A, Bt, Pt, Wp, ldw, \
                                        bias, c0, h, c_last, B, S, d, shift_on)
        if (act == DMN_ACT_TANH) DMN_MP(DMN_ACT_TANH);
        else if (act == DMN_ACT_RELU) DMN_MP(DMN_ACT_RELU);
        else DMN_MP(DMN_ACT_IDENTITY);
#undef DMN_MP
        return launch_status("dmn_mix_sru_fwd");
    }
    dim3 grid((d + WAVE - 1) / WAVE, B);
#define DMN_MS(ACTV, KXV, LXV, WBV) hipLaunchKernelGGL((mix_sru_fwd_kernel<ACTV, KXV, LXV, WBV>), grid, dim3(WAVE), lds, s, A, Bt, Pt, \
                                                       Wp, ldw, bias, c0, h, c_last, B, L, K, S, d)
#define DMN_MSA(KXV, LXV, WBV)                                                \
    do {                                                                      \
        if (act == DMN_ACT_TANH) DMN_MS(DMN_ACT_TANH, KXV, LXV, WBV);         \
        else if (act == DMN_ACT_RELU) DMN_MS(DMN_ACT_RELU, KXV, LXV, WBV);    \
        else DMN_MS(DMN_ACT_IDENTITY, KXV, LXV, WBV);                         \
    } while (0)
#define DMN_MSW(KXV)                                  \
    do {                                              \
        if (L >= 20) DMN_MSA(KXV, 0, 20);             \
        else if (L >= 8) DMN_MSA(KXV, 0, 8);          \
        else DMN_MSA(KXV, 0, 0);                      \
    } while (0)
    if (K == 10 && L == 20) DMN_MSA(10, 20, 20);      // single-wave variant of the model's default shape
    else if (K == 10) DMN_MSW(10);
    else DMN_MSW(0);
#undef DMN_MSW
#undef DMN_MSA
#undef DMN_MS
    return launch_status("dmn_mix_sru_fwd");
}
