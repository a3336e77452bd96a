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
#include "sru_ring.h"
#include <atomic>

namespace dmn {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;

// element offset of row r = t*K + k of an (L, B, K, ld) tensor, batch entry b
__device__ __forceinline__ long long lbk_row(int r, int b, int B, int K, int ld)
{
    const int t = r / K, k = r - t * K;
    return ((long long)(t * B + b) * K + k) * ld;
}


// XCD-aware tile order.  The hardware deals consecutive workgroups to the 8 XCDs round-robin, and each XCD has its
// own L2: with (x, y, b) taken straight from blockIdx the 16 tiles of one sample -- which share its filter rows and
// its feature columns -- land on all 8 XCDs and every L2 fetches the sample's operands for itself (measured at
// B = 32: ~50 TFLOP/s whichever kernel, 700 MB of operand traffic crossing the fabric for 157 MB of tensors).
// With a 1-D grid and tile id v = (block % 8) * chunk + block / 8, XCD k works through the CONSECUTIVE ids
// [k * chunk, (k + 1) * chunk): a sample's tiles meet in one L2.  Returns false for the padding blocks.
__device__ __forceinline__ bool xcd_tile(int nx, int ny, int nz, int &x, int &y, int &z)
{
    const int total = nx * ny * nz, chunk = (total + 7) / 8;
    const int i = (int)blockIdx.x;
    const int v = (i % 8) * chunk + i / 8;
    if (v >= total) return false;
    x = v % nx;
    y = (v / nx) % ny;
    z = v / (nx * ny);
    return true;
}

// ---- forward, vector path: C % 4 == 0, S % 4 == 0, 16-byte aligned bases -------------
// The contraction over C runs in DF_NCH chunks of `spc` K-steps: every chunk is accumulated from zero and
// the chunk sums are added in chunk order.  One workgroup does all chunks of its tile (SPLIT = false), or --
// few samples: 16 tiles per sample cannot fill 256 CUs, round 2 -- one workgroup per (tile, chunk) writes its
// chunk sum to a workspace (ws[chunk][b][r][s]) and dynfilter_reduce adds them in the same order.  Both
// forms give the SAME bits, so a sample's result does not depend on the batch it arrives in.
constexpr int DF_NCH = 16;

template <bool SPLIT>
__global__ void __launch_bounds__(256)
dynfilter_fwd_vec(const float *__restrict__ V, const float *__restrict__ F, float *__restrict__ P,
                  int B, int R, int K, int C, int S, int spc, int nch)
{
    __shared__ float As[BK][BM + PAD];   // As[kk][row]  (transposed on store)
    __shared__ float Bs[BK][BN + PAD];   // Bs[kk][col]
    int bx, by, bz;
    if (!xcd_tile((S + BN - 1) / BN, (R + BM - 1) / BM, SPLIT ? B * nch : B, bx, by, bz)) return;
    const int b = SPLIT ? bz / nch : bz, ch = SPLIT ? bz % nch : 0;
    const int r0 = by * BM, s0 = bx * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float *Vb = V + (long long)b * C * S;
    const int steps = (C + BK - 1) / BK;
    const int st_lo = SPLIT ? ch * spc : 0, st_hi = SPLIT ? min(steps, (ch + 1) * spc) : steps;

    // this lane's slice of the two tiles
    const int ar = tid >> 2, ac = (tid & 3) * 4;     // A: row ar, k-offset ac..ac+3
    const int bk = tid >> 4, bc = (tid & 15) * 4;    // B: k-row bk, col bc..bc+3
    const bool a_ok = (r0 + ar) < R;
    const float *Ap = F + (a_ok ? lbk_row(r0 + ar, b, B, K, C) : 0);
    const bool b_ok = (s0 + bc) < S;                 // S % 4 == 0 -> whole float4 in or out

    // column pairs ride v_pk_fma_f32 (the filter value is broadcast by op_sel): twice the FMA rate
    f32x2_t acc2[4][2], tot2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i][0] = acc2[i][1] = tot2[i][0] = tot2[i][1] = (f32x2_t){0.0f, 0.0f};
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    auto fetch = [&](int c0) {
        av = (a_ok && c0 + ac < C) ? *reinterpret_cast<const float4 *>(Ap + c0 + ac) : make_float4(0, 0, 0, 0);
        bv = (b_ok && c0 + bk < C) ? *reinterpret_cast<const float4 *>(Vb + (long long)(c0 + bk) * S + s0 + bc)
                                   : make_float4(0, 0, 0, 0);
    };
    if (st_lo < st_hi) fetch(st_lo * BK);
    int left = spc;                                  // K-steps left in the current chunk
    for (int st = st_lo; st < st_hi; ++st) {
        As[ac + 0][ar] = av.x; As[ac + 1][ar] = av.y; As[ac + 2][ar] = av.z; As[ac + 3][ar] = av.w;
        *reinterpret_cast<float4 *>(&Bs[bk][bc]) = bv;
        __syncthreads();
        if (st + 1 < st_hi) fetch((st + 1) * BK);
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
        if (!SPLIT && (--left == 0 || st + 1 == st_hi)) {       // chunk boundary: fold the chunk sum in
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tot2[i][0] += acc2[i][0]; tot2[i][1] += acc2[i][1];
                acc2[i][0] = acc2[i][1] = (f32x2_t){0.0f, 0.0f};
            }
            left = spc;
        }
    }
    const int sc = s0 + tx * 4;
    if (sc < S) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty * 4 + i;
            if (r >= R) continue;
            if (SPLIT)
                *reinterpret_cast<float4 *>(P + (((long long)ch * B + b) * R + r) * S + sc) =
                    make_float4(acc2[i][0].x, acc2[i][0].y, acc2[i][1].x, acc2[i][1].y);
            else
                *reinterpret_cast<float4 *>(P + lbk_row(r, b, B, K, S) + sc) =
                    make_float4(tot2[i][0].x, tot2[i][0].y, tot2[i][1].x, tot2[i][1].y);
        }
    }
}

// ---- forward on the matrix cores (round 2) ----------------------------------------------------------------
// Same tiling, chunking and (measured: same) bits as dynfilter_fwd_vec; the 4x4 register tile per lane and
// its LDS reads (two 16-byte reads per 32 packed FMAs: the kernel was LDS-read bound at 33 % of the fp32
// peak) are replaced by v_mfma_f32_32x32x2_f32: each of the 4 waves owns a 32 x 32 quarter of the 64 x 64
// tile, a lane feeds one A and one B scalar per MFMA (row / column lane % 32, k parity lane / 32), read
// from LDS as 16-byte vectors that serve four MFMAs.  fp32 MFMA adds its two products in k order with a
// rounding after each, i.e. it continues the same fmaf chain as the vector kernel.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int MF_ROW = 12;               // BK / 2 = 8 k-pairs per row + padding (16-byte aligned rows)
// Round 3: the kernel was bound by its global -> LDS staging (two barriers and one exposed LDS round trip per
// 8 MFMAs: 47 % of the MFMA issue slots at B = 32).  Now a stage is TWO K-steps (32 columns of C, 16 MFMAs per
// wave), the LDS tiles are double-buffered -- the registers fetched during stage st-1 are written into the
// other buffer at the top of stage st, the loads of stage st+2 are issued, then the MFMAs of stage st run --
// and ONE barrier closes the stage.  The per-K-step order of the fmaf chain and the chunk boundaries are those
// of dynfilter_fwd_vec, so the bits are unchanged (a chunk may end between the two K-steps of a stage; a stage
// that sticks out of the chunk range is fed zeros: + 0 * 0).
constexpr int MF_KS = 2;                 // K-steps per stage

// BMT = 64: four waves, a 64 x 64 tile.  BMT = 32: two waves (128 threads), a 32 x 64 tile -- R = 200 filter rows are
// 6.25 x 32: seven 32-row tiles issue 224 rows of MFMA work where four 64-row tiles issue 256, and a sample has 28
// workgroups instead of 16.
template <bool SPLIT, int BMT>
__global__ void __launch_bounds__(4 * BMT)
dynfilter_fwd_mfma(const float *__restrict__ V, const float *__restrict__ F, float *__restrict__ P,
                   int B, int R, int K, int C, int S, int spc, int nch)
{
    __shared__ __attribute__((aligned(16))) float As2[2][MF_KS][2][BMT][MF_ROW];   // [buffer][K-step][k parity][row][k / 2]
    int bx, by, bz;
    if (!xcd_tile((S + BN - 1) / BN, (R + BMT - 1) / BMT, SPLIT ? B * nch : B, bx, by, bz)) return;
    const int b = SPLIT ? bz / nch : bz, ch = SPLIT ? bz % nch : 0;
    const int r0 = by * BMT, s0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BMT == 64 ? (wave >> 1) * 32 : 0, wn = BMT == 64 ? (wave & 1) * 32 : wave * 32;   // this wave's 32 x 32 part
    const int l32 = lane & 31, half = lane >> 5;
    const float *Vb = V + (long long)b * C * S;
    const int steps = (C + BK - 1) / BK;
    const int st_lo = SPLIT ? ch * spc : 0, st_hi = SPLIT ? min(steps, (ch + 1) * spc) : steps;
    const int c_hi = min(C, st_hi * BK);                        // columns of C past the range read as zero

    const int ar = tid >> 2, ac = (tid & 3) * 4;     // A: row ar, k-offset ac..ac+3 of a K-step
    const bool a_ok = (r0 + ar) < R;

    f32x16_t acc, tot;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0f; tot[i] = 0.0f; }
    // Operand fetches are BUFFER loads whose masks live in the descriptor and the offset, not in branches: as
    // guarded global loads (`ok ? load : 0`) hipcc branched around every load and waited vmcnt(0) right behind it
    // to merge the two values -- a synchronous "prefetch", one exposed L2 round trip per operand and stage.
    //   A (the filters): staged through LDS, 16 bytes per lane and K-step.  Its descriptor ends at column c_hi of
    //   the LAST row of the tile, and a lane past the range of its own row points past the window: zeros.
    //   B (the features): NOT staged.  The MFMA wants lane (n, k parity) to hold V[c0 + 2 kk + parity][column n]:
    //   a coalesced row read, so each wave loads its eight scalars per K-step straight into registers (through
    //   LDS they were transposed with 4-byte writes that hit 8 banks of 64).  The sample's descriptor ends with
    //   its C rows, so the rows of the last, partial K-step read as zeros; rows past a chunk's end (SPLIT) meet
    //   zero filter values.  One v_add per load is all the address arithmetic left (the kernel issued 200 VALU
    //   instructions per 16 MFMAs before: rocprofv3 SQ_INSTS_VALU, profiles/r3_dynfilter.txt).
    const rsrc_t rF = make_rsrc(F);
    const rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Vb), 0, (unsigned)((long long)C * S * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned a_off = a_ok ? (unsigned)(lbk_row(r0 + ar, b, B, K, C) * sizeof(float)) + (unsigned)ac * 4u : OOB;
    const unsigned a_lim = a_ok ? a_off - (unsigned)ac * 4u + (unsigned)c_hi * 4u : 0u;    // first byte past this row's range
    const bool n_ok = (s0 + wn + l32) < S;
    const unsigned b_off = n_ok ? (unsigned)((s0 + wn + l32) * sizeof(float)) + (unsigned)half * (unsigned)S * 4u : OOB;
    const unsigned rowB = (unsigned)S * 4u;
    // Fetch distance: ONE stage, into the register set the stage before last has finished with (two sets, the
    // stage loop unrolled by 2 so that the set index is a compile-time constant).  Two stages ahead (three B sets,
    // loop unrolled by 6) measured SLOWER: 0.1445 vs 0.1326 ms at B = 32, 63.6 vs 78.8 TFLOP/s at B = 64.
    float4 av[MF_KS];
    float bb[2][MF_KS][8];                           // B scalars: landing set / working set alternate per stage
    auto fetch = [&](int c0, auto setc) {            // the MF_KS K-steps starting at column c0, B into set `setc`
        constexpr int set = decltype(setc)::value;
        const unsigned ca = a_off + (unsigned)c0 * 4u;
        const unsigned cbase = b_off + (unsigned)c0 * rowB;
#pragma unroll
        for (int q = 0; q < MF_KS; ++q) {
            const unsigned oa = ca + (unsigned)(q * BK) * 4u;
            av[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rF, oa < a_lim ? oa : OOB, 0, 0));
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                bb[set][q][kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rV, cbase + (unsigned)(q * BK + 2 * kk) * rowB, 0, 0));
        }
    };
    auto stage_to_lds = [&](int buf) {
#pragma unroll
        for (int q = 0; q < MF_KS; ++q) {
            *reinterpret_cast<float2 *>(&As2[buf][q][0][ar][ac / 2]) = make_float2(av[q].x, av[q].z);
            *reinterpret_cast<float2 *>(&As2[buf][q][1][ar][ac / 2]) = make_float2(av[q].y, av[q].w);
        }
    };
    const int nstage = (st_hi - st_lo + MF_KS - 1) / MF_KS;
    if (nstage > 0) {
        fetch(st_lo * BK, IC<0>{});
        stage_to_lds(0);
        if (nstage > 1) fetch((st_lo + MF_KS) * BK, IC<1>{});
    }
    __syncthreads();
    int left = spc, st = st_lo;
    // stage sg computes from LDS buffer sg & 1 and B set sg & 1; behind its MFMAs it writes stage sg+1's A (fetched a
    // stage ago) into the other buffer and fetches stage sg+2 into the set it has just finished with
    auto stage = [&](int sg, auto setc) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int q = 0; q < MF_KS; ++q) {
            if (st < st_hi) {                        // (uniform) the last stage of an odd range has one K-step
                const float4 a0 = *reinterpret_cast<const float4 *>(&As2[set][q][half][wm + l32][0]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&As2[set][q][half][wm + l32][4]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bb[set][q][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bb[set][q][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bb[set][q][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bb[set][q][3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bb[set][q][4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bb[set][q][5], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bb[set][q][6], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bb[set][q][7], acc, 0, 0, 0);
                ++st;
                if (!SPLIT && (--left == 0 || st == st_hi)) {   // chunk boundary: fold the chunk sum in
#pragma unroll
                    for (int i = 0; i < 16; ++i) { tot[i] += acc[i]; acc[i] = 0.0f; }
                    left = spc;
                }
            }
        }
        if (sg + 1 < nstage) {
            stage_to_lds(set ^ 1);
            if (sg + 2 < nstage) fetch((st_lo + (sg + 2) * MF_KS) * BK, setc);
        }
        __syncthreads();
    };
    for (int sg = 0; sg < nstage; sg += 2) {
        stage(sg, IC<0>{});
        if (sg + 1 < nstage) stage(sg + 1, IC<1>{});
    }
    // D layout of 32x32x2: register i of lane (n = lane % 32, h = lane / 32) is row 8 * (i / 4) + 4 * h + i % 4
    const int sc = s0 + wn + l32;
    if (sc < S) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + wm + 8 * (i / 4) + 4 * half + (i % 4);
            if (r >= R) continue;
            if (SPLIT) P[(((long long)ch * B + b) * R + r) * S + sc] = acc[i];
            else P[lbk_row(r, b, B, K, S) + sc] = tot[i];
        }
    }
}

// ---- forward on the matrix cores, operands through an LDS-DMA ring (round 3) -------------------------------------
// Same chunk order and bits as dynfilter_fwd_mfma (tested with torch.equal against every other variant).  There a workgroup has ONE stage of operands
// in flight (registers), so with cold operands the library runs it as (tile, chunk) workgroups plus a reduction
// pass to get enough loads in flight.  Here both operands go global -> LDS directly (buffer_load_dwordx4 .. lds: no
// registers), D - 1 = 3 stages (96 columns of C) ahead of the MFMAs, retired with one counted s_waitcnt vmcnt per stage
// and one barrier; every wave issues the same IPW loads per stage (stages past the end are fetched out of bounds:
// zeros into a free slot), so the count is a constant.
// LDS layout of a stage, chosen so that both the DMA writes (lane i -> byte 16 i of a 1 KiB piece) and the MFMA operand
// reads are conflict-free:
//   A: [K-step][row][4 chunks of 4 k], chunk c of row r at position c ^ ((r >> 2) & 3);
//   B: [32 k][16 chunks of 4 columns], chunk c of row k at position c ^ (8 * (k & 1)).

__device__ __forceinline__ void df_dma_b128(unsigned lds_base, i32x4 srd, unsigned voff, unsigned soff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
}

template <int BMT, int D>
__global__ void __launch_bounds__(4 * BMT)
dynfilter_fwd_dma(const float *__restrict__ V, const float *__restrict__ F, float *__restrict__ P,
                  int B, int R, int K, int C, int S, int spc)
{
    constexpr int NWV = BMT / 16;                       // waves per workgroup: 2 or 4
    constexpr int A_INS = BMT / 8, B_INS = 8;           // 1 KiB DMA pieces per stage (A: BMT rows x 32 k, B: 32 k x 64 columns)
    constexpr int IPW = (A_INS + B_INS) / NWV;          // pieces per wave and stage: 6 or 4
    constexpr int STAGE = (A_INS + B_INS) * 1024;       // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char df_lds[];     // [D][STAGE]
    int bx, by, bz;
    if (!xcd_tile((S + BN - 1) / BN, (R + BMT - 1) / BMT, B, bx, by, bz)) return;
    const int b = bz, r0 = by * BMT, s0 = bx * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BMT == 64 ? (wave >> 1) * 32 : 0, wn = BMT == 64 ? (wave & 1) * 32 : wave * 32;
    const int l32 = lane & 31, half = lane >> 5;
    const int steps = (C + BK - 1) / BK;
    const int nstage = (steps + 1) / 2;
    constexpr unsigned OOB = 0x80000000u;

    // this wave's pieces: i = wave + NWV * n; per-lane source offsets are loop-invariant, the stage moves in soffset
    i32x4 srdF = make_srd(F);
    srdF.z = (int)((long long)R * B * C * 4);            // the last rows' look-ahead past the tensor reads zeros, not memory
    i32x4 srdV = make_srd(V + (long long)b * C * S);
    srdV.z = (int)((long long)C * S * 4);                // rows past C read as zeros
    constexpr int NA = A_INS / NWV;                      // a wave's first NA pieces are A's (i = wave + NWV n < A_INS), the rest B's
    unsigned voff[IPW];
#pragma unroll
    for (int n = 0; n < IPW; ++n) {
        const int i = wave + NWV * n;                    // wave-uniform
        if (n < NA) {
            const int slot = 64 * i + lane, q = slot / (BMT * 4), rem = slot - q * (BMT * 4);
            const int row = rem >> 2, cp = rem & 3, c = cp ^ ((row >> 2) & 3);
            voff[n] = (r0 + row) < R ? (unsigned)(lbk_row(r0 + row, b, B, K, C) * sizeof(float)) + (unsigned)(q * 16 + c * 4) * 4u : OOB;
        } else {
            const int slot = 64 * (i - A_INS) + lane, k = slot >> 4, cp = slot & 15, c = cp ^ (8 * (k & 1));
            voff[n] = (s0 + c * 4) < S ? (unsigned)(k * S + s0 + c * 4) * 4u : OOB;
        }
    }
    const unsigned lds0 = (unsigned)(unsigned long long)df_lds;
    auto issue = [&](int st) {                           // stage st -> ring slot st % D (st >= nstage: zeros)
        const bool on = st < nstage;
        const unsigned base = lds0 + (unsigned)(st % D) * STAGE;
        const unsigned sA = on ? (unsigned)st * 32u * 4u : 0u, sB = on ? (unsigned)st * 32u * (unsigned)S * 4u : 0u;
#pragma unroll
        for (int n = 0; n < IPW; ++n) {
            const int i = wave + NWV * n;
            if (n < NA) df_dma_b128(base + (unsigned)i * 1024u, srdF, on ? voff[n] : OOB, sA);
            else df_dma_b128(base + (unsigned)i * 1024u, srdV, on ? voff[n] : OOB, sB);
        }
    };
#pragma unroll
    for (int st = 0; st < D - 1; ++st) issue(st);

    // MFMA operand addresses (bytes inside a stage)
    const int arow = wm + l32;
    unsigned aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (unsigned)((arow * 4 + (c ^ ((arow >> 2) & 3))) * 16);
    const int n = wn + l32;
    const unsigned boff = (unsigned)(A_INS * 1024 + half * 256 + ((((n >> 2) ^ (8 * half)) * 4 + (n & 3)) * 4));

    f32x16_t acc, tot;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0f; tot[i] = 0.0f; }
    int left = spc, stp = 0;
    for (int st = 0; st < nstage; ++st) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IPW * (D - 2)) : "memory");   // this wave's pieces of stage st have landed
        __syncthreads();                                 // everyone's have; everyone is done with stage st - 1
        issue(st + D - 1);                               // into the slot of stage st - 1
        const unsigned char *sb = df_lds + (st % D) * STAGE;
        const bool up = half != 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (stp < steps) {                           // (uniform) C's last stage may hold one K-step
                float4 a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const float4 *>(sb + q * (BMT * 64) + aoff[c]);
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = *reinterpret_cast<const float *>(sb + boff + (q * 16 + 2 * j) * 256);
                // k = 2 j + half: the lower half-wave takes elements x / z of a chunk, the upper one y / w (v_cndmask; an
                // array indexed by `half` went to scratch memory, whose loads hipcc waits for with vmcnt(0)).
                // (Reading both K-steps' operands up front measured slower: 0.126 vs 0.118 ms at B = 32, cold.)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float e0 = up ? a[c].y : a[c].x, e1 = up ? a[c].w : a[c].z;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, bv[2 * c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e1, bv[2 * c + 1], acc, 0, 0, 0);
                }
                ++stp;
                if (--left == 0 || stp == steps) {       // chunk boundary: fold the chunk sum in (dynfilter_fwd_vec's order)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { tot[i] += acc[i]; acc[i] = 0.0f; }
                    left = spc;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead pieces past the end
    const int sc = s0 + wn + l32;
    if (sc < S) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + wm + 8 * (i / 4) + 4 * half + (i % 4);
            if (r < R) P[lbk_row(r, b, B, K, S) + sc] = tot[i];
        }
    }
}

// P[t,b,k,s] = sum over chunks (in chunk order, starting from +0) of ws[chunk][b][r][s]
__global__ void __launch_bounds__(256)
dynfilter_reduce(const float *__restrict__ ws, float *__restrict__ P, int B, int R, int K, int S, int nch)
{
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;      // float4 index into (B, R, S)
    const long long per = (long long)B * R * S;
    if (i4 * 4 >= per) return;
    float4 tot = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int ch = 0; ch < nch; ++ch) {
        const float4 v = *reinterpret_cast<const float4 *>(ws + ch * per + i4 * 4);
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
    }
    const long long e = i4 * 4;
    const int sc = (int)(e % S);
    const long long br = e / S;
    const int r = (int)(br % R), b = (int)(br / R);
    *reinterpret_cast<float4 *>(P + lbk_row(r, b, B, K, S) + sc) = tot;
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

// ---- backward, vector paths (round 2): C % 4 == 0, S % 4 == 0, 16-byte aligned bases -------------------
// The generic kernel above fetches every tile element with its own guarded 4-byte load (and a division
// for the (t, k) row address); these read 16 bytes per lane along the contiguous axis and ride
// v_pk_fma_f32 like the forward.  MODE 1: gF[r,c] = sum_s gP[r,s] V[c,s] (both operands s-contiguous:
// transposed on the way into LDS).  MODE 2: gV[c,s] = sum_r F[r,c] gP[r,s] (both operands contiguous
// along the tile axes: straight 16-byte copies into LDS).
template <int MODE>
__global__ void __launch_bounds__(256)
dynfilter_bwd_vec(const float *__restrict__ V, const float *__restrict__ F, const float *__restrict__ gP,
                  float *__restrict__ O, int B, int R, int K, int C, int S)
{
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];
    const int Md = (MODE == 2) ? C : R, Nd = (MODE == 1) ? C : S, Kd = (MODE == 1) ? S : R;
    int bx, by, b;
    if (!xcd_tile((Nd + BN - 1) / BN, (Md + BM - 1) / BM, B, bx, by, b)) return;
    const int m0 = by * BM, n0 = bx * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float *Vb = V + (long long)b * C * S;

    f32x2_t acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i][0] = acc2[i][1] = (f32x2_t){0.0f, 0.0f};
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    // MODE 1: lane -> (row tid >> 2, k-offset (tid & 3) * 4) of both operands (k-contiguous)
    // MODE 2: lane -> (k-row tid >> 4, tile offset (tid & 15) * 4) of both operands (tile-contiguous)
    const int rr = (MODE == 1) ? tid >> 2 : tid >> 4, off = (MODE == 1) ? (tid & 3) * 4 : (tid & 15) * 4;
    // branch-free fetches: buffer loads whose mask is the offset (outside -> past the 2 GiB window -> zeros); as
    // guarded global loads every fetch was a branch plus an immediate vmcnt(0) (see dynfilter_fwd_mfma)
    const rsrc_t rV = make_rsrc(Vb), rF = make_rsrc(F), rG = make_rsrc(gP);
    constexpr unsigned OOB = 0x80000000u;
    const bool a_row_ok = (MODE == 1) ? (m0 + rr) < R : true;
    const unsigned arow_off = (MODE == 1 && a_row_ok) ? (unsigned)(lbk_row(m0 + rr, b, B, K, S) * sizeof(float)) : 0u;
    auto ld = [&](rsrc_t r, bool ok, unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : OOB, 0, 0));
    };
    auto fetch = [&](int k0) {
        if (MODE == 1) {
            av = ld(rG, a_row_ok && k0 + off < S, arow_off + (unsigned)(k0 + off) * 4u);
            bv = ld(rV, (n0 + rr) < C && k0 + off < S, (unsigned)(((long long)(n0 + rr) * S + k0 + off) * 4));
        } else {
            const int r = k0 + rr;
            const int rc = r < R ? r : 0;
            av = ld(rF, r < R && m0 + off < C, (unsigned)((lbk_row(rc, b, B, K, C) + m0 + off) * 4));
            bv = ld(rG, r < R && n0 + off < S, (unsigned)((lbk_row(rc, b, B, K, S) + n0 + off) * 4));
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Kd; k0 += BK) {
        if (MODE == 1) {
            As[off + 0][rr] = av.x; As[off + 1][rr] = av.y; As[off + 2][rr] = av.z; As[off + 3][rr] = av.w;
            Bs[off + 0][rr] = bv.x; Bs[off + 1][rr] = bv.y; Bs[off + 2][rr] = bv.z; Bs[off + 3][rr] = bv.w;
        } else {
            *reinterpret_cast<float4 *>(&As[rr][off]) = av;
            *reinterpret_cast<float4 *>(&Bs[rr][off]) = bv;
        }
        __syncthreads();
        if (k0 + BK < Kd) fetch(k0 + BK);
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
    const int n = n0 + tx * 4;
    if (n < Nd) {                                    // Nd % 4 == 0: whole float4 in or out
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty * 4 + i;
            if (m >= Md) continue;
            float *dst = (MODE == 1) ? O + lbk_row(m, b, B, K, C) + n : O + ((long long)b * C + m) * S + n;
            *reinterpret_cast<float4 *>(dst) = make_float4(acc2[i][0].x, acc2[i][0].y, acc2[i][1].x, acc2[i][1].y);
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

// K-steps per chunk / number of non-empty chunks of the chunked contraction (see dynfilter_fwd_vec)
static void df_chunks(int C, int *spc, int *nch)
{
    const int steps = (C + BK - 1) / BK;
    *spc = (steps + DF_NCH - 1) / DF_NCH;
    *nch = (steps + *spc - 1) / *spc;
}

extern "C" int dmn_dynfilter_workspace_floats(int B, int L, int K, int C, int S)
{
    int spc, nch;
    if (B <= 0 || L <= 0 || K <= 0 || C <= 0 || S <= 0) return 0;
    df_chunks(C, &spc, &nch);
    const long long n = (long long)nch * B * L * K * S;
    return n > 0x7fffffffLL ? 0 : (int)n;           // 0: too large, call without a workspace
}

static int dynfilter_fwd_impl(const char *fn, const float *V, const float *F, float *P, float *workspace,
                              int B, int L, int K, int C, int S, void *stream)
{
    if (!V || !F || !P) return fail(-1, "%s: NULL pointer", fn);
    if (int rc = check_sizes(fn, B, L, K, C, S)) return rc;
    const int R = L * K;
    dim3 grid((S + BN - 1) / BN, (R + BM - 1) / BM, B);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (C % 4 == 0) && (S % 4 == 0) &&
                     !(((uintptr_t)V | (uintptr_t)F | (uintptr_t)P) & 15);
    if (!vec) {
        hipLaunchKernelGGL(dynfilter_generic<0>, grid, dim3(256), 0, s, V, F, (const float *)nullptr, P, B, R, K, C, S);
        return launch_status(fn);
    }
    int spc, nch;
    df_chunks(C, &spc, &nch);
    const char *me = getenv("DMN_DYNFILTER_MFMA");              // 0: the packed-FMA kernels (A/B runs)
    // (the matrix-core kernel addresses F through one 2 GiB buffer window)
    const bool mfma = !(me && atoi(me) == 0) && (long long)L * B * K * C * 4 < 0x7fffffffLL && (long long)C * S * 4 < 0x7fffffffLL;
    // matrix-core kernel: 32-row tiles when they pad R less than 64-row ones (R = 200: 224 vs 256 rows of MFMA work, 28
    // workgroups per sample instead of 16: 63 -> 71 TFLOP/s at B = 32, 75 -> 83 at B = 64; DMN_DYNFILTER_BM=64|32 forces)
    const char *be = getenv("DMN_DYNFILTER_BM");
    const int bmt = !mfma ? 64 : be ? (atoi(be) == 32 ? 32 : 64) : (((R + 31) / 32) * 32 < ((R + 63) / 64) * 64 ? 32 : 64);
    // one workgroup per (tile, chunk) below 4 64 x 64 tiles per CU.  Measured with COLD operands (tools/dynfilter_cold.py: a
    // 1 GB fill between calls, as inside the step): B = 32: whole 0.176 / split 0.157 ms, B = 64: 0.232 / 0.231 ms.  The
    // repeat-the-same-call microbench keeps the 157 MB of operands in the Infinity Cache and prefers whole tiles at B = 32
    // (0.123 vs 0.129 ms) -- inside the step that choice ran 0.153 ms against 0.137 ms for split.
    const char *se = getenv("DMN_DYNFILTER_SPLIT");
    const long long tiles = (long long)grid.x * ((R + bmt - 1) / bmt) * B;
    bool split = workspace && nch > 1 && !((uintptr_t)workspace & 15) && (long long)B * nch <= 65535 &&
                 (se ? atoi(se) != 0 : (long long)grid.x * grid.y * B < 4 * NUM_CU);
    auto xcd_grid = [](long long tiles) { return dim3((unsigned)(((tiles + 7) / 8) * 8)); };
    if (tiles * (split ? nch : 1) >= (1LL << 30)) return fail(-5, "%s: too many tiles", fn);
    // operands through an LDS-DMA ring, whole contraction per 64 x 64 tile, no workspace / reduction pass: from one tile per
    // CU up to 8 (cold operands, tools/dynfilter_cold.py: B = 16 0.076 vs 0.084 ms, B = 32 0.118 vs 0.153, B = 64 0.226 vs
    // 0.229, training shape B = 64, S = 100 0.121 vs 0.165; below that the (tile, chunk) split wins, above it 32-row whole
    // tiles do: B = 128 0.430 vs 0.441).  DMN_DYNFILTER_DMA=0|1 forces it off / on.
    const char *de = getenv("DMN_DYNFILTER_DMA");
    const long long tiles64 = (long long)grid.x * grid.y * B;
    const bool dma_ok = mfma && (long long)C * S * 4 + 64LL * S * 4 < 0x7fffffffLL && tiles64 < (1LL << 30);
    if (dma_ok && (de ? atoi(de) == 1 : (tiles64 >= NUM_CU && tiles64 < 8 * NUM_CU))) {
        constexpr int D = 4;
        const int ldsb = D * (64 / 8 + 8) * 1024;
        // 64 KB of dynamic LDS: opt in once per device (not per launch: a launch may be under stream capture)
        static std::atomic<unsigned long long> opted{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return fail((int)e, "%s: hipGetDevice: %s", fn, hipGetErrorString(e));
        if (dev >= 64 || !(opted.load() >> dev & 1ULL)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(dynfilter_fwd_dma<64, D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
            if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
            if (dev < 64) opted.fetch_or(1ULL << dev);
        }
        hipLaunchKernelGGL((dynfilter_fwd_dma<64, D>), xcd_grid(tiles64), dim3(256), ldsb, s, V, F, P, B, R, K, C, S, spc);
        return launch_status(fn);
    }
    if (split) {
        dim3 sgrid = xcd_grid(tiles * nch);
        if (mfma && bmt == 32) hipLaunchKernelGGL((dynfilter_fwd_mfma<true, 32>), sgrid, dim3(128), 0, s, V, F, workspace, B, R, K, C, S, spc, nch);
        else if (mfma) hipLaunchKernelGGL((dynfilter_fwd_mfma<true, 64>), sgrid, dim3(256), 0, s, V, F, workspace, B, R, K, C, S, spc, nch);
        else hipLaunchKernelGGL(dynfilter_fwd_vec<true>, sgrid, dim3(256), 0, s, V, F, workspace, B, R, K, C, S, spc, nch);
        const long long n4 = (long long)B * R * S / 4;
        hipLaunchKernelGGL(dynfilter_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, workspace, P, B, R, K, S, nch);
    } else {
        if (mfma && bmt == 32) hipLaunchKernelGGL((dynfilter_fwd_mfma<false, 32>), xcd_grid(tiles), dim3(128), 0, s, V, F, P, B, R, K, C, S, spc, nch);
        else if (mfma) hipLaunchKernelGGL((dynfilter_fwd_mfma<false, 64>), xcd_grid(tiles), dim3(256), 0, s, V, F, P, B, R, K, C, S, spc, nch);
        else hipLaunchKernelGGL(dynfilter_fwd_vec<false>, xcd_grid(tiles), dim3(256), 0, s, V, F, P, B, R, K, C, S, spc, nch);
    }
    return launch_status(fn);
}

extern "C" int dmn_dynfilter_fwd(const float *V, const float *F, float *P, int B, int L, int K,
                                 int C, int S, void *stream)
{
    return dynfilter_fwd_impl("dmn_dynfilter_fwd", V, F, P, nullptr, B, L, K, C, S, stream);
}

extern "C" int dmn_dynfilter_fwd_ws(const float *V, const float *F, float *P, float *workspace, int B, int L,
                                    int K, int C, int S, void *stream)
{
    return dynfilter_fwd_impl("dmn_dynfilter_fwd_ws", V, F, P, workspace, B, L, K, C, S, stream);
}

extern "C" int dmn_dynfilter_bwd(const float *V, const float *F, const float *gP, float *gV,
                                 float *gF, int B, int L, int K, int C, int S, void *stream)
{
    if (!V || !F || !gP) return fail(-1, "dmn_dynfilter_bwd: NULL pointer");
    if (int rc = check_sizes("dmn_dynfilter_bwd", B, L, K, C, S)) return rc;
    const int R = L * K;
    hipStream_t s = (hipStream_t)stream;
    const char *ve = getenv("DMN_DYNFILTER_BWD_VEC");
    const bool vec = (C % 4 == 0) && (S % 4 == 0) && !(ve && atoi(ve) == 0) &&
                     !(((uintptr_t)V | (uintptr_t)F | (uintptr_t)gP | (uintptr_t)gV | (uintptr_t)gF) & 15) &&
                     (long long)L * B * K * C * 4 < 0x7fffffffLL && (long long)C * S * 4 < 0x7fffffffLL;   // one 2 GiB buffer window per tensor
    if (gF) {
        dim3 grid((C + BN - 1) / BN, (R + BM - 1) / BM, B);
        if (vec) hipLaunchKernelGGL(dynfilter_bwd_vec<1>, dim3((unsigned)((((long long)grid.x * grid.y * B + 7) / 8) * 8)), dim3(256), 0, s, V, F, gP, gF, B, R, K, C, S);
        else hipLaunchKernelGGL(dynfilter_generic<1>, grid, dim3(256), 0, s, V, gP, (const float *)nullptr, gF, B, R, K, C, S);
    }
    if (gV) {
        dim3 grid((S + BN - 1) / BN, (C + BM - 1) / BM, B);
        if (vec) hipLaunchKernelGGL(dynfilter_bwd_vec<2>, dim3((unsigned)((((long long)grid.x * grid.y * B + 7) / 8) * 8)), dim3(256), 0, s, V, F, gP, gV, B, R, K, C, S);
        else hipLaunchKernelGGL(dynfilter_generic<2>, grid, dim3(256), 0, s, V, F, gP, gV, B, R, K, C, S);
    }
    return launch_status("dmn_dynfilter_bwd");
}
