// dms_amd/csrc/sru_ring.h -- pieces shared by the SRU scan kernels (forward: sru_fwd.inl, backward:
// sru_bwd.hip): packed 3-float type, buffer descriptors, streamed-operand cache policy, static_for.
#pragma once
#include "common.h"

namespace dmn {

struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

constexpr float LOG2E = 1.4426950408889634f;

// Cache policy of the streamed operands: U / x are read once and h / c written once per
// launch, so both sides are non-temporal (`nt` on the loads, aux bit 1 on the stores).
// Measured (tools/scan_bench.py policy, MI355X): in-situ k=4 63.5 -> 67.4 %, tail layer
// 75 -> 84.6 %, config-5 stress 61.6 -> 64.4 % of 8 TB/s; either side alone gives about half.
#ifndef DMN_LD_POLICY
#define DMN_LD_POLICY " nt"
#endif
#ifndef DMN_ST_AUX
#define DMN_ST_AUX 2
#endif

// Counted waits and stores: ONE policy for the library.  Vector-memory loads and stores retire IN ORDER on the one
// vmcnt counter: s_waitcnt vmcnt(N) guarantees every operation but the N youngest, whatever their kind.  That is
// hipcc's own model for this target family (no separate store counter: SIInsertWaitcnts counts loads and stores as
// one event type; for load / store / use-of-load it emits vmcnt(1) -- tools/probe/vmcnt_order.hip, checked by
// tests/test_asm_audit.py) and every compiler-scheduled kernel of this library depends on it; mix_sru.hip's
// hand-counted waits (N = the younger h stores) and mix.hip's (N = the younger LDS-DMA loads) use it directly.
// The scan rings below count the younger LOADS only (0): with the younger stores left out a wait is merely
// stricter than needed, and that setting is what measured fastest once the ring depths were re-tuned for it
// (tools/sru_bwd_bench.py, tools/scan_bench.py; 72.3 % -> 74.8 % on the roofline kernel).  1 = count the
// younger stores too (equally valid), for A/B runs.
#ifndef DMN_VMCNT_COUNT_STORES
#define DMN_VMCNT_COUNT_STORES 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}

// raw-buffer SRD words for the asm loads (base, stride 0, 2 GiB window, dword format)
__device__ __forceinline__ i32x4 make_srd(const void *base)
{
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = (int)(a & 0xffffffffull);
    r.y = (int)((a >> 32) & 0xffffull);
    r.z = 0x7fffffff;
    r.w = 0x00020000;
    return r;
}

template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace dmn
