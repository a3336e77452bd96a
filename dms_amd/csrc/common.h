// dms_amd/csrc/common.h -- shared helpers for libdmn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/dmn_hip.h"

namespace dmn {

// thread-local message of the last failure, exposed through dmn_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr int WAVE = 64;          // CDNA wavefront
constexpr int NUM_CU = 256;       // MI355X

__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

// 1 / (1 + 2^(-v*log2 e)): one v_exp_f32 + one v_rcp_f32 (both ~1 ulp)
__device__ __forceinline__ float sigmoid_f(float v)
{
    return fast_rcp(1.0f + fast_exp2(v * -1.4426950408889634f));
}

// tanh(v) = 2 / (1 + 2^(-2 v log2 e)) - 1: mul, v_exp, add, v_rcp, fma.  Saturates
// correctly (2^inf -> rcp(inf) = 0 -> -1).  Absolute error <= ~1.2e-7 (the rounding of the
// sigmoid near 1/2), i.e. ~1e-7 relative to the O(1) hidden states it feeds.
__device__ __forceinline__ float tanh_f(float v)
{
    const float s = fast_rcp(1.0f + fast_exp2(v * -2.8853900817779268f));
    return __builtin_fmaf(2.0f, s, -1.0f);
}

// streaming (write-once) 16-byte store: non-temporal, does not displace L2 / MALL lines
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream4(float *p, float a, float b, float c, float d)
{
    f32x4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t *>(p));
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float v)
{
    if (ACT == DMN_ACT_TANH) return tanh_f(v);
    if (ACT == DMN_ACT_RELU) return v > 0.0f ? v : 0.0f;
    return v;
}

// act_fwd split in two halves with act_fwd(v) == act_back<ACT>(act_front<ACT>(v)) bit for bit: for tanh the
// exponential and the reciprocal land on different sides (mix_sru_pair_kernel balances its two waves'
// quarter-rate instructions this way), for relu / identity the front half is the whole activation.
template <int ACT>
__device__ __forceinline__ float act_front(float v)
{
    if (ACT == DMN_ACT_TANH) return 1.0f + fast_exp2(v * -2.8853900817779268f);
    return act_fwd<ACT>(v);
}
template <int ACT>
__device__ __forceinline__ float act_back(float v)
{
    if (ACT == DMN_ACT_TANH) return __builtin_fmaf(2.0f, fast_rcp(v), -1.0f);
    return v;
}

// derivative expressed through the activation OUTPUT y
template <int ACT>
__device__ __forceinline__ float act_bwd(float y)
{
    if (ACT == DMN_ACT_TANH) return 1.0f - y * y;
    if (ACT == DMN_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}

}  // namespace dmn
