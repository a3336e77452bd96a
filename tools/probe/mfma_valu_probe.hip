// Do fp32 MFMAs of one wave run in the shadow of ANOTHER wave's VALU work on the same SIMD (MI355X)?
// One workgroup of 8 waves on one CU: waves w and w + 4 share SIMD w (checked with HW_ID below).  Each wave
// runs a fixed amount of one kind of work and records its own wall-clock; modes:
//   0: all 8 waves MFMA      1: all 8 waves VALU (transcendental + FMA mix of the scan)
//   2: waves 0-3 MFMA, waves 4-7 VALU      3: only waves 0-3 MFMA (4-7 idle)      4: only waves 4-7 VALU
//   5: ONE wave alternating MFMA and VALU work (the same total as an MFMA wave + a VALU wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_work(f32x16 acc, float a, float b, int n)
{
    f32x16 acc2 = acc, acc3 = acc, acc4 = acc;
    for (int i = 0; i < n; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        acc4 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc4, 0, 0, 0);
    }
    return acc + acc2 + acc3 + acc4;
}
__device__ __forceinline__ float valu_work(float x, int n)
{
    float y0 = x, y1 = x + 1.0f, y2 = x + 2.0f, y3 = x + 3.0f;
    for (int i = 0; i < n; ++i) {
        // per chain: exp2, add, rcp, fma (what a gate costs)
        y0 = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y0)), 0.5f, 0.25f);
        y1 = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y1)), 0.5f, 0.25f);
        y2 = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y2)), 0.5f, 0.25f);
        y3 = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y3)), 0.5f, 0.25f);
    }
    return y0 + y1 + y2 + y3;
}
__device__ __forceinline__ float fma_work(float x, int n)
{
    float y0 = x, y1 = x + 1.0f, y2 = x + 2.0f, y3 = x + 3.0f, y4 = x + 4.0f, y5 = x + 5.0f, y6 = x + 6.0f, y7 = x + 7.0f;
    for (int i = 0; i < n; ++i) {
        y0 = __builtin_fmaf(y0, 0.999f, 0.25f); y1 = __builtin_fmaf(y1, 0.999f, 0.25f);
        y2 = __builtin_fmaf(y2, 0.999f, 0.25f); y3 = __builtin_fmaf(y3, 0.999f, 0.25f);
        y4 = __builtin_fmaf(y4, 0.999f, 0.25f); y5 = __builtin_fmaf(y5, 0.999f, 0.25f);
        y6 = __builtin_fmaf(y6, 0.999f, 0.25f); y7 = __builtin_fmaf(y7, 0.999f, 0.25f);
    }
    return y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}

__global__ void __launch_bounds__(512) probe(int mode, int nm, int nv, long long *cycles, unsigned *hwid, float *sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = (float)lane;
    float v = (float)lane * 0.01f;
    __syncthreads();
    const long long t0 = wall_clock64();
    const bool lo = wave < 4;
    if (mode == 0 || (mode == 2 && lo) || (mode == 3 && lo)) acc = mfma_work(acc, v, 0.5f, nm);
    else if (mode == 1 || (mode == 2 && !lo) || (mode == 4 && !lo)) v = valu_work(v, nv);
    else if (mode == 5 && wave == 0) { acc = mfma_work(acc, v, 0.5f, nm); v = valu_work(v, nv); }
    else if (mode == 6 && lo) acc = mfma_work(acc, v, 0.5f, nm);
    else if (mode == 6 && !lo) v = fma_work(v, nv);
    else if (mode == 7 && !lo) v = fma_work(v, nv);
    const long long t1 = wall_clock64();
    float s = v;
    for (int i = 0; i < 16; ++i) s += acc[i];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { cycles[wave] = t1 - t0; hwid[wave] = id; }
}

int main()
{
    long long *cyc; unsigned *hw; float *sink;
    hipMalloc(&cyc, 8 * 8); hipMalloc(&hw, 8 * 4); hipMalloc(&sink, 512 * 4);
    const int nm = 20000, nv = 20000;
    const char *names[] = {"all 8 waves MFMA", "all 8 waves gate maths (exp2, rcp, fma)", "waves 0-3 MFMA + waves 4-7 gate maths",
                           "waves 0-3 MFMA alone", "waves 4-7 gate maths alone", "one wave: MFMA work, then gate maths",
                           "waves 0-3 MFMA + waves 4-7 plain FMA chains", "waves 4-7 plain FMA chains alone"};
    int freq = 0;
    hipDeviceGetAttribute(&freq, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock %d kHz; %d x 4 MFMA 32x32x2 f32 per MFMA wave, %d x 4 gates per VALU wave\n", freq, nm, nv);
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, mode, nm, nv, cyc, hw, sink);
        hipDeviceSynchronize();
        long long hc[8]; unsigned hh[8];
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost); hipMemcpy(hh, hw, sizeof(hh), hipMemcpyDeviceToHost);
        printf("mode %d (%s):\n   us per wave:", mode, names[mode]);
        for (int w = 0; w < 8; ++w) printf(" %8.1f", hc[w] * 1e3 / freq);
        printf("\n   simd of wave:");
        for (int w = 0; w < 8; ++w) printf(" %u", (hh[w] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
