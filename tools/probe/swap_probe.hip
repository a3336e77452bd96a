// what does v_permlane32_swap do, and where do the rows of a 32x32x2 f32 MFMA land?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(unsigned *o, float *rows)
{
    const unsigned lane = threadIdx.x;
    unsigned x = lane, y = 100 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[lane] = r[0]; o[64 + lane] = r[1];
    // D = A x B with A[i][k] = (k == 0) ? i : 0, B[k][j] = (k == 0) ? 1 : 0  ->  D[i][j] = i
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float a = (lane < 32) ? (float)(lane & 31) : 0.0f, b = (lane < 32) ? 1.0f : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) rows[lane * 16 + i] = acc[i];
}
int main()
{
    unsigned *o; float *rows;
    hipMalloc(&o, 128 * 4); hipMalloc(&rows, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, rows);
    unsigned ho[128]; float hr[1024];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(hr, rows, sizeof(hr), hipMemcpyDeviceToHost);
    printf("new vdst: lane0 %u lane31 %u lane32 %u lane63 %u\n", ho[0], ho[31], ho[32], ho[63]);
    printf("new src0: lane0 %u lane31 %u lane32 %u lane63 %u\n", ho[64], ho[95], ho[96], ho[127]);
    for (int l : {0, 1, 32, 33}) { printf("lane %2d rows:", l); for (int i = 0; i < 16; ++i) printf(" %g", hr[l * 16 + i]); printf("\n"); }
    return 0;
}
