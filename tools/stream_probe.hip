// tools/stream_probe.hip -- what does this MI355X stream at the scan's read : write mix when nothing is
// sequential?  Per element: read 16 B (one float4, like a k = 4 U row; or 12 + 4 B like k = 3 + x), write 4 B
// -- the 20 B per element of sru_fwd_cols -- with every element independent, so the kernel is a pure
// bandwidth probe: the ceiling the scan's `roofline.frac` should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT, int UNROLL>
__global__ void __launch_bounds__(256) mix41(const f32x4 *__restrict__ u, float *__restrict__ o, long long n)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(u + i + k * stride) : u[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const float r = (v[k].x + v[k].y) + (v[k].z + v[k].w);
            if (NT) __builtin_nontemporal_store(r, o + i + k * stride); else o[i + k * stride] = r;
        }
    }
    for (; i < n; i += stride) {
        const f32x4 v = u[i];
        o[i] = (v.x + v.y) + (v.z + v.w);
    }
}

// time-major like the scan: a wave walks T rows of its 64 columns (row t, column c) = u[t * ncols + c]
template <bool NT, int UNROLL>
__global__ void __launch_bounds__(64) rows41(const f32x4 *__restrict__ u, float *__restrict__ o, int T, int ncols)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= ncols) return;
    int t = 0;
    for (; t + UNROLL <= T; t += UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(u + (long long)(t + k) * ncols + c) : u[(long long)(t + k) * ncols + c];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const float r = (v[k].x + v[k].y) + (v[k].z + v[k].w);
            if (NT) __builtin_nontemporal_store(r, o + (long long)(t + k) * ncols + c); else o[(long long)(t + k) * ncols + c] = r;
        }
    }
}

// 4 : 1 with WIDE stores: a thread takes 4 consecutive elements (4 float4 reads, ONE float4 write)
template <bool NT>
__global__ void __launch_bounds__(256) mix41_wide(const f32x4 *__restrict__ u, f32x4 *__restrict__ o, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = NT ? __builtin_nontemporal_load(u + 4 * i + k) : u[4 * i + k];
    f32x4 r;
    r.x = (v[0].x + v[0].y) + (v[0].z + v[0].w); r.y = (v[1].x + v[1].y) + (v[1].z + v[1].w);
    r.z = (v[2].x + v[2].y) + (v[2].z + v[2].w); r.w = (v[3].x + v[3].y) + (v[3].z + v[3].w);
    if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r;
}
// 4 : 1, one element per thread but the 4-byte results of 4 neighbouring lanes gathered into one lane's
// 16-byte store is not possible without a shuffle; instead: lane-strided reads so that stores are 16 B:
// thread reads u[i], u[i + q], u[i + 2q], u[i + 3q] (q = n / 4) and writes 4 B to each quarter -> same as flat.

// the backward scan's mix: read 24 B (a float4 + two floats), write 16 B per element
template <bool NT>
__global__ void __launch_bounds__(256) mix2416(const f32x4 *__restrict__ u, const float *__restrict__ a, const float *__restrict__ b2,
                                                f32x4 *__restrict__ o, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f32x4 v = NT ? __builtin_nontemporal_load(u + i) : u[i];
    const float x = NT ? __builtin_nontemporal_load(a + i) : a[i], y = NT ? __builtin_nontemporal_load(b2 + i) : b2[i];
    v.x += x; v.y += y; v.z *= x; v.w *= y;
    if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
}

template <class F> static float timeit(F f, int iters = 10)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main()
{
    const int T = 5120, ncols = 32160;
    const long long n = (long long)T * ncols;
    f32x4 *u; float *o;
    CHECK(hipMalloc(&u, n * 16)); CHECK(hipMalloc(&o, n * 4));
    CHECK(hipMemset(u, 0, n * 16));
    const double bytes = 20.0 * n;
    printf("stream probe: %lld elements, 16 B read + 4 B written each (%.2f GB per launch)\n", n, bytes / 1e9);
    for (int grid : {2048, 4096, 8192, 65536, (int)((n + 255) / 256 / 4)}) {
        float ms = timeit([&] { hipLaunchKernelGGL((mix41<true, 4>), dim3(grid), dim3(256), 0, 0, u, o, n); });
        printf("  flat   nt    unroll 4 grid %7d : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", grid, ms, bytes / ms / 1e6, bytes / ms / 8e7);
        ms = timeit([&] { hipLaunchKernelGGL((mix41<false, 4>), dim3(grid), dim3(256), 0, 0, u, o, n); });
        printf("  flat   plain unroll 4 grid %7d : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", grid, ms, bytes / ms / 1e6, bytes / ms / 8e7);
    }
    {
        float ms = timeit([&] { hipLaunchKernelGGL((mix41<true, 8>), dim3(4096), dim3(256), 0, 0, u, o, n); });
        printf("  flat   nt    unroll 8 grid    4096 : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", ms, bytes / ms / 1e6, bytes / ms / 8e7);
        const int g = (ncols + 63) / 64;
        ms = timeit([&] { hipLaunchKernelGGL((rows41<true, 16>), dim3(g), dim3(64), 0, 0, u, o, T, ncols); });
        printf("  rows   nt    16 rows in flight, %d waves : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", g, ms, bytes / ms / 1e6, bytes / ms / 8e7);
        ms = timeit([&] { hipLaunchKernelGGL((rows41<true, 8>), dim3(g), dim3(64), 0, 0, u, o, T, ncols); });
        printf("  rows   nt     8 rows in flight, %d waves : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", g, ms, bytes / ms / 1e6, bytes / ms / 8e7);
        ms = timeit([&] { hipLaunchKernelGGL((rows41<false, 16>), dim3(g), dim3(64), 0, 0, u, o, T, ncols); });
        printf("  rows   plain 16 rows in flight, %d waves : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", g, ms, bytes / ms / 1e6, bytes / ms / 8e7);
    }
    {
        const long long n4 = n / 4;
        float ms = timeit([&] { hipLaunchKernelGGL((mix41_wide<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, u, (f32x4 *)o, n4); });
        printf("  flat   nt    4 elements per thread, one 16-byte store : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", ms, bytes / ms / 1e6, bytes / ms / 8e7);
        ms = timeit([&] { hipLaunchKernelGGL((mix41_wide<false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, u, (f32x4 *)o, n4); });
        printf("  flat   plain 4 elements per thread, one 16-byte store : %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", ms, bytes / ms / 1e6, bytes / ms / 8e7);
    }
    CHECK(hipFree(u)); CHECK(hipFree(o));
    {   // the backward scan at the training size: 2000 x 64 320 elements, 24 B read + 16 B written
        const long long nb = 2000LL * 64320;
        f32x4 *bu, *bo; float *ba, *bb;
        CHECK(hipMalloc(&bu, nb * 16)); CHECK(hipMalloc(&bo, nb * 16)); CHECK(hipMalloc(&ba, nb * 4)); CHECK(hipMalloc(&bb, nb * 4));
        CHECK(hipMemset(bu, 0, nb * 16)); CHECK(hipMemset(ba, 0, nb * 4)); CHECK(hipMemset(bb, 0, nb * 4));
        const double bbytes = 40.0 * nb;
        const int g = (int)((nb + 255) / 256);
        float ms = timeit([&] { hipLaunchKernelGGL((mix2416<true>), dim3(g), dim3(256), 0, 0, bu, ba, bb, bo, nb); });
        printf("backward mix (24 B read + 16 B written, %lld elements): nt    %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", nb, ms, bbytes / ms / 1e6, bbytes / ms / 8e7);
        ms = timeit([&] { hipLaunchKernelGGL((mix2416<false>), dim3(g), dim3(256), 0, 0, bu, ba, bb, bo, nb); });
        printf("backward mix (24 B read + 16 B written, %lld elements): plain %.3f ms  %6.0f GB/s (%.1f %% of 8 TB/s)\n", nb, ms, bbytes / ms / 1e6, bbytes / ms / 8e7);
        CHECK(hipFree(bu)); CHECK(hipFree(bo)); CHECK(hipFree(ba)); CHECK(hipFree(bb));
    }
    return 0;
}
