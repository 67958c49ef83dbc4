// Probe (round 3, VERDICT stretch item): issue rate of the int8 matrix pipe on this
// MI355X under sustained load on random operands, next to the fp64 MFMA, to bound what an
// Ozaki-style split of the fp64 rank-2048 updates into int8 slice products could reach.
// Register-only loops (the ceiling of any kernel), every CU busy, ~1 s per configuration.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_i8.hip -o tools/bin/probe_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

template <int NACC>
__global__ __launch_bounds__(256) void i8_32(int *out, int iters, unsigned seed)
{
    unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
    v4i a = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    v4i b = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    v16i acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int t = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) t += acc[i][j];
    if (t == 0x7fffffff) out[0] = t;
}

template <int NACC>
__global__ __launch_bounds__(256) void i8_16(int *out, int iters, unsigned seed)
{
    unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
    v4i a = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    v4i b = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    v4i acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int t = 0;
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 0x7fffffff) out[0] = t;
}

template <int NACC>
__global__ __launch_bounds__(256) void f64_16(int *out, int iters, unsigned seed)
{
    unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
    double a = (double)(rnd(s) >> 8) * 1e-7, b = (double)(rnd(s) >> 8) * 1e-7;
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double t = 0;
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678) out[0] = 1;
}

template <typename K> static int timeit(const char *name, K kernel, int *d, double ops_per_mfma, int nacc,
                                        int blocks_per_cu, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);   // warm: clocks settle
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 54321u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)blocks * 4 * iters * nacc;
    const double wps = blocks_per_cu;                       // waves per SIMD
    printf("%-28s waves/SIMD=%d  %8.1f ms  %9.1f Tops/s  %.2f ns per MFMA per SIMD\n", name,
           blocks_per_cu, ms, mfma * ops_per_mfma / ms * 1e-9, ms * 1e6 / (iters * nacc * wps));
    return 0;
}

int main()
{
    int *d;
    CK(hipMalloc(&d, 1024));
    // (the fp64 MFMA rate is measured by tools/probe_mfma.hip: 69-73 TFLOP/s; a loop of this
    // shape on v4d accumulators compiles to VGPR <-> AGPR copies around every MFMA and
    // under-reports it by half)
    if (timeit("i8 32x32x32 (4 acc)", i8_32<4>, d, 65536.0, 4, 1, 8000000)) return 1;
    if (timeit("i8 32x32x32 (4 acc)", i8_32<4>, d, 65536.0, 4, 2, 4000000)) return 1;
    if (timeit("i8 32x32x32 (7 acc, Ozaki)", i8_32<7>, d, 65536.0, 7, 1, 4000000)) return 1;
    return 0;
}
