// Hardware probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) checks the A/B/C/D
// lane maps gemm_f64.hip relies on with asymmetric integer data, (2) measures
// the issue rate (TFLOP/s chip-wide, ns per MFMA per SIMD) at 1 and 2 waves
// per SIMD with 4/8/16 independent accumulators. Build:
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/bin/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e = (x);                                                     \
        if (e != hipSuccess) {                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__,   \
                   __LINE__);                                                   \
            return 1;                                                           \
        }                                                                       \
    } while (0)

__global__ void layout_kernel(double *out)
{
    const int l = threadIdx.x;
    const double a = (double)((l & 15) * 4 + (l >> 4) + 1);          // A[i][k]
    const double b = (double)(((l >> 4) + 1) * 100 + (l & 15));      // B[k][j]
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int NACC>
__global__ __launch_bounds__(512) void rate_kernel(double *out, int iters, double seed)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;      // keep the chain alive
}

template <int NACC>
static int run_rate(double *dout, int threads, int blocks_per_cu)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, dout, 16, 0.5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, dout, iters, 0.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * threads / 64;
    const double mfma = waves * iters * NACC;
    const double flops = mfma * 2048.0;
    const double waves_per_simd = (double)threads / 64 * blocks_per_cu / 4;
    printf("RATE nacc=%2d threads=%d blocks/CU=%d waves/SIMD=%.1f: %.3f ms  %.2f TFLOP/s  "
           "%.1f ns per MFMA per SIMD\n",
           NACC, threads, blocks_per_cu, waves_per_simd, ms, flops / ms * 1e-9,
           ms * 1e6 / (iters * NACC * waves_per_simd));
    return 0;
}

int main()
{
    double *dout;
    CK(hipMalloc(&dout, 256 * sizeof(double)));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dout);
    std::vector<double> h(256);
    CK(hipMemcpy(h.data(), dout, 256 * sizeof(double), hipMemcpyDeviceToHost));
    double E[16][16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += (double)(i * 4 + k + 1) * ((k + 1) * 100 + j);
            E[i][j] = s;
        }
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const double v = h[l * 4 + r];
            if (v != E[(l >> 4) + 4 * r][l & 15]) okA = 0;     // f64 map of the guide
            if (v != E[4 * (l >> 4) + r][l & 15]) okB = 0;     // f32-style map
        }
    printf("LAYOUT row=(lane>>4)+4*reg col=lane&15 : %s\n", okA ? "MATCH" : "no");
    printf("LAYOUT row=4*(lane>>4)+reg col=lane&15 : %s\n", okB ? "MATCH" : "no");
    if (!okA && !okB) {
        printf("lane reg value (expected table E[i][j] = sum_k (4i+k+1)((k+1)100+j))\n");
        for (int l = 0; l < 64; l += 5)
            for (int r = 0; r < 4; ++r) printf("  l=%d r=%d v=%.0f\n", l, r, h[l * 4 + r]);
    }
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("DEVICE %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount,
           p.clockRate);
    if (run_rate<4>(dout, 256, 1)) return 1;
    if (run_rate<8>(dout, 256, 1)) return 1;
    if (run_rate<16>(dout, 256, 1)) return 1;
    if (run_rate<16>(dout, 256, 2)) return 1;
    if (run_rate<16>(dout, 512, 1)) return 1;
    if (run_rate<1>(dout, 256, 1)) return 1;
    if (run_rate<2>(dout, 256, 1)) return 1;
    return 0;
}
