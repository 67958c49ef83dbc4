// Probe (round 3): is a 131 000-workgroup store kernel of the shape of the config-5 build
// (64 x 64 fp32 tile per 256-thread workgroup, stored twice: 32 KB per workgroup, 4.3 GB
// in all) limited by the workgroup launch rate? Same bytes with 1, 2, 4 tiles per
// workgroup (wider tiles, no serial loop) and with a little or a lot of ALU work per
// element. Build: hipcc --offload-arch=gfx950 -O3 tools/probe_dispatch.hip -o tools/bin/probe_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int W, int ALU>
__global__ __launch_bounds__(256) void store_kernel(float *out, long long ld, int T, float seed)
{
    __shared__ float pad[64 * 65];
    // 1-D triangular grid over (row tile r, column tile group c): W 64-col tiles per group
    const int G = T / W;                       // groups per row
    long long k = blockIdx.x;
    int r = 0;
    // rows come in blocks; plain search (uniform, scalar)
    while (true) {
        const int live = G - r / W;
        if (k < live) break;
        k -= live;
        ++r;
    }
    const int c = r / W + (int)k;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    if (tid == 0) pad[0] = seed;
    float v[W][4][4];
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float x = seed + (float)(tid + a + b + w);
#pragma unroll
                for (int i = 0; i < ALU; ++i) x = __builtin_amdgcn_exp2f(x * 0.5f - 1.0f);
                v[w][a][b] = x;
            }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const long long i0 = (long long)r * 64, j0 = ((long long)c * W + w) * 64;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 o = {v[w][a][0], v[w][a][1], v[w][a][2], v[w][a][3]};
            *reinterpret_cast<float4 *>(out + (i0 + ty + 16 * a) * ld + j0 + 4 * tx) = o;
            if (j0 / 64 != r)
                *reinterpret_cast<float4 *>(out + (j0 + ty + 16 * a) * ld + i0 + 4 * tx) = o;
        }
    }
}

template <int W, int ALU> static int run(float *d, int N)
{
    const int T = N / 64, G = T / W;
    long long wgs = 0;
    for (int r = 0; r < T; ++r) wgs += G - r / W;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((store_kernel<W, ALU>), dim3((unsigned)wgs), dim3(256), 0, 0, d, (long long)N, T, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((store_kernel<W, ALU>), dim3((unsigned)wgs), dim3(256), 0, 0, d, (long long)N, T, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("tiles/WG=%d exp2/elem=%2d workgroups=%7lld  %.3f ms  %.2f TB/s  %.1f WG/us\n", W, ALU, wgs, ms,
           4.0 * N * N / ms * 1e-9, wgs / ms * 1e-3);
    return 0;
}

int main()
{
    const int N = 32768;
    float *d;
    CK(hipMalloc(&d, (size_t)N * N * 4));
    if (run<1, 0>(d, N) || run<2, 0>(d, N) || run<4, 0>(d, N)) return 1;
    if (run<1, 2>(d, N) || run<2, 2>(d, N) || run<4, 2>(d, N)) return 1;
    if (run<1, 6>(d, N) || run<2, 6>(d, N) || run<4, 6>(d, N)) return 1;
    return 0;
}
