// Probe (round 3): which streams of a process share a hardware queue, and which share
// something coarser (a dispatcher pipe)? Ten plain streams, created and touched in order,
// beside a high- and a low-priority one (as a handle's look-ahead streams are). For every
// pair: (1) the serialisation test of gpx_api.hip (a 0.3-ms spinner on one, an empty kernel
// on the other: does the empty one queue behind it?), (2) a dispatch-bound kernel (8192
// one-wave workgroups that do nothing) on both at once: time of the pair over time of one.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_queues.hip -o tools/bin/probe_queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void nothing(int *p)
{
    if (p && threadIdx.x == 999) *p = 1;
}

static double now()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const int NS = 10;
    hipStream_t s[NS], hi, lo;
    int plo = 0, phi = 0;
    CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    for (int i = 0; i < NS; ++i) {
        CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        if (i == 0) {       // like a handle: its stream, then the look-ahead streams
            CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, phi));
            CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, plo));
        }
    }
    // touch in creation order
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, s[0], nullptr);
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, hi, nullptr);
    hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, lo, nullptr);
    for (int i = 1; i < NS; ++i) hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, s[i], nullptr);
    CK(hipDeviceSynchronize());
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    printf("serialised pairs (1 = the empty kernel queued behind the spinner):\n    ");
    for (int j = 0; j < NS; ++j) printf(" %2d", j);
    printf("\n");
    for (int i = 0; i < NS; ++i) {
        printf("  %2d", i);
        for (int j = 0; j < NS; ++j) {
            if (i == j) { printf("  ."); continue; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 30000LL);
            CK(hipEventRecord(ea, s[i]));
            hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, s[j], nullptr);
            CK(hipEventRecord(eb, s[j]));
            CK(hipEventSynchronize(eb));
            const int same = hipEventQuery(ea) == hipSuccess;
            CK(hipEventSynchronize(ea));
            printf("  %d", same);
        }
        printf("\n");
    }
    // dispatch-bound pairs
    const int WG = 16384;
    auto one = [&](hipStream_t a, hipStream_t b) -> double {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipDeviceSynchronize();
            const double t0 = now();
            hipLaunchKernelGGL(nothing, dim3(WG), dim3(64), 0, a, nullptr);
            if (b) hipLaunchKernelGGL(nothing, dim3(WG), dim3(64), 0, b, nullptr);
            (void)hipStreamSynchronize(a);
            if (b) (void)hipStreamSynchronize(b);
            best = std::min(best, now() - t0);
        }
        return best;
    };
    const double alone = one(s[0], nullptr);
    printf("dispatch-bound kernel alone: %.1f us; pairs, time / alone:\n    ", alone);
    for (int j = 0; j < NS; ++j) printf("   %2d", j);
    printf("\n");
    for (int i = 0; i < NS; ++i) {
        printf("  %2d", i);
        for (int j = 0; j < NS; ++j) {
            if (j <= i) { printf("    ."); continue; }
            printf(" %4.2f", one(s[i], s[j]) / alone);
        }
        printf("\n");
    }
    printf("with the high-priority stream: %.2f, the low-priority one: %.2f\n", one(s[0], hi) / alone, one(s[0], lo) / alone);
    // a handle's look-ahead: its stream (0), crit (high), bulk (plain, 1), aux (low)
    printf("handle-like set: stream-crit %.2f stream-bulk %.2f stream-aux %.2f crit-bulk %.2f crit-aux %.2f bulk-aux %.2f\n",
           one(s[0], hi) / alone, one(s[0], s[1]) / alone, one(s[0], lo) / alone, one(hi, s[1]) / alone,
           one(hi, lo) / alone, one(s[1], lo) / alone);
    return 0;
}
