// Which CUs does a hipExtStreamCreateWithCUMask stream run on? Every workgroup records
// its XCC and CU (HW_ID) and then holds its CU for a while so that the grid spreads over
// everything the mask allows. usage: cumask_probe <first_cleared_bit> <count> [stride]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned *out, long long spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}
int main(int argc, char **argv)
{
    const int first = argc > 1 ? atoi(argv[1]) : 0, count = argc > 2 ? atoi(argv[2]) : 8;
    const int stride = argc > 3 ? atoi(argv[3]) : 1;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    uint32_t mask[32] = {};
    for (int i = 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
    for (int c = 0; c < count; ++c) {
        const int b = first + c * stride;
        mask[b / 32] &= ~(1u << (b % 32));
    }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (ncu + 31) / 32, mask) != hipSuccess) return 1;
    const int nwg = 4096;
    unsigned *d;
    hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(1024), 0, s, d, 20000LL /* 200 us */);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nwg);
    hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    printf("cleared bits first=%d count=%d stride=%d: CUs seen per XCC:", first, count, stride);
    int total = 0;
    for (auto &kv : per_xcc) {
        printf(" x%d=%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("  total=%d of %d\n", total, ncu);
    return 0;
}
