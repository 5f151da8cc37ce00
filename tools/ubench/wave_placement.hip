// Where does the dispatcher put the waves of a launch that is smaller than the machine?  N workgroups of one wave each spin
// for ~100 us and report (XCC, SE, CU, SIMD).  Usage: wave_placement [workgroups=1024] [lds_bytes=0] [waves per workgroup=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(1024) where(unsigned *out, long long spin) {
    extern __shared__ float pad[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    const long long t0 = wall_clock64();
    float a = threadIdx.x;
    while (wall_clock64() - t0 < spin) a = a * 1.0001f + 0.5f;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
        if (a == 12345.f) pad[0] = a;
    }
}
int main(int argc, char **argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t lds = argc > 2 ? (size_t)atol(argv[2]) : 0;
    const int wpg = argc > 3 ? atoi(argv[3]) : 1;
    const int n = groups * wpg;
    unsigned *d;
    hipMalloc(&d, n * 8);
    if (lds > 65536) hipFuncSetAttribute((const void *)where, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        where<<<groups, 64 * wpg, lds>>>(d, 10000);   // wall_clock64 ticks at 100 MHz: 100 us
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu, per_xcc;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++;
        per_simd[(cukey << 2) | simd]++;
        per_xcc[xcc]++;
    }
    std::map<int, int> hist_cu, hist_simd;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    printf("%d workgroups x %d waves, lds=%zu: %zu CUs used, %zu SIMDs used\n", groups, wpg, lds, per_cu.size(), per_simd.size());
    printf("  waves per XCC:");
    for (auto &kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n  CUs holding k waves:");
    for (auto &kv : hist_cu) printf(" k=%d:%d", kv.first, kv.second);
    printf("\n  SIMDs holding k waves:");
    for (auto &kv : hist_simd) printf(" k=%d:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
