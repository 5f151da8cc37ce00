#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    float v = (float)(lane + 1);
    unsigned b = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    float s16 = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    float s32 = __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
    float p16 = (lane & 16) ? __builtin_bit_cast(float, (unsigned)r[0]) : __builtin_bit_cast(float, (unsigned)r[1]);
    float p32 = (lane & 32) ? __builtin_bit_cast(float, (unsigned)q[0]) : __builtin_bit_cast(float, (unsigned)q[1]);
    out[lane] = s16; out[64 + lane] = s32; out[128 + lane] = p16; out[192 + lane] = p32;
}
int main() {
    float *d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float v = l + 1, v16 = (l ^ 16) + 1, v32 = (l ^ 32) + 1;
        if (h[l] != v + v16 || h[64 + l] != v + v32 || h[128 + l] != v16 || h[192 + l] != v32) ++bad;
    }
    printf("bad %d  s16[0]=%g s16[17]=%g s32[5]=%g p16[3]=%g p32[40]=%g\n", bad, h[0], h[17], h[64 + 5], h[128 + 3], h[192 + 40]);
    return bad != 0;
}
