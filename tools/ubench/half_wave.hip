// Does a wavefront whose upper 32 lanes are switched off (EXEC = 0x00000000ffffffff) issue its vector instructions faster?
// v_fma_f32 / v_pk_fma_f32 chains with all 64 lanes and with the low 32 only, one and two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 half_wave.hip -o half_wave ; run: ./half_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(64) kern(float *out, int iters, float a, float b, int active) {
    if ((int)threadIdx.x >= active) return;
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r1, r2}, p2 = {r2, r3}, p3 = {r3, r4}, p4 = {r4, r5}, p5 = {r5, r6}, p6 = {r6, r7}, p7 = {r7, r0};
    f2 aa = {a, a}, bb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));)
        } else {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa), "v"(bb));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND>
void run(const char *name, int waves_per_simd, int active, float *d) {
    const int iters = 2000, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<KIND><<<blocks, 64>>>(d, 10, 1.0001f, 0.5f, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<KIND><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 128 * waves_per_simd);
    printf("%-14s waves/SIMD=%d active lanes=%2d  %.3f ms  %.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, active, ms, per, per * 2.4);
}
int main() {
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2, 4})
        for (int active : {64, 32, 16}) {
            run<0>("v_fma_f32", w, active, d);
            run<1>("v_pk_fma_f32", w, active, d);
        }
    return 0;
}
