// VALU issue/throughput microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_mov_dpp etc.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(64) kern(float *out, int iters, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = r0 * 2, q1 = r1 * 2, q2 = r2 * 2, q3 = r3 * 2, q4 = r4 * 2, q5 = r5 * 2, q6 = r6 * 2, q7 = r7 * 2;
    unsigned long long msk = __ballot(threadIdx.x & 1); unsigned vm = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { // 16 independent v_fma_f32 per REP
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));)
        } else if (KIND == 1) { // v_pk_fma_f32 on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {r0, q0}, p1 = {r1, q1}, p2 = {r2, q2}, p3 = {r3, q3}, p4 = {r4, q4}, p5 = {r5, q5}, p6 = {r6, q6}, p7 = {r7, q7};
            f2 aa = {a, a}, bb = {b, b};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa), "v"(bb));)
            r0 = p0.x; q0 = p0.y; r1 = p1.x; q1 = p1.y; r2 = p2.x; q2 = p2.y; r3 = p3.x; q3 = p3.y;
            r4 = p4.x; q4 = p4.y; r5 = p5.x; q5 = p5.y; r6 = p6.x; q6 = p6.y; r7 = p7.x; q7 = p7.y;
        } else if (KIND == 2) { // v_mov_b32_dpp quad_perm
            REP16(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                               "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                               "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (KIND == 3) { // v_fmac_f32_dpp (fused dpp operand)
            REP16(asm volatile("v_fmac_f32_dpp %0, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %4, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %0, %1, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %3, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %4, %5, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %7, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if (KIND == 4) { // v_rcp_f32
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (KIND == 5) { // v_cndmask_b32 (vcc)
            REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) :: "vcc");)
        } else if (KIND == 6) { // v_pk_mul_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {r0, q0}, p1 = {r1, q1}, p2 = {r2, q2}, p3 = {r3, q3}, p4 = {r4, q4}, p5 = {r5, q5}, p6 = {r6, q6}, p7 = {r7, q7};
            f2 aa = {a, a};
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa));)
            r0 = p0.x; q0 = p0.y; r1 = p1.x; q1 = p1.y; r2 = p2.x; q2 = p2.y; r3 = p3.x; q3 = p3.y;
            r4 = p4.x; q4 = p4.y; r5 = p5.x; q5 = p5.y; r6 = p6.x; q6 = p6.y; r7 = p7.x; q7 = p7.y;
        } else if (KIND == 8) { // v_cndmask_b32_e64 with an SGPR-pair mask
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %8\n v_cndmask_b32_e64 %1, %1, %2, %8\n v_cndmask_b32_e64 %2, %2, %3, %8\n v_cndmask_b32_e64 %3, %3, %4, %8\n"
                               "v_cndmask_b32_e64 %4, %4, %5, %8\n v_cndmask_b32_e64 %5, %5, %6, %8\n v_cndmask_b32_e64 %6, %6, %7, %8\n v_cndmask_b32_e64 %7, %7, %0, %8\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(msk));)
        } else if (KIND == 9) { // v_bfi_b32 with a VGPR lane mask
            REP16(asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                               "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(vm));)
        } else if (KIND == 10) { // v_cndmask_b32 vcc, independent (each reads only itself and a constant)
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");)
        } else if (KIND == 11) { // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if (KIND == 12) { // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                               "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (KIND == 7) { // dependent chain v_fma_f32 (latency)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               : "+v"(r0) : "v"(a), "v"(b));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
}

template <int KIND>
void run(const char *name, int waves_per_simd, float *d) {
    const int iters = 2000;
    const int blocks = 256 * 4 * waves_per_simd; // one wave per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<KIND><<<blocks, 64>>>(d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<KIND><<<blocks, 64>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * 8;
    // ns per wave-instruction on one SIMD with `waves_per_simd` waves sharing it
    const double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_instr_simd, ns_per_instr_simd * 2.4);
}

int main() {
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    for (int w : {1, 2}) {
        run<0>("v_fma_f32", w, d);
        run<1>("v_pk_fma_f32", w, d);
        run<6>("v_pk_mul_f32", w, d);
        run<2>("v_mov_b32_dpp+nop", w, d);
        run<3>("v_fmac_f32_dpp", w, d);
        run<4>("v_rcp_f32", w, d);
        run<5>("v_cndmask_b32", w, d);
        run<7>("v_fma_f32 dependent", w, d);
        run<8>("v_cndmask_e64 sgpr", w, d);
        run<9>("v_bfi_b32", w, d);
        run<10>("v_cndmask vcc indep", w, d);
        run<11>("v_add_f32", w, d);
        run<12>("v_mov_b32", w, d);
    }
    return 0;
}
