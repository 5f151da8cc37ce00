// Store-pattern microbenchmark for gfx950: the cfg-3 solve writes 16384 trajectories x 366 rows x 544 B; every
// wave owns 8 trajectories (one per 8-lane group) and per save round each lane stores 4 x 16 B + 4 B into its
// trajectory's current row.  How long do exactly those stores take with no arithmetic in between, and how
// does it change when the rounds are spaced out by VALU work (the solver steps between saves)?
// Build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int SPACING>
__global__ void __launch_bounds__(64) kern(float *out, int n_save, int D, float seed) {
    const int lane = threadIdx.x & 63, a = lane & 7, grp = lane >> 3;
    const long traj = (long)blockIdx.x * 8 + grp;
    float *row = out + traj * (long)n_save * D;
    float4 v = {seed + lane, seed, seed * 2, seed * 3};
    float acc = seed;
    for (int j = 0; j < n_save; ++j) {
        if (SPACING > 0) { // dependent FMAs standing in for the solver's step between save rounds
#pragma unroll 16
            for (int q = 0; q < SPACING; ++q) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.x = acc;
        }
        row[a] = v.x;
        *reinterpret_cast<float4 *>(row + 8 + a * 4) = v;
        *reinterpret_cast<float4 *>(row + 40 + a * 4) = v;
        *reinterpret_cast<float4 *>(row + 72 + a * 4) = v;
        *reinterpret_cast<float4 *>(row + 104 + a * 4) = v;
        row += D;
    }
}

// the same stores, but spread over the round's arithmetic: one store after every fifth of the FMAs
template <int SPACING>
__global__ void __launch_bounds__(64) kern_spread(float *out, int n_save, int D, float seed) {
    const int lane = threadIdx.x & 63, a = lane & 7, grp = lane >> 3;
    const long traj = (long)blockIdx.x * 8 + grp;
    float *row = out + traj * (long)n_save * D;
    float4 v = {seed + lane, seed, seed * 2, seed * 3};
    float acc = seed;
    for (int j = 0; j < n_save; ++j) {
#pragma unroll
        for (int part = 0; part < 5; ++part) {
#pragma unroll 16
            for (int q = 0; q < SPACING / 5; ++q) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
            v.x = acc;
            if (part == 0) row[a] = v.x;
            else *reinterpret_cast<float4 *>(row + 8 + (part - 1) * 32 + a * 4) = v;
        }
        row += D;
    }
}

template <int SPACING, bool SPREAD = false>
static void run(float *out, int B, int n_save, int D) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = SPREAD ? kern_spread<SPACING> : kern<SPACING>;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(B / 8), dim3(64), 0, 0, out, n_save, D, 1.0f);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(B / 8), dim3(64), 0, 0, out, n_save, D, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%s spacing %5d dependent FMAs per round: %.3f ms, %.0f GB/s written\n", SPREAD ? "spread " : "burst  ", SPACING, ms, (double)B * n_save * D * 4 / ms / 1e6);
}

int main() {
    const int B = 16384, n_save = 366, D = 136;
    float *out;
    if (hipMalloc(&out, (size_t)B * n_save * D * 4) != hipSuccess) return 1;
    run<0>(out, B, n_save, D);
    run<64>(out, B, n_save, D);
    run<256>(out, B, n_save, D);
    run<512>(out, B, n_save, D);
    run<1024>(out, B, n_save, D);
    run<2048>(out, B, n_save, D);
    run<512, true>(out, B, n_save, D);
    run<1024, true>(out, B, n_save, D);
    run<2048, true>(out, B, n_save, D);
    return 0;
}
