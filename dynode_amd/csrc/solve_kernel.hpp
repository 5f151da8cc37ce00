// solve_kernel.hpp -- fused adaptive RK5(4) + compartmental RHS kernel for gfx950 (MI355X).
//
// Replaces, for B independent parameter samples, the whole of
//   diffrax.diffeqsolve(ODETerm(ode), Tsit5()|Dopri5(), t0, t1, dt0=None, y0, args,
//                       ClipStepSizeController(PIDController(rtol, atol)), SaveAt(ts), max_steps)
// as called by dynode.simulation.simulate (reference src/dynode/simulation/odes.py:107-144)
// with the RHS family of the reference's examples/*.py.
//
// Mapping (CDNA4, 64-lane wavefronts): a trajectory is owned by a group of G lanes
// (G = power of two >= n_age); lane `a` of the group holds EVERY compartment value of age
// bin `a` in VGPRs (NV = 1 + S*(E + 1 + W + C) values), together with all 7 RK stages.
// A wavefront therefore integrates 64/G trajectories with all lanes busy.  The only
// cross-lane traffic is (1) the age-contact contraction sum_b C[a][b] x_b -- an all-gather
// inside the lane group by xor-exchange against a pre-permuted contact row held in VGPRs --
// and (2) the RMS error norm (xor-butterfly).  No LDS allocation, no barriers, no atomics;
// HBM traffic is the compulsory I/O only: P + D floats in, n_save*D_saved floats out.
// The whole [t0,t1] solve (all steps, accept/reject, dense output) runs inside ONE launch.
#pragma once
// (diagnostic builds, tools/diag_build.sh W3: every solve kernel compiled for three waves per SIMD)
#ifdef DYN_DIAG_W3
#define DYN_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#ifndef DYN_KERNEL_ATTR
#define DYN_KERNEL_ATTR
#endif
#include "nuts_device.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace dyn {

enum { ST_OK = 0, ST_MAX_STEPS = 1, ST_NONFINITE = 2 };

template <typename T>
struct KArgs {
    const T *y0;
    const T *params;
    const T *contact;
    const T *save_ts;
    T *out;
    int32_t *status;
    int32_t *n_acc;
    int32_t *n_rej;
    int64_t B;
    int64_t max_steps;
    T t0, t1, rtol, atol, constant_dt;
    int32_t y0_batched, n_save, A, P, normalize, seasonal, d_saved, vec_ok;
    int32_t save_off[5]; // offset of s,e,i,r,c inside a saved row; -1 = not saved
    uint64_t intro_mask[8]; // per strain: bit a set = age bin a receives external introductions
    int32_t n_vax_tiers, n_vax_knots; // vaccination tiers actually used (<= KV) and spline knots (<= 4)
    int32_t seasonal_vax, has_intro;  // SEIP family (seip_kernel.hpp): yearly reset of the top vaccination tier; introductions on
    // fused observation likelihood (tangent kernels only; obs == nullptr: off)
    const T *obs;        // [n_obs][ll_row] observed counts, shared by the batch
    double *ll_out;      // [B] sum of obs * log(rate) - rate
    double *dll_out;     // [B][ND] its directional derivatives
    int32_t ll_slot;     // observed compartment: 0 s, 1 e, 2 i, 3 r, 4 c
    int32_t ll_mode;     // 0: rate = value at each save time; 1: rate = increment between save times
    int32_t ll_row;      // values per observation row
    T ll_floor;          // rate = max(rate, floor) (the reference clamps incidence at 1e-6)
    // forward-mode tangents (dyn_solve_batch_jvp); unused when the kernel's ND == 0
    const T *dparams; // [B][ND][P] seed directions of the parameter vector
    const T *dy0;     // [ND][D] or [B][ND][D] seeds of the initial state, or nullptr (= 0)
    T *dout;          // [B][n_save][ND][D_saved]
    int32_t dy0_batched;
    // discontinuity_points (ClipStepSizeController jump_ts), passed by value in the kernarg
    int32_t n_jump;
    T jump_ts[64];
    // Step schedules (SEIP family, dyn_solve_batch_record / _replay): an adaptive solve can write down the (t_prev, t_next)
    // of every accepted step, and a later launch can make other parameter rows take exactly those steps -- the discrete
    // map is then smooth in the parameters, which is what differentiating through the reference's solve assumes (the
    // step-size controller is not differentiated) and what central differences need.
    T *sched_out;               // [B][sched_cap][2] accepted steps of this launch, or nullptr
    int32_t *sched_n_out;       // [B] how many (-1: more than sched_cap)
    const T *sched_in;          // [n_leaders][sched_cap][2] steps to replay, or nullptr (adaptive / constant stepping)
    const int32_t *sched_n_in;  // [n_leaders]
    const int64_t *sched_leader; // [B] schedule row of each trajectory, or nullptr (= its own index)
    int32_t sched_cap;
    // Replication for small batches: 2^rep_log2 lane groups integrate the SAME trajectory
    // (bit-identical redundant stepping on otherwise idle SIMDs) and split its save times
    // round-robin, which divides the serial dense-output latency of a trajectory.
    int32_t rep_log2;
    // Dispatch order (dyn_solve_batch_ordered): slot i of the grid integrates trajectory order[i] -- a permutation of
    // 0..B-1, or nullptr for the identity.  Results do not depend on it (every trajectory is computed from its own inputs
    // alone and written to its own rows); launch TIME does: the lane groups of a wave step in lock-step and waves are
    // dispatched in index order, so neighbours with similar step counts and the expensive ones first cost 10-15 % less.
    const int32_t *order;
    // Work pulling (dyn_solver_opts::work_counter): two zeroed int32 words in device memory, or nullptr for a static launch.
    // work[0] hands out queue entries beyond the first one of every slot, work[1] counts retired slots; the kernel leaves
    // both at zero again (Solver::run).
    int32_t *work;
    // Fused sampler iteration (dyn_solver_opts::nuts_tail; FUSED instances only): the sampler run whose chains this batch's
    // trajectories belong to (HOST memory: launch() hands the struct to the kernel by value; the kernel only tests the
    // pointer), or nullptr.
    const dynnuts::Tail *nuts_tail;
    // dyn_solver_opts::hints.pull / .pull_waves, for launch() (host side only; the kernels never read them)
    int32_t pull_mode, pull_waves;
};
constexpr int kMaxJumps = 64;   // (a year of weekly interventions: 52)
// both arguments of a fused launch travel by value: 4 KB of kernel-argument segment hold them (3144 bytes in float64)
static_assert(sizeof(KArgs<double>) + sizeof(dynnuts::Tail) + 64 <= 4096, "KArgs + Tail exceed the kernel-argument segment");

// ---------------------------------------------------------------- math per precision
template <typename T>
struct Mth;
template <>
struct Mth<float> {
    static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
    // max(|a|, |b|) as ONE instruction: written with fmaxf / fabsf the compiler first quiets each operand (v_max x, x), three
    // instructions per element of the error norm; v_max_f32 itself already returns the other operand for a NaN
    static __device__ __forceinline__ float max_abs(float a, float b) {
        float r;
        asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    // sin(x) for the seasonal forcing 1 + amp sin(w t + phase), evaluated in every right-hand side.  The library sinf is ~32
    // vector instructions and two branches (octant reduction, BOTH polynomials, selection, the Payne-Hanek path for huge
    // arguments behind a branch): a fifth of the instructions of the seasonal 8 x 4 kernel (BASELINE cfg 5).  Here: x - n pi in
    // three Cody-Waite FMAs (exact products for |n| < 2^16, i.e. |x| < 2e5 rad: tens of thousands of years of a yearly
    // cycle), the odd Taylor polynomial to x^13 on [-pi/2, pi/2] (truncation 7e-10), the sign from n's parity: 17
    // instructions, no branch.  Max |error| 1.2e-7 (1.9 ulp at |sin| ~ 1, against libm's 1.5), checked against float64 over
    // [-2e4, 2e4] (tests/test_tableau.py: the same arithmetic in NumPy; tests/test_gpu_parity.py: the kernel against it).
    static __device__ __forceinline__ float sin(float x) {
        const float n = __builtin_rintf(x * 0.318309886183790671538f);
        float r = __builtin_fmaf(n, -3.140625f, x);
        r = __builtin_fmaf(n, -9.67502593994140625e-4f, r);
        r = __builtin_fmaf(n, -1.509957990978376432e-7f, r);
        const float r2 = r * r;
        float p = 1.0f / 6227020800.0f;
        p = __builtin_fmaf(p, r2, -1.0f / 39916800.0f);
        p = __builtin_fmaf(p, r2, 1.0f / 362880.0f);
        p = __builtin_fmaf(p, r2, -1.0f / 5040.0f);
        p = __builtin_fmaf(p, r2, 1.0f / 120.0f);
        p = __builtin_fmaf(p, r2, -1.0f / 6.0f);
        const float s = __builtin_fmaf(r, p * r2, r);
        const unsigned flip = (unsigned)(int)n << 31;
        return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s) ^ flip);
    }
    static __device__ __forceinline__ float sin_lib(float x) { return sinf(x); }   // (the SEIP family: its seasonal terms sit under kinks)
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    // 1-ulp hardware reciprocal: used only for the error-norm scaling
    static __device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
    // 1/x to ~1 ulp: v_rcp_f32 + one Newton step (3 VALU ops instead of the 10-op IEEE sequence)
    static __device__ __forceinline__ float recip(float x) {
        const float r = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    }
    // x^e through v_log_f32 / v_exp_f32: only ever sets the next step size
    // a / b for quantities that only steer the step-size controller (starting-step norms): multiply by the refined reciprocal
    static __device__ __forceinline__ float div_ctl(float a, float r) { return a * r; }
    static __device__ __forceinline__ float rcp_ctl(float b) { return recip(b); }
    static __device__ __forceinline__ float pow_fast(float x, float e) {
        return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));
    }
    static __device__ __forceinline__ float exp(float x) { return expf(x); } // external-introduction pulse
    // natural log through v_log_f32 (fused Poisson likelihood terms; sums are kept in float64)
    static __device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float next(float x, float to) { return nextafterf(x, to); }
    static constexpr float clip_tol = 1e-6f; // diffrax _clip_to_end tolerance, float32
};
template <>
struct Mth<double> {
    static __device__ __forceinline__ double abs(double x) { return fabs(x); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return fmin(a, b); }
    static __device__ __forceinline__ double max_abs(double a, double b) {
        double r;
        asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double sin_lib(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ double rcp_fast(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double recip(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double div_ctl(double a, double r) { return a / r; }   // (float64: the oracle's division)
    static __device__ __forceinline__ double rcp_ctl(double b) { return b; }
    static __device__ __forceinline__ double pow_fast(double x, double e) { return ::pow(x, e); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double next(double x, double to) { return nextafter(x, to); }
    static constexpr double clip_tol = 1e-10;
};

// ---------------------------------------------------------------- tableaux
// Constants identical to oracle/dynode_oracle.c (checked against the RK order conditions
// in tests/test_tableau.py).
template <int METHOD>
struct Tab;
template <>
struct Tab<0> { // Tsit5
    static constexpr double c[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
    static constexpr double a[7][7] = {
        {0},
        {0.161},
        {-0.008480655492356989, 0.335480655492357},
        {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
        {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
        {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
         -0.028269050394068383},
        {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
         2.324710524099774}};
    static constexpr double berr[7] = {0.00178001105222577714, 0.0008164344596567469,
                                       -0.007880878010261995,  0.1447110071732629,
                                       -0.5823571654525552,    0.45808210592918697,
                                       -0.015151515151515152};
    // dense output: the weights b_i(theta) of Tsitouras' interpolant (the factored forms in Solver::dense_prepare) expanded
    // in powers of theta; bp[i][m - 2] is the coefficient of theta^m, m = 2, 3, 4.  The theta^1 coefficient is 1 for i = 0
    // (y'(t_prev) = f_0) and 0 for every other stage.  tests/test_tableau.py checks the expansion against the factored forms.
    static constexpr double bp[7][3] = {{-2.763706197274826, 2.9132554618219126, -1.0530884977290216},
                                        {0.13169999999999998, -0.2234, 0.1017},
                                        {3.9302962368947516, -5.941033872131505, 2.490627285651253},
                                        {-12.411077166933676, 30.33818863028232, -16.548102889244902},
                                        {37.50931341651104, -88.1789048947664, 47.37952196281928},
                                        {-27.896526289197286, 65.09189467479366, -34.87065786149661},
                                        {1.5, -4.0, 2.5}};
};
template <>
struct Tab<1> { // Dopri5
    static constexpr double c[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
    static constexpr double a[7][7] = {
        {0},
        {1.0 / 5},
        {3.0 / 40, 9.0 / 40},
        {44.0 / 45, -56.0 / 15, 32.0 / 9},
        {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
        {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
    static constexpr double berr[7] = {35.0 / 384 - 5179.0 / 57600,
                                       0.0,
                                       500.0 / 1113 - 7571.0 / 16695,
                                       125.0 / 192 - 393.0 / 640,
                                       -2187.0 / 6784 + 92097.0 / 339200,
                                       11.0 / 84 - 187.0 / 2100,
                                       -1.0 / 40};
    static constexpr double cmid[7] = {6025192743.0 / 30085553152.0 / 2,
                                       0.0,
                                       51252292925.0 / 65400821598.0 / 2,
                                       -2691868925.0 / 45128329728.0 / 2,
                                       187940372067.0 / 1594534317056.0 / 2,
                                       -1776094331.0 / 19743644256.0 / 2,
                                       11237099.0 / 235043384.0 / 2};
};

// ---------------------------------------------------------------- cross-lane helpers
// xchg_xor<K>(v): lane l receives the value held by lane l ^ K.  Built from DPP row
// permutations (VALU operand modifiers: no LDS pipe, no lgkmcnt wait) composed as
//   K in 1..3  : quad_perm            K in 4..7  : row_half_mirror (l -> l^7) then quad_perm
//   K = 8      : row_ror:8;   K in 9..15 : row_mirror (l -> l^15) then the K^15 case
//   K in 16..31: ds_swizzle xor 16 first;   K >= 32: ds_bpermute xor 32 first.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float swz_xor16(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ double swz_xor16(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)b, 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), 0x401F);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int K, typename T>
__device__ __forceinline__ T xchg_xor(T v) {
    static_assert(K >= 0 && K < 64, "lane xor out of range");
    if constexpr (K == 0) return v;
    else if constexpr (K == 1) return dpp_mov<0xB1>(v); // quad_perm [1,0,3,2]
    else if constexpr (K == 2) return dpp_mov<0x4E>(v); // quad_perm [2,3,0,1]
    else if constexpr (K == 3) return dpp_mov<0x1B>(v); // quad_perm [3,2,1,0]
    else if constexpr (K < 8) return xchg_xor<(K ^ 7)>(dpp_mov<0x141>(v));   // row_half_mirror
    else if constexpr (K == 8) return dpp_mov<0x128>(v);                     // row_ror:8 -- in a row of 16, l + 8 mod 16 = l ^ 8: one move, not two
    else if constexpr (K < 16) return xchg_xor<(K ^ 15)>(dpp_mov<0x140>(v)); // row_mirror
    else if constexpr (K < 32) return xchg_xor<(K ^ 16)>(swz_xor16(v));
    else return xchg_xor<(K ^ 32)>(__shfl_xor(v, 32, 64));
}

// acc_a += sum_j c_j * xa[lane ^ j], acc_b likewise for xb, j = 0..3: the quad_perm rides on
// the FMA itself (v_fmac_f32_dpp, ~the cost of a plain FMA; a separate v_mov_dpp costs 2x).
// The two plain FMAs in front give the >= 2 wait states a DPP read needs after a VALU write
// of its source (hipcc pads nothing inside an asm statement).  EXEC must be all ones.
__device__ __forceinline__ void fma_quad_pair(float &acc_a, float &acc_b, float xa, float xb,
                                              float c0, float c1, float c2, float c3) {
    asm("v_fmac_f32_e32 %0, %2, %4\n\t"
        "v_fmac_f32_e32 %1, %3, %4\n\t"
        "v_fmac_f32_dpp %0, %2, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %7 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %7 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
        : "+v"(acc_a), "+v"(acc_b)
        : "v"(xa), "v"(xb), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
}
// the same with acc_a = c0 * xa, acc_b = c0 * xb as the first term (base 0 of the gather: no zero-initialised accumulator)
__device__ __forceinline__ void mul_quad_pair(float &acc_a, float &acc_b, float xa, float xb,
                                              float c0, float c1, float c2, float c3) {
    asm("v_mul_f32_e32 %0, %2, %4\n\t"
        "v_mul_f32_e32 %1, %3, %4\n\t"
        "v_fmac_f32_dpp %0, %2, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %7 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %3, %7 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
        : "=&v"(acc_a), "=&v"(acc_b)
        : "v"(xa), "v"(xb), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
}
__device__ __forceinline__ void fma_quad_one(float &acc, float x, float c0, float c1, float c2,
                                             float c3) {
    asm("v_fmac_f32_e32 %0, %1, %2\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(x), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
}
// generic (double / fallback) versions of the same contraction
template <typename T>
__device__ __forceinline__ void fma_quad_one(T &acc, T x, T c0, T c1, T c2, T c3) {
    acc += c0 * x;
    acc += c1 * xchg_xor<1>(x);
    acc += c2 * xchg_xor<2>(x);
    acc += c3 * xchg_xor<3>(x);
}
template <typename T>
__device__ __forceinline__ void fma_quad_pair(T &acc_a, T &acc_b, T xa, T xb, T c0, T c1, T c2,
                                              T c3) {
    fma_quad_one<T>(acc_a, xa, c0, c1, c2, c3);
    fma_quad_one<T>(acc_b, xb, c0, c1, c2, c3);
}

template <typename T>
__device__ __forceinline__ void mul_quad_pair(T &acc_a, T &acc_b, T xa, T xb, T c0, T c1, T c2, T c3) {
    acc_a = T(0);
    acc_b = T(0);
    fma_quad_pair<T>(acc_a, acc_b, xa, xb, c0, c1, c2, c3);
}

template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
    if constexpr (G >= 2) v += xchg_xor<1>(v);
    if constexpr (G >= 4) v += xchg_xor<2>(v);
    if constexpr (G >= 8) v += xchg_xor<4>(v);
    if constexpr (G >= 16) v += xchg_xor<8>(v);
    if constexpr (G >= 32) v += xchg_xor<16>(v);
    if constexpr (G >= 64) v += xchg_xor<32>(v);
    return v;
}

// store CNT contiguous values; 16-byte (or 8-byte) vector stores when the host proved alignment
template <typename T, int CNT>
__device__ __forceinline__ void store_run(T *p, const T (&v)[CNT], bool vec_ok) {
    constexpr int PER = 16 / sizeof(T);
    if constexpr (sizeof(T) == 4 && CNT % 4 != 0 && CNT % 2 == 0) {
        if (vec_ok) {
            typedef T vec2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < CNT / 2; ++q) {
                vec2_t pk;
                pk[0] = v[2 * q];
                pk[1] = v[2 * q + 1];
                *reinterpret_cast<vec2_t *>(p + 2 * q) = pk;
            }
            return;
        }
    }
    if constexpr (CNT % PER == 0) {
        if (vec_ok) {
            typedef T vec_t __attribute__((ext_vector_type(PER)));
#pragma unroll
            for (int q = 0; q < CNT / PER; ++q) {
                vec_t pk;
#pragma unroll
                for (int z = 0; z < PER; ++z) pk[z] = v[q * PER + z];
                *reinterpret_cast<vec_t *>(p + q * PER) = pk;
            }
            return;
        }
    }
#pragma unroll
    for (int q = 0; q < CNT; ++q) p[q] = v[q];
}

// ---------------------------------------------------------------- step-size control (shared by the kernel families)
// diffrax's defaults as called from the reference (odes.py:107-144): I-controller with safety 0.9 and factor in
// [0.2, 10], Hairer-Norsett-Wanner starting step, clip-to-end.  solve_kernel.hpp and seip_kernel.hpp both step with
// these, so the two families cannot drift apart.
template <typename T>
struct Control {
    using M = Mth<T>;
    // second half of the starting-step heuristic: h1 from max(d1, d2)
    template <bool STRICT = false>
    static __device__ __forceinline__ T initial_h1(T max_d, T h0) {
        if constexpr (STRICT && sizeof(T) == 4) return (max_d <= T(1e-15)) ? M::max(T(1e-6), h0 * T(1e-3)) : powf(T(0.01) / max_d, T(0.2));
        else return (max_d <= T(1e-15)) ? M::max(T(1e-6), h0 * T(1e-3)) : M::pow_fast(T(0.01) / max_d, T(0.2));
    }
    // error norm -> accept?, can the solve go on?, next step factor.  A NaN / inf estimate is a rejected step with
    // infinite error (diffeqsolve replaces NaN by inf before the controller); the solve only fails when the step cannot
    // shrink any further.  factor = clip(safety * err^(-1/order), keep ? 1 : factormin, factormax); err == 0 -> factormax
    // STRICT (test-only instances, FEAT bit 16 / Seip OPT bit 1): the oracle's float32 arithmetic to the letter -- libm's
    // correctly rounded powf instead of v_log_f32 / v_exp_f32 -- so that what the fast controller arithmetic costs in
    // accept / reject decisions can be measured (tests/test_gpu_parity.py)
    template <bool STRICT = false>
    static __device__ __forceinline__ void decide(T err, T tprev, T dt, bool &keep, bool &finite, T &factor) {
        if (!(err == err)) err = M::inf();
        keep = err < T(1);
        finite = !(err == M::inf() && !(tprev + T(0.2) * dt > tprev));
        T f;
        if constexpr (STRICT && sizeof(T) == 4) f = T(0.9) * powf(err, T(-0.2));
        else f = T(0.9) * M::pow_fast(err, T(-0.2));
        f = M::max(f, keep ? T(1) : T(0.2));
        factor = M::min(f, T(10));
    }
    // the same from the MEAN SQUARE of the scaled error, err = sqrt(ms): float32 goes without the square root --
    // err < 1 <=> ms < 1, err^(-1/5) = ms^(-1/10) through the same v_log_f32 / v_exp_f32 pair -- which is ten instructions of
    // every step attempt (the correctly rounded sqrtf is a v_sqrt_f32 plus denormal scaling and two refinement steps);
    // float64 keeps the oracle's operation order (its step counts are compared one for one)
    static __device__ __forceinline__ void decide_ms(T ms, T tprev, T dt, bool &keep, bool &finite, T &factor) {
        if constexpr (sizeof(T) == 4) {
            if (!(ms == ms)) ms = M::inf();
            keep = ms < T(1);
            finite = !(ms == M::inf() && !(tprev + T(0.2) * dt > tprev));
            T f = T(0.9) * M::pow_fast(ms, T(-0.1));
            f = M::max(f, keep ? T(1) : T(0.2));
            factor = M::min(f, T(10));
        } else {
            decide(M::sqrt(ms), tprev, dt, keep, finite, factor);
        }
    }
    // diffeqsolve's clip-to-end; returns true when the step was clipped (a pending jump clip is then void)
    static __device__ __forceinline__ bool clip_to_end(T &next_t1, T tp, bool accept, T t_end) {
        if (next_t1 > t_end - M::clip_tol) {
            next_t1 = accept ? t_end : tp + T(0.5) * (t_end - tp);
            return true;
        }
        return false;
    }
};

// ---------------------------------------------------------------- kernel arguments of the cold paths
// The stepping loop is entered thousands of times per launch, the prologue / write-off of a trajectory once per trajectory.
// Arguments only those need (the input pointers, the batch size, the status arrays, ...) would otherwise sit in scalar
// registers -- or in spilled ones, a v_readlane each -- through every iteration.  cold_args() hands back the kernel's argument
// block (the KArgs struct is the kernel's only parameter, at offset 0 of the kernarg segment) through a pointer the compiler
// cannot see through, so that those fields are re-read with scalar loads where they are used.
template <typename T>
__device__ __forceinline__ const KArgs<T> __attribute__((address_space(4))) *cold_args() {
    auto p = (const KArgs<T> __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// ---------------------------------------------------------------- per-lane state in register pairs
// One plane of the per-lane state (or of a stage derivative): N values kept as ceil(N / 2) two-element vectors, so that the
// register allocator holds them in aligned register pairs from the start and the linear algebra of the stepper -- the
// stage combinations, the embedded error, the dense output -- runs on v_pk_fma_f32 without the pair-packing moves the
// vectoriser otherwise inserts (12 % of the vector instructions of the D = 360 kernel).  operator[] gives the right-hand
// side element access; an odd N leaves one pad element, which stays 0.
template <typename T, int N>
struct PairState {
    static constexpr int NP = (N + 1) / 2;
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 p[NP];
    struct Ref {
        V2 &v;
        int i;
        __device__ __forceinline__ operator T() const { return v[i]; }
        __device__ __forceinline__ Ref &operator=(T x) {
            v[i] = x;
            return *this;
        }
        __device__ __forceinline__ Ref &operator=(const Ref &o) {
            v[i] = (T)o;
            return *this;
        }
        __device__ __forceinline__ Ref &operator+=(T x) {
            v[i] += x;
            return *this;
        }
        __device__ __forceinline__ Ref &operator-=(T x) {
            v[i] -= x;
            return *this;
        }
    };
    __device__ __forceinline__ T operator[](int e) const { return p[e >> 1][e & 1]; }
    __device__ __forceinline__ Ref operator[](int e) { return Ref{p[e >> 1], e & 1}; }
    __device__ __forceinline__ void clear_pad() {
        if constexpr (N % 2 == 1) p[NP - 1][1] = T(0);
    }
};

// ---------------------------------------------------------------- the kernel
// ND > 0 adds forward-mode tangents: NC = 1 + ND "planes" of every state array, plane 0 the
// primal.  RK stages, FSAL and the dense output are linear in (y, k), so they are applied
// plane by plane; only the RHS is non-linear and carries an explicit JVP.  Step sizes and the
// accept/reject decision come from the primal alone (what jax.grad through diffrax does: the
// controller is under stop_gradient), so the tangents are the exact derivative of the computed
// trajectory ("discretise-then-optimise"), see dyn_solve_batch_jvp in include/dynode_hip.h.
// Lane mapping: a trajectory owns GA x GS lanes -- GA = power of two >= n_age "age lanes" (low bits
// of the lane id) times GS = S / SPL "strain lanes"; lane (a, h) holds s_a (replicated over h) and
// the e/i/r/c values of age a for its SPL strains h*SPL .. h*SPL+SPL-1.  SPL == S is the plain
// one-lane-per-age mapping; splitting the strains (SPL < S) divides the per-lane state, trading
// replicated control arithmetic for occupancy and smaller lock-step groups.
// FEAT (optional model features): bit 0 = externally introduced strains; bits 1.. = KV, the lane width
// of the vaccination-tier axis (0 = none, 2 or 4): the contact axis then enumerates (age, tier) groups,
// tier in the low bits, and susceptibles flow from tier k to k + 1 at a spline-in-time rate.
template <typename T, int METHOD, int GA, int ST, bool HAS_E, bool HAS_WANE, bool HAS_C, int W, int ND, int SPL,
          int FEAT = 0>
struct Solver {
    static constexpr bool INTRO = (FEAT & 1) != 0;
    static constexpr int KV = (FEAT >> 1) & 0x7f;
    // bit 14: every compartment is saved into 16-byte aligned rows (the host checks) -- the per-round tests of
    // the save offsets and of the store width fold away (the save loop is half of a daily-output solve)
    static constexpr bool SAVE_ALL = (FEAT & 0x4000) != 0;
    // bit 15: stepping and dense output on two waves of one workgroup ("Producer / consumer" at Solver::run) -- for launches of
    // at most one wave per SIMD, where a lone wave leaves a third of the SIMD's issue slots empty
    static constexpr bool PC = (FEAT & 0x8000) != 0;
    static_assert(!PC || (ND == 0 && KV == 0), "producer / consumer: primal kernels with step-scaled rates");
    // bit 13: the gradient-solve of the reference's inference example as compile-time facts instead of run-time switches --
    // normalised force of infection, no seasonal forcing, no discontinuity points, adaptive steps, the Poisson likelihood of
    // the increments of r fused in (examples/sir_infer_parameters.py:21-39).  A sampler iteration is ONE wave's serial
    // instruction stream (128 chains on 1024 SIMDs): every switch the wave does not have to evaluate is latency.
    static constexpr bool LEAN = (FEAT & 0x2000) != 0;
    static_assert(!LEAN || (ND > 0 && KV == 0 && !INTRO), "lean instance: tangent kernels of the plain family");
    // bits 17-19 (with bit 13): WHICH compartment the lean instance scores -- 0: r (the reference's inference example), 1: c
    // (the multi-strain inference example, examples/infer_multi_strain.py), 2: i, 3: e, 4: s -- and bit 20: its VALUES at the
    // save times instead of its increments between them.  Twins other than the two examples' are built on first use
    // (dynode_amd/jit.py ensure_lean_twin).
    static constexpr int LEAN_CODE = (FEAT >> 17) & 0x7;
    static constexpr int LEAN_SLOT = LEAN_CODE == 0 ? 3 : LEAN_CODE == 1 ? 4 : LEAN_CODE == 2 ? 2 : LEAN_CODE == 3 ? 1 : 0;
    static constexpr int LEAN_MODE = (FEAT & 0x100000) != 0 ? 0 : 1;
    static_assert((FEAT & 0x1E0000) == 0 || LEAN, "bits 17-20 qualify a lean instance");
    static_assert(LEAN_CODE <= 4 && (LEAN_SLOT != 4 || HAS_C) && (LEAN_SLOT != 1 || HAS_E), "lean instance: the scored compartment exists");
    // FEAT bit 12: a tangent instance (lean or general: any observed compartment / likelihood mode the fused likelihood takes)
    // whose waves, once their trajectories are scored, run the NUTS state machine of the chains those trajectories belong to
    // (nuts_device.hpp) -- the sampler iteration as one launch.  Static grids only.
    static constexpr bool FUSED = (FEAT & 0x1000) != 0;
    static_assert(!FUSED || (ND > 0 && KV == 0), "the fused sampler tail rides on a gradient-solve (tangent instance)");
    // ... with the state machines up to eight dimensions, except on the cfg 4 lean instance (four: nuts_device.hpp)
    static constexpr int kTailMaxDim = LEAN && LEAN_CODE == 0 && LEAN_MODE == 1 ? dynnuts::kFusedLeanMaxDim : dynnuts::kFusedMaxDim;
    // FEAT bit 16 (test-only instances): the step controller in the oracle's arithmetic -- IEEE division, sqrtf, powf (stepper.hpp)
    static constexpr bool STRICT_CONTROL = (FEAT & 0x10000) != 0;
    // where solve_kernel_fused's second argument (dynnuts::Tail, by value) sits in the kernel-argument segment
    static constexpr size_t kTailOffset = (sizeof(KArgs<T>) + alignof(dynnuts::Tail) - 1) / alignof(dynnuts::Tail) * alignof(dynnuts::Tail);
    static_assert(KV == 0 || ((KV == 2 || KV == 4) && GA % KV == 0), "vaccination tiers: 2 or 4 lanes per age");
    static_assert(ST % SPL == 0, "strains per lane must divide the strain count");
    static constexpr int S = SPL;        // strains held by one lane (all per-lane arrays use S)
    static constexpr int GS = ST / SPL;  // strain lanes
    static constexpr int G = GA * GS;    // lanes per trajectory
    static_assert(G <= 64 && (GS & (GS - 1)) == 0, "lane group must be a power of two <= 64");
    static constexpr int NE = HAS_E ? S : 0;
    static constexpr int NCU = HAS_C ? S : 0;
    static constexpr int NV = 1 + NE + S + S * W + NCU;
    // per-lane element order: e | i | r | c | s -- the strain blocks first, so that they start on even indices (pairs for
    // the packed stepper arithmetic, whole 16-byte runs for the stores), the lone s last (next to the pad when NV is odd)
    static constexpr int IE = 0, II = NE, IR = II + S, IC = IR + S * W, IS = IC + NCU;
    static constexpr int TPW = 64 / G;
    static constexpr int NC = 1 + ND;          // planes
    // Small states are latency-bound by the serial save rounds: interpolate SU save times per
    // round (all computed for ILP, stores predicated) instead of one.
    static constexpr int SU = (NV * NC <= 4) ? 4 : (NV * NC <= 9 ? 2 : 1);
    static constexpr int NDA = ND > 0 ? ND : 1; // array extent for tangent-only data
    using Scalar = T;
    using M = Mth<T>;
    using TB = Tab<METHOD>;

    // the per-lane state, the stage state and the stage derivatives live in register pairs (PairState above)
    static constexpr int NP = (NV + 1) / 2;
    typedef T V2 __attribute__((ext_vector_type(2)));
    using State = PairState<T, NV>;

    // per-lane model data
    T beta[S], gamma[S], sigma[S], omega[S];
    // the same rates as register pairs over the strains (rhs_strain_pairs; an instantiation keeps whichever form it reads)
    static constexpr bool PAIRED_RHS = sizeof(T) == 4 && S % 2 == 0 && W == 1 && ND == 0 && KV == 0 && !INTRO;
    static constexpr int SP = PAIRED_RHS ? S / 2 : 1;
    V2 beta2[SP], gamma2[SP], sigma2[SP], omega2[SP];
    T Cx[GA]; // Cx[k] = contact[a][a ^ k] (0 outside the matrix)
    T amp, phase, w_season;
    // external introductions (INTRO): peak day, 1 / scale, percentage / (scale sqrt(2 pi)) * [age receives it]
    T itime[INTRO ? S : 1], iinv[INTRO ? S : 1], iamp[INTRO ? S : 1];
    T ibase[INTRO ? S : 1];  // iamp / percentage (tangents only)
    // parameter seeds per direction
    T ditime[INTRO ? NDA : 1][INTRO ? S : 1], discale[INTRO ? NDA : 1][INTRO ? S : 1], dipct[INTRO ? NDA : 1][INTRO ? S : 1];
    T dbeta[NDA][S], dgamma[NDA][S], dsigma[NDA][S], domega[NDA][S];
    T damp[NDA], dphase[NDA], dw_season[NDA];
    // vaccination tiers (KV): susceptibility 1 - efficacy of this group's tier per strain; this group's
    // vaccination-rate spline a + b t + c t^2 + d t^3 + sum_i coef_i (t - knot_i)^3 [t > knot_i]
    T sus[KV ? S : 1], vbase[KV ? 4 : 1], vknot[KV ? 4 : 1], vcoef[KV ? 4 : 1];
    T dsus[KV ? NDA : 1][KV ? S : 1], dvbase[KV ? NDA : 1][KV ? 4 : 1], dvknot[KV ? NDA : 1][KV ? 4 : 1],
        dvcoef[KV ? NDA : 1][KV ? 4 : 1]; // their seeds per direction
    int vnk;
    bool vax_top, vax_first;  // last tracked tier (nobody leaves) / tier 0 (nobody arrives)
    bool pad, normalize, seasonal;
    bool lead; // strain lane 0: owns the replicated s for norms and stores

    // sum over the strain lanes of the group (identical result in every lane)
    __device__ __forceinline__ static T strain_sum(T v) {
        if constexpr (GS >= 2) v += xchg_xor<GA>(v);
        if constexpr (GS >= 4) v += xchg_xor<2 * GA>(v);
        if constexpr (GS >= 8) v += xchg_xor<4 * GA>(v);
        if constexpr (GS >= 16) v += xchg_xor<8 * GA>(v);
        return v;
    }

    // ---- step-scaled rates (PRESCALE).  Every term of the right-hand side is a rate times state values, so with the
    // rates multiplied by the step size beforehand the right-hand side returns K = dt f directly: the stage combinations
    // become y + sum a K (one packed FMA less per register pair and stage), the embedded error and the dense output lose
    // their dt factors as well.  The rates as loaded are parked in LDS ([quad][lane] rows of four: conflict-free 16-byte
    // reads) and re-scaled from there for every step attempt -- scaling the scaled copy by dt_new / dt_old instead would
    // let rounding accumulate in the PARAMETERS over a hundred steps.  FSAL: k[0] carries the factor of the step it was
    // computed in and is rescaled by dt_new / dt_old (Solver::run).  Every kernel without vaccination tiers (whose dose
    // cap min(doses, s) is not linear in a rate).
#ifdef DYN_DIAG_NOPRESCALE_ND      // diagnostic build (tools/diag_build.sh NOPRESCALE_ND): tangent kernels keep k = f
    static constexpr bool PRESCALE = KV == 0 && ND == 0;
#else
    static constexpr bool PRESCALE = KV == 0;
#endif
    typedef T V4 __attribute__((ext_vector_type(4)));
    // quads per lane: beta[S] | gamma[S] | sigma[S] | omega[S], then the same four blocks of every seed direction (the
    // tangent of a rate-times-state term is d(rate) state + rate d(state): both carry the factor)
    static constexpr int NRQ = S * (1 + ND);
    __device__ __forceinline__ T &rate_ref(int i) {
        const int plane = i / (4 * S), r = i % (4 * S), j = plane > 0 ? plane - 1 : 0;
        if (plane == 0) return r < S ? beta[r] : (r < 2 * S ? gamma[r - S] : (r < 3 * S ? sigma[r - 2 * S] : omega[r - 3 * S]));
        return r < S ? dbeta[j][r] : (r < 2 * S ? dgamma[j][r - S] : (r < 3 * S ? dsigma[j][r - 2 * S] : domega[j][r - 3 * S]));
    }
    __device__ __forceinline__ void park_rates(V4 *tab, int lane) {
#pragma unroll
        for (int q = 0; q < NRQ; ++q)
            tab[q * 64 + lane] = V4{rate_ref(4 * q), rate_ref(4 * q + 1), rate_ref(4 * q + 2), rate_ref(4 * q + 3)};
    }
    __device__ __forceinline__ void scale_rates(const V4 *tab, int lane, T h) {
#pragma unroll
        for (int q = 0; q < NRQ; ++q) {
            const V4 v = tab[q * 64 + lane];
            const V2 lo = V2{v[0], v[1]} * h, hi = V2{v[2], v[3]} * h;   // two packed multiplies per quad
            rate_ref(4 * q) = lo[0];
            rate_ref(4 * q + 1) = lo[1];
            rate_ref(4 * q + 2) = hi[0];
            rate_ref(4 * q + 3) = hi[1];
        }
        if constexpr (PAIRED_RHS) {
#pragma unroll
            for (int q = 0; q < SP; ++q) {
                beta2[q] = V2{beta[2 * q], beta[2 * q + 1]};
                gamma2[q] = V2{gamma[2 * q], gamma[2 * q + 1]};
                sigma2[q] = V2{sigma[2 * q], sigma[2 * q + 1]};
                omega2[q] = V2{omega[2 * q], omega[2 * q + 1]};
            }
        }
    }

    // acc_l = sum_k Cx[k] * x_l[lane ^ k]: all-gather over the lane group fused with the
    // pre-permuted contact row
    __device__ __forceinline__ void contract(const T (&x)[S], T (&acc)[S]) const {
#pragma unroll
        for (int l = 0; l < S; ++l) acc[l] = T(0);
        if constexpr (GA < 4) {
#pragma unroll
            for (int l = 0; l < S; ++l) {
                acc[l] = Cx[0] * x[l];
                if constexpr (GA == 2) acc[l] += Cx[1] * xchg_xor<1>(x[l]);
            }
        } else {
            gather_base<0>(x, acc);
        }
    }

    // ---- the age's population, once per step attempt.  Every flow of these models moves people between compartments of the
    // same age (infection, progression, recovery, waning; visitors are never added to the state), so s + e + i + r of an age --
    // and the sum of every tangent plane -- is the same at the six stage states of a step in exact arithmetic: each stage
    // derivative sums to zero.  The reference recomputes it in every right-hand side (examples/*.py: `pop = s + e + i + r`);
    // here it is formed from the step's starting state (Stepper: begin_attempt) and reused by the stages, which tracks the
    // state's rounding drift like the reference does, one step late, and drops ~12 of each right-hand side's instructions
    // (a cross-lane sum and a reciprocal among them).  Not with vaccination tiers: there the contact groups are (age, tier)
    // and doses move people between them.
    static constexpr bool POP_PER_ATTEMPT = KV == 0;
    T pop_N, pop_invN, pop_dN[NDA];
    __device__ __forceinline__ T population(const State &y0) const {
        if constexpr (PAIRED_RHS) {
            constexpr int PE = IE / 2, PI = II / 2, PR = IR / 2;
            V2 tot = y0.p[PI];
#pragma unroll
            for (int q = 1; q < SP; ++q) tot += y0.p[PI + q];
#pragma unroll
            for (int q = 0; q < SP; ++q) {
                if constexpr (HAS_E) tot += y0.p[PE + q];
                tot += y0.p[PR + q];
            }
            return y0[IS] + strain_sum(tot[0] + tot[1]);
        } else {
            // everyone of this lane's strains who is not susceptible: the elements e | i | r sit in front of c and s (0 .. IC - 1);
            // whole register pairs are added pairwise (packed adds), the two halves and an odd last element at the end
            T others;
            if constexpr (IC >= 4) {
                V2 tot = y0.p[0];
#pragma unroll
                for (int q = 1; q < IC / 2; ++q) tot += y0.p[q];
                others = tot[0] + tot[1];
                if constexpr (IC % 2 == 1) others += y0[IC - 1];
            } else {
                others = y0[0];
#pragma unroll
                for (int v = 1; v < IC; ++v) others += y0[v];
            }
            return y0[IS] + strain_sum(others);
        }
    }
    __device__ __forceinline__ T population_tangent(const State &u) const {
        T dse = 0, dsi = 0, dsr = 0;
#pragma unroll
        for (int l = 0; l < S; ++l) {
            if constexpr (HAS_E) dse += u[IE + l];
            dsi += u[II + l];
#pragma unroll
            for (int w = 0; w < W; ++w) dsr += u[IR + l * W + w];
        }
        return u[IS] + strain_sum((dse + dsi) + dsr);
    }
    // a b + c d with the association pinned (ONE rounding of c d, then an FMA): left to the compiler, which of the two products
    // is fused depends on the code around the expression, and two instances of the same arithmetic (one / two tangent
    // directions per trajectory, the fused sampler launch and the plain one) would differ in the last bit of a gradient
    __device__ __forceinline__ static T two_products(T a, T b, T c, T d) {
        T cd;
        {
#pragma clang fp contract(off)
            cd = c * d;
        }
        return M::fma(a, b, cd);
    }
    // Stepper hook: the state a step attempt starts from
    __device__ __forceinline__ void begin_attempt(const State (&y)[NC]) {
        if constexpr (POP_PER_ATTEMPT) {
            pop_N = population(y[0]);
            pop_invN = T(1);
            if (normalize) pop_invN = pad ? T(0) : M::recip(pop_N);
            if constexpr (ND > 0) {
#pragma unroll
                for (int j = 0; j < ND; ++j) pop_dN[j] = population_tangent(y[1 + j]);
            }
        }
    }

    // f(t, y) for this lane's age bin (plane 0) and its JVP (planes 1..ND);
    // reference RHS: see include/dynode_hip.h
    // f(t, y) written on register pairs over the strains (plain multi-strain shapes in float32: the 8 age x 4 strain models).
    // The same expression tree as rhs() strain by strain; only the sums over the strains of a lane -- the population and the
    // net flow of s -- are formed pairwise first and across the pair last (a different, equally valid summation order).
    __device__ __forceinline__ void rhs_strain_pairs(T t, const State &y0, State &dy) const {
        constexpr int PE = IE / 2, PI = II / 2, PR = IR / 2, PC = IC / 2;
        const T s = y0[IS];
        const T invN = pop_invN;   // (PAIRED_RHS implies KV == 0: the population of the step's starting state, begin_attempt)
        T season = T(1);
        if (__builtin_expect(seasonal, 0)) season = T(1) + amp * M::sin(w_season * t + phase);
        T x[S], acc[S];
#pragma unroll
        for (int q = 0; q < SP; ++q) {
            const V2 xq = y0.p[PI + q] * invN;
            x[2 * q] = xq[0];
            x[2 * q + 1] = xq[1];
        }
        contract(x, acc);
        V2 net = V2{T(0), T(0)};   // what returns to s minus what leaves it, per strain pair
#pragma unroll
        for (int q = 0; q < SP; ++q) {
            const V2 foi = (beta2[q] * season) * V2{acc[2 * q], acc[2 * q + 1]};
            const V2 flux = foi * s;
            const V2 g_i = gamma2[q] * y0.p[PI + q];
            net -= flux;
            if constexpr (HAS_E) {
                const V2 s_e = sigma2[q] * y0.p[PE + q];
                dy.p[PE + q] = flux - s_e;
                dy.p[PI + q] = s_e - g_i;
            } else {
                dy.p[PI + q] = flux - g_i;
            }
            if constexpr (HAS_WANE) {
                const V2 o = omega2[q] * y0.p[PR + q];
                dy.p[PR + q] = g_i - o;
                net += o;
            } else {
                dy.p[PR + q] = g_i;
            }
            if constexpr (HAS_C) dy.p[PC + q] = flux;
        }
        dy[IS] = strain_sum(net[0] + net[1]);
    }

    __device__ __forceinline__ void rhs(T t, const State (&y)[NC], State (&dy)[NC]) const {
        if constexpr (PAIRED_RHS) {
            rhs_strain_pairs(t, y[0], dy[0]);
            return;
        }
        const State &y0 = y[0];
        T N, invN;
        if constexpr (POP_PER_ATTEMPT) { // the population of the step's starting state (begin_attempt)
            N = pop_N;
            invN = pop_invN;
        } else {
            N = population(y0);
            invN = T(1);
            if (normalize) invN = pad ? T(0) : M::recip(N);
        }
        T season = T(1), sin_arg = T(0), cos_arg = T(0);
        if (__builtin_expect(seasonal, 0)) {
            const T arg = w_season * t + phase;
            sin_arg = M::sin(arg);
            if constexpr (ND > 0) cos_arg = M::cos(arg);
            season = T(1) + amp * sin_arg;
        }
        T x[S], acc[S], foi[S];
#pragma unroll
        for (int l = 0; l < S; ++l) x[l] = y0[II + l] * invN;
        T iu[INTRO ? S : 1], ibell[INTRO ? S : 1]; // (t - time) / scale and exp(-u^2 / 2), reused by the tangents
        if constexpr (INTRO) {
            // infectious visitors from an untracked population, mixed in around the introduction day:
            // I_b + Normal(t; time, scale) * percentage * P_b in the force of infection (ode_model.md)
            const T visitors = normalize ? T(1) : N;
#pragma unroll
            for (int l = 0; l < S; ++l) {
                iu[l] = (t - itime[l]) * iinv[l];
                ibell[l] = M::exp(T(-0.5) * iu[l] * iu[l]);
                x[l] += (iamp[l] * ibell[l]) * visitors;
            }
        }
        contract(x, acc);
        T out_s = 0, back_s = 0;
#pragma unroll
        for (int l = 0; l < S; ++l) {
            foi[l] = (beta[l] * season) * acc[l];
            if constexpr (KV > 0) foi[l] *= sus[l];
            const T flux = foi[l] * y0[IS];
            const T g_i = gamma[l] * y0[II + l];
            out_s += flux;
            if constexpr (HAS_E) {
                const T s_e = sigma[l] * y0[IE + l];
                dy[0][IE + l] = flux - s_e;
                dy[0][II + l] = s_e - g_i;
            } else {
                dy[0][II + l] = flux - g_i;
            }
            if constexpr (HAS_WANE) {
                const T wrate = T(W) * omega[l];
                T inflow = g_i;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const T o = wrate * y0[IR + l * W + w];
                    dy[0][IR + l * W + w] = inflow - o;
                    inflow = o;
                }
                back_s += inflow;
            } else {
                dy[0][IR + l * W] = g_i;
#pragma unroll
                for (int w = 1; w < W; ++w) dy[0][IR + l * W + w] = T(0);
            }
            if constexpr (HAS_C) dy[0][IC + l] = flux;
        }
        dy[0][IS] = strain_sum(back_s - out_s);
        T v_rate = T(0), v_nage = T(0);       // max(nu, 0) and the age's population, reused by the tangents
        bool v_nu_pos = false, v_capped = false, v_has_s = false;
        if constexpr (KV > 0) {
            // vaccination (ode_model.md): per day nu_{age,tier}(t) * (population of the age) doses go to
            // the susceptibles of the tier, at most as many as there are; they move up one tier
            T nu = vbase[0] + t * (vbase[1] + t * (vbase[2] + t * vbase[3]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T lag = t - vknot[q];
                if (q < vnk && lag > T(0)) nu += vcoef[q] * (lag * lag * lag);
            }
            T n_age = N + xchg_xor<1>(N);
            if constexpr (KV == 4) n_age += xchg_xor<2>(n_age);
            const T doses = M::max(nu, T(0)) * n_age;
            const T leave = vax_top ? T(0) : M::min(doses, M::max(y0[IS], T(0)));
            v_rate = M::max(nu, T(0));
            v_nage = n_age;
            v_nu_pos = nu > T(0);
            v_has_s = y0[IS] > T(0);
            v_capped = !(doses < M::max(y0[IS], T(0))); // the tier runs empty: everyone left is vaccinated
            // lane of tier k receives what tier k - 1 of the same age gives up
            const T from_below = KV == 2 ? xchg_xor<1>(leave) : dpp_mov<0x90>(leave); // quad_perm [0,0,1,2]
            dy[0][IS] += (vax_first ? T(0) : from_below) - leave;
        }

        // ---- JVP planes: the same expression tree, linearised
        if constexpr (ND > 0) {
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const State &u = y[1 + j];
                State &du = dy[1 + j];
                const T dN = POP_PER_ATTEMPT ? pop_dN[j] : population_tangent(u);
                const T dinvN = normalize ? -(invN * invN) * dN : T(0); // pad: invN == 0
                T dseason = T(0);
                if (seasonal)
                    dseason = damp[j] * sin_arg + amp * cos_arg * (dw_season[j] * t + dphase[j]);
                T dx[S], dacc[S];
#pragma unroll
                for (int l = 0; l < S; ++l) dx[l] = two_products(u[II + l], invN, y0[II + l], dinvN);
                if constexpr (INTRO) {
                    // pulse = pct * base * bell(u), base = mask / (scale sqrt(2 pi)), u = (t - time) / scale:
                    // d/dtime = pulse * u / scale, d/dscale = pulse * (u^2 - 1) / scale, d/dpct = base * bell
#pragma unroll
                    for (int l = 0; l < S; ++l) {
                        const T pulse = iamp[l] * ibell[l];
                        T dp = pulse * iinv[l] * (iu[l] * ditime[j][l] + (iu[l] * iu[l] - T(1)) * discale[j][l]) +
                               (ibase[l] * ibell[l]) * dipct[j][l];
                        dx[l] += normalize ? dp : dp * N + pulse * dN;
                    }
                }
                contract(dx, dacc);
                T dout_s = 0, dback_s = 0;
#pragma unroll
                for (int l = 0; l < S; ++l) {
                    const T bs = beta[l] * season;
                    const T dbs = dbeta[j][l] * season + beta[l] * dseason;
                    T dfoi = two_products(dbs, acc[l], bs, dacc[l]);
                    if constexpr (KV > 0) dfoi = dfoi * sus[l] + (bs * acc[l]) * dsus[j][l];
                    const T dflux = two_products(dfoi, y0[IS], foi[l], u[IS]);
                    const T dg_i = two_products(dgamma[j][l], y0[II + l], gamma[l], u[II + l]);
                    dout_s += dflux;
                    if constexpr (HAS_E) {
                        const T ds_e = two_products(dsigma[j][l], y0[IE + l], sigma[l], u[IE + l]);
                        du[IE + l] = dflux - ds_e;
                        du[II + l] = ds_e - dg_i;
                    } else {
                        du[II + l] = dflux - dg_i;
                    }
                    if constexpr (HAS_WANE) {
                        T dinflow = dg_i;
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            const T d_o = T(W) * two_products(domega[j][l], y0[IR + l * W + w], omega[l], u[IR + l * W + w]);
                            du[IR + l * W + w] = dinflow - d_o;
                            dinflow = d_o;
                        }
                        dback_s += dinflow;
                    } else {
                        du[IR + l * W] = dg_i;
#pragma unroll
                        for (int w = 1; w < W; ++w) du[IR + l * W + w] = T(0);
                    }
                    if constexpr (HAS_C) du[IC + l] = dflux;
                }
                du[IS] = strain_sum(dback_s - dout_s);
                if constexpr (KV > 0) {
                    // d min(max(nu, 0) * P_a, max(s, 0)): doses while they last, else the remaining susceptibles
                    T dnu = dvbase[j][0] + t * (dvbase[j][1] + t * (dvbase[j][2] + t * dvbase[j][3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const T lag = t - vknot[q];
                        if (q < vnk && lag > T(0))
                            dnu += (dvcoef[j][q] * lag - T(3) * vcoef[q] * dvknot[j][q]) * (lag * lag);
                    }
                    T dn_age = dN + xchg_xor<1>(dN);
                    if constexpr (KV == 4) dn_age += xchg_xor<2>(dn_age);
                    const T ddoses = (v_nu_pos ? dnu : T(0)) * v_nage + v_rate * dn_age;
                    const T dleave = vax_top ? T(0) : (v_capped ? (v_has_s ? u[IS] : T(0)) : ddoses);
                    const T dfrom_below = KV == 2 ? xchg_xor<1>(dleave) : dpp_mov<0x90>(dleave);
                    du[IS] += (vax_first ? T(0) : dfrom_below) - dleave;
                }
            }
        }
    }

    // Base b of the all-gather covers the four lane offsets {o, o^1, o^2, o^3} with
    // o = 0 (x itself), 7 (row_half_mirror), 15 (row_mirror), 8 (= 15 then 7), and for wider
    // groups the same four bases of the xor-16 / xor-32 images.
    template <int BASE>
    __device__ __forceinline__ void gather_base(const T (&x)[S], T (&acc)[S]) const {
        constexpr int o = (BASE & 3) == 0 ? 0 : (BASE & 3) == 1 ? 7 : (BASE & 3) == 2 ? 15 : 8;
        constexpr int hi = (BASE >> 2) * 16; // 0, 16, 32, 48
        constexpr int off = o ^ hi;
        T xb[S];
#pragma unroll
        for (int l = 0; l < S; ++l) xb[l] = xchg_xor<off>(x[l]);
#pragma unroll
        for (int l = 0; l + 1 < S; l += 2) {
            if constexpr (BASE == 0)
                mul_quad_pair(acc[l], acc[l + 1], xb[l], xb[l + 1], Cx[off], Cx[off ^ 1], Cx[off ^ 2], Cx[off ^ 3]);
            else
                fma_quad_pair(acc[l], acc[l + 1], xb[l], xb[l + 1], Cx[off], Cx[off ^ 1], Cx[off ^ 2], Cx[off ^ 3]);
        }
        if constexpr (S % 2 == 1)
            fma_quad_one(acc[S - 1], xb[S - 1], Cx[off], Cx[off ^ 1], Cx[off ^ 2], Cx[off ^ 3]);
        if constexpr ((BASE + 1) * 4 < GA) gather_base<BASE + 1>(x, acc);
    }

    // dense output at theta in [0,1] for the accepted step (y -> y1), k = stage derivatives.  The Tsit5 weights b_i(theta) are
    // kept splatted over a register pair each: the packed form multiplies whole pairs by them, and a scalar weight read
    // back from the struct and splatted at the use is widened by instcombine into overlapping two-float loads that pin the
    // struct in scratch memory.
    struct Dense {
        V2 b0, b1, b2, b3, b4, b5, b6;
        T theta;
    };
    __device__ __forceinline__ static void dense_prepare(T th, Dense &d) {
        d.theta = th;
        if constexpr (METHOD == 0) {
            const T t2 = th * th;
            const T w0 = T(-1.0530884977290216) * th * (th - T(1.3299890189751412)) *
                         (t2 - T(1.4364028541716351) * th + T(0.7139816917074209));
            const T w1 = T(0.1017) * t2 * (t2 - T(2.1966568338249754) * th + T(1.2949852507374631));
            const T w2 = T(2.490627285651252793) * t2 * (t2 - T(2.38535645472061657) * th + T(1.57803468208092486));
            const T w3 = T(-16.54810288924490272) * (th - T(1.21712927295533244)) * (th - T(0.61620406037800089)) * t2;
            const T w4 = T(47.37952196281928122) * (th - T(1.203071208372362603)) * (th - T(0.658047292653547382)) * t2;
            const T w5 = T(-34.87065786149660974) * (th - T(1.2)) * (th - T(0.666666666666666667)) * t2;
            const T w6 = T(2.5) * (th - T(1.0)) * (th - T(0.6)) * t2;
            d.b0 = V2{w0, w0};
            d.b1 = V2{w1, w1};
            d.b2 = V2{w2, w2};
            d.b3 = V2{w3, w3};
            d.b4 = V2{w4, w4};
            d.b5 = V2{w5, w5};
            d.b6 = V2{w6, w6};
        }
    }
    __device__ __forceinline__ static T wt(const V2 &w, T) { return w[0]; }
    __device__ __forceinline__ static V2 wt(const V2 &w, V2) { return w; }
    template <typename U>   // U = T (one value) or V2 (a register pair of the packed state)
    __device__ __forceinline__ static U dense_eval(const Dense &d, T dt, U y0v, U y1v, U k0, U k1,
                                                   U k2, U k3, U k4, U k5, U k6) {
        if constexpr (METHOD == 0) {
            U a = wt(d.b0, U()) * k0;
            a += wt(d.b1, U()) * k1;
            a += wt(d.b2, U()) * k2;
            a += wt(d.b3, U()) * k3;
            a += wt(d.b4, U()) * k4;
            a += wt(d.b5, U()) * k5;
            a += wt(d.b6, U()) * k6;
            return y0v + dt * a;
        } else {
            // quartic through y0, y1, ymid, f0, f1 (Shampine midpoint weights)
            U mid = T(TB::cmid[0]) * k0;
            mid += T(TB::cmid[2]) * k2;
            mid += T(TB::cmid[3]) * k3;
            mid += T(TB::cmid[4]) * k4;
            mid += T(TB::cmid[5]) * k5;
            mid += T(TB::cmid[6]) * k6;
            const U ymid = y0v + dt * mid;
            const U f0 = dt * k0, f1 = dt * k6;
            const U ca = T(2) * (f1 - f0) - T(8) * (y1v + y0v) + T(16) * ymid;
            const U cb = T(5) * f0 - T(3) * f1 + T(18) * y0v + T(14) * y1v - T(32) * ymid;
            const U cc = f1 - T(4) * f0 - T(11) * y0v - T(5) * y1v + T(16) * ymid;
            const T th = d.theta;
            return (((ca * th + cb) * th + cc) * th + f0) * th + y0v;
        }
    }

    // ---- the same interpolant as a polynomial in theta, formed ONCE per accepted step (the s/e/i/r/c kernels; the SEIP
    // kernels, which save about one row per step, keep the weight form above):
    //     y(theta) = y + scale * (q1 + theta (q2 + theta (q3 + theta q4)))
    // Tsit5: every weight b_i(theta) is a quartic without constant term and only b_1 has a theta^1 term (coefficient 1:
    //     y'(t_prev) = f_1), so q1 = k[0], q_m = sum_i bp[i][m] k_i (Tab<0>::bp, 21 packed FMAs per register pair and step)
    //     and scale = dt theta; a saved row then costs 4 packed FMAs per pair instead of 8 plus the 35 scalar operations of
    //     the weights (at 3.3-5 rows per step: 9 % of the D = 360 kernel's instructions, more for D = 136).
    // Dopri5: Shampine's quartic through y, y1, ymid, f0, f1 -- the coefficients the round-2 kernel formed per ROW, hoisted
    //     (same operations, same order: bit-identical rows): q1 = dt k[0], q2 = cc, q3 = cb, q4 = ca, scale = theta.
    // The coefficient planes overwrite stage derivatives that are dead once the error estimate exists:
    //     k[1] <- q2, k[2] <- q3, k[3] <- q4 (Dopri5 also k[4] <- q1); k[0] and k[6] (FSAL) are never touched.
    static constexpr int QP1 = (METHOD == 0 || PRESCALE) ? 0 : 4;   // (PRESCALE: k[] hold dt f already, q1 = k[0] for both methods)
    struct Poly {
        T theta, scale;
    };
    __device__ __forceinline__ static void poly_prepare(T th, T dt, Poly &d) {
        d.theta = th;
        d.scale = (METHOD == 0 && !PRESCALE) ? dt * th : th;
    }
    template <typename U>   // U = T (one value) or V2 (a register pair of the packed state)
    __device__ __forceinline__ static U poly_eval(const Poly &d, U y0v, U q1, U q2, U q3, U q4) {
        U a = q4 * d.theta + q3;
        a = a * d.theta + q2;
        a = a * d.theta + q1;
        return a * d.scale + y0v;
    }
    // (PP0 .. PP1 - 1: the register pairs whose rows will be evaluated -- all of them, or, lean instance, the pairs that hold
    // the scored compartment: the other elements' coefficients would be 21 packed FMAs per pair and plane for nobody)
    // (`pairs`: bit pp set = register pair pp holds an element of a compartment that is saved / scored -- wave-uniform, from
    // the save mask (Solver::carve); a sub-save, the reference's `sub_save_indices`, leaves the other pairs' 21 packed FMAs out)
    template <int PP0 = 0, int PP1 = NP>
    __device__ __forceinline__ static void dense_coefficients(T dt, const State (&y)[NC], const State (&y1)[NC],
                                                              State (&k)[7][NC], uint32_t pairs = ~0u) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int pp = PP0; pp < PP1; ++pp) {
                if (!((pairs >> pp) & 1u)) continue;
                const V2 k0 = k[0][c].p[pp], k1 = k[1][c].p[pp], k2 = k[2][c].p[pp], k3 = k[3][c].p[pp],
                         k4 = k[4][c].p[pp], k5 = k[5][c].p[pp], k6 = k[6][c].p[pp];
                if constexpr (METHOD == 0) {
                    V2 q[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        V2 a = T(TB::bp[0][m]) * k0;
                        a += T(TB::bp[1][m]) * k1;
                        a += T(TB::bp[2][m]) * k2;
                        a += T(TB::bp[3][m]) * k3;
                        a += T(TB::bp[4][m]) * k4;
                        a += T(TB::bp[5][m]) * k5;
                        a += T(TB::bp[6][m]) * k6;
                        q[m] = a;
                    }
                    k[1][c].p[pp] = q[0];
                    k[2][c].p[pp] = q[1];
                    k[3][c].p[pp] = q[2];
                } else {
                    const V2 y0v = y[c].p[pp], y1v = y1[c].p[pp];
                    V2 mid = T(TB::cmid[0]) * k0;
                    mid += T(TB::cmid[2]) * k2;
                    mid += T(TB::cmid[3]) * k3;
                    mid += T(TB::cmid[4]) * k4;
                    mid += T(TB::cmid[5]) * k5;
                    mid += T(TB::cmid[6]) * k6;
                    V2 ymid, f0, f1;
                    if constexpr (PRESCALE) {
                        ymid = y0v + mid;
                        f0 = k0;
                        f1 = k6;
                    } else {
                        ymid = y0v + dt * mid;
                        f0 = dt * k0;
                        f1 = dt * k6;
                    }
                    const V2 ca = T(2) * (f1 - f0) - T(8) * (y1v + y0v) + T(16) * ymid;
                    const V2 cb = T(5) * f0 - T(3) * f1 + T(18) * y0v + T(14) * y1v - T(32) * ymid;
                    const V2 cc = f1 - T(4) * f0 - T(11) * y0v - T(5) * y1v + T(16) * ymid;
                    k[1][c].p[pp] = cc;
                    k[2][c].p[pp] = cb;
                    k[3][c].p[pp] = ca;
                    if constexpr (!PRESCALE) k[4][c].p[pp] = f0;
                }
            }
    }

    // interpolate + store one compartment block [FIRST, FIRST+CNT) of one plane of this lane
    template <int FIRST, int CNT>
    __device__ __forceinline__ static void save_block(const Poly &d, const State &y, const State (&k)[7][NC],
                                                      int plane, T *dst, bool vec_ok) {
        if constexpr (FIRST % 2 == 0 && CNT % 2 == 0) { // whole register pairs: packed arithmetic, stored as they come
            V2 o[CNT / 2];
#pragma unroll
            for (int q = 0; q < CNT / 2; ++q) {
                const int j = FIRST / 2 + q;
                o[q] = poly_eval<V2>(d, y.p[j], k[QP1][plane].p[j], k[1][plane].p[j], k[2][plane].p[j], k[3][plane].p[j]);
            }
            if (vec_ok) {
                if constexpr (sizeof(T) == 4 && CNT % 4 == 0) {
                    typedef T V4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int q = 0; q < CNT / 4; ++q)
                        *reinterpret_cast<V4 *>(dst + 4 * q) = V4{o[2 * q][0], o[2 * q][1], o[2 * q + 1][0], o[2 * q + 1][1]};
                } else {
#pragma unroll
                    for (int q = 0; q < CNT / 2; ++q) *reinterpret_cast<V2 *>(dst + 2 * q) = o[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < CNT / 2; ++q) {
                    dst[2 * q] = o[q][0];
                    dst[2 * q + 1] = o[q][1];
                }
            }
        } else {
            T v[CNT];
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                const int j = FIRST + q;
                v[q] = poly_eval<T>(d, y[j], k[QP1][plane][j], k[1][plane][j], k[2][plane][j], k[3][plane][j]);
            }
            store_run<T, CNT>(dst, v, vec_ok);
        }
    }

    // ---- fused observation likelihood (Poisson log-likelihood of the reference's model(),
    // examples/sir_infer_parameters.py:30-38: incidence = max(diff(R), 1e-6), obs ~ Poisson(incidence);
    // lgamma(obs + 1) is the host's).  Unreplicated trajectories accumulate while they save (each
    // group walks its save times in order); replicated ones (small states, save times interleaved
    // over the replicas) park the interpolated values in an LDS table and score them after the solve.
    static constexpr int LLMAX = S * W;
    struct LL {
        T prev[NC][LLMAX];
        double acc;
        double dacc[NC];   // [0] unused
    };
    // one Poisson term and its tangents: terms in the solve's own precision (fp32: v_log_f32 /
    // v_rcp_f32), sums in float64; the gradient passes the floor where inc >= floor (clamp convention)
    __device__ __forceinline__ static void ll_term(const KArgs<T> &ka, LL &ll, T o, const T (&inc)[NC]) {
        const T rate = inc[0] > ka.ll_floor ? inc[0] : ka.ll_floor;
        ll.acc += (double)(o * M::log(rate) - rate);
        const double coef = inc[0] >= ka.ll_floor ? (double)(o * M::recip(rate) - T(1)) : 0.0;
#pragma unroll
        for (int c = 1; c < NC; ++c) ll.dacc[c] += coef * (double)inc[c];
    }
    template <int FIRST, int CNT>
    __device__ __forceinline__ static void ll_block(const KArgs<T> &ka, const Poly &d, const State (&y)[NC],
                                                    const State (&k)[7][NC], LL &ll, int j, int off, T *tab_row) {
        const int ll_mode = LEAN ? LEAN_MODE : ka.ll_mode;
        const bool have = ll_mode == 0 || j > 0;
        const T *orow = ka.obs + (int64_t)(ll_mode == 0 ? j : (j > 0 ? j - 1 : 0)) * ka.ll_row + off;
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const int e = FIRST + q;
            T v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c)
                v[c] = poly_eval<T>(d, y[c][e], k[QP1][c][e], k[1][c][e], k[2][c][e], k[3][c][e]);
            // (the two destinations are written in separate statements, `prev` by value: merged into one store through a
            // pointer that is either the LDS row or &ll.prev, the whole LL struct -- accumulators included -- was pinned in
            // scratch memory, a dependent scratch load / add / store chain in every save round of a latency-bound kernel)
            const bool table = tab_row != nullptr;
            if (table) {
#pragma unroll
                for (int c = 0; c < NC; ++c) tab_row[q * NC + c] = v[c];
            } else if (have) {
                T inc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) inc[c] = ll_mode == 0 ? v[c] : v[c] - ll.prev[c][q];
                ll_term(ka, ll, orow[q], inc);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) ll.prev[c][q] = table ? ll.prev[c][q] : v[c];
        }
    }
    __device__ __forceinline__ static void ll_row(const KArgs<T> &ka, const Poly &d, const State (&y)[NC],
                                                  const State (&k)[7][NC], LL &ll, int j, int a, int as, bool lead,
                                                  T *tab_row) {
        switch (LEAN ? LEAN_SLOT : ka.ll_slot) {
        case 0:
            if (lead) ll_block<IS, 1>(ka, d, y, k, ll, j, a, tab_row);
            break;
        case 1:
            if constexpr (HAS_E) ll_block<IE, S>(ka, d, y, k, ll, j, as, tab_row);
            break;
        case 2:
            ll_block<II, S>(ka, d, y, k, ll, j, as, tab_row);
            break;
        case 3:
            ll_block<IR, S * W>(ka, d, y, k, ll, j, as * W, tab_row);
            break;
        default:
            if constexpr (HAS_C) ll_block<IC, S>(ka, d, y, k, ll, j, as, tab_row);
            break;
        }
    }
    // table mode, after the solve: replica `rep` scores observation rows rep, rep + R, ...
    __device__ __forceinline__ static void ll_from_table(const KArgs<T> &ka, LL &ll, const T *tab_lane, int row_stride,
                                                         int n_save, int rep, int R, int a, int as, bool lead) {
        const int slot = LEAN ? LEAN_SLOT : ka.ll_slot, ll_mode = LEAN ? LEAN_MODE : ka.ll_mode;
        const int cnt = slot == 0 ? 1 : (slot == 3 ? S * W : S);
        const int off = slot == 0 ? a : (slot == 3 ? as * W : as);
        if (slot == 0 && !lead) return;
        const int n_obs = n_save - ll_mode;
        for (int r = rep; r < n_obs; r += R) {
            const int j = r + ll_mode;
            const T *now = tab_lane + (int64_t)j * row_stride, *before = tab_lane + (int64_t)(j - ll_mode) * row_stride;
            for (int q = 0; q < cnt; ++q) {
                T inc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    inc[c] = ll_mode == 0 ? now[q * NC + c] : now[q * NC + c] - before[q * NC + c];
                ll_term(ka, ll, ka.obs[(int64_t)r * ka.ll_row + off + q], inc);
            }
        }
    }

    // CNT consecutive elements of an interpolated plane to aligned memory (SAVE_ALL rows)
    template <int FIRST, int CNT>
    __device__ __forceinline__ static void store_elements(const State &o, T *dst) {
        if constexpr (CNT > 0) {
            T v[CNT];
#pragma unroll
            for (int q = 0; q < CNT; ++q) v[q] = o[FIRST + q];
            store_run<T, CNT>(dst, v, true);
        }
    }

    // one saved row of one plane
    // `as` = a * ST + h * SPL: position of this lane's first strain inside an [A, ST] block
    template <int PLANE>
    __device__ __forceinline__ static void save_row(const KArgs<T> &ka, const Poly &dn, const State (&y)[NC],
                                                    const State (&k)[7][NC], T *row, int a, int as, bool lead,
                                                    bool vec_ok) {
        if constexpr (SAVE_ALL && sizeof(T) == 4 && S % 2 == 1) {
            // every compartment is saved and the strain blocks are single elements (the strain-split shapes): interpolate
            // ALL register pairs at once -- e and i, c and s share pairs -- and store the elements from the result
            State o;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                o.p[j] = poly_eval<V2>(dn, y[PLANE].p[j], k[QP1][PLANE].p[j], k[1][PLANE].p[j], k[2][PLANE].p[j], k[3][PLANE].p[j]);
            if (lead) row[ka.save_off[0] + a] = o[IS];
            store_elements<IE, NE>(o, row + ka.save_off[1] + as);
            store_elements<II, S>(o, row + ka.save_off[2] + as);
            store_elements<IR, S * W>(o, row + ka.save_off[3] + as * W);
            store_elements<IC, NCU>(o, row + ka.save_off[4] + as);
            return;
        }
        if constexpr (SAVE_ALL) vec_ok = true;
        if ((SAVE_ALL || ka.save_off[0] >= 0) && lead)
            save_block<IS, 1>(dn, y[PLANE], k, PLANE, row + ka.save_off[0] + a, false);
        if constexpr (HAS_E)
            if (SAVE_ALL || ka.save_off[1] >= 0)
                save_block<IE, S>(dn, y[PLANE], k, PLANE, row + ka.save_off[1] + as, vec_ok);
        if (SAVE_ALL || ka.save_off[2] >= 0)
            save_block<II, S>(dn, y[PLANE], k, PLANE, row + ka.save_off[2] + as, vec_ok);
        if (SAVE_ALL || ka.save_off[3] >= 0)
            save_block<IR, S * W>(dn, y[PLANE], k, PLANE, row + ka.save_off[3] + as * W, vec_ok);
        if constexpr (HAS_C)
            if (SAVE_ALL || ka.save_off[4] >= 0)
                save_block<IC, S>(dn, y[PLANE], k, PLANE, row + ka.save_off[4] + as, vec_ok);
    }

    __device__ __forceinline__ static void fill_row(const KArgs<T> &ka, T *row, int a, int as,
                                                    bool lead, T v) {
        if (ka.save_off[0] >= 0 && lead) row[ka.save_off[0] + a] = v;
        if constexpr (HAS_E)
            if (ka.save_off[1] >= 0)
                for (int q = 0; q < S; ++q) row[ka.save_off[1] + as + q] = v;
        if (ka.save_off[2] >= 0)
            for (int q = 0; q < S; ++q) row[ka.save_off[2] + as + q] = v;
        if (ka.save_off[3] >= 0)
            for (int q = 0; q < S * W; ++q) row[ka.save_off[3] + as * W + q] = v;
        if constexpr (HAS_C)
            if (ka.save_off[4] >= 0)
                for (int q = 0; q < S; ++q) row[ka.save_off[4] + as + q] = v;
    }

    template <int PLANE>
    __device__ __forceinline__ static void save_tangents(const KArgs<T> &ka, const Poly &dn, const State (&y)[NC],
                                                         const State (&k)[7][NC], T *drow, int a, int as, bool lead,
                                                         bool vec_ok) {
        if constexpr (PLANE < NC) {
            save_row<PLANE>(ka, dn, y, k, drow + (int64_t)(PLANE - 1) * ka.d_saved, a, as, lead, vec_ok);
            save_tangents<PLANE + 1>(ka, dn, y, k, drow, a, as, lead, vec_ok);
        }
    }

    // everything of one trajectory that lives in this lane besides the state: rates, seasonal numbers, introductions,
    // vaccination rows and their seed directions (KArgs::params row `traj`, broadcast loads inside the lane group)
    template <typename KA>   // KA: KArgs<T> in whatever address space the caller reads it from (run(): cold_args)
    __device__ __forceinline__ void load_parameters(const KA &ka, int64_t traj, int a, int aa, int s0) {
        const int A = ka.A;
        const T *p = ka.params + traj * ka.P;
        constexpr int oS = 2, oW = 2 + (HAS_E ? 1 : 0), oI = 2 + (HAS_E ? 1 : 0) + (HAS_WANE ? 1 : 0),
                      oSe = oI + (INTRO ? 3 : 0);
#pragma unroll
        for (int l = 0; l < S; ++l) {
            beta[l] = p[s0 + l];
            gamma[l] = p[ST + s0 + l];
            sigma[l] = HAS_E ? p[oS * ST + s0 + l] : T(0);
            omega[l] = HAS_WANE ? p[oW * ST + s0 + l] : T(0);
        }
        if constexpr (PAIRED_RHS) {
#pragma unroll
            for (int q = 0; q < SP; ++q) {
                beta2[q] = V2{beta[2 * q], beta[2 * q + 1]};
                gamma2[q] = V2{gamma[2 * q], gamma[2 * q + 1]};
                sigma2[q] = V2{sigma[2 * q], sigma[2 * q + 1]};
                omega2[q] = V2{omega[2 * q], omega[2 * q + 1]};
            }
        }
        if constexpr (INTRO) {
#pragma unroll
            for (int l = 0; l < S; ++l) {
                const T scale = p[(oI + 1) * ST + s0 + l];
                const bool here = !pad && ((ka.intro_mask[s0 + l] >> aa) & 1ull);
                itime[l] = p[oI * ST + s0 + l];
                iinv[l] = T(1) / scale;
                ibase[l] = here ? T(1) / (scale * T(2.5066282746310002)) : T(0);
                iamp[l] = ibase[l] * p[(oI + 2) * ST + s0 + l];
            }
#pragma unroll
            for (int j = 0; j < NDA; ++j)
#pragma unroll
                for (int l = 0; l < S; ++l) ditime[j][l] = discale[j][l] = dipct[j][l] = T(0);
        }
        if constexpr (KV > 0) {
            // per-trajectory vaccination block after the seasonal numbers:
            //   susceptibility [groups][ST], then per group: base[4], knot[n_knots], coef[n_knots]
            const int nk = ka.n_vax_knots;
            const T *vp = p + oSe * ST + (ka.seasonal ? 3 : 0);   // (KV > 0: never a lean instance)
            const T *sp_ = vp + A * ST + aa * (4 + 2 * nk);
            vnk = nk;
#pragma unroll
            for (int l = 0; l < S; ++l) sus[l] = pad ? T(0) : vp[aa * ST + s0 + l];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vbase[q] = pad ? T(0) : sp_[q];
                vknot[q] = (!pad && q < nk) ? sp_[4 + q] : T(0);
                vcoef[q] = (!pad && q < nk) ? sp_[4 + nk + q] : T(0);
            }
            const int tier = a % KV;
            vax_first = tier == 0;
            vax_top = tier >= ka.n_vax_tiers - 1;
        }
        amp = T(0);
        phase = T(0);
        w_season = T(0);
        T period = T(1);
        if (seasonal) {
            const T *sp = p + oSe * ST;
            amp = sp[0];
            phase = sp[1];
            period = sp[2];
            w_season = T(6.283185307179586476925286766559) / period;
        }
#pragma unroll
        for (int j = 0; j < NDA; ++j) {
#pragma unroll
            for (int l = 0; l < S; ++l) dbeta[j][l] = dgamma[j][l] = dsigma[j][l] = domega[j][l] = T(0);
            damp[j] = dphase[j] = dw_season[j] = T(0);
        }
        if constexpr (ND > 0) {
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const T *dp = ka.dparams + (traj * ND + j) * ka.P;
#pragma unroll
                for (int l = 0; l < S; ++l) {
                    dbeta[j][l] = dp[s0 + l];
                    dgamma[j][l] = dp[ST + s0 + l];
                    if constexpr (HAS_E) dsigma[j][l] = dp[oS * ST + s0 + l];
                    if constexpr (HAS_WANE) domega[j][l] = dp[oW * ST + s0 + l];
                    if constexpr (INTRO) {
                        ditime[j][l] = dp[oI * ST + s0 + l];
                        discale[j][l] = dp[(oI + 1) * ST + s0 + l];
                        dipct[j][l] = dp[(oI + 2) * ST + s0 + l];
                    }
                }
                if constexpr (KV > 0) {
                    const int nk = ka.n_vax_knots;
                    const T *dvp = dp + oSe * ST + (ka.seasonal ? 3 : 0);
                    const T *dsp_ = dvp + A * ST + aa * (4 + 2 * nk);
#pragma unroll
                    for (int l = 0; l < S; ++l) dsus[j][l] = pad ? T(0) : dvp[aa * ST + s0 + l];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        dvbase[j][q] = pad ? T(0) : dsp_[q];
                        dvknot[j][q] = (!pad && q < nk) ? dsp_[4 + q] : T(0);
                        dvcoef[j][q] = (!pad && q < nk) ? dsp_[4 + nk + q] : T(0);
                    }
                }
                if (seasonal) {
                    const T *dsp = dp + oSe * ST;
                    damp[j] = dsp[0];
                    dphase[j] = dsp[1];
                    dw_season[j] = -w_season / period * dsp[2]; // d(2 pi / period)
                }
            }
        }
    }

    // ---- Producer / consumer (PC): a workgroup of TWO waves per TPW trajectories.  Wave 0 steps (stages, error control,
    // dense-output coefficients) and hands every accepted step -- y, k[0] and the three coefficient planes, five planes of NP
    // register pairs -- to wave 1 through one LDS buffer; wave 1 evaluates and stores the rows while wave 0 is already on the
    // next step.  Same polynomial, same operands, same instructions: the rows are bit-identical to the one-wave kernel's.
    // Barrier protocol (both waves execute the same count, 2 n + 2 for n hand-overs; no other barrier after the role split):
    //     producer:  [ A  write buffer  B ] per accepted step ...            A  set `fin`  B
    //     consumer:    A  [ B  read buffer  A  evaluate + store rows ] ...   B  (sees `fin`)
    // A = "the buffer is free" (the consumer arrives right after copying it to registers), B = "a step is published".
    struct Handoff {
        V2 *planes;   // [5][NP][64]: y | k[0] | q2 | q3 | q4
        T *tprev, *tnext;   // [64] each: the accepted step's interval (tnext < tprev: this lane's group did not accept)
        int *fin;
    };
    __device__ __forceinline__ static void consume(const KArgs<T> &ka, const Handoff &h, const T *ts_tab, int lane, int a, int as,
                                                   bool lead, bool writer, bool valid, int64_t traj) {
        const int n_save = ka.n_save;
        const bool vec_ok = ka.vec_ok != 0;
        T *const out_traj = ka.out + traj * (int64_t)n_save * ka.d_saved;
        int save_idx = 0;
        T ts_next = n_save > 0 ? ts_tab[0] : M::inf();
        T ts_next2 = n_save > 1 ? ts_tab[1] : M::inf();
        State y[1], k[7][1];
        __syncthreads();                       // A (the first one: nothing to wait for)
        for (;;) {
            __syncthreads();                   // B: a step is published, or the producer is through
            if (*h.fin) break;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                y[0].p[pp] = h.planes[(0 * NP + pp) * 64 + lane];
                k[0][0].p[pp] = h.planes[(1 * NP + pp) * 64 + lane];
                k[1][0].p[pp] = h.planes[(2 * NP + pp) * 64 + lane];
                k[2][0].p[pp] = h.planes[(3 * NP + pp) * 64 + lane];
                k[3][0].p[pp] = h.planes[(4 * NP + pp) * 64 + lane];
            }
            const T tprev = h.tprev[lane], tnext = h.tnext[lane];
            __syncthreads();                   // A: the buffer is free again
            const bool accept = valid && tnext > tprev;
            const T dt = tnext - tprev;
            const T inv_dt = M::recip(dt);
            bool pending = accept && (save_idx < n_save) && (ts_next <= tnext);
            while (__any(pending)) {
                if (pending) {
                    Poly dn;
                    poly_prepare((ts_next - tprev) * inv_dt, dt, dn);
                    if (writer) save_row<0>(ka, dn, y, k, out_traj + (int64_t)save_idx * ka.d_saved, a, as, lead, vec_ok);
                    save_idx += 1;
                    ts_next = ts_next2;
                    ts_next2 = save_idx + 1 < n_save ? ts_tab[save_idx + 1] : M::inf();
                }
                pending = accept && (save_idx < n_save) && (ts_next <= tnext);
            }
        }
        // rows never reached (failed solves): +inf, like diffrax's unfilled SaveAt buffer
        if (valid && writer)
            for (; save_idx < n_save; ++save_idx) fill_row(ka, out_traj + (int64_t)save_idx * ka.d_saved, a, as, lead, M::inf());
    }
    // ================================================================ the family interface of Stepper<F> (stepper.hpp)
    static constexpr int NDIR = ND;                 // tangent directions (planes 1..ND of the state)
    static constexpr int GW = G;                    // lanes of one wave that hold one trajectory
    // FEAT bit 11: adaptive steps without discontinuity points as compile-time facts (what every BASELINE ensemble is): the
    // constant-step and jump bookkeeping -- four per-lane values and a dozen scalar ones, a handful of uniform branches per
    // attempt -- leaves the loop.  enqueue() picks the variant when the call has neither (lean instances imply it).
    static constexpr bool ADAPTIVE_NO_JUMPS = LEAN || (FEAT & 0x0800) != 0;
    static constexpr bool ROOTLESS_NORM = KV == 0;  // the controller works on the mean square of the error (Control::decide_ms); not with the dose cap's kinks
    // FEAT bit 10: a static-grid-only instance -- prologue in front of the stepping loop, write-off behind it, nothing of the
    // queue in between (Stepper: PULLS = false).  For shapes whose launches are static anyway (launch(): two trajectories per wave).
    static constexpr bool PULLS = (FEAT & 0x0400) == 0 && !LEAN && !FUSED;   // slots may draw further trajectories from KArgs::work (a lean instance is only dispatched without a caller's queue: a static grid)
    static constexpr bool REPLAYS = false;          // (recorded step schedules: the SEIP family)
    static constexpr bool IDLE_SLOTS_LOAD = false;  // a slot beyond the batch idles without data
    int a, as;            // age lane; (age, first global strain of this lane) = a ST + s0: the lane's place in a compartment
    bool writer, leader;  // the lane stores rows (not a pad lane) / reports the trajectory's status (age 0, strain lane 0)

    // lane indices and what every trajectory of the launch shares (contact row, model switches); -> the lane's slot in the wave
    __device__ __forceinline__ int init(const KArgs<T> &ka, int lane) {
        a = lane % GA;                    // age lane
        const int h = (lane / GA) % GS;   // strain lane: strains h*SPL .. h*SPL+SPL-1
        const int A = ka.A;
        pad = a >= A;
        lead = h == 0;
        normalize = LEAN ? true : ka.normalize != 0;
        seasonal = LEAN ? false : ka.seasonal != 0;
        const int aa = pad ? 0 : a;
        writer = !pad;
        leader = a == 0 && lead;
        as = a * ST + h * SPL;
        // contact row, pre-permuted to the xor-exchange order
#pragma unroll
        for (int k = 0; k < GA; ++k) {
            const int b = a ^ k;
            Cx[k] = (!pad && b < A) ? ka.contact[aa * A + b] : T(0);
        }
        return lane / G;
    }
    __device__ __forceinline__ static int state_dim(const KArgs<T> &ka) { // compartment-major layout
        return ka.A * (1 + ST * ((HAS_E ? 1 : 0) + 1 + W + (HAS_C ? 1 : 0)));
    }

    // LDS behind the save grid and the discontinuity points
    struct Tables {
        bool fused_ll, ll_table;   // tangent kernels: the likelihood is scored instead of rows / ... through the table of replicated trajectories
        T *ll_lane;                // likelihood table (replicated trajectories): [trajectory slot][save index][lane of group][LLMAX][planes]
        V4 *rate_tab;              // PRESCALE: the rates (and their seeds) as loaded, [quad][lane], 16-byte aligned behind the other tables
        Handoff hand;              // PC: the accepted step on its way to the row-writing wave
        uint32_t pairs;            // register pairs holding an element of a saved (or scored) compartment: dense_coefficients
    };
    // bit pp of the result: pair pp holds an element of a compartment whose rows are needed (slots: s, e, i, r, c)
    __device__ __forceinline__ static uint32_t pairs_of(bool s_, bool e_, bool i_, bool r_, bool c_) {
        uint32_t m = 0;
        auto add = [&](int first, int cnt) {
            for (int q = 0; q < cnt; ++q) m |= 1u << ((first + q) >> 1);
        };
        if (s_) add(IS, 1);
        if (e_) add(IE, NE);
        if (i_) add(II, S);
        if (r_) add(IR, S * W);
        if (c_) add(IC, NCU);
        return m;
    }
    static constexpr int LL_ROW = G * LLMAX * NC;
    __device__ __forceinline__ void carve(const KArgs<T> &ka, Tables &tb, T *ts_tab, T *free_lds, int lane, int grp, int n_save, int n_jump) const {
        tb.fused_ll = tb.ll_table = false;
        if constexpr (ND > 0) {
            tb.fused_ll = LEAN ? true : ka.obs != nullptr;
            tb.ll_table = tb.fused_ll && ka.rep_log2 > 0;
        }
        tb.ll_lane = free_lds + ((int64_t)((lane / G) >> ka.rep_log2) * n_save * LL_ROW + (lane % G) * (LLMAX * NC));
        tb.rate_tab = reinterpret_cast<V4 *>(
            ts_tab + (((n_save + (n_jump > 0 ? kMaxJumps : 0) + (tb.ll_table ? (64 >> ka.rep_log2) * n_save * LLMAX * NC : 0)) + 3) & ~3));
        if (tb.fused_ll) {
            const int slot = LEAN ? LEAN_SLOT : ka.ll_slot;
            tb.pairs = pairs_of(slot == 0, slot == 1, slot == 2, slot == 3, slot == 4);
        } else {
            tb.pairs = pairs_of(ka.save_off[0] >= 0, ka.save_off[1] >= 0, ka.save_off[2] >= 0, ka.save_off[3] >= 0, ka.save_off[4] >= 0);
        }
        tb.hand = Handoff{nullptr, nullptr, nullptr, nullptr};
        if constexpr (PC) {
            tb.hand.planes = reinterpret_cast<V2 *>(tb.rate_tab + NRQ * 64);
            tb.hand.tprev = reinterpret_cast<T *>(tb.hand.planes + 5 * NP * 64);
            tb.hand.tnext = tb.hand.tprev + 64;
            tb.hand.fin = reinterpret_cast<int *>(tb.hand.tnext + 64);
            if (threadIdx.x == 0) *tb.hand.fin = 0;
        }
    }

    // parameters and initial state (plane 0) with its seeds (planes 1..ND) of trajectory `traj`, under this group's lanes.
    // The lane's indices and the layout offsets are re-derived HERE from values the compiler cannot see through (lane_c,
    // the cold kernel arguments): hoisted out of the stepping loop, the address arithmetic of this block would hold a dozen
    // registers through every iteration (the D = 136 kernel sits at the 256-register line)
    template <typename KA>
    __device__ __forceinline__ void load_trajectory(const KA &kc, int64_t traj, int lane_c, Tables &tb, State (&y)[NC]) {
        const int Ac = kc.A;
        const int a_c = lane_c % GA, s0_c = ((lane_c / GA) % GS) * SPL, aa_c = a_c >= Ac ? 0 : a_c;
        const int cE = Ac, cI = Ac + (HAS_E ? Ac * ST : 0), cR = cI + Ac * ST, cC = cR + Ac * ST * W;
        const int Dc = cC + (HAS_C ? Ac * ST : 0);
        load_parameters(kc, traj, a_c, aa_c, s0_c);
        if constexpr (PRESCALE) park_rates(tb.rate_tab, lane_c);
        // initial state (plane 0) and its seeds (planes 1..ND)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T *src;
            bool zero = pad;
            if (c == 0) {
                src = kc.y0 + (kc.y0_batched ? traj * Dc : 0);
            } else {
                const T *const dy0 = kc.dy0;
                zero = zero || dy0 == nullptr;
                src = dy0 + ((kc.dy0_batched ? traj * ND : 0) + (c - 1)) * (int64_t)Dc;
            }
            y[c][IS] = zero ? T(0) : src[aa_c];
#pragma unroll
            for (int l = 0; l < S; ++l) {
                const int sg = aa_c * ST + s0_c + l; // (age, global strain)
                if constexpr (HAS_E) y[c][IE + l] = zero ? T(0) : src[cE + sg];
                y[c][II + l] = zero ? T(0) : src[cI + sg];
#pragma unroll
                for (int w = 0; w < W; ++w) y[c][IR + l * W + w] = zero ? T(0) : src[cR + sg * W + w];
                if constexpr (HAS_C) y[c][IC + l] = zero ? T(0) : src[cC + sg];
            }
        }
    }
    __device__ __forceinline__ static bool start_ok(bool lane_ok, int lane_c) { // no lane of the trajectory saw NaN / inf
        const unsigned long long bad_lanes = __ballot(!lane_ok);
        const unsigned long long group_mask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << ((lane_c / G) * G);
        return (bad_lanes & group_mask) == 0ull;
    }
    __device__ __forceinline__ static T traj_sum(T v) { return group_sum<G>(v); }
    // the replicated s enters every norm once: only the lead strain lane counts it
    __device__ __forceinline__ T weigh(int v, T x) const { return (v == IS ? (lead ? T(1) : T(0)) : T(1)) * x; }
    __device__ __forceinline__ void count_once(int pp, V2 &r) const {
        if (GS > 1 && pp == IS / 2) r[IS % 2] = lead ? r[IS % 2] : T(0);
    }

    // rows: where they go, the step's interpolant, one row, a row never reached
    struct Output {
        T *out_traj, *dout_traj;
        LL ll;
    };
    template <typename KA>
    __device__ __forceinline__ static void begin_output(const KA &kc, Output &o, int64_t traj, int n_save) {
        o.out_traj = kc.out + traj * (int64_t)n_save * kc.d_saved;
        o.dout_traj = nullptr;
        if constexpr (ND > 0) {
            o.dout_traj = kc.dout + traj * (int64_t)n_save * ND * kc.d_saved;
            o.ll.acc = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                o.ll.dacc[c] = 0.0;
#pragma unroll
                for (int q = 0; q < LLMAX; ++q) o.ll.prev[c][q] = T(0);
            }
        }
    }
    __device__ __forceinline__ static void dense_begin(const Tables &tb, T dt, const State (&y)[NC], const State (&y1)[NC], State (&k)[7][NC]) {
        // (a lean instance's rows are the likelihood of ONE compartment: coefficients for its register pairs only)
        if constexpr (LEAN && LEAN_SLOT == 3) dense_coefficients<IR / 2, (IR + S * W - 1) / 2 + 1>(dt, y, y1, k);
        else if constexpr (LEAN && LEAN_SLOT == 4) dense_coefficients<IC / 2, (IC + S - 1) / 2 + 1>(dt, y, y1, k);
        else if constexpr (LEAN && LEAN_SLOT == 2) dense_coefficients<II / 2, (II + S - 1) / 2 + 1>(dt, y, y1, k);
        else if constexpr (LEAN && LEAN_SLOT == 1) dense_coefficients<IE / 2, (IE + S - 1) / 2 + 1>(dt, y, y1, k);
        else if constexpr (LEAN) dense_coefficients<IS / 2, IS / 2 + 1>(dt, y, y1, k);
        else if constexpr (SAVE_ALL) dense_coefficients(dt, y, y1, k);
        else dense_coefficients(dt, y, y1, k, tb.pairs);
    }
    // the row of save time tprev + theta dt (`on`: this lane stores it): the saved compartments and their tangents, or --
    // tangent kernels with the likelihood fused in -- the row's contribution to the score
    __device__ __forceinline__ void emit_row(const KArgs<T> &ka, const Tables &tb, Output &o, T theta, T dt, const State (&y)[NC],
                                             const State (&)[NC], const State (&k)[7][NC], int save_idx, bool on, bool vec_ok) const {
        Poly dq;
        poly_prepare(theta, dt, dq);
        if (ND > 0 && tb.fused_ll) {
            if (on) ll_row(ka, dq, y, k, o.ll, save_idx, a, as, lead, tb.ll_table ? tb.ll_lane + (int64_t)save_idx * LL_ROW : nullptr);
        } else if (on) {
            save_row<0>(ka, dq, y, k, o.out_traj + (int64_t)save_idx * ka.d_saved, a, as, lead, vec_ok);
            if constexpr (ND > 0) save_tangents<1>(ka, dq, y, k, o.dout_traj + (int64_t)save_idx * ND * ka.d_saved, a, as, lead, vec_ok);
        }
    }
    __device__ __forceinline__ void fill_row(const KArgs<T> &ka, const Output &o, int save_idx, T v) const {
        fill_row(ka, o.out_traj + (int64_t)save_idx * ka.d_saved, a, as, lead, v);
        if constexpr (ND > 0)
            for (int j = 0; j < ND; ++j) fill_row(ka, o.dout_traj + ((int64_t)save_idx * ND + j) * ka.d_saved, a, as, lead, v);
    }
    // tangent kernels with the likelihood fused in: the finished trajectory's score (and status) instead of rows; -> scored
    __device__ __forceinline__ bool finish_score(const KArgs<T> &ka, const Tables &tb, Output &o, int64_t traj, int32_t st, int save_idx, int n_save, int rep,
                                                 int R, int32_t n_acc, int32_t n_rej, bool writer_now) const {
        if (!tb.fused_ll) return false; // wave-uniform
        LL &ll = o.ll;
        // sum over the lanes of the trajectory; a failed solve scores -inf (rejected by the sampler)
        // (the replicas of a trajectory are adjacent lane groups of the same wave; they step identically,
        // so they arrive here in the same iteration)
        int unfinished = (st != ST_OK || save_idx < n_save) ? 1 : 0;
        if (tb.ll_table) {
            __syncthreads(); // every replica's table entries are visible
            if (writer_now && !unfinished) ll_from_table(ka, ll, tb.ll_lane, LL_ROW, n_save, rep, R, a, as, lead);
        }
        const int span = G * R;
        double tot = writer_now ? ll.acc : 0.0;
        for (int off = span / 2; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        double dtot[NC];
#pragma unroll
        for (int c = 1; c < NC; ++c) {
            dtot[c] = writer_now ? ll.dacc[c] : 0.0;
            for (int off = span / 2; off > 0; off >>= 1) dtot[c] += __shfl_xor(dtot[c], off);
        }
        for (int off = span / 2; off > 0; off >>= 1) unfinished |= __shfl_xor(unfinished, off);
        const bool ok = unfinished == 0;
        if (writer_now && leader && rep == 0) {
            const auto &kc = *cold_args<T>();
            kc.ll_out[traj] = ok ? tot : -__builtin_inf();
            double *const dll = kc.dll_out;
#pragma unroll
            for (int c = 1; c < NC; ++c) dll[traj * ND + (c - 1)] = ok ? dtot[c] : 0.0;
            kc.status[traj] = st;
            kc.n_acc[traj] = n_acc;
            kc.n_rej[traj] = n_rej;
        }
        return true;
    }
    // PC: the row-writing wave of a two-wave workgroup
    __device__ __forceinline__ void consume(const KArgs<T> &ka, const Handoff &h, const T *ts_tab, int lane, bool valid, int64_t traj) const {
        consume(ka, h, ts_tab, lane, a, as, lead, writer, valid, traj);
    }
};

} // namespace dyn

#include "stepper.hpp"

namespace dyn {

// Waves per SIMD the kernel is compiled for, (min, max).  The D = 360 ensemble kernel (8 ages x 4 strain lanes, eight waning
// bins) needs 188 registers left alone: two waves per SIMD, whose vector units are then busy 87 % of the time.  Compiled for
// three (168 registers: the compiler parks fourteen loop-invariant values in scratch and reloads them where the right-hand side
// uses them) the same launch is 1.7 % shorter (2.54 -> 2.50 ms, one box, A/B in docs/perf-log.md).  Everything else: the
// compiler's default range.
template <typename T, int METHOD, int GA, int ST, int W, int ND, int SPL, int FEAT>
constexpr int waves_per_simd(bool upper) {
    return (sizeof(T) == 4 && METHOD == 0 && GA == 8 && ST == 4 && ND == 0 && (FEAT & ~0x14C00) == 0 &&
            ((W == 8 && SPL == 1) || (W == 1 && SPL == 2))) ? 3 : (upper ? 8 : 1);
}

template <typename T, int METHOD, int GA, int ST, bool HAS_E, bool HAS_WANE, bool HAS_C, int W, int ND, int SPL,
          int FEAT = 0>
__global__ void __launch_bounds__((FEAT & 0x8000) ? 128 : 64) DYN_KERNEL_ATTR
__attribute__((amdgpu_waves_per_eu(waves_per_simd<T, METHOD, GA, ST, W, ND, SPL, FEAT>(false), waves_per_simd<T, METHOD, GA, ST, W, ND, SPL, FEAT>(true))))
solve_kernel(const KArgs<T> ka) {
    Stepper<Solver<T, METHOD, GA, ST, HAS_E, HAS_WANE, HAS_C, W, ND, SPL, FEAT>>::run(ka);
}

// ... of a FUSED instance: the sampler run rides along as a second argument (Solver::run reads it at kTailOffset)
template <typename T, int METHOD, int GA, int ST, bool HAS_E, bool HAS_WANE, bool HAS_C, int W, int ND, int SPL,
          int FEAT = 0>
__global__ void __launch_bounds__(64)
solve_kernel_fused(const KArgs<T> ka, const dynnuts::Tail tail) {
    Stepper<Solver<T, METHOD, GA, ST, HAS_E, HAS_WANE, HAS_C, W, ND, SPL, FEAT>>::run(ka);
}

// waves of `kernel` the current device holds at once (occupancy x compute units); cached per (kernel, device, LDS size).
// Every instantiation of solve_kernel<T, ...> with the same T has the same function-pointer TYPE, so the kernel's address
// is part of the key (a 256-register D = 136 kernel and the eight-waves-per-SIMD cfg 2 kernel must not read each other's
// occupancy).
template <typename K>
static int64_t resident_waves(K kernel, size_t lds) {
    static thread_local const void *c_kernel = nullptr;
    static thread_local int c_dev = -1;
    static thread_local size_t c_lds = 0;
    static thread_local int64_t c_val = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if ((const void *)kernel == c_kernel && dev == c_dev && lds == c_lds) return c_val;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, lds) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    c_kernel = (const void *)kernel;
    c_dev = dev;
    c_lds = lds;
    c_val = (int64_t)per_cu * cus;
    return c_val;
}

// host-side launcher, one explicit instantiation per compiled shape (instances.def)
template <typename T, int METHOD, int GA, int ST, bool HAS_E, bool HAS_WANE, bool HAS_C, int W, int ND, int SPL,
          int FEAT = 0>
hipError_t launch(const KArgs<T> &ka_in, hipStream_t stream) {
    KArgs<T> ka = ka_in;
    constexpr int TPW = 64 / (GA * (ST / SPL));
    const int64_t blocks = ((ka.B << ka.rep_log2) + TPW - 1) / TPW;
    if (blocks <= 0) return hipSuccess;
    size_t lds = ((size_t)ka.n_save + (ka.n_jump > 0 ? kMaxJumps : 0)) * sizeof(T); // LDS tables (a lean instance is only dispatched with n_jump == 0)
    if (ND > 0 && ka.obs != nullptr && ka.rep_log2 > 0) // likelihood table of the replicated trajectories of a wave
        lds += (size_t)(64 >> ka.rep_log2) * ka.n_save * (SPL * W) * (1 + ND) * sizeof(T);
    if (((FEAT >> 1) & 0x7f) == 0) // PRESCALE: the parked rates and seeds, [SPL (1 + ND) quads][64 lanes] of four, behind the tables (rounded up to four elements)
        lds = ((lds / sizeof(T) + 3) & ~(size_t)3) * sizeof(T) + (size_t)SPL * (1 + ND) * 64 * 4 * sizeof(T);
    constexpr bool PC = (FEAT & 0x8000) != 0;
    if constexpr (PC) { // producer / consumer: the hand-over buffer [5][NP][64] register pairs, two [64] time arrays, a flag
        constexpr int NVL = 1 + SPL * ((HAS_E ? 1 : 0) + 1 + W + (HAS_C ? 1 : 0)), NPL = (NVL + 1) / 2;
        lds += (size_t)5 * NPL * 64 * 2 * sizeof(T) + 2 * 64 * sizeof(T) + 16;
        ka.work = nullptr;
        if (ka.rep_log2 != 0 || ka.order != nullptr) return hipErrorInvalidValue; // enqueue() never asks for this
    }
    if constexpr ((FEAT & 0x1000) != 0) { // fused sampler tail: the wave <-> chain correspondence of a static grid in the given order
        if (ka.order != nullptr || ka.nuts_tail == nullptr) return hipErrorInvalidValue; // enqueue() never asks for this
        ka.work = nullptr;
        hipLaunchKernelGGL((solve_kernel_fused<T, METHOD, GA, ST, HAS_E, HAS_WANE, HAS_C, W, ND, SPL, FEAT>), dim3((unsigned)blocks),
                           dim3(64), lds, stream, ka, *ka.nuts_tail);
        return hipGetLastError();
    } else {
    ka.nuts_tail = nullptr;
    const auto kernel = solve_kernel<T, METHOD, GA, ST, HAS_E, HAS_WANE, HAS_C, W, ND, SPL, FEAT>;
    int64_t grid = blocks;
    if ((FEAT & 0x0400) != 0 || (FEAT & 0x2000) != 0) ka.work = nullptr;   // (a static-only instance)
    if (ka.work != nullptr) {
        // Work pulling needs a batch of more waves than the chip holds at once: then the grid is exactly the resident waves
        // and every lane group draws trajectories until the queue is empty.  A batch that fits is one wave per TPW
        // trajectories as ever (nothing to pull), and replicated trajectories (small batches) are static by construction.
        // Measured (one box, MI355X, ms per launch, static grid -> pulling; docs/perf-log.md):
        //   given order    D = 360 (TPW 2) B = 16384  2.70 -> 2.67;  D = 136 (TPW 8) B = 65536  3.43-3.55 -> 3.46-3.47;
        //                  cfg 5 B = 65536  3.80 -> 3.76;  cfg 2 (TPW 8, 40-step trajectories) B = 65536  0.79-0.85 -> 1.05
        //   caller's order (most step attempts first)  D = 360  2.40-2.46 -> 2.36-2.39;  cfg 5 B = 65536  3.27 -> 3.23-3.25;
        //                  cfg 2 B = 65536  0.645-0.66 -> 0.625-0.63
        // Every pass of the prologue stalls ALL lane groups of its wave, so with eight short trajectories per wave the passes
        // cost more than the waiting they remove; with a queue sorted by cost, pulling is what makes longest-first work.
        // Since the D = 360 kernel runs three waves per SIMD (waves_per_simd below) the static grid wins there too -- the
        // hardware starts the next wave (pair of trajectories) the moment one retires, and 6144 slots leave 2.7 jobs each:
        //   D = 360 B = 16384, bench.py, three alternations on one box: static 2.332 ms (0.464), pulling 2.385 (0.454);
        //   with the caller's exact order: static 2.24-2.28, pulling 2.30-2.40;  D = 136 B = 65536 in the caller's order:
        //   static 3.186, pulling 3.167;  cfg 5 B = 65536: 3.286 -> 3.259.
        // Default therefore: pull when the caller supplied the queue and a wave holds more than two trajectories.
        // dyn_solver_opts::hints.pull = -1 / 1 forces it off / on, hints.pull_waves = <n> sets the grid (tests, tuning).
        const bool forced = ka.pull_waves > 0;
        const bool want = ka.pull_mode ? ka.pull_mode > 0 : (forced || (TPW > 2 && ka.order != nullptr));
        int64_t resident = forced ? (int64_t)ka.pull_waves : resident_waves(kernel, lds);
        if (!want || ka.rep_log2 != 0 || resident <= 0 || blocks <= resident)
            ka.work = nullptr;
        else
            grid = resident;
    }
    if (lds > 65536) {
        // a save grid beyond the default 64 KB of dynamic LDS (hourly saves over a year in float64 are 70 KB): the kernel
        // attribute has to allow it -- up to the CU's 160 KB, at one or two waves per CU instead of a full complement
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(PC ? 128 : 64), lds, stream, ka);
    return hipGetLastError();
    }
}

} // namespace dyn
