// nuts_device.hpp -- the per-chain NUTS state machine as a device function, shared by the stand-alone sampler kernel
// (nuts_kernel.hip, `nuts_advance`) and the fused gradient-solve + sampler launch (solve_kernel.hpp, FEAT bit 12).
#pragma once
#include "latent_device.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace dynnuts {

constexpr double POOL_SCALE = 1073741824.0; // 2^30 fixed point for the pooled window sums

// Philox4x32-10 (Salmon et al. 2011), counter = (ctr_lo, ctr_hi, chain, 0), key = seed
__host__ __device__ inline void philox4x32_10(const uint32_t (&c)[4], const uint32_t (&k)[2], uint32_t (&o)[4]) {
    uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

struct Philox {
    uint32_t key0, key1;
    uint64_t ctr;
    uint32_t chain;
    __device__ double uniform() { // in (0, 1), 52 random bits
        const uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), chain, 0u}, k[2] = {key0, key1};
        uint32_t o[4];
        philox4x32_10(c, k, o);
        ++ctr;
        const uint64_t bits = ((uint64_t)o[0] << 20) ^ (uint64_t)(o[1] >> 12);
        return ((double)bits + 0.5) * (1.0 / 4503599627370496.0);
    }
    __device__ double normal() {
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    }
};

__device__ inline double logaddexp(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = fmax(a, b);
    return m + log1p(exp(-fabs(a - b)));
}

// The dimension of a chain is a compile-time constant here, up to kRegDim: every loop unrolls, every vector lives in
// registers, one thread runs a chain (the helpers take D as an argument and are always inlined, so the constant propagates).
// Beyond kRegDim the state machine is a kernel of its own with a half wave per chain (nuts_kernel.hip, nuts_advance_lanes).
constexpr int kRegDim = 8;

// y = M v for a row-major D x D matrix
template <int DMAX>
__device__ __forceinline__ void matvec(const double *M, const double *v, double *y, const int D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double a = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) a += M[i * D + j] * v[j];
        y[i] = a;
    }
}
template <int DMAX>
__device__ __forceinline__ double dot(const double *a, const double *b, const int D) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) s += a[i] * b[i];
    return s;
}
template <int DMAX>
__device__ __forceinline__ bool is_turning(const double *imm, const double *rl, const double *rr, const double *rsum, const int D) {
    double rs[DMAX], vl[DMAX], vr[DMAX];
#pragma unroll
    for (int i = 0; i < D; ++i) rs[i] = rsum[i] - 0.5 * (rl[i] + rr[i]);
    matvec<DMAX>(imm, rl, vl, D);
    matvec<DMAX>(imm, rr, vr, D);
    return dot<DMAX>(vl, rs, D) <= 0.0 || dot<DMAX>(vr, rs, D) <= 0.0;
}

// mm_sqrt = chol(inv(imm)) for a symmetric positive definite D x D (Gauss-Jordan + Cholesky).  `a` and `inv`: D x D of
// workspace each.  (nuts_kernel.hip's mass_sqrt_lanes does the same operations in the same order, spread over lanes.)
__device__ inline void mass_sqrt_into(const double *imm, double *out, double *a, double *inv, const int D) {
    for (int i = 0; i < D * D; ++i) a[i] = imm[i];
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) inv[i * D + j] = i == j ? 1.0 : 0.0;
    for (int c = 0; c < D; ++c) {
        const double p = 1.0 / a[c * D + c];
        for (int j = 0; j < D; ++j) { a[c * D + j] *= p; inv[c * D + j] *= p; }
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const double f = a[r * D + c];
            for (int j = 0; j < D; ++j) { a[r * D + j] -= f * a[c * D + j]; inv[r * D + j] -= f * inv[c * D + j]; }
        }
    }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.5 * (inv[i * D + j] + inv[j * D + i]);
            for (int k = 0; k < j; ++k) s -= out[i * D + k] * out[j * D + k];
            out[i * D + j] = i == j ? sqrt(fmax(s, 1e-300)) : s / out[j * D + j];
        }
    for (int i = 0; i < D; ++i)
        for (int j = i + 1; j < D; ++j) out[i * D + j] = 0.0;
}
template <int DMAX>
__device__ inline void mass_sqrt(const double *imm, double *out, const int D) {
    double a[DMAX * DMAX], inv[DMAX * DMAX];
    mass_sqrt_into(imm, out, a, inv, D);
}

// What the solve handed over for chain c: the likelihood side of the potential at z_eval.  `nuts_advance` reads it from the
// buffers of the launch before it; the fused gradient-solve (solve_kernel.hpp, FEAT bit 12) from what its own wave just wrote.
template <int DMAX>
struct Handed {
    double ll;
    double dll[DMAX];
};

// One sampler iteration of chain c (everything between two potential evaluations), in DMAX dimensions.
// INLINE_MAP: this thread also does the map of the position it hands out (dyn_nuts_advance_mapped; the fused launch's tail).
// Without it the caller does -- nuts_kernel.hip spreads that over the lanes of a chain (dynlat::map_chain_lanes) -- and takes
// the position from `ze_out`.  Returns whether a position was handed out (false: no such chain, or a finished one).
template <int DMAX, bool INLINE_MAP = true, typename ST, typename MAP>
__device__ __forceinline__ bool advance_chain(const ST &st, const MAP &map, const int c, const Handed<DMAX> &handed, double *ze_out = nullptr) {
    constexpr int D = DMAX;
    const int C = st.n_chains, Dm = st.max_depth;
    if (c >= C) return false;
    const int total = st.num_warmup + st.num_samples;
    if (st.it[c] >= total) return false; // finished chains idle
    // Scalars of this chain live in registers for the whole call (loaded here, stored at the end);
    // the per-chain vectors are disjoint, which the restrict qualifiers tell the compiler, so loads
    // are not serialised behind the stores of earlier phases.
    double L_u = st.u[c];
    double L_eps = st.eps[c];
    double L_eps_avg = st.eps_avg[c];
    double L_da_mu = st.da_mu[c];
    double L_da_xbar = st.da_xbar[c];
    double L_da_gbar = st.da_gbar[c];
    double L_da_t = st.da_t[c];
    double L_wf_n = st.wf_n[c];
    double L_e0 = st.e0[c];
    double L_up = st.up[c];
    double L_weight = st.weight[c];
    double L_sum_acc = st.sum_acc[c];
    double L_sgn = st.sgn[c];
    double L_s_up = st.s_up[c];
    double L_s_weight = st.s_weight[c];
    double L_s_acc = st.s_acc[c];
    int L_it = st.it[c];
    int L_wi = st.wi[c];
    int L_n_prop = st.n_prop[c];
    int L_depth = st.depth[c];
    int L_right = st.right[c];
    int L_leaf = st.leaf[c];
    int L_s_turn = st.s_turn[c];
    int L_s_div = st.s_div[c];
    int L_s_n = st.s_n[c];
    int L_pend = st.pooled ? st.pend[c] : 0;
    int64_t L_rng_ctr = st.rng_ctr[c];
    double *__restrict__ const p_g = st.g + (int64_t)c * D;
    double *__restrict__ const p_gc = st.gc + (int64_t)c * D;
    double *__restrict__ const p_gl = st.gl + (int64_t)c * D;
    double *__restrict__ const p_gp = st.gp + (int64_t)c * D;
    double *__restrict__ const p_gr = st.gr + (int64_t)c * D;
    double *__restrict__ const p_r_half = st.r_half + (int64_t)c * D;
    double *__restrict__ const p_r_sum = st.r_sum + (int64_t)c * D;
    double *__restrict__ const p_rc = st.rc + (int64_t)c * D;
    double *__restrict__ const p_rl = st.rl + (int64_t)c * D;
    double *__restrict__ const p_rr = st.rr + (int64_t)c * D;
    double *__restrict__ const p_s_gp = st.s_gp + (int64_t)c * D;
    double *__restrict__ const p_s_rsum = st.s_rsum + (int64_t)c * D;
    double *__restrict__ const p_s_zp = st.s_zp + (int64_t)c * D;
    double *__restrict__ const p_wf_mean = st.wf_mean + (int64_t)c * D;
    double *__restrict__ const p_z = st.z + (int64_t)c * D;
    double *__restrict__ const p_zc = st.zc + (int64_t)c * D;
    double *__restrict__ const p_zl = st.zl + (int64_t)c * D;
    double *__restrict__ const p_zp = st.zp + (int64_t)c * D;
    double *__restrict__ const p_zr = st.zr + (int64_t)c * D;
    double *__restrict__ const p_imm = st.imm + (int64_t)c * D * D;
    double *__restrict__ const p_mm_sqrt = st.mm_sqrt + (int64_t)c * D * D;
    double *__restrict__ const p_wf_m2 = st.wf_m2 + (int64_t)c * D * D;
    int it = L_it;
    double *__restrict__ const z = p_z, *__restrict__ const g = p_g, *__restrict__ const zc = p_zc, *__restrict__ const rc = p_rc,
           *__restrict__ const gc = p_gc, *__restrict__ const imm = p_imm, *__restrict__ const mms = p_mm_sqrt;
    const double eps_signed = L_eps * L_sgn;
    Philox rng{(uint32_t)st.seed, (uint32_t)(st.seed >> 32), (uint64_t)L_rng_ctr, (uint32_t)c};

    // ---- finish the leapfrog started by the previous launch: second momentum half step
    // the potential at z_eval: handed over as (u, g), or -- folded potential, infer/folded.py -- as its parts, combined here
    // instead of in a launch of their own: u = -(lp + ll + offset), g = -(dlp + dll)
    double un, gn[DMAX];
    if (st.pot_lp != nullptr) {
        un = -(st.pot_lp[c] + handed.ll + st.pot_offset);
        for (int i = 0; i < D; ++i) gn[i] = -(st.pot_dlp[(int64_t)c * D + i] + handed.dll[i]);
    } else {
        un = st.u_new[c];
        for (int i = 0; i < D; ++i) gn[i] = st.g_new[(int64_t)c * D + i];
    }
    const double *zn = st.z_eval + (int64_t)c * D;
    double rn[DMAX], tmp[DMAX];
    bool bad = !isfinite(un);
    for (int i = 0; i < D; ++i) bad = bad || !isfinite(gn[i]);
    for (int i = 0; i < D; ++i) rn[i] = p_r_half[i] - 0.5 * eps_signed * (bad ? 0.0 : gn[i]);
    matvec<DMAX>(imm, rn, tmp, D);
    double de = (bad ? INFINITY : un) + 0.5 * dot<DMAX>(rn, tmp, D) - L_e0;
    if (isnan(de)) de = INFINITY;
    const double lw = -de;
    const bool div = de > st.max_delta_energy;
    const double acc = exp(fmin(-de, 0.0));

    // ---- grow the subtree by this leaf (multinomial choice inside the subtree)
    double s_weight = L_s_weight;
    const double new_w = logaddexp(s_weight, lw);
    if (rng.uniform() < exp(lw - new_w)) {
        for (int i = 0; i < D; ++i) { p_s_zp[i] = zn[i]; p_s_gp[i] = bad ? 0.0 : gn[i]; }
        L_s_up = bad ? INFINITY : un;
    }
    s_weight = new_w;
    double *s_rsum = p_s_rsum;
    for (int i = 0; i < D; ++i) s_rsum[i] += rn[i];
    bool s_div = L_s_div != 0 || div;
    const double s_acc = L_s_acc + acc;
    const int s_n = L_s_n + 1;
    for (int i = 0; i < D; ++i) { zc[i] = zn[i]; rc[i] = rn[i]; gc[i] = bad ? 0.0 : gn[i]; }

    // ---- checkpointed U-turn test
    int leaf = L_leaf;
    const int idx_max = __popc((unsigned)(leaf >> 1));
    int trailing = 0;
    while ((leaf >> trailing) & 1) ++trailing;
    const int idx_min = idx_max - trailing + 1;
    double *r_ck = st.r_ck + (int64_t)c * Dm * D, *rs_ck = st.rs_ck + (int64_t)c * Dm * D;
    bool s_turn = L_s_turn != 0;
    if ((leaf & 1) == 0) {
        for (int i = 0; i < D; ++i) { r_ck[idx_max * D + i] = rn[i]; rs_ck[idx_max * D + i] = s_rsum[i]; }
    } else {
        for (int l = idx_max; l >= idx_min; --l) {
            double sub[DMAX];
            for (int i = 0; i < D; ++i) sub[i] = s_rsum[i] - rs_ck[l * D + i] + r_ck[l * D + i];
            s_turn = s_turn || is_turning<DMAX>(imm, r_ck + l * D, rn, sub, D);
        }
    }
    ++leaf;

    // ---- subtree complete -> merge into the trajectory (biased progressive sampling)
    int depth = L_depth;
    const bool right = L_right != 0;
    const bool sub_done = s_turn || s_div || leaf >= (1 << depth);
    bool stop = false;
    if (sub_done) {
        const bool ok = !s_turn && !s_div;
        if (ok && rng.uniform() < exp(fmin(s_weight - L_weight, 0.0))) {
            for (int i = 0; i < D; ++i) { p_zp[i] = p_s_zp[i]; p_gp[i] = p_s_gp[i]; }
            L_up = L_s_up;
        }
        double *ze = right ? p_zr : p_zl, *re = right ? p_rr : p_rl, *ge = right ? p_gr : p_gl;
        for (int i = 0; i < D; ++i) { ze[i] = zc[i]; re[i] = rc[i]; ge[i] = gc[i]; }
        L_weight = logaddexp(L_weight, s_weight);
        for (int i = 0; i < D; ++i) p_r_sum[i] += s_rsum[i];
        L_sum_acc += s_acc;
        L_n_prop += s_n;
        ++depth;
        stop = s_turn || s_div || is_turning<DMAX>(imm, p_rl, p_rr, p_r_sum, D) || depth >= Dm;
    }

    double eps = L_eps;
    if (stop) {
        // ---- transition complete: adapt, record, next transition
        const bool warm = it < st.num_warmup;
        const int n_prop = L_n_prop;
        const double a_prob = L_sum_acc / (double)(n_prop > 0 ? n_prop : 1);
        for (int i = 0; i < D; ++i) { z[i] = p_zp[i]; g[i] = p_gp[i]; }
        L_u = L_up;
        if (warm) {
            // dual averaging (Stan / numpyro constants: t0 = 10, kappa = 0.75, gamma = 0.05)
            const double t1 = L_da_t + 1.0, w = 1.0 / (t1 + 10.0);
            const double gbar = (1.0 - w) * L_da_gbar + w * (st.target_accept - a_prob);
            const double x = L_da_mu - sqrt(t1) / 0.05 * gbar;
            const double wx = pow(t1, -0.75);
            const double xbar = (1.0 - wx) * L_da_xbar + wx * x;
            L_da_t = t1; L_da_gbar = gbar; L_da_xbar = xbar;
            eps = exp(x);
            L_eps_avg = exp(xbar);
            if (st.pooled && L_pend > 0) {
                // pooled window statistics of every chain that has closed this window so far
                // (pool_ro = the pool as it stood after the previous launch: no concurrent writers)
                const int64_t *pw = st.pool_ro + (int64_t)(L_pend - 1) * (1 + D + D * D);
                const double N = (double)pw[0], nn = fmax(N, 2.0);
                double mu[DMAX], cand[DMAX * DMAX], chol[DMAX * DMAX];
                for (int i = 0; i < D; ++i) mu[i] = (double)pw[1 + i] / POOL_SCALE / N;
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) {
                        const double cov = ((double)pw[1 + D + i * D + j] / POOL_SCALE - N * mu[i] * mu[j]) / (nn - 1.0);
                        cand[i * D + j] = (nn / (nn + 5.0)) * cov + (i == j ? 1e-3 * (5.0 / (nn + 5.0)) : 0.0);
                    }
                mass_sqrt<DMAX>(cand, chol, D);
                bool good = N >= 2.0;
                for (int i = 0; i < D * D; ++i) good = good && isfinite(cand[i]) && isfinite(chol[i]);
                for (int i = 0; i < D; ++i) good = good && chol[i * D + i] > 0.0 && cand[i * D + i] > 0.0;
                if (good) {
                    for (int i = 0; i < D * D; ++i) { imm[i] = cand[i]; mms[i] = chol[i]; }
                    eps = L_eps_avg;
                    L_da_mu = log(10.0 * eps);
                    L_da_t = 0.0; L_da_gbar = 0.0; L_da_xbar = 0.0;
                }
                L_pend = 0;
            }
            // windowed dense mass matrix (Welford), applied with its Cholesky factor at window end
            const int wi = L_wi;
            if (wi < st.n_windows && it >= st.w_start[wi] && it < st.w_end[wi]) {
                const double n1 = L_wf_n + 1.0;
                double d0[DMAX];
                double *mean = p_wf_mean, *m2 = p_wf_m2;
                for (int i = 0; i < D; ++i) { d0[i] = z[i] - mean[i]; mean[i] += d0[i] / n1; }
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) m2[i * D + j] += d0[i] * (z[j] - mean[j]);
                L_wf_n = n1;
                if (it + 1 == st.w_end[wi]) {
                    if (st.pooled) {
                        // contribute this chain's window to the pool (fixed point: the sums do not
                        // depend on the order of the atomics); applied at the NEXT transition end
                        auto add = [](int64_t *p, double v) {
                            atomicAdd((unsigned long long *)p, (unsigned long long)llrint(v * POOL_SCALE));
                        };
                        int64_t *pw = st.pool + (int64_t)wi * (1 + D + D * D);
                        atomicAdd((unsigned long long *)pw, (unsigned long long)n1);
                        for (int i = 0; i < D; ++i) add(pw + 1 + i, n1 * mean[i]);
                        for (int i = 0; i < D; ++i)
                            for (int j = 0; j < D; ++j)
                                add(pw + 1 + D + i * D + j, m2[i * D + j] + n1 * mean[i] * mean[j]);
                        L_pend = wi + 1;
                    } else {
                        const double nn = fmax(n1, 2.0);
                        for (int i = 0; i < D; ++i)
                            for (int j = 0; j < D; ++j)
                                imm[i * D + j] = (nn / (nn + 5.0)) * m2[i * D + j] / (nn - 1.0) +
                                                 (i == j ? 1e-3 * (5.0 / (nn + 5.0)) : 0.0);
                        mass_sqrt<DMAX>(imm, mms, D);
                        eps = L_eps_avg; // restart dual averaging around the running average
                        L_da_mu = log(10.0 * eps);
                        L_da_t = 0.0; L_da_gbar = 0.0; L_da_xbar = 0.0;
                    }
                    L_wf_n = 0.0;
                    for (int i = 0; i < D; ++i) mean[i] = 0.0;
                    for (int i = 0; i < D * D; ++i) m2[i] = 0.0;
                    L_wi = wi + 1;
                }
            }
            if (it + 1 == st.num_warmup) {
                eps = L_eps_avg;
                if (st.pooled) {
                    // final step size = geometric mean over the chains that have finished warm-up so
                    // far (they share one mass matrix, so the ideal step is the same; the 50-transition
                    // final buffer alone leaves a factor ~2 of dual-averaging noise between chains)
                    const int64_t stride = 1 + D + D * D;
                    int64_t *pw = st.pool + (int64_t)st.n_windows * stride;
                    const int64_t *pr = st.pool_ro + (int64_t)st.n_windows * stride;
                    const double le = log(L_eps_avg);
                    atomicAdd((unsigned long long *)pw, 1ull);
                    atomicAdd((unsigned long long *)(pw + 1), (unsigned long long)llrint(le * POOL_SCALE));
                    eps = exp(((double)pr[1] / POOL_SCALE + le) / ((double)pr[0] + 1.0));
                }
            }
        } else {
            const int j = it - st.num_warmup;
            for (int i = 0; i < D; ++i) st.out_z[((int64_t)c * st.num_samples + j) * D + i] = z[i];
            st.out_acc[(int64_t)c * st.num_samples + j] = a_prob;
            st.out_n[(int64_t)c * st.num_samples + j] = n_prop;
            st.out_div[(int64_t)c * st.num_samples + j] = s_div ? 1 : 0;
        }
        L_eps = eps;
        L_it = ++it;
        // fresh momentum r0 = chol(M) * normal, new trajectory = the single point (z, r0)
        double nrm[DMAX], r0[DMAX];
        for (int i = 0; i < D; ++i) nrm[i] = rng.normal();
        matvec<DMAX>(mms, nrm, r0, D);
        matvec<DMAX>(imm, r0, tmp, D);
        L_e0 = L_u + 0.5 * dot<DMAX>(r0, tmp, D);
        for (int i = 0; i < D; ++i) {
            p_zl[i] = p_zr[i] = p_zp[i] = z[i];
            p_rl[i] = p_rr[i] = p_r_sum[i] = r0[i];
            p_gl[i] = p_gr[i] = p_gp[i] = g[i];
        }
        L_up = L_u;
        L_weight = 0.0; L_sum_acc = 0.0; L_n_prop = 0;
        depth = 0;
    }
    L_depth = depth;

    bool go_right = right;
    if (sub_done) {
        // ---- next subtree (next doubling, or the first of a new transition)
        go_right = rng.uniform() < 0.5;
        L_right = go_right ? 1 : 0;
        L_sgn = go_right ? 1.0 : -1.0;
        const double *ze = go_right ? p_zr : p_zl, *re = go_right ? p_rr : p_rl, *ge = go_right ? p_gr : p_gl;
        for (int i = 0; i < D; ++i) { zc[i] = ze[i]; rc[i] = re[i]; gc[i] = ge[i]; }
        for (int i = 0; i < D; ++i) { p_s_zp[i] = p_zp[i]; p_s_gp[i] = p_gp[i]; s_rsum[i] = 0.0; }
        L_s_up = L_up;
        s_weight = -INFINITY;
        L_s_acc = 0.0; L_s_n = 0;
        s_turn = false; s_div = false;
        leaf = 0;
        for (int i = 0; i < Dm * D; ++i) { r_ck[i] = 0.0; rs_ck[i] = 0.0; }
    } else {
        L_s_acc = s_acc; L_s_n = s_n;
    }
    L_s_weight = s_weight;
    L_s_turn = s_turn ? 1 : 0;
    L_s_div = s_div ? 1 : 0;
    L_leaf = leaf;

    // ---- first half of the next leapfrog: r_half, and the position the potential is needed at
    const double es = eps * (go_right ? 1.0 : -1.0);
    double rh[DMAX];
    for (int i = 0; i < D; ++i) rh[i] = rc[i] - 0.5 * es * gc[i];
    matvec<DMAX>(imm, rh, tmp, D);
    double ze[DMAX];
    for (int i = 0; i < D; ++i) {
        p_r_half[i] = rh[i];
        ze[i] = (it >= total) ? z[i] : zc[i] + es * tmp[i];
        st.z_eval[(int64_t)c * D + i] = ze[i];
        if constexpr (!INLINE_MAP) ze_out[i] = ze[i];
    }
    // dyn_nuts_advance_mapped: the prior side of the potential at that position and the parameter rows / tangent seeds of
    // the solve that comes next -- what dyn_latent_param_map would do in a launch of its own (the potential parts read at the
    // top of this call were consumed above, so their buffers can take the next position's values now)
    if constexpr (INLINE_MAP) if (map.enabled) {
        if (map.f64)
            dynlat::map_chain<double, DMAX>(map.tab, st.n_chains, c, ze, map.x, map.lp, map.dlp_dz, map.P, map.coef, map.expo,
                                      (double *)map.params, (double *)map.seeds, map.split);
        else
            dynlat::map_chain<float, DMAX>(map.tab, st.n_chains, c, ze, map.x, map.lp, map.dlp_dz, map.P, map.coef, map.expo,
                                     (float *)map.params, (float *)map.seeds, map.split);
    }
    st.rng_ctr[c] = (int64_t)rng.ctr;
    st.u[c] = L_u;
    st.eps[c] = L_eps;
    st.eps_avg[c] = L_eps_avg;
    st.da_mu[c] = L_da_mu;
    st.da_xbar[c] = L_da_xbar;
    st.da_gbar[c] = L_da_gbar;
    st.da_t[c] = L_da_t;
    st.wf_n[c] = L_wf_n;
    st.e0[c] = L_e0;
    st.up[c] = L_up;
    st.weight[c] = L_weight;
    st.sum_acc[c] = L_sum_acc;
    st.sgn[c] = L_sgn;
    st.s_up[c] = L_s_up;
    st.s_weight[c] = L_s_weight;
    st.s_acc[c] = L_s_acc;
    st.it[c] = L_it;
    st.wi[c] = L_wi;
    st.n_prop[c] = L_n_prop;
    st.depth[c] = L_depth;
    st.right[c] = L_right;
    st.leaf[c] = L_leaf;
    st.s_turn[c] = L_s_turn;
    st.s_div[c] = L_s_div;
    st.s_n[c] = L_s_n;
    if (st.pooled) st.pend[c] = L_pend;
    return true;
}


// The fused launch's view of a sampler run: written by dyn_nuts_tail_pack into HOST memory of the caller, handed to the
// fused kernel BY VALUE as its second argument -- in the kernel-argument segment the fields are scalar loads and the pointers
// inside are known to be global memory, as for nuts_advance.
constexpr uint64_t kTailMagic = 0x4c4941545354554eull; // "NUTSTAIL"
struct Tail {
    uint64_t magic;
    dyn_nuts_state st;
    dynlat::MapArgs map;
    int32_t rows_per_chain; // trajectories of one chain in the solve's batch (directions split: >= n_sites, rows beyond the sites are padding; else 1)
};

__device__ inline double load_written(const double *p) { // what this wave stored a moment ago (past the vector L1)
    const unsigned long long v = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(double, v);
}

template <int D, typename TL>
__device__ __forceinline__ void tail_chain(const TL &tl, const int c, const double *ll_out, const double *dll_out) {
    Handed<D> handed;
    handed.ll = load_written(ll_out + (int64_t)c * tl.rows_per_chain);
    // directions split: [C rows][1], row r of a chain carries direction r (rows beyond D are padding); else [C][D]
    const int64_t first = (int64_t)c * (tl.map.split ? tl.rows_per_chain : D);
    for (int i = 0; i < D; ++i) handed.dll[i] = load_written(dll_out + first + i);
    advance_chain<D>(tl.st, tl.map, c, handed);
}

constexpr int kFusedMaxDim = kRegDim;   // every compile-time-dimension instance of the state machine ...
constexpr int kFusedLeanMaxDim = 4;     // ... in a general tangent instance; a lean one (solve_kernel.hpp) stops here: the cases beyond
                                        // cost the cfg 4 kernel 116 bytes of scratch per lane and twice the spilled scalars

// Chain c's iteration inside the gradient-solve launch (solve_kernel.hpp): called by ONE lane per chain, after the wave's
// stores of ll_out / dll_out have completed.  TL: `Tail` in the address space the caller reads it from (the kernarg segment).
template <int MAXD, typename TL>
__device__ __forceinline__ void fused_tail(const TL &tl, const int c, const double *ll_out, const double *dll_out) {
    static_assert(MAXD == kFusedLeanMaxDim || MAXD == kFusedMaxDim, "fused tail: four or eight dimensions");
    switch (tl.st.dim) {
    case 1: tail_chain<1>(tl, c, ll_out, dll_out); break;
    case 2: tail_chain<2>(tl, c, ll_out, dll_out); break;
    case 3: tail_chain<3>(tl, c, ll_out, dll_out); break;
    case 4: tail_chain<4>(tl, c, ll_out, dll_out); break;
    default:
        if constexpr (MAXD > 4) {
            switch (tl.st.dim) {
            case 5: tail_chain<5>(tl, c, ll_out, dll_out); break;
            case 6: tail_chain<6>(tl, c, ll_out, dll_out); break;
            case 7: tail_chain<7>(tl, c, ll_out, dll_out); break;
            case 8: tail_chain<8>(tl, c, ll_out, dll_out); break;
            default: break; // (dyn_nuts_tail_pack refuses other sizes)
            }
        }
        break; // (enqueue() never hands a lean instance more than kFusedLeanMaxDim)
    }
}

} // namespace dynnuts

#pragma clang fp contract(fast)   // (back to hipcc's default: see latent_device.hpp)
