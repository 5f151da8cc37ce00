// nuts_kernel.hip -- one HIP kernel per sampler iteration: the asynchronous batched NUTS state
// machine of dynode_amd/infer/nuts.py (`GraphNUTS._step`), one thread per chain.
//
// Under numpyro (reference src/dynode/infer/inference.py:149-163) the NUTS transition is traced
// into the same XLA program as the model.  Here the model's potential/gradient stays a torch
// program (user Python + the fused gradient-solve kernel, captured in a HIP graph), and everything
// else a chain does between two potential evaluations -- finishing the leapfrog, energy error,
// multinomial / biased-progressive proposal selection, the checkpointed U-turn test, tree and
// transition bookkeeping, dual-averaging step size, windowed dense mass-matrix adaptation with
// its Cholesky factor, recording the draw, fresh momentum and direction, the first half of the
// next leapfrog -- is this single launch (about 400 tiny torch kernels before).  Randomness is
// Philox4x32-10 keyed by (seed, chain) with a per-chain counter, so a chain's stream does not
// depend on the other chains.
#include "../../include/dynode_hip.h"

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

// y = M v for a row-major D x D matrix (D is a compile-time constant: everything stays in registers)
template <int D>
__device__ inline void matvec(const double *M, const double *v, double *y) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double a = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) a += M[i * D + j] * v[j];
        y[i] = a;
    }
}
template <int D>
__device__ inline double dot(const double *a, const double *b) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) s += a[i] * b[i];
    return s;
}
template <int D>
__device__ inline bool is_turning(const double *imm, const double *rl, const double *rr, const double *rsum) {
    double rs[D], vl[D], vr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) rs[i] = rsum[i] - 0.5 * (rl[i] + rr[i]);
    matvec<D>(imm, rl, vl);
    matvec<D>(imm, rr, vr);
    return dot<D>(vl, rs) <= 0.0 || dot<D>(vr, rs) <= 0.0;
}

// mm_sqrt = chol(inv(imm)) for a symmetric positive definite D x D (Gauss-Jordan + Cholesky)
template <int D>
__device__ inline void mass_sqrt(const double *imm, double *out) {
    double a[D * D], inv[D * D];
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
    for (int i = 0; i < D * D; ++i) out[i] = 0.0;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.5 * (inv[i * D + j] + inv[j * D + i]);
            for (int k = 0; k < j; ++k) s -= out[i * D + k] * out[j * D + k];
            out[i * D + j] = i == j ? sqrt(fmax(s, 1e-300)) : s / out[j * D + j];
        }
}

template <int D>
__global__ void __launch_bounds__(64) nuts_advance(const dyn_nuts_state st) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int C = st.n_chains, Dm = st.max_depth;
    if (c >= C) return;
    const int total = st.num_warmup + st.num_samples;
    int it = st.it[c];
    if (it >= total) return; // finished chains idle
#define V(f) (st.f + (int64_t)c * D)
#define M2(f) (st.f + (int64_t)c * D * D)
    double *z = V(z), *g = V(g), *zc = V(zc), *rc = V(rc), *gc = V(gc);
    double *imm = M2(imm), *mms = M2(mm_sqrt);
    const double eps_signed = st.eps[c] * st.sgn[c];
    Philox rng{(uint32_t)st.seed, (uint32_t)(st.seed >> 32), (uint64_t)st.rng_ctr[c], (uint32_t)c};

    // ---- finish the leapfrog started by the previous launch: second momentum half step
    const double un = st.u_new[c];
    const double *gn = st.g_new + (int64_t)c * D;
    const double *zn = st.z_eval + (int64_t)c * D;
    double rn[D], tmp[D];
    bool bad = !isfinite(un);
    for (int i = 0; i < D; ++i) bad = bad || !isfinite(gn[i]);
    for (int i = 0; i < D; ++i) rn[i] = V(r_half)[i] - 0.5 * eps_signed * (bad ? 0.0 : gn[i]);
    matvec<D>(imm, rn, tmp);
    double de = (bad ? INFINITY : un) + 0.5 * dot<D>(rn, tmp) - st.e0[c];
    if (isnan(de)) de = INFINITY;
    const double lw = -de;
    const bool div = de > st.max_delta_energy;
    const double acc = exp(fmin(-de, 0.0));

    // ---- grow the subtree by this leaf (multinomial choice inside the subtree)
    double s_weight = st.s_weight[c];
    const double new_w = logaddexp(s_weight, lw);
    if (rng.uniform() < exp(lw - new_w)) {
        for (int i = 0; i < D; ++i) { V(s_zp)[i] = zn[i]; V(s_gp)[i] = bad ? 0.0 : gn[i]; }
        st.s_up[c] = bad ? INFINITY : un;
    }
    s_weight = new_w;
    double *s_rsum = V(s_rsum);
    for (int i = 0; i < D; ++i) s_rsum[i] += rn[i];
    bool s_div = st.s_div[c] != 0 || div;
    const double s_acc = st.s_acc[c] + acc;
    const int s_n = st.s_n[c] + 1;
    for (int i = 0; i < D; ++i) { zc[i] = zn[i]; rc[i] = rn[i]; gc[i] = bad ? 0.0 : gn[i]; }

    // ---- checkpointed U-turn test
    int leaf = st.leaf[c];
    const int idx_max = __popc((unsigned)(leaf >> 1));
    int trailing = 0;
    while ((leaf >> trailing) & 1) ++trailing;
    const int idx_min = idx_max - trailing + 1;
    double *r_ck = st.r_ck + (int64_t)c * Dm * D, *rs_ck = st.rs_ck + (int64_t)c * Dm * D;
    bool s_turn = st.s_turn[c] != 0;
    if ((leaf & 1) == 0) {
        for (int i = 0; i < D; ++i) { r_ck[idx_max * D + i] = rn[i]; rs_ck[idx_max * D + i] = s_rsum[i]; }
    } else {
        for (int l = idx_max; l >= idx_min; --l) {
            double sub[D];
            for (int i = 0; i < D; ++i) sub[i] = s_rsum[i] - rs_ck[l * D + i] + r_ck[l * D + i];
            s_turn = s_turn || is_turning<D>(imm, r_ck + l * D, rn, sub);
        }
    }
    ++leaf;

    // ---- subtree complete -> merge into the trajectory (biased progressive sampling)
    int depth = st.depth[c];
    const bool right = st.right[c] != 0;
    const bool sub_done = s_turn || s_div || leaf >= (1 << depth);
    bool stop = false;
    if (sub_done) {
        const bool ok = !s_turn && !s_div;
        if (ok && rng.uniform() < exp(fmin(s_weight - st.weight[c], 0.0))) {
            for (int i = 0; i < D; ++i) { V(zp)[i] = V(s_zp)[i]; V(gp)[i] = V(s_gp)[i]; }
            st.up[c] = st.s_up[c];
        }
        double *ze = right ? V(zr) : V(zl), *re = right ? V(rr) : V(rl), *ge = right ? V(gr) : V(gl);
        for (int i = 0; i < D; ++i) { ze[i] = zc[i]; re[i] = rc[i]; ge[i] = gc[i]; }
        st.weight[c] = logaddexp(st.weight[c], s_weight);
        for (int i = 0; i < D; ++i) V(r_sum)[i] += s_rsum[i];
        st.sum_acc[c] += s_acc;
        st.n_prop[c] += s_n;
        ++depth;
        stop = s_turn || s_div || is_turning<D>(imm, V(rl), V(rr), V(r_sum)) || depth >= Dm;
    }

    double eps = st.eps[c];
    if (stop) {
        // ---- transition complete: adapt, record, next transition
        const bool warm = it < st.num_warmup;
        const int n_prop = st.n_prop[c];
        const double a_prob = st.sum_acc[c] / (double)(n_prop > 0 ? n_prop : 1);
        for (int i = 0; i < D; ++i) { z[i] = V(zp)[i]; g[i] = V(gp)[i]; }
        st.u[c] = st.up[c];
        if (warm) {
            // dual averaging (Stan / numpyro constants: t0 = 10, kappa = 0.75, gamma = 0.05)
            const double t1 = st.da_t[c] + 1.0, w = 1.0 / (t1 + 10.0);
            const double gbar = (1.0 - w) * st.da_gbar[c] + w * (st.target_accept - a_prob);
            const double x = st.da_mu[c] - sqrt(t1) / 0.05 * gbar;
            const double wx = pow(t1, -0.75);
            const double xbar = (1.0 - wx) * st.da_xbar[c] + wx * x;
            st.da_t[c] = t1; st.da_gbar[c] = gbar; st.da_xbar[c] = xbar;
            eps = exp(x);
            st.eps_avg[c] = exp(xbar);
            if (st.pooled && st.pend[c] > 0) {
                // pooled window statistics of every chain that has closed this window so far
                // (pool_ro = the pool as it stood after the previous launch: no concurrent writers)
                const int64_t *pw = st.pool_ro + (int64_t)(st.pend[c] - 1) * (1 + D + D * D);
                const double N = (double)pw[0], nn = fmax(N, 2.0);
                double mu[D], cand[D * D], chol[D * D];
                for (int i = 0; i < D; ++i) mu[i] = (double)pw[1 + i] / POOL_SCALE / N;
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) {
                        const double cov = ((double)pw[1 + D + i * D + j] / POOL_SCALE - N * mu[i] * mu[j]) / (nn - 1.0);
                        cand[i * D + j] = (nn / (nn + 5.0)) * cov + (i == j ? 1e-3 * (5.0 / (nn + 5.0)) : 0.0);
                    }
                mass_sqrt<D>(cand, chol);
                bool good = N >= 2.0;
                for (int i = 0; i < D * D; ++i) good = good && isfinite(cand[i]) && isfinite(chol[i]);
                for (int i = 0; i < D; ++i) good = good && chol[i * D + i] > 0.0 && cand[i * D + i] > 0.0;
                if (good) {
                    for (int i = 0; i < D * D; ++i) { imm[i] = cand[i]; mms[i] = chol[i]; }
                    eps = st.eps_avg[c];
                    st.da_mu[c] = log(10.0 * eps);
                    st.da_t[c] = 0.0; st.da_gbar[c] = 0.0; st.da_xbar[c] = 0.0;
                }
                st.pend[c] = 0;
            }
            // windowed dense mass matrix (Welford), applied with its Cholesky factor at window end
            const int wi = st.wi[c];
            if (wi < st.n_windows && it >= st.w_start[wi] && it < st.w_end[wi]) {
                const double n1 = st.wf_n[c] + 1.0;
                double d0[D];
                double *mean = V(wf_mean), *m2 = M2(wf_m2);
                for (int i = 0; i < D; ++i) { d0[i] = z[i] - mean[i]; mean[i] += d0[i] / n1; }
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) m2[i * D + j] += d0[i] * (z[j] - mean[j]);
                st.wf_n[c] = n1;
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
                        st.pend[c] = wi + 1;
                    } else {
                        const double nn = fmax(n1, 2.0);
                        for (int i = 0; i < D; ++i)
                            for (int j = 0; j < D; ++j)
                                imm[i * D + j] = (nn / (nn + 5.0)) * m2[i * D + j] / (nn - 1.0) +
                                                 (i == j ? 1e-3 * (5.0 / (nn + 5.0)) : 0.0);
                        mass_sqrt<D>(imm, mms);
                        eps = st.eps_avg[c]; // restart dual averaging around the running average
                        st.da_mu[c] = log(10.0 * eps);
                        st.da_t[c] = 0.0; st.da_gbar[c] = 0.0; st.da_xbar[c] = 0.0;
                    }
                    st.wf_n[c] = 0.0;
                    for (int i = 0; i < D; ++i) mean[i] = 0.0;
                    for (int i = 0; i < D * D; ++i) m2[i] = 0.0;
                    st.wi[c] = wi + 1;
                }
            }
            if (it + 1 == st.num_warmup) eps = st.eps_avg[c];
        } else {
            const int j = it - st.num_warmup;
            for (int i = 0; i < D; ++i) st.out_z[((int64_t)c * st.num_samples + j) * D + i] = z[i];
            st.out_acc[(int64_t)c * st.num_samples + j] = a_prob;
            st.out_n[(int64_t)c * st.num_samples + j] = n_prop;
            st.out_div[(int64_t)c * st.num_samples + j] = s_div ? 1 : 0;
        }
        st.eps[c] = eps;
        st.it[c] = ++it;
        // fresh momentum r0 = chol(M) * normal, new trajectory = the single point (z, r0)
        double nrm[D], r0[D];
        for (int i = 0; i < D; ++i) nrm[i] = rng.normal();
        matvec<D>(mms, nrm, r0);
        matvec<D>(imm, r0, tmp);
        st.e0[c] = st.u[c] + 0.5 * dot<D>(r0, tmp);
        for (int i = 0; i < D; ++i) {
            V(zl)[i] = V(zr)[i] = V(zp)[i] = z[i];
            V(rl)[i] = V(rr)[i] = V(r_sum)[i] = r0[i];
            V(gl)[i] = V(gr)[i] = V(gp)[i] = g[i];
        }
        st.up[c] = st.u[c];
        st.weight[c] = 0.0; st.sum_acc[c] = 0.0; st.n_prop[c] = 0;
        depth = 0;
    }
    st.depth[c] = depth;

    bool go_right = right;
    if (sub_done) {
        // ---- next subtree (next doubling, or the first of a new transition)
        go_right = rng.uniform() < 0.5;
        st.right[c] = go_right ? 1 : 0;
        st.sgn[c] = go_right ? 1.0 : -1.0;
        const double *ze = go_right ? V(zr) : V(zl), *re = go_right ? V(rr) : V(rl), *ge = go_right ? V(gr) : V(gl);
        for (int i = 0; i < D; ++i) { zc[i] = ze[i]; rc[i] = re[i]; gc[i] = ge[i]; }
        for (int i = 0; i < D; ++i) { V(s_zp)[i] = V(zp)[i]; V(s_gp)[i] = V(gp)[i]; s_rsum[i] = 0.0; }
        st.s_up[c] = st.up[c];
        s_weight = -INFINITY;
        st.s_acc[c] = 0.0; st.s_n[c] = 0;
        s_turn = false; s_div = false;
        leaf = 0;
        for (int i = 0; i < Dm * D; ++i) { r_ck[i] = 0.0; rs_ck[i] = 0.0; }
    } else {
        st.s_acc[c] = s_acc; st.s_n[c] = s_n;
    }
    st.s_weight[c] = s_weight;
    st.s_turn[c] = s_turn ? 1 : 0;
    st.s_div[c] = s_div ? 1 : 0;
    st.leaf[c] = leaf;

    // ---- first half of the next leapfrog: r_half, and the position the potential is needed at
    const double es = eps * (go_right ? 1.0 : -1.0);
    double rh[D];
    for (int i = 0; i < D; ++i) rh[i] = rc[i] - 0.5 * es * gc[i];
    matvec<D>(imm, rh, tmp);
    for (int i = 0; i < D; ++i) {
        V(r_half)[i] = rh[i];
        st.z_eval[(int64_t)c * D + i] = (it >= total) ? z[i] : zc[i] + es * tmp[i];
    }
    st.rng_ctr[c] = (int64_t)rng.ctr;
#undef V
#undef M2
}

} // namespace dynnuts

extern "C" int32_t dyn_nuts_state_size(void) { return (int32_t)sizeof(dyn_nuts_state); }

extern "C" void dyn_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    const uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    uint32_t o[4];
    dynnuts::philox4x32_10(c, k, o);
    for (int i = 0; i < 4; ++i) out[i] = o[i];
}

extern "C" int dyn_nuts_advance(const dyn_nuts_state *st, void *stream) {
    if (!st) return DYN_ERR_NULL;
    if (st->n_chains < 0 || st->dim < 1 || st->dim > DYN_NUTS_MAX_DIM || st->max_depth < 1 ||
        st->max_depth > DYN_NUTS_MAX_DEPTH || st->n_windows < 0 || st->n_windows > DYN_NUTS_MAX_WINDOWS)
        return DYN_ERR_SIZE;
    if (st->n_chains == 0) return 0;
    const unsigned blocks = (unsigned)((st->n_chains + 63) / 64);
    if (st->pooled && (!st->pool || !st->pool_ro || !st->pend)) return DYN_ERR_NULL;
    using Kern = void (*)(const dyn_nuts_state);
    static const Kern kernels[DYN_NUTS_MAX_DIM] = {
        dynnuts::nuts_advance<1>, dynnuts::nuts_advance<2>, dynnuts::nuts_advance<3>, dynnuts::nuts_advance<4>,
        dynnuts::nuts_advance<5>, dynnuts::nuts_advance<6>, dynnuts::nuts_advance<7>, dynnuts::nuts_advance<8>};
    hipLaunchKernelGGL(kernels[st->dim - 1], dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st);
    if (hipGetLastError() != hipSuccess) return DYN_ERR_LAUNCH;
    if (st->pooled && st->n_windows > 0) {
        // readers of the next launch see the pool as it stands now, never a half-updated one
        const size_t bytes = sizeof(int64_t) * (size_t)st->n_windows * (size_t)(1 + st->dim + st->dim * st->dim);
        if (hipMemcpyAsync(st->pool_ro, st->pool, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return DYN_ERR_LAUNCH;
    }
    return 0;
}
