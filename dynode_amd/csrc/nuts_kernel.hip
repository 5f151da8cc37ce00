// nuts_kernel.hip -- one HIP kernel per sampler iteration: the asynchronous batched NUTS state
// machine of dynode_amd/infer/nuts.py (`GraphNUTS._step`): one thread per chain up to eight dimensions (`nuts_advance`,
// state machine in nuts_device.hpp), a half wave per chain beyond (`nuts_advance_lanes` below).  tests/nuts_twin.py restates a
// launch in NumPy; the GPU suite holds every form to it launch by launch.
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
#include "nuts_device.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string.h>

int dynlat_rows_per_chain(int32_t split_directions, int32_t n_sites);   // latent_kernel.hip

namespace dynnuts {

// NS lanes per chain: lane 0 of the group runs the chain's state machine; with a map behind it (dyn_nuts_advance_mapped) the NS
// lanes then share the map of the position handed out, one site each (dynlat::map_chain_lanes).  NS = 1: one thread per chain.
template <int NS, typename MAP>
__device__ __forceinline__ void map_lanes(const MAP &map, bool go, const double *ze, int dim, int c, int sub) {
    const int leader = (int)(threadIdx.x & 63) - sub;
    go = __shfl((int)go, leader) != 0;
    double z_sub = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double zi = __shfl(ze[i], leader);
        if (i == sub) z_sub = zi;
    }
    if (!go) return;
    if (map.f64)
        dynlat::map_chain_lanes<double, NS>(map.tab, c, sub, leader, z_sub, map.x, map.lp, map.dlp_dz, map.P, map.coef, map.expo,
                                            (double *)map.params, (double *)map.seeds, map.split);
    else
        dynlat::map_chain_lanes<float, NS>(map.tab, c, sub, leader, z_sub, map.x, map.lp, map.dlp_dz, map.P, map.coef, map.expo,
                                           (float *)map.params, (float *)map.seeds, map.split);
}

template <int D, int NS>
__global__ void __launch_bounds__(64) nuts_advance(const dyn_nuts_state st, const dynlat::MapArgs map) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int c = t / NS, sub = t % NS;
    bool go = false;
    double ze[NS > D ? NS : D];
#pragma unroll
    for (int i = 0; i < (NS > D ? NS : D); ++i) ze[i] = 0.0;
    if (sub == 0 && c < st.n_chains) {
        Handed<D> handed;
        handed.ll = 0.0;
        for (int i = 0; i < D; ++i) handed.dll[i] = 0.0;
        if (st.pot_lp != nullptr) {
            handed.ll = st.pot_ll[(int64_t)c * st.pot_ll_stride];
            const int64_t first = (int64_t)c * (st.pot_dll_stride > 0 ? st.pot_dll_stride : D);
            for (int i = 0; i < D; ++i) handed.dll[i] = st.pot_dll[first + i];
        }
        go = advance_chain<D, NS == 1>(st, map, c, handed, ze);
    }
    if constexpr (NS > 1)
        if (map.enabled) map_lanes<NS>(map, go, ze, D, c, sub);     // (wave-uniform)
}

// ---- beyond kRegDim dimensions: DYN_NUTS_MAX_DIM (32) lanes per chain.  Lane l of a chain's group holds element l of every
// per-chain vector in a register (loaded and stored coalesced) and row l of the matrices; every lane of the group runs the
// chain's scalar state machine redundantly on identical values (sums over the dimension are butterfly reductions, which leave
// the same bits in every lane), so control flow is uniform within a group and the dozens of short run-time loops of
// a one-thread form with a run-time dimension -- each a chain of scratch / L2 round trips -- become single instructions.
// A matrix-vector product is D fused multiply-adds per lane, the vector going round by lane shuffles.  The same transitions
// as `advance_chain` makes up to the order of those sums (per-chain adaptation only; pooled windows are refused below).
constexpr int kGroup = DYN_NUTS_MAX_DIM;
static_assert(kGroup == 32, "one half wave per chain");

__device__ __forceinline__ double lane_value(double v, int j) { return __shfl(v, j, kGroup); }   // the group's lane j
__device__ __forceinline__ double group_sum(double x) {
#pragma unroll
    for (int o = kGroup / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kGroup);
    return x;
}
__device__ __forceinline__ bool group_any(bool b) {
    const unsigned long long m = __ballot(b);
    return ((m >> (threadIdx.x & 32)) & 0xffffffffull) != 0;
}
// y_l = sum_j M[l][j] v_j (lanes beyond D: 0)
__device__ __forceinline__ double matvec_lanes(const double *M, double v, int D, int l) {
    const double *row = M + (l < D ? l : 0) * D;
    double a = 0.0;
    for (int j = 0; j < D; ++j) a += row[j] * lane_value(v, j);
    return l < D ? a : 0.0;
}
__device__ __forceinline__ bool is_turning_lanes(const double *imm, double rl, double rr, double rsum, int D, int l) {
    const double rs = rsum - 0.5 * (rl + rr);
    const double *row = imm + (l < D ? l : 0) * D;
    double vl = 0.0, vr = 0.0;
    for (int j = 0; j < D; ++j) {
        const double m = row[j];
        vl += m * lane_value(rl, j);
        vr += m * lane_value(rr, j);
    }
    if (l >= D) { vl = 0.0; vr = 0.0; }
    return group_sum(vl * rs) <= 0.0 || group_sum(vr * rs) <= 0.0;
}

// chol(inv(imm)) by the lanes of a chain: the operations of mass_sqrt_into (nuts_device.hpp), element by element in the same
// order -- Gauss-Jordan with lane l on column l, Cholesky column by column with lane i on row i.  A (the matrix, filled by the
// caller) and I: LDS, row stride kLdsStride (both access patterns conflict-free); the factor goes to `out` in global memory.
constexpr int kLdsStride = kGroup + 1;
__device__ __forceinline__ void wave_sync() {   // LDS operations of a wave execute in order: only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void mass_sqrt_lanes(double *out, double *A, double *I, int D, int l) {
    constexpr int S = kLdsStride;
    const bool own = l < D;
    if (own)
        for (int r = 0; r < D; ++r) I[r * S + l] = r == l ? 1.0 : 0.0;
    wave_sync();
    for (int c = 0; c < D; ++c) {
        const double p = 1.0 / A[c * S + c];
        wave_sync();
        double ac = 0.0, ic = 0.0;
        if (own) {
            ac = A[c * S + l] * p;
            ic = I[c * S + l] * p;
            A[c * S + l] = ac;
            I[c * S + l] = ic;
        }
        const double below = own ? A[l * S + c] : 0.0;   // lane r: a[r][c], read before any row changes (row c's is not used)
        wave_sync();
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const double f = lane_value(below, r);
            if (own) {
                A[r * S + l] -= f * ac;
                I[r * S + l] -= f * ic;
            }
        }
        wave_sync();
    }
    for (int j = 0; j < D; ++j) {     // (in place: the lower triangle of I becomes the factor, the upper one is only read)
        double s = 0.0;
        const bool mine = own && l >= j;
        if (mine) {
            s = 0.5 * (I[l * S + j] + I[j * S + l]);
            for (int k = 0; k < j; ++k) s -= I[l * S + k] * I[j * S + k];
        }
        const double djj = sqrt(fmax(lane_value(s, j), 1e-300));
        if (mine) I[l * S + j] = l == j ? djj : s / djj;
        wave_sync();
    }
    if (own)
        for (int r = 0; r < D; ++r) out[r * D + l] = l <= r ? I[r * S + l] : 0.0;
}

template <bool MAPPED>
__global__ void __launch_bounds__(64) nuts_advance_lanes(const dyn_nuts_state st, const dynlat::MapArgs map) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int c = t / kGroup, l = t % kGroup;
    const int D = st.dim, Dm = st.max_depth;
    const int total = st.num_warmup + st.num_samples;
    __shared__ double window_ws[2][2][kGroup * kLdsStride];   // (a window's end: the matrix and its inverse, mass_sqrt_lanes)
    if (c >= st.n_chains || st.it[c] >= total) return;   // (a whole group leaves together; finished chains idle)
    const bool own = l < D;
    const int64_t cD = (int64_t)c * D;
    auto ld = [&](const double *p) { return own ? p[cD + l] : 0.0; };
    auto sv = [&](double *p, double v) { if (own) p[cD + l] = v; };

    double L_u = st.u[c], L_eps = st.eps[c], L_eps_avg = st.eps_avg[c], L_da_mu = st.da_mu[c], L_da_xbar = st.da_xbar[c],
           L_da_gbar = st.da_gbar[c], L_da_t = st.da_t[c], L_wf_n = st.wf_n[c], L_e0 = st.e0[c], L_up = st.up[c],
           L_weight = st.weight[c], L_sum_acc = st.sum_acc[c], L_sgn = st.sgn[c], L_s_up = st.s_up[c], L_s_acc = st.s_acc[c];
    int L_it = st.it[c], L_wi = st.wi[c], L_n_prop = st.n_prop[c], L_s_n = st.s_n[c];
    const bool right = st.right[c] != 0;
    int depth = st.depth[c], leaf = st.leaf[c];
    bool s_turn = st.s_turn[c] != 0;
    double s_weight = st.s_weight[c];
    uint64_t ctr = (uint64_t)st.rng_ctr[c];
    auto uniform = [&]() {
        Philox r{(uint32_t)st.seed, (uint32_t)(st.seed >> 32), ctr, (uint32_t)c};
        ++ctr;
        return r.uniform();
    };
    double *const imm = st.imm + cD * D, *const mms = st.mm_sqrt + cD * D, *const m2 = st.wf_m2 + cD * D;
    // this lane's element of the chain's vectors (z, g: written at a transition's end, never read before)
    double gl = ld(st.gl), gp = ld(st.gp), gr = ld(st.gr), r_sum = ld(st.r_sum), rl = ld(st.rl), rr = ld(st.rr),
           s_gp = ld(st.s_gp), s_rsum = ld(st.s_rsum), s_zp = ld(st.s_zp), zl = ld(st.zl), zp = ld(st.zp), zr = ld(st.zr);
    double z = 0.0, g = 0.0;
    const double eps_signed = L_eps * L_sgn;

    // ---- finish the leapfrog started by the previous launch (advance_chain has the commentary of every phase)
    double un, gn;
    if (st.pot_lp != nullptr) {
        const int64_t first = (int64_t)c * (st.pot_dll_stride > 0 ? st.pot_dll_stride : D);
        un = -(st.pot_lp[c] + st.pot_ll[(int64_t)c * st.pot_ll_stride] + st.pot_offset);
        gn = own ? -(st.pot_dlp[cD + l] + st.pot_dll[first + l]) : 0.0;
    } else {
        un = st.u_new[c];
        gn = ld(st.g_new);
    }
    const double zn = ld(st.z_eval);
    const bool bad = group_any(!isfinite(un) || !isfinite(gn));
    const double rn = ld(st.r_half) - 0.5 * eps_signed * (bad ? 0.0 : gn);
    double tmp = matvec_lanes(imm, rn, D, l);
    double de = (bad ? INFINITY : un) + 0.5 * group_sum(rn * tmp) - L_e0;
    if (isnan(de)) de = INFINITY;
    const double lw = -de;
    const bool div = de > st.max_delta_energy;
    const double acc = exp(fmin(-de, 0.0));

    // ---- grow the subtree by this leaf
    const double new_w = logaddexp(s_weight, lw);
    if (uniform() < exp(lw - new_w)) {
        s_zp = zn;
        s_gp = bad ? 0.0 : gn;
        L_s_up = bad ? INFINITY : un;
    }
    s_weight = new_w;
    s_rsum += rn;
    bool s_div = st.s_div[c] != 0 || div;
    const double s_acc = L_s_acc + acc;
    const int s_n = L_s_n + 1;
    double zc = zn, rc = rn, gc = bad ? 0.0 : gn;

    // ---- checkpointed U-turn test
    const int idx_max = __popc((unsigned)(leaf >> 1));
    int trailing = 0;
    while ((leaf >> trailing) & 1) ++trailing;
    const int idx_min = idx_max - trailing + 1;
    double *const r_ck = st.r_ck + cD * Dm, *const rs_ck = st.rs_ck + cD * Dm;
    if ((leaf & 1) == 0) {
        if (own) { r_ck[idx_max * D + l] = rn; rs_ck[idx_max * D + l] = s_rsum; }
    } else {
        for (int k = idx_max; k >= idx_min && !s_turn; --k) {
            const double rk = own ? r_ck[k * D + l] : 0.0, rsk = own ? rs_ck[k * D + l] : 0.0;
            s_turn = is_turning_lanes(imm, rk, rn, s_rsum - rsk + rk, D, l);
        }
    }
    ++leaf;

    // ---- subtree complete -> merge into the trajectory
    const bool sub_done = s_turn || s_div || leaf >= (1 << depth);
    bool stop = false;
    if (sub_done) {
        const bool ok = !s_turn && !s_div;
        if (ok && uniform() < exp(fmin(s_weight - L_weight, 0.0))) { zp = s_zp; gp = s_gp; L_up = L_s_up; }
        if (right) { zr = zc; rr = rc; gr = gc; } else { zl = zc; rl = rc; gl = gc; }
        L_weight = logaddexp(L_weight, s_weight);
        r_sum += s_rsum;
        L_sum_acc += s_acc;
        L_n_prop += s_n;
        ++depth;
        stop = s_turn || s_div || is_turning_lanes(imm, rl, rr, r_sum, D, l) || depth >= Dm;
    }

    int it = L_it;
    double eps = L_eps;
    if (stop) {
        // ---- transition complete: adapt, record, next transition
        const bool warm = it < st.num_warmup;
        const int n_prop = L_n_prop;
        const double a_prob = L_sum_acc / (double)(n_prop > 0 ? n_prop : 1);
        z = zp; g = gp;
        L_u = L_up;
        if (warm) {
            const double t1 = L_da_t + 1.0, w = 1.0 / (t1 + 10.0);
            const double gbar = (1.0 - w) * L_da_gbar + w * (st.target_accept - a_prob);
            const double x = L_da_mu - sqrt(t1) / 0.05 * gbar;
            const double wx = pow(t1, -0.75);
            const double xbar = (1.0 - wx) * L_da_xbar + wx * x;
            L_da_t = t1; L_da_gbar = gbar; L_da_xbar = xbar;
            eps = exp(x);
            L_eps_avg = exp(xbar);
            const int wi = L_wi;
            if (wi < st.n_windows && it >= st.w_start[wi] && it < st.w_end[wi]) {
                // Welford: this lane the mean's element l and row l of the co-moment matrix
                const double n1 = L_wf_n + 1.0;
                double mean = ld(st.wf_mean);
                const double d0 = z - mean;
                mean += d0 / n1;
                const double zm = z - mean;
                double *const row = m2 + (own ? l : 0) * D;
                const bool last = it + 1 == st.w_end[wi];
                const double nn = fmax(n1, 2.0);
                double *const A = window_ws[(threadIdx.x >> 5) & 1][0], *const I = window_ws[(threadIdx.x >> 5) & 1][1];
                for (int j = 0; j < D; ++j) {
                    const double zj = lane_value(zm, j);
                    if (own) {
                        const double m = row[j] + d0 * zj;
                        row[j] = last ? 0.0 : m;
                        if (last) {   // the window's matrix: regularised as numpyro does, row l
                            const double v = (nn / (nn + 5.0)) * m / (nn - 1.0) + (l == j ? 1e-3 * (5.0 / (nn + 5.0)) : 0.0);
                            imm[l * D + j] = v;
                            A[l * kLdsStride + j] = v;
                        }
                    }
                }
                L_wf_n = n1;
                if (last) {
                    mass_sqrt_lanes(mms, A, I, D, l);
                    __threadfence();    // (the factor's columns were written by other lanes than the ones that read its rows below)
                    eps = L_eps_avg;
                    L_da_mu = log(10.0 * eps);
                    L_da_t = 0.0; L_da_gbar = 0.0; L_da_xbar = 0.0;
                    L_wf_n = 0.0;
                    mean = 0.0;
                    L_wi = wi + 1;
                }
                sv(st.wf_mean, mean);
            }
            if (it + 1 == st.num_warmup) eps = L_eps_avg;
        } else {
            const int64_t j = (int64_t)c * st.num_samples + (it - st.num_warmup);
            if (own) st.out_z[j * D + l] = z;
            if (l == 0) {
                st.out_acc[j] = a_prob;
                st.out_n[j] = n_prop;
                st.out_div[j] = s_div ? 1 : 0;
            }
        }
        L_eps = eps;
        L_it = ++it;
        // fresh momentum r0 = chol(M) * normal (normal i from counters ctr + 2 i, ctr + 2 i + 1, as the serial draw order has them)
        Philox r{(uint32_t)st.seed, (uint32_t)(st.seed >> 32), ctr + 2ull * (uint64_t)(own ? l : 0), (uint32_t)c};
        const double nrm = own ? r.normal() : 0.0;
        ctr += 2ull * (uint64_t)D;
        const double r0 = matvec_lanes(mms, nrm, D, l);
        tmp = matvec_lanes(imm, r0, D, l);
        L_e0 = L_u + 0.5 * group_sum(r0 * tmp);
        zl = zr = zp = z;
        rl = rr = r_sum = r0;
        gl = gr = gp = g;
        L_up = L_u;
        L_weight = 0.0; L_sum_acc = 0.0; L_n_prop = 0;
        depth = 0;
    }

    bool go_right = right;
    if (sub_done) {
        // ---- next subtree
        go_right = uniform() < 0.5;
        L_sgn = go_right ? 1.0 : -1.0;
        zc = go_right ? zr : zl; rc = go_right ? rr : rl; gc = go_right ? gr : gl;
        s_zp = zp; s_gp = gp; s_rsum = 0.0;
        L_s_up = L_up;
        s_weight = -INFINITY;
        L_s_acc = 0.0; L_s_n = 0;
        s_turn = false; s_div = false;
        leaf = 0;
        if (own)
            for (int k = 0; k < Dm; ++k) { r_ck[k * D + l] = 0.0; rs_ck[k * D + l] = 0.0; }
    } else {
        L_s_acc = s_acc; L_s_n = s_n;
    }

    // ---- first half of the next leapfrog, and the position the potential is needed at
    const double es = eps * (go_right ? 1.0 : -1.0);
    const double rh = rc - 0.5 * es * gc;
    tmp = matvec_lanes(imm, rh, D, l);
    const double ze = (it >= total) ? z : zc + es * tmp;
    sv(st.r_half, rh);
    sv(st.z_eval, ze);
    sv(st.gc, gc); sv(st.gl, gl); sv(st.gp, gp); sv(st.gr, gr);
    sv(st.r_sum, r_sum); sv(st.rc, rc); sv(st.rl, rl); sv(st.rr, rr);
    sv(st.s_gp, s_gp); sv(st.s_rsum, s_rsum); sv(st.s_zp, s_zp);
    sv(st.zc, zc); sv(st.zl, zl); sv(st.zp, zp); sv(st.zr, zr);
    if (stop) { sv(st.z, z); sv(st.g, g); }
    if (l == 0) {
        st.rng_ctr[c] = (int64_t)ctr;
        st.u[c] = L_u; st.eps[c] = L_eps; st.eps_avg[c] = L_eps_avg; st.da_mu[c] = L_da_mu; st.da_xbar[c] = L_da_xbar;
        st.da_gbar[c] = L_da_gbar; st.da_t[c] = L_da_t; st.wf_n[c] = L_wf_n; st.e0[c] = L_e0; st.up[c] = L_up;
        st.weight[c] = L_weight; st.sum_acc[c] = L_sum_acc; st.sgn[c] = L_sgn; st.s_up[c] = L_s_up; st.s_weight[c] = s_weight;
        st.s_acc[c] = L_s_acc; st.it[c] = L_it; st.wi[c] = L_wi; st.n_prop[c] = L_n_prop; st.depth[c] = depth;
        st.right[c] = go_right ? 1 : 0; st.leaf[c] = leaf; st.s_turn[c] = s_turn ? 1 : 0; st.s_div[c] = s_div ? 1 : 0;
        st.s_n[c] = L_s_n;
    }
    // the map of the position handed out: the group's first DYN_MAX_SITES lanes, one site each (dyn_nuts_advance_mapped)
    if constexpr (MAPPED) {
        if (l < DYN_MAX_SITES) {
            const int leader = (int)(threadIdx.x & 63) - l;
            if (map.f64)
                dynlat::map_chain_lanes<double, DYN_MAX_SITES>(map.tab, c, l, leader, ze, map.x, map.lp, map.dlp_dz, map.P, map.coef,
                                                               map.expo, (double *)map.params, (double *)map.seeds, map.split);
            else
                dynlat::map_chain_lanes<float, DYN_MAX_SITES>(map.tab, c, l, leader, ze, map.x, map.lp, map.dlp_dz, map.P, map.coef,
                                                              map.expo, (float *)map.params, (float *)map.seeds, map.split);
        }
    }
}

} // namespace dynnuts

extern "C" int32_t dyn_nuts_state_size(void) { return (int32_t)sizeof(dyn_nuts_state); }

extern "C" void dyn_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    const uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    uint32_t o[4];
    dynnuts::philox4x32_10(c, k, o);
    for (int i = 0; i < 4; ++i) out[i] = o[i];
}

static int advance(const dyn_nuts_state *st, const dynlat::MapArgs &map, void *stream) {
    if (!st) return DYN_ERR_NULL;
    if (st->n_chains < 0 || st->dim < 1 || st->dim > DYN_NUTS_MAX_DIM || st->max_depth < 1 ||
        st->max_depth > DYN_NUTS_MAX_DEPTH || st->n_windows < 0 || st->n_windows > DYN_NUTS_MAX_WINDOWS)
        return DYN_ERR_SIZE;
    if (st->n_chains == 0) return 0;
    if (st->pooled && (!st->pool || !st->pool_ro || !st->pend)) return DYN_ERR_NULL;
    using Kern = void (*)(const dyn_nuts_state, const dynlat::MapArgs);
    static_assert(dynnuts::kRegDim == 8, "one compile-time instance per dimension up to kRegDim");
    // one thread per chain; with a map behind the state machine, the next power of two of lanes per chain (they share the map)
    static const Kern plain[dynnuts::kRegDim] = {
        dynnuts::nuts_advance<1, 1>, dynnuts::nuts_advance<2, 1>, dynnuts::nuts_advance<3, 1>, dynnuts::nuts_advance<4, 1>,
        dynnuts::nuts_advance<5, 1>, dynnuts::nuts_advance<6, 1>, dynnuts::nuts_advance<7, 1>, dynnuts::nuts_advance<8, 1>};
    static const Kern mapped[dynnuts::kRegDim] = {
        dynnuts::nuts_advance<1, 1>, dynnuts::nuts_advance<2, 2>, dynnuts::nuts_advance<3, 4>, dynnuts::nuts_advance<4, 4>,
        dynnuts::nuts_advance<5, 8>, dynnuts::nuts_advance<6, 8>, dynnuts::nuts_advance<7, 8>, dynnuts::nuts_advance<8, 8>};
    static const int lanes_of[dynnuts::kRegDim] = {1, 2, 4, 4, 8, 8, 8, 8};
    if (st->dim > dynnuts::kRegDim) {
        // a half wave per chain: per-chain adaptation; (u_new, g_new), or a folded potential's parts and map
        if (st->pooled) return DYN_ERR_UNSUPPORTED;
        const unsigned blocks = (unsigned)(((int64_t)st->n_chains * dynnuts::kGroup + 63) / 64);
        if (map.enabled)
            hipLaunchKernelGGL(dynnuts::nuts_advance_lanes<true>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
        else
            hipLaunchKernelGGL(dynnuts::nuts_advance_lanes<false>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
        return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
    }
    const int ns = map.enabled ? lanes_of[st->dim - 1] : 1;
    const unsigned blocks = (unsigned)(((int64_t)st->n_chains * ns + 63) / 64);
    hipLaunchKernelGGL((map.enabled ? mapped : plain)[st->dim - 1], dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
    if (hipGetLastError() != hipSuccess) return DYN_ERR_LAUNCH;
    if (st->pooled) {
        // readers of the next launch see the pool as it stands now, never a half-updated one
        const size_t bytes = sizeof(int64_t) * (size_t)(st->n_windows + 1) * (size_t)(1 + st->dim + st->dim * st->dim);
        if (hipMemcpyAsync(st->pool_ro, st->pool, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return DYN_ERR_LAUNCH;
    }
    return 0;
}

extern "C" int dyn_nuts_advance(const dyn_nuts_state *st, void *stream) {
    dynlat::MapArgs map;
    map.enabled = 0;
    return advance(st, map, stream);
}

static int build_map(const dyn_nuts_state *st, const dyn_site_desc *sites, int32_t n_sites, int32_t P, const double *coef,
                     const double *expo, int32_t dtype, int32_t split_directions, double *x, double *lp, double *dlp_dz,
                     void *params, void *seeds, dynlat::MapArgs &map) {
    if (!st || !sites || !coef || !expo || !x || !lp || !dlp_dz || !params || !seeds) return DYN_ERR_NULL;
    if (n_sites != st->dim || n_sites < 1 || n_sites > DYN_MAX_SITES || P < 1) return DYN_ERR_SIZE;
    if (dtype != DYN_F32 && dtype != DYN_F64) return DYN_ERR_OPTS;
    map.tab.n = n_sites;
    for (int i = 0; i < n_sites; ++i) {
        const dyn_site_desc &d = sites[i];
        if (d.dist < DYN_DIST_NORMAL || d.dist > DYN_DIST_TRUNCNORMAL || d.aff_scale == 0.0 || !(d.lo < d.hi)) return DYN_ERR_OPTS;
        map.tab.s[i] = d;
    }
    map.enabled = 1;
    map.P = P;
    map.f64 = dtype == DYN_F64;
    map.split = dynlat_rows_per_chain(split_directions, n_sites);
    if (map.split < 0) return DYN_ERR_SIZE;
    map.coef = coef;
    map.expo = expo;
    map.x = x;
    map.lp = lp;
    map.dlp_dz = dlp_dz;
    map.params = params;
    map.seeds = seeds;
    return 0;
}

extern "C" int dyn_nuts_advance_mapped(const dyn_nuts_state *st, const dyn_site_desc *sites, int32_t n_sites, int32_t P,
                                       const double *coef, const double *expo, int32_t dtype, int32_t split_directions, double *x,
                                       double *lp, double *dlp_dz, void *params, void *seeds, void *stream) {
    dynlat::MapArgs map;
    const int rc = build_map(st, sites, n_sites, P, coef, expo, dtype, split_directions, x, lp, dlp_dz, params, seeds, map);
    if (rc) return rc;
    return advance(st, map, stream);
}

// ---- the fused launch's blob (dyn_solver_opts::nuts_tail)
extern "C" int32_t dyn_nuts_tail_size(void) { return (int32_t)sizeof(dynnuts::Tail); }

extern "C" int dyn_nuts_tail_pack(const dyn_nuts_state *st, const dyn_site_desc *sites, int32_t n_sites, int32_t P,
                                  const double *coef, const double *expo, int32_t dtype, int32_t split_directions, double *x,
                                  double *lp, double *dlp_dz, void *params, void *seeds, void *blob) {
    if (!blob) return DYN_ERR_NULL;
    dynnuts::Tail t;
    memset(&t, 0, sizeof(t));
    const int rc = build_map(st, sites, n_sites, P, coef, expo, dtype, split_directions, x, lp, dlp_dz, params, seeds, t.map);
    if (rc) return rc;
    if (st->n_chains < 0 || st->max_depth < 1 || st->max_depth > DYN_NUTS_MAX_DEPTH || st->n_windows < 0 ||
        st->n_windows > DYN_NUTS_MAX_WINDOWS)
        return DYN_ERR_SIZE;
    if (st->dim > dynnuts::kFusedMaxDim) return DYN_ERR_UNSUPPORTED;
    if (!st->pot_lp || !st->pot_dlp) return DYN_ERR_NULL; // the folded potential's prior side
    if (st->pooled && (!st->pool || !st->pool_ro || !st->pend)) return DYN_ERR_NULL;
    t.magic = dynnuts::kTailMagic;
    t.st = *st;
    t.st.pot_ll = nullptr; // (the fused launch reads its own outputs)
    t.st.pot_dll = nullptr;
    t.rows_per_chain = t.map.split ? t.map.split : 1;
    memcpy(blob, &t, sizeof(t));
    return 0;
}
