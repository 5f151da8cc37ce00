// latent_device.hpp -- device code shared by latent_kernel.hip (the prior side of the potential as its own launch) and
// nuts_kernel.hip (the same work done for the NEXT position at the end of a sampler iteration, dyn_nuts_advance_mapped).
#pragma once
#include "../../include/dynode_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

// Floating-point contraction: only inside an expression as written (the language's rule), not across statements.  hipcc's
// default (`fast`) lets the backend fuse a multiply with a later add wherever instruction selection finds one, which
// depends on the code AROUND the expression -- the same sampler source inlined into the gradient-solve's tail, compiled as the
// stand-alone kernel, or with the map of the next position shared between lanes then differs in a last bit here and there
// (found as 1-ulp step sizes between the one- and the two-launch iteration), and a sampler run is a chaotic map of its bits.
// From here to the end of nuts_device.hpp (which restores the default for the solve kernels that include it).
#pragma clang fp contract(on)

namespace dynlat {

struct SiteTable {
    dyn_site_desc s[DYN_MAX_SITES];
    int32_t n;
};

__device__ inline double log_sigmoid(double z) { return z < 0 ? z - log1p(exp(z)) : -log1p(exp(-z)); }

// one scalar site of one chain: bijection onto the support, log density, and their derivatives in z
struct SiteValue {
    double x, dx;      // constrained value, d x / d z
    double lp, dlp;    // log prior(x) + log |dx/dz|, and its derivative in z
};

__device__ inline SiteValue eval_site(const dyn_site_desc &d, double zi) {
    // ---- bijection onto the support [lo, hi] (numpyro biject_to: identity / sigmoid / exp)
    double x, dx, ladj, dladj;
    const bool lo_inf = isinf(d.lo), hi_inf = isinf(d.hi);
    if (lo_inf && hi_inf) {
        x = zi; dx = 1.0; ladj = 0.0; dladj = 0.0;
    } else if (!lo_inf && !hi_inf) {
        const double s = 1.0 / (1.0 + exp(-zi)), w = d.hi - d.lo;
        x = d.lo + w * s; dx = w * s * (1.0 - s);
        ladj = log(w) + log_sigmoid(zi) + log_sigmoid(-zi); dladj = 1.0 - 2.0 * s;
    } else if (hi_inf) {
        const double e = exp(zi);
        x = d.lo + e; dx = e; ladj = zi; dladj = 1.0;
    } else {
        const double e = exp(zi);
        x = d.hi - e; dx = -e; ladj = zi; dladj = 1.0;
    }
    // ---- log density of y = aff_loc + aff_scale * base at x, and its derivative in x
    const double xb = (x - d.aff_loc) / d.aff_scale;
    double lp, dlp;
    switch (d.dist) {
    case DYN_DIST_NORMAL: {
        const double u = (xb - d.p[0]) / d.p[1];
        lp = -0.5 * u * u - log(d.p[1]) - 0.91893853320467274178; dlp = -u / d.p[1];
    } break;
    case DYN_DIST_UNIFORM: {
        const bool in = xb >= d.p[0] && xb <= d.p[1];
        lp = in ? -log(d.p[1] - d.p[0]) : -INFINITY; dlp = 0.0;
    } break;
    case DYN_DIST_BETA: {   // p = {a, b, log B(a, b)}
        lp = (d.p[0] - 1.0) * log(xb) + (d.p[1] - 1.0) * log1p(-xb) - d.p[2];
        dlp = (d.p[0] - 1.0) / xb - (d.p[1] - 1.0) / (1.0 - xb);
    } break;
    default: {              // DYN_DIST_TRUNCNORMAL: p = {loc, scale, log Z, unused}; support = [lo, hi] of the base
        const double u = (xb - d.p[0]) / d.p[1];
        const bool in = xb >= d.base_lo && xb <= d.base_hi;
        lp = in ? -0.5 * u * u - 0.91893853320467274178 - log(d.p[1]) - d.p[2] : -INFINITY; dlp = -u / d.p[1];
    } break;
    }
    lp -= log(fabs(d.aff_scale));
    dlp /= d.aff_scale;
    return {x, dx, lp + ladj, dlp * dx + dladj};
}

// One chain of `latent_param_map`: sites, log prior, parameter row(s) and tangent seeds of the monomial parameter map
// p_j = coef_j prod_i x_i^expo[j][i] at the unconstrained position `zrow` (comments at the kernel in latent_kernel.hip).
// NMAX: the most sites the caller can have (the arrays below are sized by it: a compile-time-dimension sampler instance passes
// its dimension, so that an eight-dimension kernel does not carry sixteen-element arrays)
template <typename T, int NMAX = DYN_MAX_SITES, typename TAB>   // TAB: SiteTable in whatever address space the caller holds it (kernel argument, kernarg segment)
__device__ inline void map_chain(const TAB &tab, int64_t C, int64_t c, const double *zrow, double *__restrict__ x_out,
                                 double *__restrict__ lp_out, double *__restrict__ dlp_dz, int P, const double *__restrict__ coef,
                                 const double *__restrict__ expo, T *__restrict__ params, T *__restrict__ seeds, int split) {
    // Every loop over the sites runs to the compile-time NMAX under an `i < n` guard and is unrolled: x, rel and a parameter's
    // exponents then live in registers (run-time trip counts put them into scratch memory behind dynamic indices, and made the
    // exponent loads a chain of dependent L2 round trips: 34 us of the six-site sampler kernel's 79)
    const int n = tab.n < NMAX ? tab.n : NMAX;
    double total = 0.0, x[NMAX], rel[NMAX];   // rel_i = (dx_i/dz_i) / x_i (0 where x_i == 0: see the seeds below)
    bool at_zero = false;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        x[i] = 1.0;
        rel[i] = 0.0;
        if (i < n) {
            dyn_site_desc site;
            __builtin_memcpy(&site, &tab.s[i], sizeof(site));
            const SiteValue v = eval_site(site, zrow[i]);
            total += v.lp;
            x[i] = v.x;
            rel[i] = v.x != 0.0 ? v.dx / v.x : 0.0;
            at_zero = at_zero || v.x == 0.0;
            x_out[c * n + i] = v.x;
            dlp_dz[c * n + i] = v.dlp;
        }
    }
    lp_out[c] = total;
    for (int j = 0; j < P; ++j) {
        double e[NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) e[i] = i < n ? expo[j * n + i] : 0.0;     // (independent loads: issued together)
        double p = coef[j];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            if (e[i] == 1.0) p *= x[i];
            else if (e[i] == -1.0) p /= x[i];
            else if (e[i] != 0.0) p *= pow(x[i], e[i]);
        }
        // seed d p_j / d z_i = expo_ji p_j (dx_i/dz_i) / x_i.  A parameter that does not depend on site i (expo == 0) gets an exact
        // 0 whatever x_i is -- 0 * inf would be NaN when a site value underflows to 0 (a bounded site at its lower end, an
        // identity site at 0) and would reach every gradient through the solve; at x_i == 0 itself the derivative of a power
        // is 0 (e > 1), the coefficient (e == 1) or unbounded: the first two are formed from the product without x_i
        double sd[NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) sd[i] = e[i] == 0.0 ? 0.0 : e[i] * p * rel[i];   // (x_i == 0 with e_i < 1: NaN / inf, the honest answer)
        if (__builtin_expect(at_zero, 0)) {      // a site value exactly at 0: the rare, slow form
#pragma unroll
            for (int i = 0; i < NMAX; ++i) {
                if (!(i < n) || !(e[i] >= 1.0) || x[i] != 0.0) continue;
                double q = coef[j] * e[i];
#pragma unroll
                for (int k = 0; k < NMAX; ++k) {
                    const double ek = k == i ? e[i] - 1.0 : e[k];
                    if (k < n && ek != 0.0) q *= pow(x[k], ek);
                }
                dyn_site_desc site;
                __builtin_memcpy(&site, &tab.s[i], sizeof(site));
                sd[i] = q * eval_site(site, zrow[i]).dx;
            }
        }
        if (split) { // one direction per trajectory: chain c becomes rows c R .. c R + n - 1 of an R C batch with one seed row each
            const int64_t R = split;          // (R >= n rows per chain: rows beyond the sites are padding -- the chain's parameters, zero seeds)
            for (int i = 0; i < R; ++i) params[(c * R + i) * P + j] = (T)p;   // (neighbours: the copies of a chain take the same steps, so they share a wave for free)
#pragma unroll
            for (int i = 0; i < NMAX; ++i)
                if (i < n) seeds[(c * R + i) * P + j] = (T)sd[i];
            for (int i = n; i < R; ++i) seeds[(c * R + i) * P + j] = (T)0;
        } else {
            params[c * P + j] = (T)p;
#pragma unroll
            for (int i = 0; i < NMAX; ++i)
                if (i < n) seeds[(c * n + i) * P + j] = (T)sd[i];
        }
    }
}

// The same map with the NS lanes of a chain sharing it: lane `sub` of the group (lanes leader .. leader + NS - 1 of one wave,
// NS a power of two >= the number of sites) evaluates site `sub` -- the float64 logarithms, exponentials and error functions of
// a bijection and its prior are the bulk of the serial map: 35 of the six-site sampler kernel's 68 us --, the values go
// round by lane shuffles, and the parameter columns j = sub, sub + NS, ... are formed and stored by lane `sub`.  Every value is
// produced by the operations of map_chain in the same order: the two are bit-identical (tests/test_gpu_infer.py).  `z_sub`:
// this lane's unconstrained coordinate (lanes sub >= n: ignored).  Must be called by all NS lanes of the group together.
template <typename T, int NS, typename TAB>
__device__ inline void map_chain_lanes(const TAB &tab, int64_t c, int sub, int leader, double z_sub, double *__restrict__ x_out,
                                       double *__restrict__ lp_out, double *__restrict__ dlp_dz, int P,
                                       const double *__restrict__ coef, const double *__restrict__ expo, T *__restrict__ params,
                                       T *__restrict__ seeds, int split) {
    static_assert(NS == 1 || NS == 2 || NS == 4 || NS == 8 || NS == 16, "lanes per chain");
    const int n = tab.n < NS ? tab.n : NS;
    SiteValue v{1.0, 0.0, 0.0, 0.0};
    if (sub < n) {
        dyn_site_desc site;
        __builtin_memcpy(&site, &tab.s[sub], sizeof(site));
        v = eval_site(site, z_sub);
        x_out[c * n + sub] = v.x;
        dlp_dz[c * n + sub] = v.dlp;
    }
    const double my_rel = v.x != 0.0 ? v.dx / v.x : 0.0;
    double total = 0.0, x[NS], rel[NS], zs[NS];
    bool at_zero = false;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        x[i] = __shfl(v.x, leader + i);
        rel[i] = __shfl(my_rel, leader + i);
        zs[i] = __shfl(z_sub, leader + i);
        const double lp_i = __shfl(v.lp, leader + i);
        if (i < n) {
            total += lp_i;                      // (in site order, as map_chain sums them)
            at_zero = at_zero || x[i] == 0.0;
        } else {
            x[i] = 1.0;
            rel[i] = 0.0;
        }
    }
    if (sub == 0) lp_out[c] = total;
    for (int j = sub; j < P; j += NS) {
        double e[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) e[i] = i < n ? expo[j * n + i] : 0.0;
        double p = coef[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (e[i] == 1.0) p *= x[i];
            else if (e[i] == -1.0) p /= x[i];
            else if (e[i] != 0.0) p *= pow(x[i], e[i]);
        }
        double sd[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) sd[i] = e[i] == 0.0 ? 0.0 : e[i] * p * rel[i];
        if (__builtin_expect(at_zero, 0)) {      // a site value exactly at 0: the rare, slow form (see map_chain)
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (!(i < n) || !(e[i] >= 1.0) || x[i] != 0.0) continue;
                double q = coef[j] * e[i];
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const double ek = k == i ? e[i] - 1.0 : e[k];
                    if (k < n && ek != 0.0) q *= pow(x[k], ek);
                }
                dyn_site_desc site;
                __builtin_memcpy(&site, &tab.s[i], sizeof(site));
                sd[i] = q * eval_site(site, zs[i]).dx;
            }
        }
        if (split) {
            const int64_t R = split;
            for (int i = 0; i < R; ++i) params[(c * R + i) * P + j] = (T)p;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (i < n) seeds[(c * R + i) * P + j] = (T)sd[i];
            for (int i = n; i < R; ++i) seeds[(c * R + i) * P + j] = (T)0;
        } else {
            params[c * P + j] = (T)p;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (i < n) seeds[(c * n + i) * P + j] = (T)sd[i];
        }
    }
}

// what dyn_nuts_advance_mapped hands to the sampler kernel (by value)
struct MapArgs {
    SiteTable tab;
    int32_t enabled, P, f64, split;      // f64: params / seeds are double (else float); split: rows per chain (0: all directions in one row)
    const double *coef, *expo;
    double *x, *lp, *dlp_dz;
    void *params, *seeds;
};

} // namespace dynlat
