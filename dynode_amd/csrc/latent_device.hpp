// latent_device.hpp -- device code shared by latent_kernel.hip (the prior side of the potential as its own launch) and
// nuts_kernel.hip (the same work done for the NEXT position at the end of a sampler iteration, dyn_nuts_advance_mapped).
#pragma once
#include "../../include/dynode_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

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
    const int n = tab.n < NMAX ? tab.n : NMAX;
    double total = 0.0, x[NMAX], rel[NMAX];   // rel_i = (dx_i/dz_i) / x_i (0 where x_i == 0: see the seeds below)
    for (int i = 0; i < n; ++i) {
        dyn_site_desc site;
        __builtin_memcpy(&site, &tab.s[i], sizeof(site));
        const SiteValue v = eval_site(site, zrow[i]);
        total += v.lp;
        x[i] = v.x;
        rel[i] = v.x != 0.0 ? v.dx / v.x : 0.0;
        x_out[c * n + i] = v.x;
        dlp_dz[c * n + i] = v.dlp;
    }
    lp_out[c] = total;
    for (int j = 0; j < P; ++j) {
        double p = coef[j];
        for (int i = 0; i < n; ++i) {
            const double e = expo[j * n + i];
            if (e == 1.0) p *= x[i];
            else if (e == -1.0) p /= x[i];
            else if (e != 0.0) p *= pow(x[i], e);
        }
        // seed d p_j / d z_i = expo_ji p_j (dx_i/dz_i) / x_i.  A parameter that does not depend on site i (expo == 0) gets an exact
        // 0 whatever x_i is -- 0 * inf would be NaN when a site value underflows to 0 (a bounded site at its lower end, an
        // identity site at 0) and would reach every gradient through the solve; at x_i == 0 itself the derivative of a power
        // is 0 (e > 1), the coefficient (e == 1) or unbounded: the first two are formed from the product without x_i
        auto seed = [&](int i) -> double {
            const double e = expo[j * n + i];
            if (e == 0.0) return 0.0;
            if (x[i] != 0.0) return e * p * rel[i];
            if (e < 1.0) return e * p * rel[i];      // unbounded derivative at the boundary: NaN / inf is the honest answer
            double q = coef[j] * e;
            for (int k = 0; k < n; ++k) {
                const double ek = k == i ? e - 1.0 : expo[j * n + k];
                if (ek != 0.0) q *= pow(x[k], ek);
            }
            dyn_site_desc site;
            __builtin_memcpy(&site, &tab.s[i], sizeof(site));
            return q * eval_site(site, zrow[i]).dx;
        };
        if (split) { // one direction per trajectory: chain c becomes rows c R .. c R + n - 1 of an R C batch with one seed row each
            const int64_t R = split;          // (R >= n rows per chain: rows beyond the sites are padding -- the chain's parameters, zero seeds)
            for (int i = 0; i < R; ++i) {     // (neighbours: the copies of a chain take the same steps, so they share a wave for free)
                params[(c * R + i) * P + j] = (T)p;
                seeds[(c * R + i) * P + j] = i < n ? (T)seed(i) : (T)0;
            }
        } else {
            params[c * P + j] = (T)p;
            for (int i = 0; i < n; ++i) seeds[(c * n + i) * P + j] = (T)seed(i);
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
