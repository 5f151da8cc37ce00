// latent_kernel.hip -- the prior side of the NUTS potential as ONE launch.
//
// A numpyro model's latent sample sites (reference examples/sir_infer_parameters.py:47-58:
// r0 ~ 1.5 + Beta(0.5, 0.5), infectious_period ~ TruncatedNormal(8, 2, 2, 15)) are traced by XLA
// together with the solve.  Evaluated op by op on a GPU each site costs about 50 tiny launches per
// gradient (bijection, log-density, log-Jacobian, and their backward).  This kernel maps the
// unconstrained coordinates of every chain to the constrained values, and returns the summed
// log prior + log |dx/dz| with the analytic derivatives the backward pass needs: one thread per
// chain, a loop over the (at most DYN_MAX_SITES) scalar sites.
#include "../../include/dynode_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

namespace dynlat {

struct SiteTable {
    dyn_site_desc s[DYN_MAX_SITES];
    int32_t n;
};

__device__ inline double log_sigmoid(double z) { return z < 0 ? z - log1p(exp(z)) : -log1p(exp(-z)); }

__global__ void __launch_bounds__(64) latent_sites(const SiteTable tab, int64_t C, const double *__restrict__ z,
                                                   double *__restrict__ x_out, double *__restrict__ lp_out,
                                                   double *__restrict__ dx_dz, double *__restrict__ dlp_dz) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int n = tab.n;
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        const dyn_site_desc &d = tab.s[i];
        const double zi = z[c * n + i];
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
        total += lp + ladj;
        x_out[c * n + i] = x;
        dx_dz[c * n + i] = dx;
        dlp_dz[c * n + i] = dlp * dx + dladj;
    }
    lp_out[c] = total;
}

} // namespace dynlat

extern "C" int dyn_latent_sites(const dyn_site_desc *sites, int32_t n_sites, int64_t C, const double *z, double *x,
                                double *lp, double *dx_dz, double *dlp_dz, void *stream) {
    if (!sites || (C > 0 && (!z || !x || !lp || !dx_dz || !dlp_dz))) return DYN_ERR_NULL;
    if (n_sites < 1 || n_sites > DYN_MAX_SITES || C < 0) return DYN_ERR_SIZE;
    dynlat::SiteTable tab;
    tab.n = n_sites;
    for (int i = 0; i < n_sites; ++i) {
        const dyn_site_desc &d = sites[i];
        if (d.dist < DYN_DIST_NORMAL || d.dist > DYN_DIST_TRUNCNORMAL || d.aff_scale == 0.0 || !(d.lo < d.hi))
            return DYN_ERR_OPTS;
        tab.s[i] = d;
    }
    if (C == 0) return 0;
    hipLaunchKernelGGL(dynlat::latent_sites, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tab, C,
                       z, x, lp, dx_dz, dlp_dz);
    return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
}
