// latent_kernel.hip -- the prior side of the NUTS potential as ONE launch.
//
// A numpyro model's latent sample sites (reference examples/sir_infer_parameters.py:47-58:
// r0 ~ 1.5 + Beta(0.5, 0.5), infectious_period ~ TruncatedNormal(8, 2, 2, 15)) are traced by XLA
// together with the solve.  Evaluated op by op on a GPU each site costs about 50 tiny launches per
// gradient (bijection, log-density, log-Jacobian, and their backward).  This kernel maps the
// unconstrained coordinates of every chain to the constrained values, and returns the summed
// log prior + log |dx/dz| with the analytic derivatives the backward pass needs: one thread per
// chain, a loop over the (at most DYN_MAX_SITES) scalar sites.
#include "latent_device.hpp"

namespace dynlat {

__global__ void __launch_bounds__(64) latent_sites(const SiteTable tab, int64_t C, const double *__restrict__ z,
                                                   double *__restrict__ x_out, double *__restrict__ lp_out,
                                                   double *__restrict__ dx_dz, double *__restrict__ dlp_dz) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int n = tab.n;
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        const SiteValue v = eval_site(tab.s[i], z[c * n + i]);
        total += v.lp;
        x_out[c * n + i] = v.x;
        dx_dz[c * n + i] = v.dx;
        dlp_dz[c * n + i] = v.dlp;
    }
    lp_out[c] = total;
}

// The same with the model's parameter map behind it, for models whose ODE parameter row is a monomial in the site values:
// p_j = coef_j * prod_i x_i^expo[j][i] (the reference's get_odeparams family, SURVEY rows A6: beta = r0 / T_inf,
// gamma = 1 / T_inf, sigma = 1 / T_lat, omega = 1 / T_wane).  Writes the parameter rows and the tangent seeds
// d p_j / d z_i = expo[j][i] p_j / x_i * dx_i/dz_i in the solve's dtype: what `dyn_solve_batch_loglik` reads next.
template <typename T, int NS>   // NS lanes per chain: 1, 2, 4, 8 or DYN_MAX_SITES, the smallest power of two that holds the model's sites
__global__ void __launch_bounds__(64) latent_param_map(const SiteTable tab, int64_t C, const double *__restrict__ z,
                                                       double *__restrict__ x_out, double *__restrict__ lp_out,
                                                       double *__restrict__ dlp_dz, int P, const double *__restrict__ coef,
                                                       const double *__restrict__ expo, T *__restrict__ params,
                                                       T *__restrict__ seeds, int split) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t c = t / NS;
    const int sub = (int)(t % NS);
    if (c >= C) return;                 // (whole lane groups: NS divides the wavefront)
    const double z_sub = sub < tab.n ? z[c * tab.n + sub] : 0.0;
    map_chain_lanes<T, NS>(tab, c, sub, (int)(threadIdx.x & 63) - sub, z_sub, x_out, lp_out, dlp_dz, P, coef, expo, params, seeds, split);
}

// u = -(lp + ll + offset), g = -(dlp_dz + dll): the potential and its gradient from the prior side and the solve's
// log-likelihood (one thread per chain)
__global__ void __launch_bounds__(64) potential_combine(int64_t C, int n, const double *__restrict__ lp,
                                                        const double *__restrict__ dlp_dz, const double *__restrict__ ll,
                                                        const double *__restrict__ dll, double offset, int split,
                                                        double *__restrict__ u, double *__restrict__ g) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    // (split = rows per chain: the copies of a chain carry the same log-likelihood, the first is read; row c rows + i holds direction i)
    u[c] = -(lp[c] + ll[split ? c * split : c] + offset);
    for (int i = 0; i < n; ++i) g[c * n + i] = -(dlp_dz[c * n + i] + dll[c * (split ? split : n) + i]);
}

} // namespace dynlat

// dyn_latent_param_map's split_directions as rows per chain: 0 = all directions in one row, 1 = n_sites rows, r >= 2 = r rows
// (the chains padded to r >= n_sites trajectories); -1 = invalid
int dynlat_rows_per_chain(int32_t split_directions, int32_t n_sites) {
    if (split_directions == 0) return 0;
    if (split_directions == 1) return n_sites;
    return split_directions >= n_sites && split_directions <= 64 ? split_directions : -1;
}

static int fill_table(dynlat::SiteTable &tab, const dyn_site_desc *sites, int32_t n_sites) {
    if (n_sites < 1 || n_sites > DYN_MAX_SITES) return DYN_ERR_SIZE;
    tab.n = n_sites;
    for (int i = 0; i < n_sites; ++i) {
        const dyn_site_desc &d = sites[i];
        if (d.dist < DYN_DIST_NORMAL || d.dist > DYN_DIST_TRUNCNORMAL || d.aff_scale == 0.0 || !(d.lo < d.hi))
            return DYN_ERR_OPTS;
        tab.s[i] = d;
    }
    return 0;
}

extern "C" int dyn_latent_sites(const dyn_site_desc *sites, int32_t n_sites, int64_t C, const double *z, double *x,
                                double *lp, double *dx_dz, double *dlp_dz, void *stream) {
    if (!sites || (C > 0 && (!z || !x || !lp || !dx_dz || !dlp_dz))) return DYN_ERR_NULL;
    if (C < 0) return DYN_ERR_SIZE;
    dynlat::SiteTable tab;
    if (int rc = fill_table(tab, sites, n_sites)) return rc;
    if (C == 0) return 0;
    hipLaunchKernelGGL(dynlat::latent_sites, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, tab, C,
                       z, x, lp, dx_dz, dlp_dz);
    return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
}

extern "C" int dyn_latent_param_map(const dyn_site_desc *sites, int32_t n_sites, int64_t C, const double *z, double *x,
                                    double *lp, double *dlp_dz, int32_t P, const double *coef, const double *expo,
                                    int32_t dtype, int32_t split_directions, void *params, void *seeds, void *stream) {
    if (!sites || !coef || !expo || (C > 0 && (!z || !x || !lp || !dlp_dz || !params || !seeds))) return DYN_ERR_NULL;
    if (C < 0 || P < 1) return DYN_ERR_SIZE;
    if (dtype != DYN_F32 && dtype != DYN_F64) return DYN_ERR_OPTS;
    dynlat::SiteTable tab;
    if (int rc = fill_table(tab, sites, n_sites)) return rc;
    const int rows = dynlat_rows_per_chain(split_directions, n_sites);
    if (rows < 0) return DYN_ERR_SIZE;
    if (C == 0) return 0;
    const int ns = n_sites <= 1 ? 1 : n_sites <= 2 ? 2 : n_sites <= 4 ? 4 : n_sites <= 8 ? 8 : DYN_MAX_SITES;
    const dim3 grid((unsigned)((C * ns + 63) / 64)), block(64);
#define DYN_LAUNCH_MAP(T, NS)                                                                                              \
    hipLaunchKernelGGL((dynlat::latent_param_map<T, NS>), grid, block, 0, (hipStream_t)stream, tab, C, z, x, lp, dlp_dz, \
                       (int)P, coef, expo, (T *)params, (T *)seeds, rows)
#define DYN_LAUNCH_MAP_N(T)                                                                                                \
    do {                                                                                                                   \
        if (ns == 1) DYN_LAUNCH_MAP(T, 1);                                                                                 \
        else if (ns == 2) DYN_LAUNCH_MAP(T, 2);                                                                            \
        else if (ns == 4) DYN_LAUNCH_MAP(T, 4);                                                                            \
        else if (ns == 8) DYN_LAUNCH_MAP(T, 8);                                                                            \
        else DYN_LAUNCH_MAP(T, DYN_MAX_SITES);                                                                             \
    } while (0)
    if (dtype == DYN_F32) DYN_LAUNCH_MAP_N(float);
    else DYN_LAUNCH_MAP_N(double);
#undef DYN_LAUNCH_MAP_N
#undef DYN_LAUNCH_MAP
    return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
}

extern "C" int dyn_potential_combine(int64_t C, int32_t n, const double *lp, const double *dlp_dz, const double *ll,
                                     const double *dll, double offset, int32_t split_directions, double *u, double *g, void *stream) {
    if (C > 0 && (!lp || !dlp_dz || !ll || !dll || !u || !g)) return DYN_ERR_NULL;
    if (C < 0 || n < 1 || n > DYN_MAX_SITES) return DYN_ERR_SIZE;
    const int rows = dynlat_rows_per_chain(split_directions, n);
    if (rows < 0) return DYN_ERR_SIZE;
    if (C == 0) return 0;
    hipLaunchKernelGGL(dynlat::potential_combine, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, C,
                       (int)n, lp, dlp_dz, ll, dll, offset, rows, u, g);
    return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
}
