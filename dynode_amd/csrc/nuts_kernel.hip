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
#include "nuts_device.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string.h>

int dynlat_rows_per_chain(int32_t split_directions, int32_t n_sites);   // latent_kernel.hip

namespace dynnuts {

template <int D>
__global__ void __launch_bounds__(64) nuts_advance(const dyn_nuts_state st, const dynlat::MapArgs map) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= st.n_chains) return;
    Handed<D> handed;
    handed.ll = 0.0;
    for (int i = 0; i < D; ++i) handed.dll[i] = 0.0;
    if (st.pot_lp != nullptr) {
        handed.ll = st.pot_ll[(int64_t)c * st.pot_ll_stride];
        const int64_t first = (int64_t)c * (st.pot_dll_stride > 0 ? st.pot_dll_stride : D);
        for (int i = 0; i < D; ++i) handed.dll[i] = st.pot_dll[first + i];
    }
    advance_chain<D, false>(st, map, c, handed);
}

// ... beyond kRegDim dimensions: the same state machine with the dimension a run-time number (nuts_device.hpp).  The
// potential arrives as (u_new, g_new) or -- up to DYN_MAX_SITES dimensions -- in the parts of a folded potential (pot_*),
// with the map of the next position behind it (dyn_nuts_advance_mapped).
__global__ void __launch_bounds__(64) nuts_advance_any_dim(const dyn_nuts_state st, const dynlat::MapArgs map) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= st.n_chains) return;
    Handed<DYN_NUTS_MAX_DIM> handed;
    handed.ll = 0.0;
    if (st.pot_lp != nullptr) {
        const int D = st.dim;
        handed.ll = st.pot_ll[(int64_t)c * st.pot_ll_stride];
        const int64_t first = (int64_t)c * (st.pot_dll_stride > 0 ? st.pot_dll_stride : D);
        for (int i = 0; i < D; ++i) handed.dll[i] = st.pot_dll[first + i];
    }
    advance_chain<DYN_NUTS_MAX_DIM, true>(st, map, c, handed);
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
    const unsigned blocks = (unsigned)((st->n_chains + 63) / 64);
    if (st->pooled && (!st->pool || !st->pool_ro || !st->pend)) return DYN_ERR_NULL;
    using Kern = void (*)(const dyn_nuts_state, const dynlat::MapArgs);
    static_assert(dynnuts::kRegDim == 8, "one compile-time instance per dimension up to kRegDim");
    static const Kern kernels[dynnuts::kRegDim] = {
        dynnuts::nuts_advance<1>, dynnuts::nuts_advance<2>, dynnuts::nuts_advance<3>, dynnuts::nuts_advance<4>,
        dynnuts::nuts_advance<5>, dynnuts::nuts_advance<6>, dynnuts::nuts_advance<7>, dynnuts::nuts_advance<8>};
    if (st->dim > dynnuts::kRegDim) {
        // the run-time-dimension instance: per-chain adaptation; (u_new, g_new), or a folded potential's parts and map
        if (st->pooled) return DYN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(dynnuts::nuts_advance_any_dim, dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
        return hipGetLastError() == hipSuccess ? 0 : DYN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernels[st->dim - 1], dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
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
