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
        go = advance_chain<D, false, NS == 1>(st, map, c, handed, ze);
    }
    if constexpr (NS > 1)
        if (map.enabled) map_lanes<NS>(map, go, ze, D, c, sub);     // (wave-uniform)
}

// ... beyond kRegDim dimensions: the same state machine with the dimension a run-time number (nuts_device.hpp).  The
// potential arrives as (u_new, g_new) or -- up to DYN_MAX_SITES dimensions -- in the parts of a folded potential (pot_*),
// with the map of the next position behind it (dyn_nuts_advance_mapped).
template <int NS>
__global__ void __launch_bounds__(64) nuts_advance_any_dim(const dyn_nuts_state st, const dynlat::MapArgs map) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int c = t / NS, sub = t % NS;
    bool go = false;
    double ze[DYN_NUTS_MAX_DIM];
    if constexpr (NS > 1)
        for (int i = 0; i < NS; ++i) ze[i] = 0.0;
    if (sub == 0 && c < st.n_chains) {
        Handed<DYN_NUTS_MAX_DIM> handed;
        handed.ll = 0.0;
        if (st.pot_lp != nullptr) {
            const int D = st.dim;
            handed.ll = st.pot_ll[(int64_t)c * st.pot_ll_stride];
            const int64_t first = (int64_t)c * (st.pot_dll_stride > 0 ? st.pot_dll_stride : D);
            for (int i = 0; i < D; ++i) handed.dll[i] = st.pot_dll[first + i];
        }
        go = advance_chain<DYN_NUTS_MAX_DIM, true, NS == 1>(st, map, c, handed, ze);
    }
    if constexpr (NS > 1)
        if (map.enabled) map_lanes<NS>(map, go, ze, st.dim, c, sub);
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
        // the run-time-dimension instance: per-chain adaptation; (u_new, g_new), or a folded potential's parts and map
        if (st->pooled) return DYN_ERR_UNSUPPORTED;
        const int ns = map.enabled ? DYN_MAX_SITES : 1;
        const unsigned blocks = (unsigned)(((int64_t)st->n_chains * ns + 63) / 64);
        if (map.enabled)
            hipLaunchKernelGGL(dynnuts::nuts_advance_any_dim<DYN_MAX_SITES>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
        else
            hipLaunchKernelGGL(dynnuts::nuts_advance_any_dim<1>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, *st, map);
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
