// dynode_hip.hip -- C-ABI of libdynode_hip.so (see include/dynode_hip.h): argument
// validation, shape dispatch and kernel enqueue.  No allocation, no synchronisation,
// no global mutable state (the last-error text is thread-local), nothing read from the
// process environment: what tests and tuning tools want pinned travels in
// dyn_solver_opts::hints.
#include "../../include/dynode_hip.h"
#include "solve_kernel.hpp"
#include "seip_kernel.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

namespace dyn {

#define X(T, METHOD, G, S, E, WN, C, W, ND, SPL) \
    extern template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL>(const KArgs<T> &, hipStream_t);
#define XI(T, METHOD, G, S, E, WN, C, W, ND, SPL) \
    extern template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, 1>(const KArgs<T> &, hipStream_t);
#define XF(T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT) \
    extern template hipError_t launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT>(const KArgs<T> &, hipStream_t);
#include "instances.def"
#undef XF
#undef XI
#undef X
#define Y(T, METHOD, GA, L, K1, M1) \
    extern template hipError_t launch_seip<T, METHOD, GA, L, K1, M1>(const KArgs<T> &, hipStream_t);
#define YT(T, METHOD, GA, L, K1, M1) \
    extern template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, 2>(const KArgs<T> &, hipStream_t);
#define YW(T, METHOD, GA, L, K1, M1, KT, NW) \
    extern template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW>(const KArgs<T> &, hipStream_t);
#define YP(T, METHOD, GA, L, K1, M1, KT, NW) \
    extern template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 1>(const KArgs<T> &, hipStream_t);
#define YS(T, METHOD, GA, L, K1, M1, KT, NW) \
    extern template hipError_t launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 3>(const KArgs<T> &, hipStream_t);
#include "seip_instances.def"
#undef YS
#undef YP
#undef YW
#undef YT
#undef Y

template <typename T>
struct DType;
template <>
struct DType<float> {
    static constexpr int id = DYN_F32;
};
template <>
struct DType<double> {
    static constexpr int id = DYN_F64;
};

struct Entry {
    int dtype, method, G, S, E, WN, C, W, ND, SPL, FEAT; // FEAT: bit 0 introductions, bits 1.. vaccination-tier lanes
    void *fn; // hipError_t (*)(const KArgs<T>&, hipStream_t)
};
// SEIP shapes (seip_kernel.hpp) share the table: G = age lanes, S = strains, W = waning states,
// FEAT = kSeip | tiers; the lane group is G * 2^S
constexpr int kSeip = 0x100;
constexpr int kSeipTierLanes = 0x20; // SEIP entry with the tiers dealt over two lanes (seip_kernel.hpp, KT = 2)
constexpr int kSeipWaves2 = 0x40, kSeipWaves4 = 0x80; // ... whose trajectory is owned by a workgroup of 2 / 4 waves (NW)
constexpr int kSeipTierWaves = 0x200; // ... or one tier per tier lane with whole waves as tier lanes: KT = K1, NW = K1 x (lanes / 64)
// ... compiled without seasonal terms, introduced strains, recorded schedules, discontinuity points and a dose spline's third
// and fourth knot (Seip OPT bit 0);
// enqueue swaps it in for a call that uses none of them
constexpr int kSeipPlain = 0x1000;
// ... a plain instance with the step controller in the oracle's arithmetic (Seip OPT bit 1; test-only, hints.strict_control)
constexpr int kSeipStrict = 0x2000;
// FEAT bit 16 (solve_kernel.hpp STRICT_CONTROL): the same for the s/e/i/r/c kernels
constexpr int kStrictControl = 0x10000;
// FEAT bit 14 (solve_kernel.hpp SAVE_ALL): variant without the per-round save-offset / store-width tests, picked by
// enqueue when every compartment is saved into 16-byte aligned rows
constexpr int kSaveAll = 0x4000;
// FEAT bit 15 (solve_kernel.hpp PC): stepping and dense output on two waves of a workgroup; picked by enqueue for launches of
// at most one trajectory-wave per SIMD
constexpr int kProducerConsumer = 0x8000;
// FEAT bit 13 (solve_kernel.hpp LEAN): tangent kernel of the plain family with the options of the reference's inference example
// (normalised, no seasonal forcing, no discontinuity points, adaptive steps, Poisson likelihood of the increments of r) fixed at
// compile time; picked by enqueue when the call is exactly that
constexpr int kLean = 0x2000;
constexpr int kLeanFacts = 0x1E0000;   // (with kLean) bits 17-19: which compartment the lean instance scores, bit 20: its values (solve_kernel.hpp LEAN_CODE / LEAN_MODE)
// the FEAT bits of the lean instance that scores compartment `slot` (0 s .. 4 c) in `mode` (0 values, 1 increments)
static int lean_bits(int slot, int mode) {
    const int code = slot == 3 ? 0 : slot == 4 ? 1 : slot == 2 ? 2 : slot == 1 ? 3 : 4;
    return kLean | (code << 17) | (mode == 0 ? 0x100000 : 0);
}
// FEAT bit 11 (solve_kernel.hpp ADAPTIVE_NO_JUMPS): adaptive steps and no discontinuity points as compile-time facts; picked
// by enqueue on top of a SAVE_ALL variant when the call has neither
constexpr int kAdaptiveNoJumps = 0x0800;
// FEAT bit 10 (solve_kernel.hpp PULLS = false): static-grid-only instance; picked where launch() would not pull anyway
constexpr int kStaticOnly = 0x0400;
// FEAT bit 12 (solve_kernel.hpp FUSED): a lean instance that also runs the sampler's state machine for the chains its waves
// scored; picked when the call carries dyn_solver_opts::nuts_tail
constexpr int kFused = 0x1000;

static const Entry kEntries[] = {
#define X(T, METHOD, G, S, E, WN, C, W, ND, SPL)              \
    {DType<T>::id, METHOD, G, S, E, WN, C, W, ND, SPL, 0,     \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & \
         launch<T, METHOD, G, S, E, WN, C, W, ND, SPL>},
#define XI(T, METHOD, G, S, E, WN, C, W, ND, SPL)             \
    {DType<T>::id, METHOD, G, S, E, WN, C, W, ND, SPL, 1,     \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & \
         launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, 1>},
#define XF(T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT)       \
    {DType<T>::id, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT,  \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & \
         launch<T, METHOD, G, S, E, WN, C, W, ND, SPL, FEAT>},
#include "instances.def"
#undef XF
#undef XI
#undef X
#define Y(T, METHOD, GA, L, K1, M1)                                          \
    {DType<T>::id, METHOD, GA, L, 1, 1, 1, M1, 0, 1, kSeip | K1,             \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & launch_seip<T, METHOD, GA, L, K1, M1>},
#define YT(T, METHOD, GA, L, K1, M1)                                         \
    {DType<T>::id, METHOD, GA, L, 1, 1, 1, M1, 0, 1, kSeip | kSeipTierLanes | K1, \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & launch_seip<T, METHOD, GA, L, K1, M1, 2>},
#define YW(T, METHOD, GA, L, K1, M1, KT, NW)                                                                     \
    {DType<T>::id, METHOD, GA, L, 1, 1, 1, M1, 0, 1,                                                             \
     kSeip | (KT > 2 ? kSeipTierWaves : ((KT == 2 ? kSeipTierLanes : 0) | (NW == 2 ? kSeipWaves2 : kSeipWaves4))) | K1, \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & launch_seip<T, METHOD, GA, L, K1, M1, KT, NW>},
#define YP(T, METHOD, GA, L, K1, M1, KT, NW)                                                                     \
    {DType<T>::id, METHOD, GA, L, 1, 1, 1, M1, 0, 1,                                                             \
     kSeip | kSeipPlain |                                                                                        \
         (KT > 2 ? kSeipTierWaves : ((KT == 2 ? kSeipTierLanes : 0) | (NW == 2 ? kSeipWaves2 : (NW == 4 ? kSeipWaves4 : 0)))) | K1, \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 1>},
#define YS(T, METHOD, GA, L, K1, M1, KT, NW)                                                                     \
    {DType<T>::id, METHOD, GA, L, 1, 1, 1, M1, 0, 1,                                                             \
     kSeip | kSeipPlain | kSeipStrict |                                                                          \
         (KT > 2 ? kSeipTierWaves : ((KT == 2 ? kSeipTierLanes : 0) | (NW == 2 ? kSeipWaves2 : (NW == 4 ? kSeipWaves4 : 0)))) | K1, \
     (void *)(hipError_t(*)(const KArgs<T> &, hipStream_t)) & launch_seip<T, METHOD, GA, L, K1, M1, KT, NW, 3>},
#include "seip_instances.def"
#undef YS
#undef YP
#undef YW
#undef YT
#undef Y
};
static constexpr int kNumEntries = sizeof(kEntries) / sizeof(kEntries[0]);

static int group_width(int A) {
    int g = 1;
    while (g < A) g <<= 1;
    return g;
}

// shapes added at run time (dyn_register_instance): a fixed table, appended under a mutex, entries
// never move or disappear, so readers need no lock beyond the count's acquire load
static Entry g_extra[256];
static std::atomic<int> g_n_extra{0};
static std::mutex g_extra_mutex;

// vaccination tiers occupy 2 or 4 lanes per age (0 = the model has no tier axis, -1 = too many)
static int vax_lanes(const dyn_model_desc *m) {
    return m->n_vax_tiers <= 1 ? 0 : (m->n_vax_tiers <= 2 ? 2 : (m->n_vax_tiers <= 4 ? 4 : -1));
}
static int seip_tiers(const dyn_model_desc *m) { return m->n_vax_tiers > 1 ? m->n_vax_tiers : 1; }
static int model_features(const dyn_model_desc *m) {
    if (m->family == 1) return kSeip | seip_tiers(m);
    return (m->has_intro ? 1 : 0) | (vax_lanes(m) << 1);
}
// lanes one trajectory occupies in a wave
static int entry_waves(const Entry *e) {
    if (!(e->FEAT & kSeip)) return 1;
    if (e->FEAT & kSeipTierWaves) return (e->FEAT & 0x1f) * (((e->G << e->S) + 63) / 64);
    return (e->FEAT & kSeipWaves4) ? 4 : (e->FEAT & kSeipWaves2) ? 2 : 1;
}
static int entry_tier_lanes(const Entry *e) { return (e->FEAT & kSeipTierWaves) ? (e->FEAT & 0x1f) : (e->FEAT & kSeipTierLanes) ? 2 : 1; }
// lanes of a WAVE one trajectory occupies (a wave group owns whole waves: 64)
static int entry_lanes(const Entry *e) {
    if ((e->FEAT & kSeip) && (e->FEAT & kSeipTierWaves) && (e->G << e->S) < 64) return e->G << e->S; // packed: planes side by side
    if ((e->FEAT & kSeip) && entry_waves(e) > 1) return 64;
    return (e->FEAT & kSeip) ? (e->G << e->S) * ((e->FEAT & kSeipTierLanes) ? 2 : 1) : e->G * (e->S / e->SPL);
}

static bool matches(const Entry &e, const dyn_model_desc *m, int G, int dtype, int method, int nd) {
    return e.dtype == dtype && e.method == method && e.G == G && e.S == m->n_strain &&
           e.E == (m->has_e != 0) && e.WN == (m->has_wane != 0) && e.C == (m->has_c != 0) &&
           e.W == m->n_wane && e.ND == nd && e.FEAT == model_features(m);
}

static const dyn_dispatch_hints kNoHints = {0, 0, 0, 0, 0, 0, 0, 0, 0};

static const Entry *find_entry(const dyn_model_desc *m, int dtype, int method, int nd = 0, const dyn_dispatch_hints &h = kNoHints) {
    const int G = group_width(m->n_age);
    // hints.strains_per_lane = n (tuning aid): prefer the variant with n strains per lane
    const int want_spl = h.strains_per_lane;
    const Entry *first = nullptr;
    for (int i = 0; i < kNumEntries; ++i) {
        const Entry &e = kEntries[i];
        if (matches(e, m, G, dtype, method, nd)) {
            if (!first) first = &e;
            if (want_spl > 0 && e.SPL == want_spl) return &e;
        }
    }
    const int n_extra = g_n_extra.load(std::memory_order_acquire);
    for (int i = 0; i < n_extra; ++i) {
        const Entry &e = g_extra[i];
        if (matches(e, m, G, dtype, method, nd)) {
            if (!first) first = &e;
            if (want_spl > 0 && e.SPL == want_spl) return &e;
        }
    }
    return first;
}

// Batch-aware lane mapping.  A launch of fewer waves than SIMDs is as long as its slowest wave's instruction stream, and a
// trajectory whose strains are spread over more lanes (SPL < S) has a shorter one -- at the price of replicated control
// arithmetic, which is free on an otherwise empty GPU.  Measured on the 8 x 4 seasonal model, ms per launch, SPL = 4 / 2 / 1
// (the SPL = 2 kernel fits three waves per SIMD, the SPL = 1 kernel four): B = 4096 (512 waves of 8 trajectories) 0.597 / 0.505 /
// 0.46; B = 8192 0.55-0.58 / 0.54-0.57 / 0.63 (a wash: which of the first two wins changes from build to build and box to box);
// B = 16384 (D = 136) 0.87-0.91 / 0.90 / 1.03; B = 65536 3.50 / 3.73 / 4.58.  Rule: when the default mapping fills at most half a
// wave per SIMD, take the finest split compiled in that still fits two waves per SIMD; otherwise the first entry
// (instances.def order).  hints.strains_per_lane overrides.
// SIMDs of the current device (cached per thread AND device: a thread may move between GPUs)
static int device_simds() {
    static thread_local int c_dev = -1, c_simds = 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev == c_dev) return c_simds;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    c_dev = dev;
    c_simds = 4 * cus;
    return c_simds;
}

static const Entry *entry_for_batch(const Entry *e, const dyn_model_desc *m, int dtype, int method, int nd, int64_t B,
                                    const dyn_dispatch_hints &h) {
    if (h.strains_per_lane > 0 || (e->FEAT & kSeip) || e->SPL == 1 || nd != 0) return e;
    const int simds = device_simds();
    if (simds <= 0) return e;
    const int64_t waves = (B * entry_lanes(e) + 63) / 64;
    if (waves * 2 > simds) return e;   // (more than half a wave per SIMD: the widest per-lane state has the fewest instructions per trajectory)
    const int G = group_width(m->n_age);
    const Entry *best = e;
    auto consider = [&](const Entry &c) {
        if (!matches(c, m, G, dtype, method, nd) || c.SPL >= best->SPL) return;
        if ((B * entry_lanes(&c) + 63) / 64 > 2 * (int64_t)simds) return;
        best = &c;
    };
    for (int i = 0; i < kNumEntries; ++i) consider(kEntries[i]);
    const int n_extra = g_n_extra.load(std::memory_order_acquire);
    for (int i = 0; i < n_extra; ++i) consider(g_extra[i]);
    return best;
}

// the same shape with another feature word (nullptr if it is not compiled in / registered)
static const Entry *find_variant(const Entry *e, int feat) {
    auto same = [&](const Entry &c) {
        return c.dtype == e->dtype && c.method == e->method && c.G == e->G && c.S == e->S && c.E == e->E && c.WN == e->WN &&
               c.C == e->C && c.W == e->W && c.ND == e->ND && c.SPL == e->SPL && c.FEAT == feat;
    };
    for (int i = 0; i < kNumEntries; ++i)
        if (same(kEntries[i])) return &kEntries[i];
    const int n_extra = g_n_extra.load(std::memory_order_acquire);
    for (int i = 0; i < n_extra; ++i)
        if (same(g_extra[i])) return &g_extra[i];
    return nullptr;
}

// SEIP: which lane mapping runs a model.  States that would spill (more than 32 values per lane) take the tiers dealt
// over two lanes when that variant exists; it is also the fallback when only it is compiled in.  A lane group beyond a
// wavefront (8 ages x 8 histories x 2 tier lanes; 16 histories) runs as a wave group: NW waves per trajectory.
static const Entry *select_seip_entry(const dyn_model_desc *m, int dtype, int method, const Entry *e, const dyn_dispatch_hints &h = kNoHints) {
    Entry probe{dtype, method, group_width(m->n_age), m->n_strain, 1, 1, 1, m->n_wane, 0, 1, 0, nullptr};
    const int k1 = seip_tiers(m), lanes = group_width(m->n_age) << m->n_strain;
    const int per_tier = m->n_wane + 3 * m->n_strain, per_lane = k1 * per_tier;
    const int force = h.seip_tier_lanes; // tuning aid: 1 on, -1 off, 0 by state size
    // one tier per wave (K1 >= 3 tiers, a wavefront or more per tier): a third to a half of the state per lane, no padded tier
    // slot, two to three waves per SIMD instead of one -- measured on D = 2496: see DESIGN.md
    if (h.seip_tier_waves >= 0) { // (hints.seip_tier_waves = -1 switches the mapping off)
        const Entry *tw = find_variant(&probe, kSeip | kSeipTierWaves | k1);
        if (tw) return tw;
    }
    if (lanes > 64) { // histories across waves: tier lanes on top (four waves) for big per-lane states, else two waves
        const Entry *w2 = find_variant(&probe, kSeip | kSeipWaves2 | k1);
        const Entry *w4 = find_variant(&probe, kSeip | kSeipTierLanes | kSeipWaves4 | k1);
        if (w4 && (!w2 || (force ? force > 0 : per_lane > 32))) return w4;
        return w2;
    }
    const Entry *two = find_variant(&probe, kSeip | kSeipTierLanes | k1);
    if (lanes == 64) { // tier lanes would need 128: a wave group of two
        const Entry *w2 = find_variant(&probe, kSeip | kSeipTierLanes | kSeipWaves2 | k1);
        if (w2 && (!e || (force ? force > 0 : per_lane > 32))) return w2;
        return e;
    }
    // ... and small float states: half the tiers per lane fit 256 registers, so two waves share a SIMD (measured on the
    // D = 960 shape: 6.5 vs 6.7 ms at 4096 trajectories, 23.3 vs 25.4 ms at 16384) -- only when the one-lane mapping is
    // down to two trajectories per wave: with 4 ages x 4 histories it keeps four per wave and wins (6.15 vs 6.70 ms at 8192)
    const bool small = dtype == DYN_F32 && k1 > 1 && ((k1 + 1) / 2) * per_tier <= 20 && lanes >= 32;
    if (two && (!e || (force ? force > 0 : (per_lane > 32 || small)))) return two;
    return e;
}

static thread_local char tl_error[512] = "";
static thread_local char tl_kernel[160] = "";

// the instance an entry launches, in rocprofv3's spelling (dyn_last_kernel_name)
static void note_kernel(const Entry *e) {
    const char *t = e->dtype == DYN_F64 ? "double" : "float";
    if (e->FEAT & kSeip) {
        const int k1 = e->FEAT & 0x1f, kt = entry_tier_lanes(e);
        const int nv = ((k1 + kt - 1) / kt) * (e->W + 3 * e->S);
        if (entry_waves(e) > 1)
            snprintf(tl_kernel, sizeof(tl_kernel), "dyn::seip_kernel_wave_group<%s, %d, %d, %d, %d, %d, %d, %d, %d>", t, e->method, e->G,
                     e->S, k1, e->W, kt, entry_waves(e), ((e->FEAT & kSeipPlain) ? 1 : 0) | ((e->FEAT & kSeipStrict) ? 2 : 0));
        else
            snprintf(tl_kernel, sizeof(tl_kernel), "dyn::%s<%s, %d, %d, %d, %d, %d, %d, %d>",
                     (e->dtype == DYN_F32 && nv <= 20) ? "seip_kernel_two_waves" : "seip_kernel", t, e->method, e->G, e->S, k1,
                     e->W, kt, (e->FEAT & kSeipPlain) ? 1 : 0);
        return;
    }
    snprintf(tl_kernel, sizeof(tl_kernel), "dyn::solve_kernel%s<%s, %d, %d, %d, %s, %s, %s, %d, %d, %d, %d>", (e->FEAT & kFused) ? "_fused" : "", t, e->method,
             e->G, e->S, e->E ? "true" : "false", e->WN ? "true" : "false", e->C ? "true" : "false", e->W, e->ND, e->SPL,
             e->FEAT);
}

static int check_model(const dyn_model_desc *m) {
    if (!m) return DYN_ERR_NULL;
    if (m->n_age < 1 || m->n_age > 64 || m->n_strain < 1 || m->n_wane < 1) return DYN_ERR_MODEL;
    if (m->n_wane > 1 && !m->has_wane) return DYN_ERR_MODEL;
    if (m->has_intro && m->n_strain > DYN_MAX_STRAINS) return DYN_ERR_MODEL;
    if (m->family != 0 && m->family != 1) return DYN_ERR_MODEL;
    if (m->family == 1) { /* SEIP: groups = age x 2^strains lanes */
        if (m->n_strain > 4 || (group_width(m->n_age) << m->n_strain) > 128 || m->n_vax_tiers < 0 || m->n_vax_tiers > 4 ||
            m->n_vax_knots < 0 || m->n_vax_knots > 4 || !m->has_e || !m->has_c || !m->has_wane || m->normalize)
            return DYN_ERR_MODEL;
        return 0;
    }
    if (m->seasonal_vax) return DYN_ERR_MODEL;
    if (m->n_vax_tiers < 0 || vax_lanes(m) < 0 || m->n_vax_knots < 0 || m->n_vax_knots > 4) return DYN_ERR_MODEL;
    if (vax_lanes(m) > 0 && m->n_age % vax_lanes(m) != 0) return DYN_ERR_MODEL; /* groups = ages x tier lanes */
    return 0;
}

// step schedules (dyn_solve_batch_record / dyn_solve_batch_replay)
struct SchedArgs {
    void *out;                 // record: [B][cap][2]
    int32_t *n_out;            // record: [B]
    const void *in;            // replay: [n_leaders][cap][2]
    const int32_t *n_in;       // replay: [n_leaders]
    const int64_t *leader;     // replay: [B] or nullptr
    int32_t cap;
};

// fused observation likelihood (dyn_solve_batch_loglik)
struct LLArgs {
    const void *obs;
    int32_t slot, mode, row;
    double floor;
    double *ll_out, *dll_out;
};

template <typename T>
static int enqueue(const Entry *e, const dyn_model_desc *m, const dyn_solver_opts *o,
                   const void *y0, int32_t y0_is_batched, const void *params, const void *contact,
                   int64_t B, double t0, double t1, const void *save_ts, int32_t n_save,
                   const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                   int32_t *n_reject, hipStream_t stream, const void *dparams = nullptr,
                   const void *dy0 = nullptr, int32_t dy0_is_batched = 0, void *dys_out = nullptr,
                   const LLArgs *ll = nullptr, const SchedArgs *sc = nullptr, const int32_t *order = nullptr) {
    KArgs<T> ka;
    ka.order = order;
    const dyn_dispatch_hints &h = o->hints;
    ka.pull_mode = h.pull;
    ka.pull_waves = h.pull_waves > 0 ? h.pull_waves : 0;
    // work pulling (dyn_solver_opts::work_counter): the s/e/i/r/c kernels; launch() keeps it only for grids beyond one
    // resident round (the SEIP kernels have a static grid)
    ka.work = (e->FEAT & kSeip) ? nullptr : o->work_counter;
    ka.sched_out = sc ? (T *)sc->out : nullptr;
    ka.sched_n_out = sc ? sc->n_out : nullptr;
    ka.sched_in = sc ? (const T *)sc->in : nullptr;
    ka.sched_n_in = sc ? sc->n_in : nullptr;
    ka.sched_leader = sc ? sc->leader : nullptr;
    ka.sched_cap = sc ? sc->cap : 0;
    ka.obs = ll ? (const T *)ll->obs : nullptr;
    ka.ll_out = ll ? ll->ll_out : nullptr;
    ka.dll_out = ll ? ll->dll_out : nullptr;
    ka.ll_slot = ll ? ll->slot : 0;
    ka.ll_mode = ll ? ll->mode : 0;
    ka.ll_row = ll ? ll->row : 0;
    ka.ll_floor = ll ? (T)ll->floor : (T)0;
    ka.dparams = (const T *)dparams;
    ka.dy0 = (const T *)dy0;
    ka.dout = (T *)dys_out;
    ka.dy0_batched = dy0_is_batched ? 1 : 0;
    ka.y0 = (const T *)y0;
    ka.params = (const T *)params;
    ka.contact = (const T *)contact;
    ka.save_ts = (const T *)save_ts;
    ka.out = (T *)ys_out;
    ka.status = status;
    ka.n_acc = n_accept;
    ka.n_rej = n_reject;
    ka.B = B;
    ka.max_steps = o->max_steps;
    ka.t0 = (T)t0;
    ka.t1 = (T)t1;
    ka.rtol = (T)o->rtol;
    ka.atol = (T)o->atol;
    ka.constant_dt = o->constant_dt > 0.0 ? (T)o->constant_dt : (T)0;
    ka.y0_batched = y0_is_batched ? 1 : 0;
    ka.n_save = n_save;
    ka.A = m->n_age;
    ka.P = dyn_param_dim(m);
    ka.normalize = m->normalize ? 1 : 0;
    ka.seasonal = m->seasonal ? 1 : 0;
    for (int l = 0; l < DYN_MAX_STRAINS; ++l) ka.intro_mask[l] = m->has_intro ? m->intro_age_mask[l] : 0;
    ka.n_vax_tiers = m->n_vax_tiers;
    ka.n_vax_knots = m->n_vax_knots;
    ka.seasonal_vax = m->seasonal_vax ? 1 : 0;
    ka.has_intro = m->has_intro ? 1 : 0;

    // saved-row layout: saved compartments concatenated in state order
    int32_t off[8];
    const int ncomp = dyn_compartment_offsets(m, off);
    // compartment slots of the kernel: 0 s, 1 e, 2 i, 3 r, 4 c
    int slot_of[5], n = 0;
    slot_of[n++] = 0;
    if (m->has_e) slot_of[n++] = 1;
    slot_of[n++] = 2;
    if (m->family != 1) slot_of[n++] = 3; /* SEIP has no r: s e i c */
    if (m->has_c) slot_of[n++] = 4;
    for (int s = 0; s < 5; ++s) ka.save_off[s] = -1;
    int pos = 0;
    bool aligned = true, all_saved = true;
    const int per16 = 16 / (int)sizeof(T);
    for (int c = 0; c < ncomp; ++c) {
        if (save_mask && !save_mask[c]) {
            all_saved = false;
            continue;
        }
        ka.save_off[slot_of[c]] = pos;
        if (pos % per16) aligned = false;
        pos += off[c + 1] - off[c];
    }
    ka.d_saved = pos;
    ka.vec_ok = (aligned && pos % per16 == 0 && ((uintptr_t)ys_out % 16) == 0 &&
                 ((uintptr_t)dys_out % 16) == 0) ? 1 : 0;
    if (pos == 0 || n_save == 0) {
        // nothing to write: still run (status / step counts are outputs too)
        ka.d_saved = pos;
    }
    ka.n_jump = 0;
    for (int j = 0; j < dyn::kMaxJumps; ++j) ka.jump_ts[j] = (T)0;
    for (int j = 0; j < o->n_jump; ++j) {
        // only points strictly inside (t0, t1) matter; sorted input is required (checked by caller)
        if (o->jump_ts[j] > t0 && o->jump_ts[j] < t1) ka.jump_ts[ka.n_jump++] = (T)o->jump_ts[j];
    }
    // Small, save-heavy batches leave most SIMDs idle and are bound by the serial latency of one
    // trajectory's dense output: replicate each trajectory over 2^r lane groups (<= 8) while the
    // grid still fits in one resident round of 2 waves per SIMD on 1024 SIMDs (cfg 2, 512 trajectory-waves: 4 replicas
    // 0.178 ms, 8 replicas 0.204 ms, 2 replicas 0.232 ms, none 0.39 ms -- every replica repeats the stepping).
    {
        const int tpw = 64 / entry_lanes(e);
        const int64_t waves = (B + tpw - 1) / tpw;
        int r = 0;
        if (h.replicas_log2 > 0) {
            r = h.replicas_log2 - 1;
        } else if (n_save >= 64) {
            // measured: tiny states (<= 5 values per lane: SIR, SEIRS) gain up to 4 waves per SIMD
            // (cfg 2: 0.37 -> 0.19 ms); register-heavy VALU-bound shapes do not (cfg 5: 0.72 -> 0.89 ms)
            const int nv = 1 + e->SPL * (e->E + 1 + e->W + e->C);
            if (nv <= 5 && !(e->FEAT & kSeip))
                while (r < 3 && (waves << (r + 1)) <= 2048) ++r;
            // ... and a gradient-solve scored in the kernel is its rows (one Poisson term per saved value and plane), whatever
            // the state's size: up to eight lane groups while the launch stays at half a wave per SIMD (the six-site 2-age x
            // 3-strain model, 128 chains x 8 rows = 32 trajectory-waves: 310 us unreplicated, 239 / 229 with four / eight;
            // two do not fit their likelihood table into LDS).  A call that carries the sampler (nuts_tail) is refused below
            // when whole chains no longer fit a wave at this replication -- the six-site model's eight rows at eight lane
            // groups: two launches per iteration at 229 us beat one at 239 (1.44 against 1.57 s for 128 chains x 200)
            else if (ll && !(e->FEAT & kSeip))
                while (r < 3 && (waves << (r + 1)) <= 512) ++r;
        }
        ka.rep_log2 = r < 0 ? 0 : (r > 3 ? 3 : r);
        if (ll) { // the replicas of a trajectory must share a wave (LDS table, lane reductions)
            const int lanes = entry_lanes(e);
            while (ka.rep_log2 > 0 && (lanes << ka.rep_log2) > 64) --ka.rep_log2;
        }
        if (ll && ka.rep_log2 > 0) {
            // replicated trajectories park their interpolated observable in LDS (solve_kernel.hpp);
            // fall back to unreplicated in-order scoring if that table does not fit
            const size_t tab = (size_t)(64 >> ka.rep_log2) * (size_t)n_save * (size_t)(e->SPL * e->W) *
                               (size_t)(1 + e->ND) * sizeof(T);
            if (tab + (size_t)(n_save + dyn::kMaxJumps) * sizeof(T) > 60 * 1024) ka.rep_log2 = 0;
        }
    }
    if (all_saved && ka.vec_ok && n_save > 0 && !ll && !(e->FEAT & kSaveAll)) {
        const Entry *fast = find_variant(e, e->FEAT | kSaveAll);
        // one trajectory-wave per SIMD or fewer, nothing replicated, the given order, no tangents: the two-wave kernel
        // (hints.producer_consumer = 1 runs it where the variant exists)
        const Entry *pc = find_variant(e, e->FEAT | kSaveAll | kProducerConsumer);
        if (pc && ka.rep_log2 == 0 && !order && !dparams) {
            // Measured (profiles/r03_producer_consumer_ab.md): it LOSES -- cfg 5's share 0.617 -> 0.68-0.75 ms, cfg 3 without the
            // bins axis at B = 8192 0.53-0.55 -> 0.58, cfg 2 0.233 -> 0.289 (replicated one-wave kernel: 0.148): two rendezvous
            // and 45 LDS writes per accepted step cost the stepping wave what the row arithmetic it sheds was worth.
            // Off by default; the bit-for-bit test asks for it.
            if (h.producer_consumer > 0) fast = pc;
        }
        if (fast) e = fast;
        if ((e->FEAT & kSaveAll) && !(e->FEAT & kProducerConsumer)) {
            // further compile-time facts of the call: adaptive steps without discontinuity points (bit 11), a static grid
            // (bit 10: launch() pulls only with a caller's queue on waves of more than two trajectories; hints.pull /
            // .pull_waves keep the pulling instances); hints.general_instance keeps the general one.  The most specific
            // variant compiled in wins.
            const bool nojump = ka.n_jump == 0 && !(o->constant_dt > 0.0) && !h.general_instance;
            const bool stat = (64 / entry_lanes(e) <= 2 || !order) && h.pull == 0 && h.pull_waves <= 0 && !h.general_instance;
            const int want[3] = {(nojump ? kAdaptiveNoJumps : 0) | (stat ? kStaticOnly : 0), stat ? kStaticOnly : 0,
                                 nojump ? kAdaptiveNoJumps : 0};
            for (int i = 0; i < 3; ++i) {
                if (want[i] == 0) continue;
                const Entry *v = find_variant(e, e->FEAT | want[i]);
                if (v) {
                    e = v;
                    break;
                }
            }
        }
    }
    if (ll && ll->slot >= 0 && ll->slot <= 4 && (ll->mode == 0 || ll->mode == 1) && m->normalize && !m->seasonal && !m->has_intro &&
        ka.n_jump == 0 && !(o->constant_dt > 0.0) && !sc && !order && !(e->FEAT & kLean)) {
        const Entry *lean = h.general_instance ? nullptr : find_variant(e, e->FEAT | lean_bits(ll->slot, ll->mode));
        if (lean) e = lean;
    }
    // Tangent instances: gradient-solves are small batches in the given order, so the static grid (bit 10) and -- without
    // discontinuity points or constant steps -- "adaptive, no jumps" (bit 11) can be compile-time facts where the twin is
    // compiled: the general 2-age x 3-strain tangent instance spills 120 scalars, this twin 68 (the lean ones 16)
    if (e->ND > 0 && !(e->FEAT & (kLean | kSeip | kStaticOnly | kAdaptiveNoJumps)) && !order && !sc && h.pull == 0 && h.pull_waves <= 0 &&
        !h.general_instance && ka.n_jump == 0 && !(o->constant_dt > 0.0)) {
        const Entry *v = find_variant(e, e->FEAT | kStaticOnly | kAdaptiveNoJumps);
        if (v) e = v;
    }
    if (h.strict_control && !(e->FEAT & kSeip)) {   // test-only twin with the oracle's controller arithmetic, where compiled
        const Entry *strict = find_variant(e, e->FEAT | kStrictControl);
        if (strict) e = strict;
    }
    ka.nuts_tail = nullptr;
    if (o->nuts_tail) {
        // one launch per sampler iteration: only where the wave <-> chain correspondence holds (include/dynode_hip.h)
        const dynnuts::Tail *const t = static_cast<const dynnuts::Tail *>(o->nuts_tail);
        if (t->magic != dynnuts::kTailMagic) return DYN_ERR_OPTS; // not a dyn_nuts_tail_pack blob
        const int rows = t->rows_per_chain;
        // (a lean instance carries the state machine up to four dimensions; beyond, the general instance of the shape takes the call)
        // (... and so does it where the lean instance has no twin that carries the sampler)
        const int lean_dims = (e->FEAT & kLeanFacts) ? dynnuts::kFusedMaxDim : dynnuts::kFusedLeanMaxDim;   // (solve_kernel.hpp kTailMaxDim)
        if ((e->FEAT & kLean) && (t->st.dim > lean_dims || !find_variant(e, e->FEAT | kFused))) {
            const Entry *general = find_variant(e, e->FEAT & ~(kLean | kLeanFacts));
            if (general) e = general;
        }
        if ((e->FEAT & kStaticOnly) && !find_variant(e, e->FEAT | kFused)) {   // (the static twin above: its base may have the fused one)
            const Entry *base = find_variant(e, e->FEAT & ~(kStaticOnly | kAdaptiveNoJumps));
            if (base) e = base;
        }
        const Entry *fused = e->ND > 0 ? find_variant(e, e->FEAT | kFused) : nullptr;
        const int nt = (64 / entry_lanes(e)) >> ka.rep_log2; // trajectories per wave
        // directions split over the rows of a chain (one each; rows beyond the sites are padding) or all in its one row
        const bool dirs_ok = t->map.split ? (e->ND == 1 && rows >= t->st.dim) : (rows == 1 && e->ND == t->st.dim);
        if (!fused || !ll || order || rows < 1 || nt < rows || nt % rows != 0 || B != (int64_t)t->st.n_chains * rows || !dirs_ok ||
            (t->map.f64 != 0) != (sizeof(T) == 8) || t->map.params != params || t->map.seeds != dparams) {
            snprintf(tl_error, sizeof(tl_error), "nuts_tail: this call cannot carry the sampler's side (see dyn_solver_opts::nuts_tail)");
            return DYN_ERR_UNSUPPORTED;
        }
        e = fused;
        ka.nuts_tail = t;
    }
    typedef hipError_t (*fn_t)(const KArgs<T> &, hipStream_t);
    const hipError_t err = ((fn_t)e->fn)(ka, stream);
    if (err != hipSuccess) {
        snprintf(tl_error, sizeof(tl_error), "kernel launch failed: %s", hipGetErrorString(err));
        return DYN_ERR_LAUNCH;
    }
    note_kernel(e);
    return 0;
}

} // namespace dyn

extern "C" {

int32_t dyn_abi_version(void) { return DYN_ABI_VERSION; }
int32_t dyn_model_desc_size(void) { return (int32_t)sizeof(dyn_model_desc); }
int32_t dyn_solver_opts_size(void) { return (int32_t)sizeof(dyn_solver_opts); }

int32_t dyn_n_compartments(const dyn_model_desc *m) {
    if (m->family == 1) return 4; /* s e i c */
    return 3 + (m->has_e ? 1 : 0) + (m->has_c ? 1 : 0);
}

int32_t dyn_compartment_offsets(const dyn_model_desc *m, int32_t *off) {
    if (m->family == 1) {
        const int groups = (m->n_age << m->n_strain) * dyn::seip_tiers(m);
        off[0] = 0;
        off[1] = groups * m->n_wane;
        for (int c = 2; c <= 4; ++c) off[c] = off[c - 1] + groups * m->n_strain;
        return 4;
    }
    const int A = m->n_age, AS = m->n_age * m->n_strain;
    int n = 0, pos = 0;
    off[n++] = pos;
    pos += A;
    if (m->has_e) {
        off[n++] = pos;
        pos += AS;
    }
    off[n++] = pos;
    pos += AS;
    off[n++] = pos;
    pos += AS * m->n_wane;
    if (m->has_c) {
        off[n++] = pos;
        pos += AS;
    }
    off[n] = pos;
    return n;
}

int32_t dyn_state_dim(const dyn_model_desc *m) {
    if (m->family == 1) return (m->n_age << m->n_strain) * dyn::seip_tiers(m) * (m->n_wane + 3 * m->n_strain);
    return m->n_age *
           (1 + m->n_strain * ((m->has_e ? 1 : 0) + 1 + m->n_wane + (m->has_c ? 1 : 0)));
}

int32_t dyn_param_dim(const dyn_model_desc *m) {
    if (m->family == 1) {
        const int L = m->n_strain, K1 = dyn::seip_tiers(m);
        return 3 * L + m->n_wane + (m->has_intro ? 3 * L : 0) + (m->seasonal ? 3 : 0) + (m->seasonal_vax ? 1 : 0) + m->n_age +
               (1 << L) * K1 * m->n_wane * L + m->n_age * K1 * (4 + 2 * m->n_vax_knots);
    }
    return m->n_strain * (2 + (m->has_e ? 1 : 0) + (m->has_wane ? 1 : 0) + (m->has_intro ? 3 : 0)) +
           (m->seasonal ? 3 : 0) +
           (m->n_vax_tiers > 1 ? m->n_age * (m->n_strain + 4 + 2 * m->n_vax_knots) : 0);
}

int32_t dyn_trajectories_per_wave(const dyn_model_desc *m) {
    if (dyn::check_model(m)) return 0;
    const dyn::Entry *e = dyn::find_entry(m, DYN_F32, DYN_TSIT5, 0);
    if (m->family == 1) {
        e = dyn::select_seip_entry(m, DYN_F32, DYN_TSIT5, e);
        return e ? 64 / dyn::entry_lanes(e) : 64 / (dyn::group_width(m->n_age) << m->n_strain);
    }
    const int gs = e ? e->S / e->SPL : 1;
    return 64 / (dyn::group_width(m->n_age) * gs);
}

int32_t dyn_trajectories_per_wave_for_batch(const dyn_model_desc *m, const dyn_solver_opts *o, int64_t B) {
    if (dyn::check_model(m) || !o) return 0;
    if (m->family == 1) return dyn_trajectories_per_wave(m);
    const dyn::Entry *e = dyn::find_entry(m, o->dtype, o->method, 0, o->hints);
    if (!e) return 0;
    e = dyn::entry_for_batch(e, m, o->dtype, o->method, 0, B, o->hints);
    return 64 / dyn::entry_lanes(e);
}

int32_t dyn_is_supported(const dyn_model_desc *m, const dyn_solver_opts *o) {
    if (dyn::check_model(m) || !o) return 0;
    if (m->family == 1) // any lane mapping will do
        return dyn::select_seip_entry(m, o->dtype, o->method, dyn::find_entry(m, o->dtype, o->method)) ? 1 : 0;
    return dyn::find_entry(m, o->dtype, o->method) ? 1 : 0;
}

const char *dyn_last_error(void) { return dyn::tl_error; }
const char *dyn_last_kernel_name(void) { return dyn::tl_kernel; }

static int solve_impl(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0,
                      int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                      double t0, double t1, const void *save_ts, int32_t n_save,
                      const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                      int32_t *n_reject, void *stream, int32_t n_dir, const void *dparams,
                      const void *dy0, int32_t dy0_is_batched, void *dys_out,
                      const dyn::LLArgs *ll = nullptr, const dyn::SchedArgs *sc = nullptr, const int32_t *order = nullptr) {
    dyn::tl_error[0] = 0;
    int rc = dyn::check_model(m);
    if (rc) return rc;
    if (!o || !y0 || !params || !contact || !status || !n_accept || !n_reject) return DYN_ERR_NULL;
    if (B < 0 || n_save < 0 || (n_save > 0 && (!save_ts || (!ys_out && !ll)))) return DYN_ERR_SIZE;
    if (n_dir < 0 || (n_dir > 0 && (!dparams || (n_save > 0 && !dys_out && !ll)))) return DYN_ERR_NULL;
    if ((o->method != DYN_TSIT5 && o->method != DYN_DOPRI5) ||
        (o->dtype != DYN_F32 && o->dtype != DYN_F64))
        return DYN_ERR_OPTS;
    if (!(o->constant_dt > 0.0) && (!(o->rtol > 0.0) || !(o->atol > 0.0))) return DYN_ERR_TOL;
    if (o->max_steps < 1 || !(t1 >= t0)) return DYN_ERR_TOL;
    if (o->n_jump < 0 || (o->n_jump > 0 && !o->jump_ts)) return DYN_ERR_JUMP;
    if ((size_t)n_save * (o->dtype == DYN_F64 ? 8 : 4) > DYN_MAX_SAVE_BYTES) {
        snprintf(dyn::tl_error, sizeof(dyn::tl_error),
                 "save grid of %d points exceeds the %d-byte LDS table: split the solve in time",
                 n_save, DYN_MAX_SAVE_BYTES);
        return DYN_ERR_UNSUPPORTED;
    }
    if (o->n_jump > dyn::kMaxJumps) {
        snprintf(dyn::tl_error, sizeof(dyn::tl_error), "at most %d discontinuity_points are supported",
                 dyn::kMaxJumps);
        return DYN_ERR_UNSUPPORTED;
    }
    for (int j = 1; j < o->n_jump; ++j)
        if (!(o->jump_ts[j] > o->jump_ts[j - 1])) return DYN_ERR_JUMP; /* must be strictly increasing */
    if (m->family == 1 && n_dir > 0) {
        snprintf(dyn::tl_error, sizeof(dyn::tl_error), "the SEIP kernels have no tangent planes: differentiate them with "
                 "dyn_solve_batch_record + dyn_solve_batch_replay (central differences on the recorded step sequence)");
        return DYN_ERR_UNSUPPORTED;
    }
    const dyn_dispatch_hints &h = o->hints;
    if (h.replicas_log2 < 0 || h.replicas_log2 > 4 || h.pull_waves < 0 || h.strains_per_lane < 0) return DYN_ERR_OPTS;
    const dyn::Entry *e = dyn::find_entry(m, o->dtype, o->method, n_dir, h);
    if (m->family == 1) {
        e = dyn::select_seip_entry(m, o->dtype, o->method, e, h);
        // a call with no seasonal term, no introduced strain, dose splines of at most two knots, no recorded schedule and adaptive
        // steps without discontinuity points takes the variant compiled without them, when that shape has one
        // (hints.general_instance keeps the general one)
        if (e && !m->seasonal && !m->seasonal_vax && !m->has_intro && m->n_vax_knots <= 2 && !sc && o->n_jump == 0 && !(o->constant_dt > 0) &&
            !h.general_instance) {
            const dyn::Entry *plain = dyn::find_variant(e, e->FEAT | dyn::kSeipPlain);
            if (plain) e = plain;
            if (plain && h.strict_control) {   // test-only twin with the oracle's controller arithmetic, where compiled
                const dyn::Entry *strict = dyn::find_variant(e, e->FEAT | dyn::kSeipStrict);
                if (strict) e = strict;
            }
        }
    } else if (e) e = dyn::entry_for_batch(e, m, o->dtype, o->method, n_dir, B, h);
    if (!e && m->family == 1) {
        snprintf(dyn::tl_error, sizeof(dyn::tl_error),
                 "no SEIP kernel compiled for A=%d strains=%d tiers=%d waning states=%d dtype=%d method=%d; to add it "
                 "put  Y(%s, %d, %d, %d, %d, %d)  into dynode_amd/csrc/seip_instances.def and rebuild",
                 m->n_age, m->n_strain, dyn::seip_tiers(m), m->n_wane, o->dtype, o->method,
                 o->dtype == DYN_F64 ? "double" : "float", o->method, dyn::group_width(m->n_age), m->n_strain,
                 dyn::seip_tiers(m), m->n_wane);
        return DYN_ERR_UNSUPPORTED;
    }
    if (sc && m->family != 1) {
        snprintf(dyn::tl_error, sizeof(dyn::tl_error), "step schedules are implemented for the SEIP family (the s/e/i/r/c "
                 "kernels have tangent planes: dyn_solve_batch_jvp)");
        return DYN_ERR_UNSUPPORTED;
    }
    if (sc && (sc->cap < 1 || (sc->out && !sc->n_out) || (sc->in && !sc->n_in))) return DYN_ERR_SIZE;
    if (e && m->family == 1) { // susceptibility table and splines of every trajectory of a wave sit in LDS
        const size_t per_traj = (size_t)(1 << m->n_strain) * dyn::seip_tiers(m) * m->n_wane * m->n_strain +
                                (size_t)m->n_age * dyn::seip_tiers(m) * 12 /* Seip::kSplRow: padded spline rows */ +
                                (sc && sc->in ? (size_t)2 * sc->cap : 0); /* replayed schedule */
        const int nw = dyn::entry_waves(e), ktl = dyn::entry_tier_lanes(e), kl = (dyn::seip_tiers(m) + ktl - 1) / ktl;
        const size_t mailbox = nw > 1 ? (size_t)2 * nw * 64 * (m->n_strain + kl * 4 + m->n_wane + 2 * m->n_strain) : 0; /* >= 2 NW NSLOT 64 */
        const size_t bytes = ((size_t)n_save + (o->n_jump > 0 ? dyn::kMaxJumps : 0) /* as launch_seip sizes it */ + (64 / dyn::entry_lanes(e)) * per_traj + mailbox) * (o->dtype == DYN_F64 ? 8 : 4);
        const size_t limit = nw > 1 ? 160 * 1024 : 64 * 1024; /* a wave group is alone (or two) on its CU */
        if (bytes > limit) {
            snprintf(dyn::tl_error, sizeof(dyn::tl_error), "SEIP tables need %zu bytes of LDS per workgroup (limit %zu)", bytes, limit);
            return DYN_ERR_UNSUPPORTED;
        }
    }
    if (!e) {
        const int ga = dyn::group_width(m->n_age);
        snprintf(dyn::tl_error, sizeof(dyn::tl_error),
                 "no kernel compiled for A=%d S=%d e=%d wane=%d c=%d W=%d intro=%d dtype=%d method=%d tangent "
                 "directions=%d; to add it put  %s(%s, %d, %d, %d, %s, %s, %s, %d, %d, %d)  into "
                 "dynode_amd/csrc/instances.def and rebuild (make -C dynode_amd/csrc)",
                 m->n_age, m->n_strain, m->has_e, m->has_wane, m->has_c, m->n_wane, m->has_intro, o->dtype,
                 o->method, n_dir, dyn::model_features(m) > 1 ? "XF" : (m->has_intro ? "XI" : "X"),
                 o->dtype == DYN_F64 ? "double" : "float", o->method,
                 ga, m->n_strain, m->has_e ? "true" : "false", m->has_wane ? "true" : "false",
                 m->has_c ? "true" : "false", m->n_wane, n_dir, m->n_strain);
        return DYN_ERR_UNSUPPORTED;
    }
    if (B == 0) return 0;
    if (o->dtype == DYN_F64)
        return dyn::enqueue<double>(e, m, o, y0, y0_is_batched, params, contact, B, t0, t1,
                                    save_ts, n_save, save_mask, ys_out, status, n_accept, n_reject,
                                    (hipStream_t)stream, dparams, dy0, dy0_is_batched, dys_out, ll, sc, order);
    return dyn::enqueue<float>(e, m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts,
                               n_save, save_mask, ys_out, status, n_accept, n_reject,
                               (hipStream_t)stream, dparams, dy0, dy0_is_batched, dys_out, ll, sc, order);
}

int dyn_solve_batch(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0,
                    int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                    double t0, double t1, const void *save_ts, int32_t n_save,
                    const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                    int32_t *n_reject, void *stream) {
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save,
                      save_mask, ys_out, status, n_accept, n_reject, stream, 0, nullptr, nullptr, 0,
                      nullptr);
}

int dyn_solve_batch_ordered(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0, int32_t y0_is_batched,
                            const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                            int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                            int32_t *n_reject, const int32_t *order, void *stream) {
    if (order && B > 0x7fffffffLL) return DYN_ERR_SIZE;
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save, save_mask, ys_out, status,
                      n_accept, n_reject, stream, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, order);
}

int dyn_solve_batch_record(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0, int32_t y0_is_batched,
                           const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                           int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, void *sched_out, int32_t *sched_n_out, int32_t sched_cap, void *stream) {
    if (!sched_out || !sched_n_out) return DYN_ERR_NULL;
    const dyn::SchedArgs sc{sched_out, sched_n_out, nullptr, nullptr, nullptr, sched_cap};
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save, save_mask, ys_out, status,
                      n_accept, n_reject, stream, 0, nullptr, nullptr, 0, nullptr, nullptr, &sc);
}

int dyn_solve_batch_replay(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0, int32_t y0_is_batched,
                           const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                           int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, const void *sched, const int32_t *sched_n, const int64_t *leader,
                           int32_t sched_cap, void *stream) {
    if (!sched || !sched_n) return DYN_ERR_NULL;
    const dyn::SchedArgs sc{nullptr, nullptr, sched, sched_n, leader, sched_cap};
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save, save_mask, ys_out, status,
                      n_accept, n_reject, stream, 0, nullptr, nullptr, 0, nullptr, nullptr, &sc);
}

int dyn_solve_batch_jvp(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0,
                        int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                        double t0, double t1, const void *save_ts, int32_t n_save,
                        const uint8_t *save_mask, int32_t n_dir, const void *dparams,
                        const void *dy0, int32_t dy0_is_batched, void *ys_out, void *dys_out,
                        int32_t *status, int32_t *n_accept, int32_t *n_reject, void *stream) {
    if (n_dir < 1) return DYN_ERR_SIZE;
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save,
                      save_mask, ys_out, status, n_accept, n_reject, stream, n_dir, dparams, dy0,
                      dy0_is_batched, dys_out);
}

int dyn_solve_batch_loglik(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0,
                           int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                           double t0, double t1, const void *save_ts, int32_t n_save,
                           int32_t obs_compartment, int32_t obs_mode, double rate_floor, const void *obs,
                           int32_t n_dir, const void *dparams, const void *dy0, int32_t dy0_is_batched,
                           double *logp_out, double *dlogp_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, void *stream) {
    if (n_dir < 1) return DYN_ERR_SIZE;
    if (!m || !obs || !logp_out || !dlogp_out) return DYN_ERR_NULL;
    int rc = dyn::check_model(m);
    if (rc) return rc;
    if (obs_mode < 0 || obs_mode > 1 || !(rate_floor > 0.0)) return DYN_ERR_OPTS;
    if (n_save < 1 + obs_mode) return DYN_ERR_SIZE;
    int32_t off[8];
    const int ncomp = dyn_compartment_offsets(m, off);
    if (obs_compartment < 0 || obs_compartment >= ncomp) return DYN_ERR_MODEL;
    // compartment index (reference order s, (e), i, r, (c)) -> kernel slot 0 s, 1 e, 2 i, 3 r, 4 c
    int slot_of[5], n = 0;
    slot_of[n++] = 0;
    if (m->has_e) slot_of[n++] = 1;
    slot_of[n++] = 2;
    if (m->family != 1) slot_of[n++] = 3; /* SEIP has no r: s e i c */
    if (m->has_c) slot_of[n++] = 4;
    dyn::LLArgs ll;
    ll.obs = obs;
    ll.slot = slot_of[obs_compartment];
    ll.mode = obs_mode;
    ll.row = off[obs_compartment + 1] - off[obs_compartment];
    ll.floor = rate_floor;
    ll.ll_out = logp_out;
    ll.dll_out = dlogp_out;
    return solve_impl(m, o, y0, y0_is_batched, params, contact, B, t0, t1, save_ts, n_save, nullptr,
                      nullptr, status, n_accept, n_reject, stream, n_dir, dparams, dy0, dy0_is_batched,
                      nullptr, &ll);
}

int dyn_register_instance(int32_t dtype, int32_t method, int32_t ga, int32_t n_strain, int32_t has_e,
                          int32_t has_wane, int32_t has_c, int32_t n_wane, int32_t n_dir, int32_t spl,
                          int32_t features, void *launch_fn) {
    if (!launch_fn) return DYN_ERR_NULL;
    if ((dtype != DYN_F32 && dtype != DYN_F64) || (method != DYN_TSIT5 && method != DYN_DOPRI5)) return DYN_ERR_OPTS;
    if (ga < 1 || ga > 64 || (ga & (ga - 1)) || n_strain < 1 || n_strain > 64 || spl < 1 || n_strain % spl ||
        n_wane < 1 || n_dir < 0 || ga * (n_strain / spl) > 64)
        return DYN_ERR_MODEL;
    std::lock_guard<std::mutex> lock(dyn::g_extra_mutex);
    const int n = dyn::g_n_extra.load(std::memory_order_relaxed);
    if (n >= (int)(sizeof(dyn::g_extra) / sizeof(dyn::g_extra[0]))) return DYN_ERR_SIZE;
    dyn::g_extra[n] = dyn::Entry{dtype, method, ga, n_strain, has_e != 0, has_wane != 0, has_c != 0, n_wane, n_dir,
                                 spl, features, launch_fn};
    dyn::g_n_extra.store(n + 1, std::memory_order_release);
    return 0;
}

int32_t dyn_is_supported_jvp(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir) {
    if (dyn::check_model(m) || !o) return 0;
    return dyn::find_entry(m, o->dtype, o->method, n_dir) ? 1 : 0;
}

int32_t dyn_lean_twin(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir, int32_t obs_compartment, int32_t obs_mode,
                      int32_t *spl, int32_t *features) {
    if (dyn::check_model(m) || !o || !spl || !features || n_dir < 1 || m->family != 0 || obs_mode < 0 || obs_mode > 1) return 0;
    if (!m->normalize || m->seasonal || m->has_intro) return 0;      // (what a lean instance takes as facts)
    // compartment index (reference order s, (e), i, r, (c)) -> kernel slot 0 s, 1 e, 2 i, 3 r, 4 c, as dyn_solve_batch_loglik
    const int slots[5] = {0, m->has_e ? 1 : -1, 2, 3, m->has_c ? 4 : -1};
    int slot = -1, idx = 0;
    for (int k = 0; k < 5; ++k)
        if (slots[k] >= 0 && idx++ == obs_compartment) slot = slots[k];
    if (slot < 0) return 0;
    const dyn::Entry *e = dyn::find_entry(m, o->dtype, o->method, n_dir);
    if (!e || (e->FEAT & dyn::kSeip)) return 0;
    *spl = e->SPL;
    *features = e->FEAT | dyn::lean_bits(slot, obs_mode);
    return dyn::find_variant(e, *features) ? 1 : -1;
}

int32_t dyn_fused_twin(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir) {
    if (dyn::check_model(m) || !o || n_dir < 1 || m->family != 0) return 0;
    const dyn::Entry *e = dyn::find_entry(m, o->dtype, o->method, n_dir);
    if (!e) return 0;
    return dyn::find_variant(e, e->FEAT | dyn::kFused) ? 1 : -e->SPL;
}

} // extern "C"
