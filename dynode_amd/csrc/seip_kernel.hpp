// seip_kernel.hpp -- the SEIP family (include/dynode_hip.h "SEIP"; ode_model.md:15-53, 70-105, 176-211):
// age x immune history x vaccination tier x waning state, as a model family of the library's one stepping loop
// (stepper.hpp, Stepper<Seip<...>>::run): this file is the lane mapping, the per-trajectory tables, the right-hand side
// and the row writer; starting step, step control, discontinuity points, SaveAt rounds and recorded schedules are the
// stepper's (the dense-output weights and the age contraction are Solver's own functions; same status codes).
//
// Lane mapping: a trajectory owns GA x H lanes, GA = power of two >= n_age in the low lane bits (so the
// age contraction is the DPP gather of solve_kernel.hpp, unchanged), H = 2^L immune histories above
// them.  Lane (a, j) keeps in registers everything of age a with history j:
//     s[K1][M1] | e[K1][L] | i[K1][L] | c[K1][L]        NV = K1 * (M1 + 3 L) values
// Cross-lane traffic per right-hand side, all xor-shaped:
//     sum over histories of the infectious and of each tier's susceptibles   (butterfly over the high bits)
//     age contraction  lambda_a = sum_b C[a][b] x_b                          (DPP, low bits)
//     recovery  eta(j, l) = j | 2^l: lanes whose bit l is set take gamma_l i from lane j ^ 2^l (and keep their own)
// Infection, waning, vaccination and the seasonal reset move people inside a lane.  The susceptibility
// table sus[H][K1][M1][L] and the dose-rate splines of a trajectory are staged in LDS.
// Primal only (no tangent planes), no replication: one group per trajectory.
#pragma once
#include "solve_kernel.hpp"

#ifndef DYN_SEIP_CACHE_TW
#define DYN_SEIP_CACHE_TW 1
#endif
#ifndef DYN_SEIP_CACHE_6
#define DYN_SEIP_CACHE_6 1
#endif
#ifndef DYN_SEIP_CACHE_SUS
#define DYN_SEIP_CACHE_SUS 1
#endif
namespace dyn {

// KT = 2 ("tier lanes"): the vaccination tiers of an (age, history) pair are dealt over two lanes -- lane bit above the
// history bits, tier k on lane k % 2, slot k / 2 -- so a lane holds ceil(K1 / 2) * (M1 + 3 L) values instead of
// K1 * (M1 + 3 L): three strains x three tiers no longer spill.  Every tier-to-tier flow (vaccination k -> k + 1, the
// seasonal reset K -> K - 1) then crosses to the partner lane with one xor exchange.
// KT = K1 = 3, 4 ("one tier per wave"): when an (age, history) plane already fills a wavefront (8 ages x 8 histories), every
// tier gets its own wave(s) and a lane holds ONE tier: M1 + 3 L values (13 for the D = 2496 model against 26 with two tier
// lanes), no padded tier slot, and 2 waves per SIMD instead of 1.  Measured on D = 2496: 21.2 -> 16.0 ms per 4096 trajectories.
// NW > 1 ("wave groups"): a trajectory whose G = GA * H * KT lanes exceed a wavefront is owned by a WORKGROUP of NW = G / 64
// waves (one trajectory per workgroup).  The lane bits above 64 -- the top immune-history bit(s) and / or the tier-lane
// bit -- select the wave; what crossed lanes with an xor exchange crosses waves through a small LDS mailbox, with ONE
// workgroup barrier per right-hand side (two when the vaccination flow needs a cross-wave tier total first); the error
// norm takes one more per step.  All waves of a trajectory see bit-identical norms, so their control flow is identical.
template <typename T, int METHOD, int GA, int L, int K1, int M1, int KT = 1, int NW = 1, int OPT = 0>
struct Seip {
    // OPT bit 0 ("plain"): no seasonal forcing, no seasonal vaccination reset, no introduced strains, no recorded schedules,
    // adaptive steps, no discontinuity points, dose splines of at most two knots -- as compile-time facts (the entry point
    // picks the variant when the call is that): their fields and branches leave the right-hand side and the stepping loop
    // (the D = 960 kernel sat at its register line: docs/perf-log.md, "SEIP plain instances")
    static constexpr bool PLAIN = (OPT & 1) != 0;
    // OPT bit 1 (test-only instances): the step controller in the oracle's arithmetic -- IEEE division, powf (stepper.hpp)
    static constexpr bool STRICT_CONTROL = (OPT & 2) != 0;
    // ... and at most two knots per dose spline (the rows hold four: knots beyond the model's own are +inf with coefficient
    // 0, two truncated-power terms that add exactly zero to every evaluation)
    static constexpr int NKC = PLAIN ? 2 : 4;
    static constexpr int H = 1 << L, G = GA * H * KT, K = K1 - 1;
    // one tier per wave with an (age, history) plane smaller than a wavefront: the planes of 64 / (GA H) trajectories sit side
    // by side in every wave of the group ("packed": 4 ages x 8 histories = 32 lanes, two trajectories per group of three waves)
    static constexpr bool PACKED = KT > 2 && GA * H < 64;
    static constexpr int GWL = PACKED ? GA * H : (NW > 1 ? 64 : G);    // lanes of ONE wave that belong to one trajectory
    static constexpr int TPW = PACKED ? 64 / GWL : (NW > 1 ? 1 : 64 / G);
    static_assert((NW == 1 ? G <= 64 : (PACKED || G == 64 * NW)) && L >= 1 && L <= 4 && K1 >= 1 && K1 <= 4 && M1 >= 1 && KT >= 1 &&
                  KT <= 4 && NW >= 1 && NW <= 8, "SEIP lane group");
    // KT = 3, 4: one tier per tier lane (KT = K1), and the tier lanes are whole waves (below: TIER_X)
    static_assert(KT <= 2 || (KT == K1 && NW > 1), "SEIP: more than two tier lanes only as one tier per wave");
    static constexpr int KL = (K1 + KT - 1) / KT; // tier slots per lane
    static constexpr int NS = KL * M1, NE = KL * L, NV = NS + 3 * NE;
    static constexpr int IE = NS, II = NS + NE, IC = NS + 2 * NE;
    using M = Mth<T>;
    using TB = Tab<METHOD>;
    // the age contraction and the dense output are the ones of the s/e/i/r/c kernels
    using Lanes = Solver<T, METHOD, GA, L, true, true, true, 1, 0, L>;
    using Dense = typename Lanes::Dense;
    // the state, the stage state and the seven stage derivatives in register pairs (solve_kernel.hpp: PairState); the
    // stepper's linear algebra runs on the pairs, the right-hand side below indexes elements
    using PS = PairState<T, NV>;
    typedef T V2 __attribute__((ext_vector_type(2)));
    static constexpr int NP = PS::NP;
    // ---- wave groups: which lane bits of the trajectory lie above the wavefront
    static constexpr int LOGA = GA == 1 ? 0 : GA == 2 ? 1 : GA == 4 ? 2 : GA == 8 ? 3 : GA == 16 ? 4 : GA == 32 ? 5 : 6;
    static constexpr int HB_IN = (6 - LOGA) < L ? (6 - LOGA) : L;   // immune-history bits inside a wave
    static constexpr int HB_X = L - HB_IN;                           // ... selecting the wave (low wave bits)
    static constexpr bool TIER_X = KT >= 2 && (GA * H >= 64 || PACKED); // the tier lane selects the wave (the top wave bits)
    static constexpr int NXH = 1 << HB_X;                            // waves that differ in history bits only
    static_assert(NW == 1 || NW == NXH * (TIER_X ? KT : 1), "SEIP wave group: NW = 2^(cross-wave history bits) * (tier lanes across waves)");
    // mailbox slots of one round: infectious sums [L], tier totals [KL], recovery partners [HB_X][KL], tier flow [KL],
    // seasonal fall-back of the top tier [M1 + 2 L]
    //   + the doses [KL] when the tier flow is formed by the RECEIVING wave (tiers and histories both across waves)
    static constexpr int NSLOT = L + (K1 + KT - 1) / KT * (2 + HB_X) + M1 + 2 * L + ((TIER_X && HB_X > 0) ? (K1 + KT - 1) / KT : 0);
    T *xw;               // LDS mailbox [2][NW][NSLOT][64] (NW > 1)
    int wv;              // this wave's index inside the trajectory's workgroup
    mutable int xbuf;    // which half of the mailbox the next round writes
    __device__ __forceinline__ T *xslot(int buf, int w, int slot) const {
        return xw + ((buf * NW + w) * NSLOT + slot) * 64 + (threadIdx.x & 63);
    }
    // sum of v over ALL waves of the trajectory, the same bits in every wave (fixed order); one barrier
    __device__ __forceinline__ T wg_sum(T v) const {
        if constexpr (NW == 1) return v;
        else {
            const int b = xbuf;
            xbuf ^= 1;
            *xslot(b, wv, 0) = v;
            __syncthreads();
            T t = *xslot(b, 0, 0);
#pragma unroll
            for (int w = 1; w < NW; ++w) t += *xslot(b, w, 0);
            return t;
        }
    }
    // any lane of any wave of the trajectory has `flag` set
    __device__ __forceinline__ bool wg_any(bool flag) const {
        const int grp_in_wave = GWL >= 64 ? 0 : (int)(threadIdx.x & 63) / GWL;
        const unsigned long long mask = (GWL >= 64 ? ~0ull : ((1ull << (GWL & 63)) - 1ull)) << (grp_in_wave * (GWL & 63));
        const bool mine = (__ballot(flag) & mask) != 0ull;   // this trajectory's lanes of this wave
        if constexpr (NW == 1) return mine;
        else return wg_sum(mine ? T(1) : T(0)) > T(0);
    }
    // sum over the lanes of a trajectory
    __device__ __forceinline__ T traj_sum(T v) const {
        if constexpr (NW == 1) return group_sum<G>(v);
        else return wg_sum(group_sum<GWL>(v));
    }

    T beta[L], gamma[L], sigma[L], omega[M1];
    Lanes ages;        // only Cx (pre-permuted contact row) is used
    T amp, phase, w_season, tau;
    T itime[L], iinv[L], iamp[L]; // external introductions: day, 1 / scale, pct / (scale sqrt(2 pi)) * pop * [age receives it]
    T pop;             // population of this lane's age (doses per day = nu * pop)
    const T *sus;      // LDS: sus[K1][M1][L] of this lane's history
    // ... and a copy in registers of the rows this lane multiplies by in EVERY right-hand side (its tiers x waning states x
    // strains), where the register file has the room: the LDS reads sit behind the mailbox writes of a wave group (nothing
    // hoists them), and with one wave per SIMD nobody hides their latency
    static constexpr bool CACHE_SUS = DYN_SEIP_CACHE_SUS && (NW == 2 || (KT > 2 && (NW == 3 || (NW == 6 && PLAIN && DYN_SEIP_CACHE_6)) && DYN_SEIP_CACHE_TW)) && (9 * NV + KL * (M1 * L + 12) + 70) * (int)(sizeof(T) / 4) <= 500;
    T susr[CACHE_SUS ? KL * M1 * L : 1];
    T splr[CACHE_SUS ? KL * 12 : 1];   // per slot: the cubic's 4 coefficients, 4 knots, 4 knot coefficients (0 beyond nk)
    // nu(t) of one dose tier: cubic + truncated-power terms (reference utils/splines.py:10-109 conditional_knots), from the LDS
    // table row `c` or from this lane's register copy of its slot
    __device__ __forceinline__ T dose_rate_at(int slot, const T *c, T t) const {
        T nu;
        if constexpr (CACHE_SUS) {
            const T *r = splr + slot * 12;
            nu = r[0] + t * (r[1] + t * (r[2] + t * r[3]));
#pragma unroll
            for (int n = 0; n < NKC; ++n) {
                const T lag = M::max(t - r[4 + n], T(0));
                nu += r[8 + n] * (lag * lag * lag);
            }
        } else { // LDS row: the cubic's 4 coefficients | 4 knots | 4 knot coefficients, padded beyond nk by knots never reached and zeros
            nu = c[0] + t * (c[1] + t * (c[2] + t * c[3]));
#pragma unroll
            for (int n = 0; n < NKC; ++n) {
                const T lag = M::max(t - c[4 + n], T(0));
                nu += c[8 + n] * (lag * lag * lag);
            }
        }
        return nu;
    }
    __device__ __forceinline__ T sus_at(int slot, int kc, int m, int l) const {
        if constexpr (CACHE_SUS) return susr[(slot * M1 + m) * L + l];
        else return sus[(kc * M1 + m) * L + l];
    }
    mutable const T *spl;      // LDS: spline[K1][4 + 2 nk] of this lane's age
    int spl_off;               // ... as an element offset into the workgroup's LDS (rhs)
    int nk, hist, tl; // tl: tier lane (KT = 2), 0 otherwise
    bool pad, seasonal_rt, seasonal_vax_rt, intro_rt;
    __device__ __forceinline__ bool seasonal() const { return PLAIN ? false : seasonal_rt; }
    __device__ __forceinline__ bool seasonal_vax() const { return PLAIN ? false : seasonal_vax_rt; }
    __device__ __forceinline__ bool intro() const { return PLAIN ? false : intro_rt; }

    // sum over the immune histories that live in this wave (all of them unless NW > 1 splits the history bits)
    __device__ __forceinline__ static T hist_sum(T v) {
        if constexpr (HB_IN >= 1) v += xchg_xor<GA>(v);
        if constexpr (HB_IN >= 2) v += xchg_xor<2 * GA>(v);
        if constexpr (HB_IN >= 3) v += xchg_xor<4 * GA>(v);
        if constexpr (HB_IN >= 4) v += xchg_xor<8 * GA>(v);
        return v;
    }

    // recovery of strain l: gamma_l i_{a,j,k,l} enters s_{a, j | 2^l, k, 0}
    template <int l>
    __device__ __forceinline__ void recover(const PS &y, PS &dy) const {
        if constexpr (l < L) {
            const bool has = (hist >> l) & 1;
#pragma unroll
            for (int k = 0; k < K1; ++k) {
                const T g_i = gamma[l] * y[II + k * L + l];
                const T partner = xchg_xor<(GA << l)>(g_i); // every lane takes part in the exchange
                dy[k * M1] += has ? partner + g_i : T(0);
            }
            recover<l + 1>(y, dy);
        }
    }

    __device__ __forceinline__ void rhs(T t, const PS &y, PS &dy) const {
        // The dose splines stay in LDS and are read where they are used, in EVERY evaluation: shared between the evaluations of
        // a step attempt (or hoisted out of the stepping loop) their coefficients would be dozens of values held in registers
        // next to nine copies of the state -- the two-waves-per-SIMD kernels sit at their 256-register line (D = 960: 79
        // spilled registers and 7.27 ms without this, 34 and 6.5 ms with it).  The table's OFFSET passes through an empty asm
        // statement, which common-subexpression elimination cannot see through (the pointer stays visibly LDS: ds_read).
        if constexpr (!CACHE_SUS) {
            extern __shared__ __attribute__((aligned(32))) unsigned char dyn_smem[];
            int po = spl_off;
            asm volatile("" : "+v"(po));
            spl = reinterpret_cast<const T *>(dyn_smem) + po;
        }
        if constexpr (NW > 1) rhs_wave_group(t, y, dy);
        else if constexpr (KT == 2) rhs_tier_lanes(t, y, dy);
        else rhs_one_lane(t, y, dy);
    }

    // every tier of an (age, history) pair in one lane (KT = 1, NW = 1)
    __device__ __forceinline__ void rhs_one_lane(T t, const PS &y, PS &dy) const {
        T x[L], lam[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            T a = y[II + l];
#pragma unroll
            for (int k = 1; k < K1; ++k) a += y[II + k * L + l];
            x[l] = hist_sum(a);
        }
        if (intro()) { // infectious visitors: I_b + Normal(t; time, scale) * pct * P_b (ode_model.md:176-183)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T u = (t - itime[l]) * iinv[l];
                x[l] += iamp[l] * M::exp(T(-0.5) * u * u);
            }
        }
        ages.contract(x, lam);
        T season = T(1), phi = T(0);
        if (seasonal()) season = T(1) + amp * M::sin_lib(w_season * t + phase);
        if (seasonal_vax()) {
            // sin^1000 by squaring (the oracle multiplies in the same order): 1000 = 2 * (256+128+64+32+16+4)
            const T sn = M::sin_lib(T(6.283185307179586476925286766559) * (t + tau) / T(730));
            const T u = sn * sn, u2 = u * u, u4 = u2 * u2, u8 = u4 * u4, u16 = u8 * u8, u32 = u16 * u16,
                    u64 = u32 * u32, u128 = u64 * u64, u256 = u128 * u128;
            phi = ((((u256 * u128) * u64) * u32) * u16) * u4;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) lam[l] = (beta[l] * season) * lam[l];

        // share of each tier's susceptibles vaccinated per day
        T rate[K1];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const T nu = dose_rate_at(k, spl + k * kSplRow, t);
            T tot = y[k * M1];
#pragma unroll
            for (int m = 1; m < M1; ++m) tot += y[k * M1 + m];
            tot = hist_sum(tot);
            const T doses = M::max(nu, T(0)) * pop;
            const T share = doses * M::recip(tot > T(0) ? tot : T(1));
            rate[k] = tot > T(0) ? (doses < tot ? share : T(1)) : T(0);
        }

#pragma unroll
        for (int v = 0; v < NS; ++v) dy[v] = T(0);
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            T inflow[L];
#pragma unroll
            for (int l = 0; l < L; ++l) inflow[l] = T(0);
#pragma unroll
            for (int m = 0; m < M1; ++m) {
                const T S = y[k * M1 + m];
                T out = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T f = (lam[l] * sus_at(k, k, m, l)) * S;
                    inflow[l] += f;
                    out += f;
                }
                dy[k * M1 + m] -= out;
                if (m + 1 < M1) { // waning; the last state keeps its people
                    const T wn = omega[m] * S;
                    dy[k * M1 + m] -= wn;
                    dy[k * M1 + m + 1] += wn;
                }
                if (!(k == K && m == 0)) { // vaccination: up one tier, freshest state; top tier: refreshed in place
                    const T v = rate[k] * S;
                    dy[k * M1 + m] -= v;
                    dy[(k < K ? k + 1 : K) * M1] += v;
                }
                if (k == K && K > 0) { // seasonal vaccination: the top tier falls back one
                    const T f = phi * S;
                    dy[k * M1 + m] -= f;
                    dy[(K > 0 ? K - 1 : 0) * M1 + m] += f;
                }
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int q = k * L + l;
                const T s_e = sigma[l] * y[IE + q], g_i = gamma[l] * y[II + q];
                dy[IE + q] = inflow[l] - s_e;
                dy[II + q] = s_e - g_i;
                dy[IC + q] = inflow[l];
            }
        }
        recover<0>(y, dy);
        if constexpr (K > 0) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T fe = phi * y[IE + K * L + l], fi = phi * y[II + K * L + l];
                dy[IE + K * L + l] -= fe;
                dy[II + K * L + l] -= fi;
                dy[IE + (K - 1) * L + l] += fe;
                dy[II + (K - 1) * L + l] += fi;
            }
        }
    }

    // ---- the same right-hand side with the tiers dealt over two lanes (KT = 2): slot s of this lane is tier 2 s + tl
    __device__ __forceinline__ void rhs_tier_lanes(T t, const PS &y, PS &dy) const {
        constexpr int TB_ = GA * H; // lane distance to the tier partner
        T x[L], lam[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            T a = y[II + l];
#pragma unroll
            for (int sl = 1; sl < KL; ++sl) a += y[II + sl * L + l];
            a = hist_sum(a);
            x[l] = a + xchg_xor<TB_>(a);
        }
        if (intro()) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T u = (t - itime[l]) * iinv[l];
                x[l] += iamp[l] * M::exp(T(-0.5) * u * u);
            }
        }
        ages.contract(x, lam);
        T season = T(1), phi = T(0);
        if (seasonal()) season = T(1) + amp * M::sin_lib(w_season * t + phase);
        if (seasonal_vax()) {
            const T sn = M::sin_lib(T(6.283185307179586476925286766559) * (t + tau) / T(730));
            const T u = sn * sn, u2 = u * u, u4 = u2 * u2, u8 = u4 * u4, u16 = u8 * u8, u32 = u16 * u16,
                    u64 = u32 * u32, u128 = u64 * u64, u256 = u128 * u128;
            phi = ((((u256 * u128) * u64) * u32) * u16) * u4;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) lam[l] = (beta[l] * season) * lam[l];
#pragma unroll
        for (int v = 0; v < NS; ++v) dy[v] = T(0);
        T send[KL];
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            const int k = sl * 2 + tl;          // this slot's tier
            const bool live = k < K1, top = k == K;
            const int kc = live ? k : K;        // padded slots hold nobody: any valid table row will do
            const T nu = dose_rate_at(sl, spl + kc * kSplRow, t);
            T tot = y[sl * M1];
#pragma unroll
            for (int m = 1; m < M1; ++m) tot += y[sl * M1 + m];
            tot = hist_sum(tot);
            const T doses = M::max(nu, T(0)) * pop;
            const T share = doses * M::recip(tot > T(0) ? tot : T(1));
            const T rate = (live && tot > T(0)) ? (doses < tot ? share : T(1)) : T(0);
            T inflow[L], moved = T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) inflow[l] = T(0);
#pragma unroll
            for (int m = 0; m < M1; ++m) {
                const T S = y[sl * M1 + m];
                T out = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T f = (lam[l] * sus_at(sl, kc, m, l)) * S;
                    inflow[l] += f;
                    out += f;
                }
                dy[sl * M1 + m] -= out;
                if (m + 1 < M1) {
                    const T wn = omega[m] * S;
                    dy[sl * M1 + m] -= wn;
                    dy[sl * M1 + m + 1] += wn;
                }
                const T v = (top && m == 0) ? T(0) : rate * S; // the freshest state of the top tier stays
                dy[sl * M1 + m] -= v;
                moved += v;
            }
            dy[sl * M1] += top ? moved : T(0);  // top tier: refreshed in place
            send[sl] = top ? T(0) : moved;      // everyone else: up one tier = over to the partner lane
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int q = sl * L + l;
                const T s_e = sigma[l] * y[IE + q], g_i = gamma[l] * y[II + q];
                dy[IE + q] = inflow[l] - s_e;
                dy[II + q] = s_e - g_i;
                dy[IC + q] = inflow[l];
            }
        }
        // tier 2 s (lane 0) -> 2 s + 1 (lane 1, same slot); tier 2 s + 1 (lane 1) -> 2 s + 2 (lane 0, next slot)
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            const T got = xchg_xor<TB_>(send[sl]);
            dy[sl * M1] += tl ? got : T(0);
            if (sl + 1 < KL) dy[(sl + 1) * M1] += tl ? T(0) : got;
        }
        recover_slots<0>(y, dy);
        if constexpr (K > 0) {
            if (seasonal_vax()) { // wave-uniform: the top tier falls back one -- to the partner lane
                constexpr int sK = K / 2, tK = K % 2, sD = (K - 1) / 2;
                const bool holder = tl == tK;
#pragma unroll
                for (int m = 0; m < M1; ++m) {
                    const T f = holder ? phi * y[sK * M1 + m] : T(0);
                    dy[sK * M1 + m] -= f; // the partner subtracts 0 from whatever it keeps in this slot
                    const T got = xchg_xor<TB_>(f);
                    dy[sD * M1 + m] += holder ? T(0) : got;
                }
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T fe = holder ? phi * y[IE + sK * L + l] : T(0), fi = holder ? phi * y[II + sK * L + l] : T(0);
                    dy[IE + sK * L + l] -= fe;
                    dy[II + sK * L + l] -= fi;
                    const T ge = xchg_xor<TB_>(fe), gi = xchg_xor<TB_>(fi);
                    dy[IE + sD * L + l] += holder ? T(0) : ge;
                    dy[II + sD * L + l] += holder ? T(0) : gi;
                }
            }
        }
    }

    // ---- the right-hand side for a trajectory spread over NW waves (see the struct comment).  Slot sl of a lane is tier
    // sl * KT + tl.  Round 1 of the mailbox carries everything that depends on the state alone: the infectious sums, the
    // tier totals (when histories cross waves), the recovery flows of the strains whose history bit selects the wave, the
    // seasonal fall-back of the top tier, and -- when the tier totals are complete inside a wave -- the vaccination flow
    // to the tier partner; otherwise that flow goes in a second round once the totals are known.
    __device__ __forceinline__ void rhs_wave_group(T t, const PS &y, PS &dy) const {
        constexpr int S_TOT = L, S_REC = L + KL, S_SEND = L + KL * (1 + HB_X), S_SV = S_SEND + KL, S_DOSE = S_SV + M1 + 2 * L;
        (void)S_DOSE;
        constexpr int sK = K / KT, tK = K % KT, sD = K > 0 ? (K - 1) / KT : 0;   // slot / tier lane of the top tier, slot below it
        const int b = xbuf;
        xbuf ^= 1;
        const int tlw = TIER_X ? (wv >> HB_X) : 0, hw = wv & (NXH - 1);
        (void)hw;
        T phi = T(0), season = T(1);
        if (seasonal()) season = T(1) + amp * M::sin_lib(w_season * t + phase);
        if (seasonal_vax()) {
            const T sn = M::sin_lib(T(6.283185307179586476925286766559) * (t + tau) / T(730));
            const T u = sn * sn, u2 = u * u, u4 = u2 * u2, u8 = u4 * u4, u16 = u8 * u8, u32 = u16 * u16,
                    u64 = u32 * u32, u128 = u64 * u64, u256 = u128 * u128;
            phi = ((((u256 * u128) * u64) * u32) * u16) * u4;
        }
        // ---- lane-local and in-wave sums
        T x[L], lam[L], totl[KL], tot[KL], dose[KL];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            T a = y[II + l];
#pragma unroll
            for (int sl = 1; sl < KL; ++sl) a += y[II + sl * L + l];
            a = hist_sum(a);
            if constexpr (KT == 2 && !TIER_X) a += xchg_xor<GA * H>(a);
            *xslot(b, wv, l) = a;
        }
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            const int k = sl * KT + tl;
            const int kc = k < K1 ? k : K;
            const T nu = dose_rate_at(sl, spl + kc * kSplRow, t);
            dose[sl] = M::max(nu, T(0)) * pop;
            T tt = y[sl * M1];
#pragma unroll
            for (int m = 1; m < M1; ++m) tt += y[sl * M1 + m];
            totl[sl] = tt;
            tot[sl] = hist_sum(tt);
            if constexpr (HB_X > 0) *xslot(b, wv, S_TOT + sl) = tot[sl];
            if constexpr (TIER_X && HB_X > 0) { // the tier above forms this slot's flow itself once the totals are in (below)
                *xslot(b, wv, S_SEND + sl) = tt;
                *xslot(b, wv, S_DOSE + sl) = dose[sl];
            }
#pragma unroll
            for (int q = 0; q < HB_X; ++q) *xslot(b, wv, S_REC + q * KL + sl) = gamma[HB_IN + q] * y[II + sl * L + HB_IN + q];
        }
        // share of a tier's susceptibles vaccinated per day, and what a slot hands to the next tier
        auto share_of = [&](int sl, T total) -> T {
            const int k = sl * KT + tl;
            const T sh = dose[sl] * M::recip(total > T(0) ? total : T(1));
            return (k < K1 && total > T(0)) ? (dose[sl] < total ? sh : T(1)) : T(0);
        };
        T rate[KL];
        if constexpr (HB_X == 0) {
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                rate[sl] = share_of(sl, tot[sl]);
                if constexpr (TIER_X) *xslot(b, wv, S_SEND + sl) = (sl * KT + tl == K) ? T(0) : rate[sl] * totl[sl];
            }
        }
        if constexpr (TIER_X && K > 0) {
            if (seasonal_vax()) { // the holder of the top tier offers what falls back one tier
                const bool holder = tl == tK;
#pragma unroll
                for (int m = 0; m < M1; ++m) *xslot(b, wv, S_SV + m) = holder ? phi * y[sK * M1 + m] : T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    *xslot(b, wv, S_SV + M1 + l) = holder ? phi * y[IE + sK * L + l] : T(0);
                    *xslot(b, wv, S_SV + M1 + L + l) = holder ? phi * y[II + sK * L + l] : T(0);
                }
            }
        }
        // ---- everything that needs no other wave goes here, between the mailbox writes and the barrier: it runs while the
        // writes land and the other waves arrive (one wave per SIMD: nobody else hides that wait)
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            T wn_prev = T(0);   // waning chain: dy_m = wn_{m-1} - wn_m, the last state keeps its people
#pragma unroll
            for (int m = 0; m < M1; ++m) {
                const T wn = m + 1 < M1 ? omega[m + 1 < M1 ? m : 0] * y[sl * M1 + m] : T(0);
                dy[sl * M1 + m] = m == 0 ? (M1 > 1 ? -wn : T(0)) : (m + 1 < M1 ? wn_prev - wn : wn_prev);
                wn_prev = wn;
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int q = sl * L + l;
                const T s_e = sigma[l] * y[IE + q], g_i = gamma[l] * y[II + q];
                dy[IE + q] = -s_e;
                dy[II + q] = s_e - g_i;
            }
        }
        recover_slots<0>(y, dy);   // recovery eta(j, l) = j | 2^l over the in-wave history bits (xor exchange)
        __syncthreads();
        // ---- collect
#pragma unroll
        for (int l = 0; l < L; ++l) {
            T a = *xslot(b, 0, l);
#pragma unroll
            for (int w = 1; w < NW; ++w) a += *xslot(b, w, l);
            x[l] = a;
        }
        if constexpr (HB_X > 0) {
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                T a = *xslot(b, tlw << HB_X, S_TOT + sl);
#pragma unroll
                for (int h = 1; h < NXH; ++h) a += *xslot(b, (tlw << HB_X) | h, S_TOT + sl);
                tot[sl] = a;
                rate[sl] = share_of(sl, a);
            }
        }
        T up[KL];   // arrivals from the tier below, per slot
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) up[sl] = T(0);
        if constexpr (KT >= 2) {
            T got[KL];
            // the tier lane below this one (cyclically): its wave, same history waves
            const int src_wave = TIER_X ? ((((tlw + KT - 1) % KT) << HB_X) | hw) : 0;
            if constexpr (TIER_X && HB_X > 0) {
                // the flow out of the tier below needs that tier's total over ALL histories, which its own waves only know
                // after the barrier too: instead of a second mailbox round (one more barrier per right-hand side) this wave
                // forms it from what the tier below posted in the first -- its dose, its lane's total and the per-wave
                // totals -- with the same operations in the same order, so the bits are the ones the sender would have sent
                const int tsrc = (tlw + KT - 1) % KT;
#pragma unroll
                for (int sl = 0; sl < KL; ++sl) {
                    T tot_src = *xslot(b, tsrc << HB_X, S_TOT + sl);
#pragma unroll
                    for (int h = 1; h < NXH; ++h) tot_src += *xslot(b, (tsrc << HB_X) | h, S_TOT + sl);
                    const T dose_src = *xslot(b, src_wave, S_DOSE + sl), totl_src = *xslot(b, src_wave, S_SEND + sl);
                    const int ksrc = sl * KT + tsrc;
                    const T sh = dose_src * M::recip(tot_src > T(0) ? tot_src : T(1));
                    const T rate_src = (ksrc < K1 && tot_src > T(0)) ? (dose_src < tot_src ? sh : T(1)) : T(0);
                    got[sl] = (ksrc == K) ? T(0) : rate_src * totl_src;
                }
            } else if constexpr (TIER_X) {
#pragma unroll
                for (int sl = 0; sl < KL; ++sl) got[sl] = *xslot(b, src_wave, S_SEND + sl);
            } else {
#pragma unroll
                for (int sl = 0; sl < KL; ++sl) got[sl] = xchg_xor<GA * H>((sl * KT + tl == K) ? T(0) : rate[sl] * totl[sl]);
            }
            // tier s KT + tl - 1 (the lane below, same slot) -> s KT + tl; the last lane's tier -> lane 0 of the next slot
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                const T same = tl ? got[sl] : T(0);
                up[sl] = sl == 0 ? same : up[sl] + same;
                if (sl + 1 < KL) up[sl + 1] = tl ? T(0) : got[sl];
            }
        } else {
#pragma unroll
            for (int sl = 0; sl + 1 < KL; ++sl) up[sl + 1] = rate[sl] * totl[sl];
        }
        if (intro()) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T u = (t - itime[l]) * iinv[l];
                x[l] += iamp[l] * M::exp(T(-0.5) * u * u);
            }
        }
        ages.contract(x, lam);
#pragma unroll
        for (int l = 0; l < L; ++l) lam[l] = (beta[l] * season) * lam[l];
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            const int k = sl * KT + tl;
            const bool top = k == K;
            const int kc = k < K1 ? k : K;
            // (sums start from their first term: `x = 0; x += a` is an add the compiler must keep, -0 + 0 is not -0)
            T inflow[L], moved = T(0);
#pragma unroll
            for (int m = 0; m < M1; ++m) {
                const T S = y[sl * M1 + m];
                T out = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const T f = (lam[l] * sus_at(sl, kc, m, l)) * S;
                    inflow[l] = m == 0 ? f : inflow[l] + f;
                    out = l == 0 ? f : out + f;
                }
                dy[sl * M1 + m] -= out;
                const T v = (top && m == 0) ? T(0) : rate[sl] * S; // the freshest state of the top tier stays
                dy[sl * M1 + m] -= v;
                moved = m == 0 ? v : moved + v;
            }
            dy[sl * M1] += (top ? moved : T(0)) + up[sl];   // top tier: refreshed in place; everyone: arrivals from below
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int q = sl * L + l;
                dy[IE + q] += inflow[l];
                dy[IC + q] = inflow[l];
            }
        }
        // ---- recovery over the history bits that live in other waves: partners' gamma I from the mailbox
#pragma unroll
        for (int q = 0; q < HB_X; ++q) {
            const bool has = (hist >> (HB_IN + q)) & 1;
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                const T g_i = gamma[HB_IN + q] * y[II + sl * L + HB_IN + q];
                const T partner = *xslot(b, wv ^ (1 << q), S_REC + q * KL + sl);
                dy[sl * M1] += has ? partner + g_i : T(0);
            }
        }
        // ---- seasonal vaccination: the top tier falls back one
        if constexpr (K > 0) {
            if (seasonal_vax()) {
                if constexpr (KT == 1) {
#pragma unroll
                    for (int m = 0; m < M1; ++m) {
                        const T f = phi * y[K * M1 + m];
                        dy[K * M1 + m] -= f;
                        dy[(K - 1) * M1 + m] += f;
                    }
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const T fe = phi * y[IE + K * L + l], fi = phi * y[II + K * L + l];
                        dy[IE + K * L + l] -= fe;
                        dy[II + K * L + l] -= fi;
                        dy[IE + (K - 1) * L + l] += fe;
                        dy[II + (K - 1) * L + l] += fi;
                    }
                } else {
                    // the lane of tier K offers, the lane of tier K - 1 takes (with two tier lanes: the other one)
                    constexpr int tR = (K - 1) % KT;
                    const bool holder = tl == tK, taker = tl == tR;
                    const int holder_wave = TIER_X ? ((tK << HB_X) | hw) : 0;
                    (void)holder_wave;
#pragma unroll
                    for (int m = 0; m < M1; ++m) {
                        const T f = holder ? phi * y[sK * M1 + m] : T(0);
                        dy[sK * M1 + m] -= f;
                        T got;
                        if constexpr (TIER_X) got = *xslot(b, holder_wave, S_SV + m);
                        else got = xchg_xor<GA * H>(f);
                        dy[sD * M1 + m] += taker ? got : T(0);
                    }
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const T fe = holder ? phi * y[IE + sK * L + l] : T(0), fi = holder ? phi * y[II + sK * L + l] : T(0);
                        dy[IE + sK * L + l] -= fe;
                        dy[II + sK * L + l] -= fi;
                        T ge, gi;
                        if constexpr (TIER_X) {
                            ge = *xslot(b, holder_wave, S_SV + M1 + l);
                            gi = *xslot(b, holder_wave, S_SV + M1 + L + l);
                        } else {
                            ge = xchg_xor<GA * H>(fe);
                            gi = xchg_xor<GA * H>(fi);
                        }
                        dy[IE + sD * L + l] += taker ? ge : T(0);
                        dy[II + sD * L + l] += taker ? gi : T(0);
                    }
                }
            }
        }
    }

    // recovery for the KL local slots (the same exchange over the history bits as recover<>)
    template <int l>
    __device__ __forceinline__ void recover_slots(const PS &y, PS &dy) const {
        if constexpr (l < HB_IN) {   // (= L unless a wave group splits the history bits: rhs_wave_group serves the rest)
            const bool has = (hist >> l) & 1;
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                const T g_i = gamma[l] * y[II + sl * L + l];
                const T partner = xchg_xor<(GA << l)>(g_i);
                dy[sl * M1] += has ? partner + g_i : T(0);
            }
            recover_slots<l + 1>(y, dy);
        }
    }

    template <int FIRST, int CNT>
    __device__ __forceinline__ static void save_block(const PS &o, T *dst, bool vec_ok) {
        T v[CNT];
#pragma unroll
        for (int q = 0; q < CNT; ++q) v[q] = o[FIRST + q];
        store_run<T, CNT>(dst, v, vec_ok);
    }

    // KT = 2: one block per live tier slot and compartment
    template <int SL>
    __device__ __forceinline__ static void save_slots(const KArgs<T> &ka, const PS &o, T *row, int g, int tl, bool vec_ok) {
        if constexpr (SL < KL) {
            const int kt = SL * KT + tl;
            if (kt < K1) {
                const int gk = g * K1 + kt;
                if (ka.save_off[0] >= 0) save_block<SL * M1, M1>(o, row + ka.save_off[0] + gk * M1, vec_ok);
                if (ka.save_off[1] >= 0) save_block<IE + SL * L, L>(o, row + ka.save_off[1] + gk * L, vec_ok);
                if (ka.save_off[2] >= 0) save_block<II + SL * L, L>(o, row + ka.save_off[2] + gk * L, vec_ok);
                if (ka.save_off[4] >= 0) save_block<IC + SL * L, L>(o, row + ka.save_off[4] + gk * L, vec_ok);
            }
            save_slots<SL + 1>(ka, o, row, g, tl, vec_ok);
        }
    }

    // ================================================================ the family interface of Stepper<F> (stepper.hpp)
    using Scalar = T;
    using State = PS;
    static constexpr int NC = 1, NDIR = 0, SU = 1;
    static constexpr int GW = GWL;                  // lanes of one wave that hold one trajectory
    static constexpr bool PRESCALE = false, PC = false, LEAN = false, FUSED = false;
    static constexpr size_t kTailOffset = 0;
    static constexpr bool ADAPTIVE_NO_JUMPS = PLAIN;
    static constexpr bool ROOTLESS_NORM = false;    // (kinks: the dose cap, the seasonal reset)
    static constexpr bool PULLS = false;            // a static grid: the waves of a group meet at barriers, slot for slot
    static constexpr bool REPLAYS = !PLAIN;         // recorded step schedules (KArgs::sched_*: dyn_solve_batch_record / _replay)
    static constexpr bool IDLE_SLOTS_LOAD = true;   // a slot beyond the batch keeps in step on the last trajectory's data
    static constexpr int SUSN = H * K1 * M1 * L;    // susceptibility table of one trajectory
    static constexpr int kSplRow = 12;              // one dose spline in LDS: 4 cubic coefficients | 4 knots | 4 knot coefficients
    int a, g, tidx;       // age lane; (age, history) group in memory order; index among the G lanes of the trajectory
    bool writer, leader;  // the lane stores rows (not a pad lane) / reports the trajectory's status

    // lane indices and what every trajectory of the launch shares; -> the lane's trajectory slot in the workgroup
    __device__ __forceinline__ int init(const KArgs<T> &ka, int lane) {
        // position inside the trajectory's lane group: the lane itself, or (NW > 1) wave * 64 + lane of the workgroup
        const int wtid = NW > 1 ? (int)threadIdx.x : lane;
        const int tlane = PACKED ? lane : wtid;
        // (when whole waves are the tier lanes the tier is the same for every lane of the wave: taken from a scalar register,
        // so that everything that depends on it -- "is this the top tier", which table row, which mailbox wave -- is scalar
        // arithmetic and scalar branches instead of per-lane selects)
        const int wave_idx = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        a = tlane % GA;
        hist = (tlane / GA) % H;
        tl = PACKED ? wave_idx : (TIER_X ? (wave_idx >> HB_X) % KT : (tlane / (GA * H)) % KT);
        tidx = PACKED ? tl * GWL + lane % GWL : tlane % G;
        const int A = ka.A;
        wv = NW > 1 ? wave_idx : 0;
        xbuf = 0;
        pad = a >= A;
        nk = ka.n_vax_knots;
        seasonal_rt = ka.seasonal != 0;
        seasonal_vax_rt = ka.seasonal_vax != 0;
        intro_rt = ka.has_intro != 0;
        const int aa = pad ? 0 : a;
        g = aa * H + hist;
        writer = !pad;
        leader = a == 0 && hist == 0 && tl == 0;
#pragma unroll
        for (int k = 0; k < GA; ++k) {
            const int b = a ^ k;
            ages.Cx[k] = (!pad && b < A) ? ka.contact[aa * A + b] : T(0);
        }
        return PACKED ? lane / GWL : (NW > 1 ? 0 : lane / G);
    }
    __device__ __forceinline__ static int state_dim(const KArgs<T> &ka) {
        return ka.A * H * K1 * M1 + 3 * ka.A * H * K1 * L;
    }

    // LDS behind the save grid and the discontinuity points: per trajectory slot the susceptibility table and the dose splines,
    // then (replay) every slot's recorded schedule, then the mailbox of a wave group
    struct Tables {
        T *tab;      // this slot's tables
        T *sch;      // this slot's schedule: the (t_prev, t_next) pairs it must take (a global load inside the stepping loop
        int n_sch;   // would wait for every store issued before it: the loads and stores share vmcnt)
    };
    __device__ __forceinline__ void carve(const KArgs<T> &ka, Tables &tb, T *, T *free_lds, int, int grp, int, int) {
        const int spln = ka.A * K1 * kSplRow;
        const bool replay = ka.sched_in != nullptr;
        tb.tab = free_lds + grp * (SUSN + spln);
        T *const sch_base = free_lds + TPW * (SUSN + spln);
        tb.sch = sch_base + grp * (2 * ka.sched_cap);
        tb.n_sch = 0;
        xw = sch_base + (replay ? TPW * 2 * ka.sched_cap : 0);   // (16-byte aligned by construction of the table sizes or not: 4-byte words)
    }

    // parameters, tables (into LDS, by the trajectory's lanes) and this lane's chunk of every compartment
    template <typename KA>
    __device__ __forceinline__ void load_trajectory(const KA &ka, int64_t traj, int, Tables &tb, PS (&ys)[1]) {
        PS &y = ys[0];
        const int A = ka.A;
        const int aa = pad ? 0 : a;
        const int spln = A * K1 * kSplRow;
        const T *p = ka.params + traj * ka.P;
        const T *q = p + 3 * L + M1;
#pragma unroll
        for (int l = 0; l < L; ++l) itime[l] = iinv[l] = iamp[l] = T(0);
        const T *intro_p = q;
        if (intro()) q += 3 * L;
        amp = phase = w_season = tau = T(0);
        if (seasonal()) {
            amp = q[0];
            phase = q[1];
            w_season = T(6.283185307179586476925286766559) / q[2];
            q += 3;
        }
        if (seasonal_vax()) {
            tau = q[0];
            q += 1;
        }
        pop = pad ? T(0) : q[aa];
        q += A;
        if (intro()) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const T scale = intro_p[L + l];
                const bool here = !pad && ((ka.intro_mask[l] >> aa) & 1ull);
                itime[l] = intro_p[l];
                iinv[l] = T(1) / scale;
                iamp[l] = here ? intro_p[2 * L + l] / (scale * T(2.5066282746310002)) * pop : T(0);
            }
        }
        T *const tab = tb.tab;
        for (int n = tidx; n < SUSN; n += G) tab[n] = q[n];
        {   // the dose splines, one row of kSplRow per (age, tier): the parameter row's 4 + 2 nk values spread to fixed places, so
            // that the right-hand side reads a row with constant offsets and no "is this knot in use" selects
            const T *const src = q + SUSN;
            const int rowlen = 4 + 2 * nk;
            for (int n = tidx; n < spln; n += G) {
                const int row = n / kSplRow, jj = n % kSplRow;
                T v;
                if (jj < 4) v = src[row * rowlen + jj];
                else if (jj < 8) v = (jj - 4 < nk) ? src[row * rowlen + jj] : M::inf();            // a knot that is never reached
                else v = (jj - 8 < nk) ? src[row * rowlen + 4 + nk + (jj - 8)] : T(0);
                tab[SUSN + n] = v;
            }
        }
        if (ka.sched_in != nullptr) { // replay: the leader's schedule, staged in LDS
            const int64_t lead = ka.sched_leader ? ka.sched_leader[traj] : traj;
            tb.n_sch = ka.sched_n_in[lead];
            const T *src = ka.sched_in + lead * (int64_t)(2 * ka.sched_cap);
            for (int n = tidx; n < 2 * (tb.n_sch > 0 ? tb.n_sch : 0); n += G) tb.sch[n] = src[n];
        }
        __syncthreads();
        sus = tab + hist * (K1 * M1 * L);
        spl = tab + SUSN + aa * K1 * kSplRow;
        {
            extern __shared__ __attribute__((aligned(32))) unsigned char dyn_smem[];
            spl_off = (int)(spl - reinterpret_cast<const T *>(dyn_smem));
        }
        if constexpr (CACHE_SUS) {
#pragma unroll
            for (int sl = 0; sl < KL; ++sl) {
                const int kt = sl * KT + tl, kc = kt < K1 ? kt : K;   // padded slots hold nobody: any valid row will do
#pragma unroll
                for (int q = 0; q < M1 * L; ++q) susr[sl * M1 * L + q] = sus[kc * M1 * L + q];
                const T *c = spl + kc * kSplRow;
#pragma unroll
                for (int n = 0; n < kSplRow; ++n) splr[sl * 12 + n] = c[n];
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            beta[l] = p[l];
            gamma[l] = p[L + l];
            sigma[l] = p[2 * L + l];
        }
#pragma unroll
        for (int m = 0; m < M1; ++m) omega[m] = p[3 * L + m];

        // memory layout: every (age, history) group holds K1 tiers; a lane holds them all (KT = 1) or tiers tl, tl + KT, ...
        const int offE = A * H * K1 * M1, nE = A * H * K1 * L;
        const int D = offE + 3 * nE;
        const T *src = ka.y0 + (ka.y0_batched ? traj * D : 0);
        if constexpr (KT == 1) { // a lane's tiers are contiguous in memory
#pragma unroll
            for (int v = 0; v < NS; ++v) y[v] = pad ? T(0) : src[g * NS + v];
#pragma unroll
            for (int v = 0; v < NE; ++v) {
                y[IE + v] = pad ? T(0) : src[offE + g * NE + v];
                y[II + v] = pad ? T(0) : src[offE + nE + g * NE + v];
                y[IC + v] = pad ? T(0) : src[offE + 2 * nE + g * NE + v];
            }
        } else
#pragma unroll
        for (int sl = 0; sl < KL; ++sl) {
            const int kt = sl * KT + tl;           // tier of this slot
            const bool live = !pad && kt < K1;
            const int gk = g * K1 + (kt < K1 ? kt : 0);
#pragma unroll
            for (int m = 0; m < M1; ++m) y[sl * M1 + m] = live ? src[gk * M1 + m] : T(0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                y[IE + sl * L + l] = live ? src[offE + gk * L + l] : T(0);
                y[II + sl * L + l] = live ? src[offE + nE + gk * L + l] : T(0);
                y[IC + sl * L + l] = live ? src[offE + 2 * nE + gk * L + l] : T(0);
            }
        }
    }
    __device__ __forceinline__ static void begin_attempt(const PS (&)[1]) {}   // (the population of an age is a parameter here)
    __device__ __forceinline__ void rhs(T t, const PS (&y)[1], PS (&dy)[1]) const { rhs(t, y[0], dy[0]); }
    __device__ __forceinline__ bool start_ok(bool lane_ok, int) const { return !wg_any(!lane_ok); } // no lane of the trajectory saw NaN / inf
    __device__ __forceinline__ static T weigh(int, T x) { return x; }   // (no element is held by more than one lane)
    __device__ __forceinline__ static void count_once(int, V2 &) {}

    // rows: weighted sums of the stages (the polynomial form does not pay at the 1.3 rows per accepted step of these models)
    struct Output {
        T *out_traj;
    };
    template <typename KA>
    __device__ __forceinline__ static void begin_output(const KA &ka, Output &o, int64_t traj, int n_save) {
        o.out_traj = ka.out + traj * (int64_t)n_save * ka.d_saved;
    }
    __device__ __forceinline__ static void dense_begin(const Tables &, T, const PS (&)[1], const PS (&)[1], PS (&)[7][1]) {}
    __device__ __forceinline__ void emit_row(const KArgs<T> &ka, const Tables &, Output &out, T theta, T dt, const PS (&y)[1],
                                             const PS (&yt)[1], const PS (&k)[7][1], int save_idx, bool on, bool vec_ok) const {
        Dense dn;
        Lanes::dense_prepare(theta, dn);
        if (on) {
            T *row = out.out_traj + (int64_t)save_idx * ka.d_saved;
            PS o;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj)
                o.p[jj] = Lanes::template dense_eval<V2>(dn, dt, y[0].p[jj], yt[0].p[jj], k[0][0].p[jj], k[1][0].p[jj], k[2][0].p[jj],
                                                         k[3][0].p[jj], k[4][0].p[jj], k[5][0].p[jj], k[6][0].p[jj]);
            if constexpr (KT == 1) {
                if (ka.save_off[0] >= 0) save_block<0, NS>(o, row + ka.save_off[0] + g * NS, vec_ok);
                if (ka.save_off[1] >= 0) save_block<IE, NE>(o, row + ka.save_off[1] + g * NE, vec_ok);
                if (ka.save_off[2] >= 0) save_block<II, NE>(o, row + ka.save_off[2] + g * NE, vec_ok);
                if (ka.save_off[4] >= 0) save_block<IC, NE>(o, row + ka.save_off[4] + g * NE, vec_ok);
            } else {
                save_slots<0>(ka, o, row, g, tl, vec_ok);
            }
        }
    }
    __device__ __forceinline__ void fill_row(const KArgs<T> &ka, const Output &out, int save_idx, T v) const { // a row never reached
        T *row = out.out_traj + (int64_t)save_idx * ka.d_saved;
        if constexpr (KT == 1) {
            if (ka.save_off[0] >= 0)
                for (int e = 0; e < NS; ++e) row[ka.save_off[0] + g * NS + e] = v;
            for (int c = 1; c <= 4; ++c)
                if (c != 3 && ka.save_off[c] >= 0)
                    for (int e = 0; e < NE; ++e) row[ka.save_off[c] + g * NE + e] = v;
        } else {
            for (int sl = 0; sl < KL; ++sl) {
                const int kt = sl * KT + tl;
                if (kt >= K1) continue;
                const int gk = g * K1 + kt;
                if (ka.save_off[0] >= 0)
                    for (int m = 0; m < M1; ++m) row[ka.save_off[0] + gk * M1 + m] = v;
                for (int c = 1; c <= 4; ++c)
                    if (c != 3 && ka.save_off[c] >= 0)
                        for (int l = 0; l < L; ++l) row[ka.save_off[c] + gk * L + l] = v;
            }
        }
    }
    // record: the accepted step into KArgs::sched_out (dyn_solve_batch_record)
    __device__ __forceinline__ void record_step(const KArgs<T> &ka, int64_t traj, int32_t n_acc, T tprev, T tnext, bool writer_now) const {
        if (__builtin_expect(ka.sched_out != nullptr, 0) && writer_now && leader && n_acc < ka.sched_cap) {
            T *rec = ka.sched_out + (traj * (int64_t)ka.sched_cap + n_acc) * 2;
            rec[0] = tprev;
            rec[1] = tnext;
        }
    }
};

template <typename T, int METHOD, int GA, int L, int K1, int M1, int KT = 1, int OPT = 0>
__global__ void __launch_bounds__(64) seip_kernel(const KArgs<T> ka) {
    Stepper<Seip<T, METHOD, GA, L, K1, M1, KT, 1, OPT>>::run(ka);
}

// small per-lane states (tier lanes, <= 20 values): cap the registers at 256 so that two waves share a SIMD
template <typename T, int METHOD, int GA, int L, int K1, int M1, int KT = 1, int OPT = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) seip_kernel_two_waves(const KArgs<T> ka) {
    Stepper<Seip<T, METHOD, GA, L, K1, M1, KT, 1, OPT>>::run(ka);
}

// wave groups: one trajectory per workgroup of NW waves (lane groups of 128 or 256)
template <typename T, int METHOD, int GA, int L, int K1, int M1, int KT, int NW, int OPT = 0>
__global__ void __launch_bounds__(64 * NW) seip_kernel_wave_group(const KArgs<T> ka) {
    Stepper<Seip<T, METHOD, GA, L, K1, M1, KT, NW, OPT>>::run(ka);
}

template <typename T, int METHOD, int GA, int L, int K1, int M1, int KT = 1, int NW = 1, int OPT = 0>
hipError_t launch_seip(const KArgs<T> &ka, hipStream_t stream) {
    using Shape = Seip<T, METHOD, GA, L, K1, M1, KT, NW, OPT>;
    constexpr int TPW = Shape::TPW;
    const int64_t blocks = (ka.B + TPW - 1) / TPW;
    if (blocks <= 0) return hipSuccess;
    if ((OPT & 1) != 0 && (ka.seasonal || ka.seasonal_vax || ka.has_intro || ka.sched_in || ka.sched_out || ka.n_jump > 0 || ka.constant_dt > T(0) || ka.n_vax_knots > 2))
        return hipErrorInvalidValue; // enqueue() never asks a plain instance for any of these
    const size_t per_traj = (size_t)(1 << L) * K1 * M1 * L + (size_t)ka.A * K1 * Shape::kSplRow;
    const size_t lds = ((size_t)ka.n_save + (ka.n_jump > 0 ? kMaxJumps : 0) + TPW * per_traj +
                        (ka.sched_in != nullptr ? (size_t)TPW * 2 * ka.sched_cap : 0) +
                        (NW > 1 ? (size_t)2 * NW * Shape::NSLOT * 64 : 0)) * sizeof(T);
    if constexpr (NW > 1) {
        // one or two workgroups per CU (one wave per SIMD at ~400 registers): beyond the default 64 KB of dynamic LDS the
        // kernel attribute has to allow it (160 KB per CU)
        if (lds > 65536) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&seip_kernel_wave_group<T, METHOD, GA, L, K1, M1, KT, NW, OPT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((seip_kernel_wave_group<T, METHOD, GA, L, K1, M1, KT, NW, OPT>), dim3((unsigned)blocks), dim3(64 * NW), lds, stream, ka);
    }
    else if constexpr (sizeof(T) == 4 && Shape::NV <= 20)
        hipLaunchKernelGGL((seip_kernel_two_waves<T, METHOD, GA, L, K1, M1, KT, OPT>), dim3((unsigned)blocks), dim3(64), lds, stream, ka);
    else
        hipLaunchKernelGGL((seip_kernel<T, METHOD, GA, L, K1, M1, KT, OPT>), dim3((unsigned)blocks), dim3(64), lds, stream, ka);
    return hipGetLastError();
}

} // namespace dyn
