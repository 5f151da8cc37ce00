// stepper.hpp -- the ONE stepping loop of the library: explicit Runge-Kutta pairs (Tsit5 / Dopri5, Tab<>) with FSAL, the
// I-controller and the Hairer-Norsett-Wanner starting step (Control<>), diffeqsolve's clip-to-end and discontinuity points,
// SaveAt(ts) by dense output, work pulling over lane-group slots -- for every right-hand side.  What diffrax's `diffeqsolve`
// is to the reference (src/dynode/simulation/odes.py:133-144: one solve call whatever the model), `Stepper<F>::run` is here:
// a model family F (solve_kernel.hpp `Solver`: the s/e/i/r/c compartment models; seip_kernel.hpp `Seip`: the immune-history
// model) supplies the lane mapping, the per-trajectory data, the right-hand side and the row writer; the loop is this file.
//
// The family interface (all `__device__ __forceinline__`; see the two implementations):
//   types / constants   Scalar, TB (tableau), State (PairState<T, NV>), NV, NP, NC (planes: 1 + tangent directions), ND,
//                       G (lanes of a wave that hold one trajectory), TPW (trajectory slots per workgroup), SU (save times
//                       per save round), PRESCALE / PC / LEAN / FUSED (see Solver), PULLS (slots may draw further
//                       trajectories from KArgs::work), REPLAYS (recorded step schedules, KArgs::sched_*)
//   F L; L.init(ka, lane)                       lane indices, shared model data (contact row, switches)
//   L.carve(ka, lds, ...)                       the family's LDS tables behind the save grid and the discontinuity points
//   L.load_trajectory(kc, traj, ..., y)         parameters, tables and initial state (+ seeds) of trajectory `traj`
//   L.begin_attempt(y); L.rhs(t, y, k)          per-attempt data from the step's starting state; k = f(t, y) (k = dt f under PRESCALE), all planes
//   L.traj_sum(x), L.start_ok(lane_ok)          reductions over the lanes (and waves) of the trajectory
//   L.dense_begin / L.write_row / L.fill_row    dense output of an accepted step, one saved row, a row never reached
#pragma once

namespace dyn {

template <class F>
struct Stepper {
    using T = typename F::Scalar;
    using M = Mth<T>;
    using TB = typename F::TB;
    using State = typename F::State;
    using V2 = typename F::V2;
    static constexpr int NV = F::NV, NP = F::NP, NC = F::NC, ND = F::NDIR, GW = F::GW, TPW = F::TPW, SU = F::SU;
    static constexpr bool PRESCALE = F::PRESCALE, PC = F::PC, LEAN = F::LEAN, FUSED = F::FUSED;

    // ---- the solve.  A lane group is a SLOT that integrates one trajectory after the other:
    //   static launches (KArgs::work == nullptr): grid = ceil(B 2^rep / TPW) waves, slot i takes trajectory i and stops;
    //   work-pulling launches (KArgs::work != nullptr; the host sizes the grid to the waves the chip can hold at once):
    //     slot i starts with trajectory i and, whenever its trajectory finishes, draws the next index from a device
    //     counter (atomicAdd) and re-runs the prologue under its group's lanes while the other groups of the wave keep
    //     stepping.  The lane groups of a wave still step in lock-step, but none of them waits for a finished partner for
    //     longer than the rest of an iteration, and the launch ends when the queue is empty -- no max-over-groups of whole
    //     trajectories, no round structure, nothing learned in advance.  With KArgs::order the queue is that permutation
    //     (most expensive first, if the caller knows): tickets index it.
    // Every trajectory is computed from its own inputs alone, with the same instructions whatever slot runs it, so the
    // results do not depend on the assignment (tests/test_gpu_parity.py: dispatch order / batch position invariance).
    __device__ __forceinline__ static void run(const KArgs<T> &ka) {
        const int lane = threadIdx.x & 63;
        F L;
        const int grp = L.init(ka, lane);               // the family's lane indices and launch-wide model data; -> slot of this lane inside the workgroup
        const int64_t gslot = (int64_t)blockIdx.x * TPW + grp;
        const int R = 1 << ka.rep_log2;                 // replicas per trajectory (static launches only)
        const int rep = (int)(gslot & (R - 1));         // this group's replica number
        const bool pull = F::PULLS && ka.work != nullptr;
        bool writer = L.writer;                         // this lane stores rows (a pad lane of the group, a slot beyond the batch: no)

        const T rtol = ka.rtol, atol = ka.atol, t_end = ka.t1;
        const T Dn = T(L.state_dim(ka));
        const bool constant = F::ADAPTIVE_NO_JUMPS ? false : ka.constant_dt > T(0);
        const bool replay = F::REPLAYS && ka.sched_in != nullptr;   // recorded step schedules (KArgs::sched_*): launch-wide
        const bool fixed = constant || replay;                      // every step is taken as given: no error estimate, no rejection
        const int n_save = ka.n_save;
        // Small states interpolate SU save times per save round (all computed, stores predicated: instruction-level parallelism
        // for a latency-bound wave) -- while a replica has that many per step.  With eight replicas per trajectory
        // a step holds about one of a replica's save times, and a round of SU candidates computes mostly discarded rows:
        // then one row per round (cfg 2 at B = 1024, eight replicas: 0.105 -> 0.1007 ms; four replicas prefer the rounds of four: 0.1144 vs 0.1178).  Launch-wide, so the two-ahead
        // save time of the single-row rounds stays consistent.
        [[maybe_unused]] const bool multi_rows = SU > 1 && ka.rep_log2 < 3;
        // The save grid lives in LDS: a global load inside the save loop would share the
        // in-order vmcnt counter with the output stores, and waiting for it would drain every
        // store of the previous round (measured: the dominant stall of the save path).
        // LDS reads count on lgkmcnt, so stores stay fire-and-forget.
        extern __shared__ __attribute__((aligned(32))) unsigned char dyn_smem[];
        T *const ts_tab = reinterpret_cast<T *>(dyn_smem); // LDS address space: ds_read only
        for (int j = lane; j < n_save; j += 64) ts_tab[j] = ka.save_ts[j];
        // discontinuity points follow the save grid in LDS (per-group index into the table)
        T *const jt_tab = ts_tab + n_save;
        const int n_jump = F::ADAPTIVE_NO_JUMPS ? 0 : ka.n_jump;
        if (n_jump > 0) {   // one per lane, straight from the kernel-argument segment (no unrolled copy through scalar registers)
            static_assert(kMaxJumps == 64, "the discontinuity points are staged one per lane");
            jt_tab[lane] = cold_args<T>()->jump_ts[lane];
        }
        // the family's own tables follow (Solver: likelihood table, parked rates, the two-wave hand-over; Seip: susceptibility
        // tables and dose splines per trajectory slot, recorded schedules, the mailbox of a wave group)
        typename F::Tables tb;
        L.carve(ka, tb, ts_tab, jt_tab + (n_jump > 0 ? kMaxJumps : 0), lane, grp, n_save, n_jump);
        __syncthreads();
        const bool vec_ok = ka.vec_ok != 0;
        if constexpr (PC) {
            if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1) { // wave 1: the rows of wave 0's trajectories
                const int64_t t = gslot;           // (static launch, no replicas, no caller's order: enqueue() guarantees it)
                L.consume(ka, tb.hand, ts_tab, lane, t < ka.B, t < ka.B ? t : 0);
                return;
            }
        }

        // ---- the slot's trajectory: state, stage derivatives, step control
        State y[NC], yt[NC], k[7][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { // (also the pad element of every plane: zero for good -- nothing but the pairwise stepper touches it)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                y[c].p[pp] = V2{T(0), T(0)};
                yt[c].p[pp] = V2{T(0), T(0)};
#pragma unroll
                for (int q = 0; q < 7; ++q) k[q][c].p[pp] = V2{T(0), T(0)};
            }
        }
        int64_t traj = 0;
        bool live = false, done = true;      // live: a trajectory is loaded and not yet written off; done: nothing to step
        T tprev = ka.t0, tnext = ka.t0, ts_next = M::inf(), ts_next2 = M::inf(), dt_unclipped = T(0);
        int save_idx = 0, jidx = 0, waited = 0;
        [[maybe_unused]] int si = 0;         // (REPLAYS) index of the recorded step being taken
        bool at_jump = false;
        int64_t steps = 0;
        int32_t n_acc = 0, n_rej = 0, st = ST_OK;
        typename F::Output out;          // where the rows of the slot's trajectory go (and, fused likelihood, its running score)
#ifdef DYN_DIAG_ROUNDS
        int diag_iters = 0, diag_rounds = 0;
#endif
        // first assignment: slot i takes entry i of the queue
        bool need_load = false, want_ticket = false;
        [[maybe_unused]] bool idle_slot = false;
        {
            int64_t t = gslot >> ka.rep_log2;
            if (t < ka.B) {
                if (ka.order) t = ka.order[t];
                // an order that is not a permutation never makes the kernel touch memory outside the batch: the entry is skipped
                if ((uint64_t)t < (uint64_t)ka.B) {
                    traj = t;
                    need_load = true;
                } else {
                    want_ticket = pull;
                }
            }
            if constexpr (F::IDLE_SLOTS_LOAD) {
                // a family whose right-hand side meets the other waves of its workgroup at barriers and reads per-trajectory
                // tables keeps the lanes of a slot beyond the batch in step on the last trajectory's data; they write nothing
                if (!need_load) {
                    traj = ka.B - 1;
                    need_load = true;
                    idle_slot = true;
                }
            }
        }

        // ---- prologue of the slot's trajectory (stepper_prologue.inc).  A family whose slots draw trajectories (PULLS) runs it
        // inside the stepping loop, whenever a slot has drawn one; a static family once, in front of the loop -- the loop then
        // carries nothing of it (the SEIP wave groups sit at their register lines).
        if constexpr (!F::PULLS) {
            if (need_load || F::IDLE_SLOTS_LOAD) {
#define DYN_STEPPER_ARGS ka
#include "stepper_prologue.inc"
#undef DYN_STEPPER_ARGS
            }
        }
        // (a static family tests at the bottom -- while any of its trajectories has steps to take -- and writes them off behind
        // the loop: with the test at the top the exit edge would carry the whole loop state through the latch)
        if (F::PULLS || __any(live && !done))
        for (;;) {
            // A slot that needs work waits an iteration or two when another group of its wave is on the last step of ITS
            // trajectory (tnext == t_end): one pass of the prologue then serves both (the pass costs the wave about a third of
            // an iteration however many of its groups it loads).
            bool draw = false;
            if (__builtin_expect(__any(want_ticket), 0)) { // wave-uniform
                const bool closing = __any(!done && !(tnext < t_end));
                draw = want_ticket && !(closing && waited < 2);
                waited += (want_ticket && !draw) ? 1 : 0;
            }
            if (__builtin_expect(draw, 0)) { // (lane-group uniform) draw the next entry of the queue
                waited = 0;
                const auto &kc = *cold_args<T>();
                int32_t *const work = kc.work;
                const int32_t *const order = kc.order;
                const int64_t B = kc.B;
                const int64_t n_slots = (int64_t)gridDim.x * TPW;
                for (;;) {
                    long long t = 0;
                    if ((lane & (GW - 1)) == 0) {
                        t = n_slots + (long long)atomicAdd(work, 1);
                        if (t >= B) {
                            // this slot retires; the last one to retire re-arms the counters, so that the caller can hand
                            // the same two words to its next launch on the stream without clearing them
                            if (atomicAdd(work + 1, 1) == (int)(n_slots - 1)) {
                                __atomic_store_n(work, 0, __ATOMIC_RELAXED);
                                __atomic_store_n(work + 1, 0, __ATOMIC_RELAXED);
                            }
                        }
                    }
                    t = __shfl(t, lane & ~(GW - 1), 64);
                    if (t >= B) break;
                    if (order) t = order[t];
                    if ((uint64_t)t < (uint64_t)B) {
                        traj = t;
                        need_load = true;
                        break;
                    }
                }
                want_ticket = false;
            }
            if constexpr (F::PULLS) {
                if (__builtin_expect(need_load, 0)) {
                    // (the cold copy of the kernel arguments: cold_args() in solve_kernel.hpp)
#define DYN_STEPPER_ARGS (*cold_args<T>())
#include "stepper_prologue.inc"
#undef DYN_STEPPER_ARGS
                }
            }
            if constexpr (F::PULLS) {
                if (!__any(live)) break;
            }
            L.begin_attempt(y);  // what the family forms once per step attempt from the state it starts from (Solver: the age's population)

#ifdef DYN_DIAG_ROUNDS
            ++diag_iters;
#endif
            const T dt = tnext - tprev;
            // ---- stages 2..7 (k[] hold f; k[0] is FSAL)
#pragma unroll
            for (int sg = 1; sg < 7; ++sg) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) { // register pairs: v_pk_fma_f32
                        if constexpr (PRESCALE) {   // k[] hold dt f: y + sum a k, the chain starts on y
                            V2 acc = T(TB::a[sg][0]) * k[0][c].p[pp] + y[c].p[pp];
#pragma unroll
                            for (int q = 1; q < sg; ++q)
                                if (TB::a[sg][q] != 0.0) acc += T(TB::a[sg][q]) * k[q][c].p[pp];
                            yt[c].p[pp] = acc;
                        } else {
                            V2 acc = T(TB::a[sg][0]) * k[0][c].p[pp];
#pragma unroll
                            for (int q = 1; q < sg; ++q)
                                if (TB::a[sg][q] != 0.0) acc += T(TB::a[sg][q]) * k[q][c].p[pp];
                            yt[c].p[pp] = y[c].p[pp] + dt * acc;
                        }
                    }
                L.rhs(tprev + T(TB::c[sg]) * dt, yt, k[sg]);
            }
            // after stage 7, yt == y1 (a[6][:] == b) and k[6] == f(tnext, y1)

            // ---- embedded error, RMS norm over the whole (primal) state, I-controller
            bool keep = true, finite = true;
            T factor = T(1);
            if (__builtin_expect(!fixed, 1)) {
                // scaled error per element, pair by pair: only |.|, max and the reciprocal are one-element instructions
                V2 ssq[2] = {V2{T(0), T(0)}, V2{T(0), T(0)}};   // two partial sums: no dependent packed FMAs back to back
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    V2 e2 = T(TB::berr[0]) * k[0][0].p[pp];
#pragma unroll
                    for (int q = 1; q < 7; ++q)
                        if (TB::berr[q] != 0.0) e2 += T(TB::berr[q]) * k[q][0].p[pp];
                    const V2 ym = V2{M::max_abs(y[0].p[pp][0], yt[0].p[pp][0]), M::max_abs(y[0].p[pp][1], yt[0].p[pp][1])};
                    const V2 sc = ym * rtol + atol;
                    V2 r;
                    if constexpr (F::STRICT_CONTROL) r = (PRESCALE ? e2 : dt * e2) / sc;          // (IEEE division, as the oracle)
                    else r = (PRESCALE ? e2 : dt * e2) * V2{M::rcp_fast(sc[0]), M::rcp_fast(sc[1])};
                    L.count_once(pp, r);        // (an element replicated over the trajectory's lanes counts once)
                    ssq[pp & 1] += r * r;       // (the pad element of an odd NV carries e = 0)
                }
                const V2 ss2 = ssq[0] + ssq[1];
                const T ss = ss2[0] + ss2[1];
                // (families whose right-hand side has kinks keep the oracle's operation order to the letter: where a step lands
                // relative to a kink is decided by float32 rounding, and their parity bars were measured with it)
                if constexpr (F::ROOTLESS_NORM && !F::STRICT_CONTROL) Control<T>::decide_ms(L.traj_sum(ss) / Dn, tprev, dt, keep, finite, factor);
                else Control<T>::template decide<F::STRICT_CONTROL>(M::sqrt(L.traj_sum(ss) / Dn), tprev, dt, keep, finite, factor);
            } else {
                T chk = 0;
#pragma unroll
                for (int v = 0; v < NV; ++v) chk += (T)yt[0][v] - (T)yt[0][v];
                chk = L.traj_sum(chk);
                finite = (chk == T(0));
            }
            const bool act = !done;
            steps += act ? 1 : 0;
            if (act && !finite) {
                st = ST_NONFINITE;
                done = true;
            }
            const bool accept = act && finite && keep;

            // ---- SaveAt(ts): dense output at every save time in (tprev, tnext]
            bool pending = accept && (save_idx < n_save) && (ts_next <= tnext);
            T inv_dt = M::recip(dt);
            // (a step clipped to length zero in front of a discontinuity point: its rows are the state itself, theta = 0, as
            // in the oracle -- not 0 * inf)
            if (__builtin_expect(n_jump > 0, 0)) inv_dt = dt > T(0) ? inv_dt : T(0);
            if constexpr (PC) { // the rows are wave 1's: publish the accepted step (see consume())
                pending = false;
                if (__any(accept)) {
                    L.dense_begin(tb, dt, y, yt, k);
                    __syncthreads();           // A: wave 1 has copied the previous step
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        tb.hand.planes[(0 * NP + pp) * 64 + lane] = y[0].p[pp];
                        tb.hand.planes[(1 * NP + pp) * 64 + lane] = k[0][0].p[pp];
                        tb.hand.planes[(2 * NP + pp) * 64 + lane] = k[1][0].p[pp];
                        tb.hand.planes[(3 * NP + pp) * 64 + lane] = k[2][0].p[pp];
                        tb.hand.planes[(4 * NP + pp) * 64 + lane] = k[3][0].p[pp];
                    }
                    tb.hand.tprev[lane] = tprev;
                    tb.hand.tnext[lane] = accept ? tnext : tprev - T(1);
                    __syncthreads();           // B: published
                }
                if (accept) save_idx = n_save; // (the producer does not track rows: nothing for it to fill at the end)
            }
            if (__any(pending)) {
                // what the family prepares once per accepted step for all its rows (Solver: the interpolant as a polynomial
                // in theta, dense_coefficients; Seip: nothing, its rows are weighted sums of the stages)
                L.dense_begin(tb, dt, y, yt, k);
                if constexpr (SU > 1) {
                    while (multi_rows && __any(pending)) {
                        if (pending) {
                            T tsu[SU];
                            bool pu[SU];
                            tsu[0] = ts_next;
#pragma unroll
                            for (int q = 1; q < SU; ++q)
                                tsu[q] = save_idx + q * R < n_save ? ts_tab[save_idx + q * R] : M::inf();
                            int cnt = 0;
#pragma unroll
                            for (int q = 0; q < SU; ++q) {
                                pu[q] = tsu[q] <= tnext; // increasing grid: the saved ones form a prefix
                                cnt += pu[q] ? 1 : 0;
                            }
#pragma unroll
                            for (int q = 0; q < SU; ++q) {
                                L.emit_row(ka, tb, out, ((pu[q] ? tsu[q] : tprev) - tprev) * inv_dt, dt, y, yt, k, save_idx + q * R,
                                           pu[q] && writer, vec_ok);
                            }
                            save_idx += cnt * R;
                            ts_next = save_idx < n_save ? ts_tab[save_idx] : M::inf();
                        }
                        pending = accept && (save_idx < n_save) && (ts_next <= tnext);
                    }
                }
                while ((SU == 1 || !multi_rows) && __any(pending)) {
#ifdef DYN_DIAG_ROUNDS
                    ++diag_rounds;
#endif
                    if (pending) {
                        L.emit_row(ka, tb, out, (ts_next - tprev) * inv_dt, dt, y, yt, k, save_idx, writer, vec_ok);
                        save_idx += R;
                        ts_next = ts_next2;
                        ts_next2 = save_idx + R < n_save ? ts_tab[save_idx + R] : M::inf();
                    }
                    pending = accept && (save_idx < n_save) && (ts_next <= tnext);
                }
            }

            // ---- commit / reject
            if (accept) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        y[c].p[pp] = yt[c].p[pp];
                        if constexpr (!PRESCALE) k[0][c].p[pp] = k[6][c].p[pp];   // (PRESCALE: FSAL is taken over below, together with its rescaling)
                    }
                if constexpr (F::REPLAYS) L.record_step(ka, traj, n_acc, tprev, tnext, writer);   // (KArgs::sched_out)
                ++n_acc;
            } else if (act && finite) {
                ++n_rej;
            }
            [[maybe_unused]] const bool fsal_from_k6 = accept;
            [[maybe_unused]] bool landed_now = false;      // this attempt ended on a discontinuity point (and was accepted)
            // ---- next interval: prev_dt * factor, then diffeqsolve's clip-to-end
            T next_t0 = accept ? tnext : tprev;
            // (the product is rounded on its own in every instance: where `constant` is a compile-time false the compiler would
            // otherwise contract dt * factor + next_t0 into one FMA, and an instance with the switch compiled in would take
            // steps that differ in the last bit from the general one -- sub-saves are compared bit for bit with full saves)
            T next_t1;
            if constexpr (LEAN) { // (no general twin to agree with: the contracted form it has always had)
                next_t1 = next_t0 + dt * factor;
            } else {
                T next_len;
                {
#pragma clang fp contract(off)
                    next_len = constant ? ka.constant_dt : dt * factor;
                }
                {
#pragma clang fp contract(off)
                    next_t1 = next_t0 + next_len;
                }
            }
            if (__builtin_expect(replay, 0)) { // launch-uniform: the next recorded step; across a discontinuity point the first stage is recomputed
                if constexpr (F::REPLAYS) {
                    if (act) ++si;
                    const bool more = si < tb.n_sch;
                    next_t0 = more ? tb.sch[2 * si] : t_end;
                    next_t1 = more ? tb.sch[2 * si + 1] : t_end;
                    const bool gap = act && accept && more && next_t0 != tnext;
                    if (__any(gap)) {
                        L.rhs(next_t0, y, k[1]);
                        if (gap) {
#pragma unroll
                            for (int c = 0; c < NC; ++c)
#pragma unroll
                                for (int pp = 0; pp < NP; ++pp) k[0][c].p[pp] = k[1][c].p[pp];
                        }
                    }
                    if (act && finite) {
                        tprev = next_t0;
                        tnext = next_t1;
                        if (!more) done = true;
                        else if (steps >= ka.max_steps) {
                            st = ST_MAX_STEPS;
                            done = true;
                        }
                    }
                }
            } else {
            if (__builtin_expect(n_jump > 0, 0)) { // wave-uniform: no cost when discontinuity_points is empty
                const bool landed = at_jump && accept;
                if (landed) {
                    // prev_dt is the controller's proposal before the jump clipped it
                    next_t0 = M::next(jt_tab[jidx], M::inf());
                    next_t1 = next_t0 + (constant ? ka.constant_dt : dt_unclipped * factor);
                    ++jidx;
                    landed_now = true;
                }
                // FSAL is invalid across a jump: k[0] = f(t_jump+, y).  (PRESCALE: formed below, with the NEXT step's rates)
                if constexpr (!PRESCALE) {
                    if (__any(landed)) {
                        L.rhs(next_t0, y, k[1]);
                        if (landed) {
#pragma unroll
                            for (int c = 0; c < NC; ++c)
#pragma unroll
                                for (int pp = 0; pp < NP; ++pp) k[0][c].p[pp] = k[1][c].p[pp];
                        }
                    }
                }
                if (act) at_jump = false;
                // (points at or behind the next step's start are done with -- also one a step happened to END on exactly:
                // constant steps on a grid the point lies on; the index would otherwise stay on it and later points be ignored)
                while (act && jidx < n_jump && jt_tab[jidx] <= next_t0) ++jidx;
                if (act && jidx < n_jump) {
                    const T tj = jt_tab[jidx];
                    if (tj < next_t1 && tj > next_t0) {
                        dt_unclipped = next_t1 - next_t0;
                        next_t1 = M::next(tj, -M::inf());
                        at_jump = true;
                    }
                }
            }
            const T tp = M::min(next_t0, t_end);
            if (Control<T>::clip_to_end(next_t1, tp, accept, t_end)) at_jump = false;
            if (!done) {
                tprev = tp;
                tnext = next_t1;
                if (!(tprev < t_end)) {
                    done = true;
                } else if (steps >= ka.max_steps) {
                    st = ST_MAX_STEPS;
                    done = true;
                }
            }
            }
            if constexpr (PRESCALE) {
                // the next attempt's step size: rates from their parked originals, FSAL by the ratio of the step sizes
                const T dt_new = tnext - tprev;
                const T ratio = act ? dt_new * inv_dt : T(1);
                if (fsal_from_k6) {   // accepted: the last stage's derivative is the next step's first one (no separate copy)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) k[0][c].p[pp] = k[6][c].p[pp] * ratio;
                } else {              // rejected (or idle): the same first stage, for another step size
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) k[0][c].p[pp] = k[0][c].p[pp] * ratio;
                }
                L.scale_rates(tb.rate_tab, lane, dt_new);
                // ... and across a discontinuity point the first stage is evaluated afresh, K0 = dt_new f(t_jump+, y), on the
                // new step's rates.  (Rescaling a derivative taken on the clipped step's rates, as the FSAL above, divides by
                // that step's length -- which is ZERO when the step before happened to end on the last representable time
                // in front of the point: the clipped step then has tnext == tprev, is accepted with error 0, as in the
                // reference's controller, and 0 * inf made every later attempt non-finite.  Found by the fuzz sweep, seed 21447.)
                if (__builtin_expect(n_jump > 0, 0)) {
                    if (__any(landed_now)) {
                        L.rhs(tprev, y, k[1]);
                        if (landed_now) {
#pragma unroll
                            for (int c = 0; c < NC; ++c)
#pragma unroll
                                for (int pp = 0; pp < NP; ++pp) k[0][c].p[pp] = k[1][c].p[pp];
                        }
                    }
                }
            }

            // ---- a trajectory that finished in this iteration is written off (stepper_writeoff.inc), and its slot asks for the next one
            if constexpr (F::PULLS) {
                if (__builtin_expect(live && done, 0)) {
#include "stepper_writeoff.inc"
                }
            } else {
                if (!__any(live && !done)) break;
            }
        }
        if constexpr (!F::PULLS) { // a static family: behind the loop
            if (live) {
#include "stepper_writeoff.inc"
            }
        }
        if constexpr (PC) { // tell wave 1 that no more steps will come
            __syncthreads();                   // A
            if (lane == 0) *tb.hand.fin = 1;
            __syncthreads();                   // B
        }
        if constexpr (FUSED) {
            // ---- the sampler's side of the iteration, for the chains whose trajectories this wave has just scored (a static
            // launch without a caller's order: the wave's slots hold trajectories t0 .. t0 + nt - 1; enqueue() checked that
            // whole chains fall into waves).  One lane per chain, as in nuts_advance.
            const auto &kc = *cold_args<T>();
            if (kc.nuts_tail != nullptr) {
                // the kernel's second argument, read where it is used (like cold_args: nothing of it lives through the stepping loop)
                const auto &tl = *reinterpret_cast<const dynnuts::Tail __attribute__((address_space(4))) *>(
                    reinterpret_cast<const char __attribute__((address_space(4))) *>(&kc) + F::kTailOffset);
                __threadfence(); // ll_out / dll_out of this wave's trajectories have reached memory
                const int rows = tl.rows_per_chain;
                const int nt = TPW >> kc.rep_log2;
                const int64_t t0 = ((int64_t)blockIdx.x * TPW) >> kc.rep_log2;
                const int64_t c = t0 / rows + lane;
                // (tl.magic: the second argument really sits where kTailOffset says -- a kernarg layout this code did not expect
                // makes the launch a plain gradient-solve whose chains stop advancing, instead of a corrupted sampler state)
                if (tl.magic == dynnuts::kTailMagic && lane < nt / rows && c < (int64_t)tl.st.n_chains)
                    dynnuts::fused_tail<F::kTailMaxDim>(tl, (int)c, kc.ll_out, kc.dll_out);
            }
        }
    }
};

} // namespace dyn
