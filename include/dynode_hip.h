/*
 * dynode_hip.h -- C-ABI of the MI355X-native batched ODE engine (libdynode_hip.so).
 *
 * The reference (CDCgov/DynODE @ 2026.01.28.1a) is pure Python and has no FFI: the hot
 * path sits behind the Python call
 *     dynode.simulation.simulate(ode, duration_days, initial_state, ode_parameters,
 *                                solver_parameters, sub_save_indices, save_step)
 * (src/dynode/simulation/odes.py:35-43), which hands everything to
 * diffrax.diffeqsolve (odes.py:133-144).  The entry points below are what a binding for
 * that call site binds instead of diffeqsolve; INTEGRATION.md shows the ctypes stub.
 *
 *   reference interface                                   entry point here
 *   ------------------------------------------------------------------------------------
 *   diffeqsolve(term, solver, t0, t1, dt0, y0, args,      dyn_solve_batch
 *               stepsize_controller, saveat, max_steps)
 *               (odes.py:133-144)
 *   ODETerm(ode) with the example RHS family               dyn_model_desc (declarative)
 *               (odes.py:107; examples RHS)
 *   SolverParams (src/dynode/config/params.py:24-67)       dyn_solver_opts
 *   SaveAt(ts)/SubSaveAt (odes.py:148-198)                 save_ts / save_mask arguments
 *   diffrax Solution.stats / RESULTS                       status / n_accept / n_reject
 *
 * Conventions: plain pointers and sizes only.  Every data pointer of dyn_solve_batch is a
 * DEVICE pointer (HBM) owned by the caller; the library allocates nothing user-visible,
 * keeps no mutable global state, reads nothing from the process environment and never
 * synchronises: work is enqueued on `stream`.
 * Return value: 0 on success, negative DYN_ERR_* on argument errors (nothing enqueued).
 */
#ifndef DYNODE_HIP_H
#define DYNODE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYN_ABI_VERSION 9
/* the save grid is staged in LDS: n_save * sizeof(real) must not exceed this (128 KB of a CU's 160: 32 k float32 or 16 k
 * float64 save times; beyond 48 KB -- the limit up to round 3 -- a launch keeps only one or two waves per CU resident) */
#define DYN_MAX_SAVE_BYTES 131072

/*
 * One member of the compartmental RHS family (the examples of the reference).
 * State layout per trajectory, "compartment-major", each block row-major:
 *     s[A] | e[A,S] (has_e) | i[A,S] | r[A,S,W] | c[A,S] (has_c)
 * Parameter vector per trajectory (length P = dyn_param_dim):
 *     beta[S] gamma[S] sigma[S] (has_e) omega[S] (has_wane)
 *     intro_time[S] intro_scale[S] intro_pct[S] (has_intro)  amp phase period (seasonal)
 *     vaccination block (n_vax_tiers > 1, see below)
 * RHS (seirs_multi_strain_age_stratified.py:213-243; S=1/no-e/no-wane reduce to
 * sir_age_stratified.py:127-142, seirs.py:88-95, sir.py:78-84):
 *     N_b = s_b + sum_l (e+i+sum_w r)_{b,l};  x_{b,l} = i_{b,l}/N_b (normalize) else i_{b,l}
 *     foi_{a,l} = beta_l * season(t) * sum_b C[a][b] x_{b,l};  flux = foi * s_a
 *     ds = -sum_l flux + sum_l W*omega_l r_{a,l,W-1};  de = flux - sigma e;  di = sigma e - gamma i
 *     dr_0 = gamma i - W omega r_0;  dr_w = W omega (r_{w-1} - r_w);  dc = flux
 *     season(t) = 1 + amp*sin(2*pi*t/period + phase)      (seirs_seasonal_forcing.py:40-55)
 * has_intro: strains seeded from an untracked external population (Strain.is_introduced,
 * introduction_time / _scale / _percentage / _ages: src/dynode/config/strains.py:53-109; the term is
 * the one of ode_model.md, "I_b + N(mu, sigma) * phi * P_b" inside the force of infection):
 *     x_{b,l} += intro_pct_l * NormalPdf(t; intro_time_l, intro_scale_l)   for ages b in intro_age_mask[l]
 *     (times N_b when normalize = 0)
 * n_vax_tiers > 1: vaccination fluxes (ode_model.md; VaccinationDimension, Strain.vaccine_efficacy,
 * utils/splines.py).  The contact axis then enumerates (age, tier) GROUPS, g = age * KV + tier with
 * KV = 2 (two tiers) or 4 (three or four; unused tiers stay empty), n_age = ages * KV, normalize = 0 and
 * a group contact matrix C[(a,k)][(b,j)] = C_age[a][b] / P_b supplied by the caller, so that
 * foi_{(a,k),l} = sus_{(a,k),l} * beta_l * sum_b C_age[a][b] * (sum_j i_{(b,j),l}) / P_b.
 * Susceptibles of tier k < n_vax_tiers - 1 move to tier k + 1:
 *     flow_{a,k} = min( max(nu_{a,k}(t), 0) * P_a , s_{a,k} ),   P_a = current population of age a,
 *     nu(t) = base0 + base1 t + base2 t^2 + base3 t^3 + sum_q coef_q (t - knot_q)^3 [t > knot_q]
 * (the cubic-spline form of utils/splines.py:evaluate_cubic_spline).  Everyone else keeps their tier.
 * Parameter block, after the seasonal numbers:  sus[groups][S] (= 1 - vaccine efficacy of the group's
 * tier against each strain), then per group base[4] knot[n_vax_knots] coef[n_vax_knots].
 *
 * SEIP (family = 1): the production model of ode_model.md:15-53 -- susceptible / exposed / infectious /
 * partially immune, stratified by age a, immune history j (a bit set over the L strains, H = 2^L), vaccination
 * tier k = 0..K (K1 = max(n_vax_tiers, 1) tiers) and, for the susceptibles, waning state m = 0..M
 * (M1 = n_wane states).  The reference describes it in prose only; this is the concrete form built here.
 * State, compartment-major, row-major blocks:   s[A][H][K1][M1] | e[A][H][K1][L] | i[A][H][K1][L] | c[A][H][K1][L]
 * Parameters:  beta[L] gamma[L] sigma[L] | omega[M1] (rate of leaving waning state m; the last entry is ignored,
 *     the last state keeps its people) | intro_time[L] intro_scale[L] intro_pct[L] (has_intro) |
 *     amp phase period (seasonal) | tau (seasonal_vax) | pop[A] |
 *     sus[H][K1][M1][L] | spline[A][K1][4 + 2 n_vax_knots] (base[4] knot[] coef[] as above)
 * Right-hand side (has_e = has_c = has_wane = 1, normalize = 0; `contact` is used as given, the caller folds
 * 1 / P into it):
 *     lambda_{a,l} = beta_l season(t) sum_b C[a][b] ( sum_{j,k} i_{b,j,k,l} + visitors_{b,l}(t) )   (ode_model.md:176-183)
 *     visitors_{b,l}(t) = intro_pct_l NormalPdf(t; intro_time_l, intro_scale_l) pop_b for ages b of intro_age_mask[l] (has_intro)
 *     infection     f_{a,j,k,m,l} = lambda_{a,l} sus_{j,k,m,l} s_{a,j,k,m}: leaves s, enters e_{a,j,k,l} and c
 *         (sus = 1 - WI of ode_model.md:185-211: the host evaluates cross-immunity, vaccine efficacy, waning
 *          protection and minimum homologous immunity into the table)
 *     progression   de = sum_m f - sigma_l e;   di = sigma_l e - gamma_l i
 *     recovery      gamma_l i_{a,j,k,l} enters s_{a, j | 2^l, k, 0}                   (eta, ode_model.md:86-105)
 *     waning        omega_m s_{a,j,k,m}: m -> m + 1 for m < M
 *     vaccination   r_{a,k} = min( max(nu_{a,k}(t), 0) pop_a / sum_{j,m} s_{a,j,k,m}, 1 )  (0 when the sum is <= 0);
 *         r_{a,k} s_{a,j,k,m} moves to (k + 1, m = 0); in the top tier K to (K, 0), and (K, 0) itself stays --
 *         the m' = 0 term of ode_model.md's top-tier gain has no matching loss and is dropped, so people are conserved
 *     seasonal vaccination (seasonal_vax)   phi(t) = sin(2 pi (t + tau) / 730)^1000 moves s, e, i of tier K to K - 1
 *         (ode_model.md:70-84; applied at all times -- the power makes it vanish outside the yearly window)
 * Limits: group_width(n_age) * 2^n_strain <= 128 lanes, n_strain <= 4, tiers <= 4.  A trajectory is one lane group of a
 * wavefront when it fits (optionally with the tiers dealt over two lanes), else a workgroup of 2..6 waves with the cross-wave
 * sums through LDS ("wave groups": 8 ages x 3 strains x 3 tiers = three waves of one tier each, D = 2496; 8 x 4 x 3 = six
 * waves, D = 6144) -- which mapping runs is the library's choice among the compiled instances (dyn_last_kernel_name).
 * All solver options of dyn_solve_batch apply (both methods, constant steps, discontinuity points, sub-save masks).
 * dyn_solve_batch_jvp / _loglik return DYN_ERR_UNSUPPORTED for this family (its kernels carry no tangent planes): gradients
 * come from dyn_solve_batch_record + dyn_solve_batch_replay (central differences on the recorded step sequence, below).
 */
#define DYN_MAX_STRAINS 8
typedef struct dyn_model_desc {
    int32_t n_age;     /* A: bins on the contact axis (1..64) */
    int32_t n_strain;  /* S */
    int32_t has_e;
    int32_t has_wane;
    int32_t has_c;
    int32_t n_wane;    /* W >= 1 (W > 1 requires has_wane) */
    int32_t normalize;
    int32_t seasonal;
    int32_t has_intro; /* ABI 2 */
    int32_t n_vax_tiers; /* ABI 3: 0/1 = no vaccination axis; 2..4 = tracked dose counts (see above) */
    uint64_t intro_age_mask[DYN_MAX_STRAINS]; /* per strain: bit a = age bin a receives introductions */
    int32_t n_vax_knots; /* knots of the vaccination-rate splines, 0..4 */
    int32_t family;       /* ABI 4: 0 = the s/e/i/r/c family, 1 = SEIP (see above; n_strain <= 4, group_width(n_age) * 2^n_strain <= 128) */
    int32_t seasonal_vax; /* ABI 4, SEIP only */
    int32_t reserved;
} dyn_model_desc;

enum { DYN_TSIT5 = 0, DYN_DOPRI5 = 1 };
enum { DYN_F32 = 0, DYN_F64 = 1 };

/* ABI 9.  Dispatch hints of ONE call.  Which compiled instance, lane mapping and grid shape runs a call is the library's
 * measured choice when every field is 0 (what every ordinary caller passes).  Tests and tuning tools pin a choice here --
 * up to ABI 8 these were DYNODE_HIP_* environment variables read inside the dispatch; the library now reads nothing from
 * the process environment, so a call's result depends on its arguments alone.  None of them changes WHAT is computed;
 * `strains_per_lane` and `replicas_log2` choose among lane mappings whose float32 summation orders differ in the last bits
 * (see dyn_solve_batch on the batch-size rule). */
typedef struct dyn_dispatch_hints {
    int32_t pull;              /* work pulling (work_counter): 0 = where measured to pay, 1 = wherever the batch exceeds one
                                  resident round, -1 = never */
    int32_t pull_waves;        /* > 0: grid of a work-pulling launch in waves, instead of the waves the chip holds at once */
    int32_t strains_per_lane;  /* > 0: take the instance with this many strains per lane (and skip the batch-size rule) */
    int32_t replicas_log2;     /* 0 = by batch size; k + 1 = exactly 2^k lane groups per trajectory (k = 0..3) */
    int32_t producer_consumer; /* 1 = the two-wave stepping / dense-output kernel where that variant is compiled (default off) */
    int32_t general_instance;  /* 1 = keep the general instance: none of the variants with call facts compiled in (adaptive
                                  steps without discontinuity points, static grid, lean gradient-solve, SEIP plain) */
    int32_t seip_tier_lanes;   /* SEIP, tiers dealt over two lanes: 0 = by state size, 1 = on, -1 = off */
    int32_t seip_tier_waves;   /* SEIP, one tier per wave: 0 = where compiled, -1 = off */
    int32_t strict_control;    /* 1 = the test-only twin of the instance with the step controller in the oracle's float32
                                  arithmetic (IEEE division, sqrtf, powf instead of v_rcp / v_log / v_exp), where one is
                                  compiled (the D = 360 ensemble shape, the plain D = 2496 SEIP shape): measures what the fast
                                  controller arithmetic costs in accept / reject decisions */
} dyn_dispatch_hints;

/* SolverParams (params.py:24-67).  jump_ts is a HOST pointer (tiny, read at enqueue). */
typedef struct dyn_solver_opts {
    int32_t method;
    int32_t dtype;
    double rtol;        /* ode_solver_rel_tolerance, default 1e-5 */
    double atol;        /* ode_solver_abs_tolerance, default 1e-6 */
    int64_t max_steps;  /* default 1e6; exhausting it sets status DYN_STATUS_MAX_STEPS */
    double constant_dt; /* constant_step_size; > 0 selects ConstantStepSize */
    const double *jump_ts; /* discontinuity_points */
    int32_t n_jump;
    /* ABI 7.  Work pulling: DEVICE pointer to two int32 words that are zero when the launch starts, or NULL.  With it, a
     * batch of more trajectory-waves than the GPU holds at once is launched as one resident grid whose lane groups draw
     * trajectories from a queue as they finish (solve_kernel.hpp, Solver::run) instead of one wave per TPW trajectories:
     * no lane group idles behind a longer-running partner, and nothing has to be known about the batch in advance.  The
     * kernel leaves both words zero again, so one pair serves every launch of a stream; launches that may overlap (other
     * streams) need pairs of their own.  The caller owns the memory, like every other buffer of this ABI.  Results never
     * depend on it.  The library pulls where that was measured to pay: when the caller supplies the queue
     * (dyn_solve_batch_ordered: longest-first needs dynamic assignment) and a wave holds more than two trajectories;
     * hints.pull = 1 pulls wherever the batch exceeds one resident round.  NULL, batches of one resident round, and the
     * SEIP family: a static grid, as before ABI 7. */
    int32_t *work_counter;
    /* Fused sampler iteration (ABI 8, dyn_solve_batch_loglik only): HOST pointer to a blob written by dyn_nuts_tail_pack,
     * or NULL.  When set, every wave of the gradient-solve goes on, after scoring its trajectories, to run the NUTS state
     * machine (what dyn_nuts_advance_mapped does in a launch of its own) for the chains whose trajectories it just scored,
     * reading log-likelihood and gradient from what it wrote and leaving the parameter rows / tangent seeds of the chains'
     * next positions in `params` / `dparams` (which must be the blob's buffers) for the next call: one launch per sampler
     * iteration instead of two, the same draws bit for bit.  The blob is read during the call (it travels as a kernel
     * argument), so a captured HIP graph holds its own copy.  Needs the chain-major batch of dyn_nuts_advance_mapped (chain c
     * = trajectories c*rows .. c*rows + rows - 1), every chain's trajectories inside one wave (rows divides the trajectories
     * per wave, dyn_trajectories_per_wave_for_batch: pad the chains to a power of two of rows, dyn_latent_param_map), no
     * caller order, at most 8 sampled dimensions -- either one per row or all in a chain's single row -- and a shape whose
     * tangent instance was compiled with the sampler behind it (ABI 9: any observed compartment / likelihood mode of
     * dyn_solve_batch_loglik; csrc/instances.def units 28, 33, 34, and every float32 shape built on demand); otherwise the
     * call returns DYN_ERR_UNSUPPORTED with nothing enqueued and the caller keeps its two launches (DYN_ERR_OPTS: the pointer
     * is not a packed blob). */
    const void *nuts_tail;
    /* ABI 9: all zero = the library's choices (see dyn_dispatch_hints) */
    dyn_dispatch_hints hints;
} dyn_solver_opts;

/* per-trajectory status */
enum { DYN_STATUS_OK = 0, DYN_STATUS_MAX_STEPS = 1, DYN_STATUS_NONFINITE = 2 };

/* argument-error return codes */
enum {
    DYN_ERR_NULL = -1,
    DYN_ERR_MODEL = -2,
    DYN_ERR_SIZE = -3,
    DYN_ERR_OPTS = -4,
    DYN_ERR_TOL = -5,
    DYN_ERR_JUMP = -6,
    DYN_ERR_UNSUPPORTED = -7, /* shape / option not compiled into the library */
    DYN_ERR_LAUNCH = -8       /* hipLaunchKernel failed */
};

int32_t dyn_abi_version(void);
/* sizeof(dyn_model_desc), sizeof(dyn_solver_opts): a binding checks its own struct layouts against these before the first call */
int32_t dyn_model_desc_size(void);
int32_t dyn_solver_opts_size(void);
int32_t dyn_state_dim(const dyn_model_desc *m);
int32_t dyn_param_dim(const dyn_model_desc *m);
int32_t dyn_n_compartments(const dyn_model_desc *m);
/* element offsets of each compartment inside the flat state; off has n_comp+1 entries */
int32_t dyn_compartment_offsets(const dyn_model_desc *m, int32_t *off);
/* 1 if a kernel for (model shape, method, dtype) is compiled in, else 0 */
int32_t dyn_is_supported(const dyn_model_desc *m, const dyn_solver_opts *o);
/* trajectories that share a 64-lane wavefront for this model (a wave group of the SEIP family: per workgroup of NW waves) */
int32_t dyn_trajectories_per_wave(const dyn_model_desc *m);
/* ... of the float32 / Tsit5 default mapping.  The mapping a call actually gets depends on its batch size: a batch that fills
 * at most half a wave per SIMD takes the finest strain split compiled in (a shorter serial instruction stream per trajectory),
 * so a trajectory's float32 summation order -- its last bits and, rarely, an accept / reject decision -- can differ between a
 * 3072-row and a 65536-row batch (float64 step counts do not).  This query answers for (opts, B) what dyn_solve_batch will
 * use; opts->hints.strains_per_lane pins the mapping for callers that need batch-size-independent bits. */
int32_t dyn_trajectories_per_wave_for_batch(const dyn_model_desc *m, const dyn_solver_opts *opts, int64_t B);
/* last launch-failure text of the calling thread ("" if none) */
const char *dyn_last_error(void);
/* name of the kernel instance the calling thread's last successful dyn_solve_batch* call enqueued, spelled the way
 * rocprofv3's kernel trace prints it without the return type and argument list, e.g.
 * "dyn::solve_kernel<float, 0, 8, 4, true, true, true, 8, 0, 1, 16384>" ("" before the first launch).  bench.py
 * quotes it in its roofline block and only attaches profiled HBM traffic recorded for the same instance. */
const char *dyn_last_kernel_name(void);

/*
 * Batched solve over [t0, t1]: replaces diffeqsolve at odes.py:133-144 for B independent
 * parameter samples.  Float type of all real arrays = opts->dtype.
 *   y0        [D] (y0_is_batched = 0, shared) or [B][D]          device
 *   params    [B][P]                                             device
 *   contact   [A][A], foi_a = sum_b contact[a][b] x_b            device
 *   save_ts   [n_save] increasing, all within [t0, t1]           device
 *   save_mask [n_comp] HOST bytes, NULL = save all (SubSaveAt, odes.py:182-193)
 *   ys_out    [B][n_save][D_saved]  rows = saved compartments concatenated   device
 *   status, n_accept, n_reject  [B] int32                         device
 *   stream    hipStream_t (NULL = default stream)
 */
int dyn_solve_batch(const dyn_model_desc *m, const dyn_solver_opts *opts, const void *y0,
                    int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                    double t0, double t1, const void *save_ts, int32_t n_save,
                    const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                    int32_t *n_reject, void *stream);

/*
 * Step schedules (family = 1, SEIP).  The SEIP kernels carry no tangent planes (270 live values per lane as it is);
 * they are differentiated the way jax differentiates the reference's solve -- with the step-size controller held
 * fixed (src/dynode/infer/inference.py:149-163: value_and_grad through diffeqsolve, whose controller is under
 * stop_gradient): dyn_solve_batch_record writes down the (t_prev, t_next) pair of every accepted step of an adaptive
 * solve, dyn_solve_batch_replay makes any batch of parameter rows take exactly the steps of their "leader" row.  On a
 * fixed step sequence the solve is a smooth map of the parameters, so central differences of replayed solves are the
 * derivative of the adaptive solve (dynode_amd/engine.py composes them into the outputs of dyn_solve_batch_jvp).
 *   sched_out    [B][sched_cap][2]  accepted steps (solve dtype), device;  sched_n_out [B]: their number, -1 if > sched_cap
 *   sched        [n_leaders][sched_cap][2], sched_n [n_leaders];  leader [B] int64 row of `sched` per trajectory
 *                (NULL: trajectory b follows row b).  A follower of a leader with sched_n < 0 ends with status 1.
 * Discontinuity points are part of the recording (the gap between two recorded steps restarts the first stage).
 * The replayed schedule is staged in LDS: 2 * sched_cap values per trajectory of a wave next to the model's tables
 * (DYN_ERR_UNSUPPORTED when that exceeds 64 KB: lower sched_cap).  Other arguments as dyn_solve_batch.
 */
int dyn_solve_batch_record(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0, int32_t y0_is_batched,
                           const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                           int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, void *sched_out, int32_t *sched_n_out, int32_t sched_cap, void *stream);
int dyn_solve_batch_replay(const dyn_model_desc *m, const dyn_solver_opts *o, const void *y0, int32_t y0_is_batched,
                           const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                           int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, const void *sched, const int32_t *sched_n, const int64_t *leader,
                           int32_t sched_cap, void *stream);

/*
 * Batched solve + forward-mode tangents: value and directional derivatives of the saved
 * trajectory with respect to the parameters / initial state, for n_dir directions at once.
 * This is the gradient-solve under the reference's NUTS loop, where numpyro takes
 * value_and_grad of the potential through diffeqsolve (src/dynode/infer/inference.py:149-163,
 * examples/sir_infer_parameters.py:21-39).  Tangents are propagated through the SAME accepted
 * steps as the primal (step sizes and error control see the primal only), i.e. they are the
 * exact derivative of the computed trajectory -- what differentiating through diffrax yields.
 *   dparams [B][n_dir][P]     seed directions of the parameter vector                 device
 *   dy0     NULL (zero), [n_dir][D] (dy0_is_batched = 0) or [B][n_dir][D]              device
 *   dys_out [B][n_save][n_dir][D_saved]                                                device
 * Everything else as dyn_solve_batch.
 */
int dyn_solve_batch_jvp(const dyn_model_desc *m, const dyn_solver_opts *opts, const void *y0,
                        int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                        double t0, double t1, const void *save_ts, int32_t n_save,
                        const uint8_t *save_mask, int32_t n_dir, const void *dparams,
                        const void *dy0, int32_t dy0_is_batched, void *ys_out, void *dys_out,
                        int32_t *status, int32_t *n_accept, int32_t *n_reject, void *stream);
/*
 * Tangent solve with the observation likelihood fused in: nothing of the trajectory is written to
 * HBM (cfg 4: "D_saved = 0, likelihood fused").  For every batch member returns
 *     logp  = sum_{j, x} obs[j][x] * log(rate[j][x]) - rate[j][x]          (Poisson, without the
 *                                                        constant -lgamma(obs + 1), the caller's)
 *     dlogp = its derivative along each of the n_dir seed directions
 * where rate = max(v, rate_floor) and v is the observed compartment at save time j (obs_mode 0,
 * n_save rows of observations) or its increment between save times j and j + 1 (obs_mode 1,
 * n_save - 1 rows) -- the reference's model(): incidence = max(diff(R), 1e-6), obs ~ Poisson
 * (examples/sir_infer_parameters.py:30-38).  The derivative passes through the floor where
 * v >= rate_floor (the usual clamp / clip autodiff convention).  A failed solve scores logp = -inf.
 *   obs_compartment  index into the state tuple (as dyn_compartment_offsets)
 *   obs   [n_obs][size of that compartment]  same float type as the solve, shared by the batch   device
 *   logp_out [B], dlogp_out [B][n_dir]  float64                                                  device
 * Everything else as dyn_solve_batch_jvp.
 */
int dyn_solve_batch_loglik(const dyn_model_desc *m, const dyn_solver_opts *opts, const void *y0,
                           int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                           double t0, double t1, const void *save_ts, int32_t n_save,
                           int32_t obs_compartment, int32_t obs_mode, double rate_floor, const void *obs,
                           int32_t n_dir, const void *dparams, const void *dy0, int32_t dy0_is_batched,
                           double *logp_out, double *dlogp_out, int32_t *status, int32_t *n_accept,
                           int32_t *n_reject, void *stream);
/*
 * Add a kernel shape at run time.  The library ships the shapes of dynode_amd/csrc/instances.def;
 * any other member of the family is one explicit instantiation of the same template
 * (dyn::launch<T, METHOD, GA, S, E, WANE, C, W, ND, SPL[, INTRO]> in csrc/solve_kernel.hpp) compiled
 * into a separate shared object -- dynode_amd/jit.py does that with hipcc on first use -- whose
 * launcher is registered here.  ga = lanes on the age axis (power of two >= n_age), spl = strains
 * per lane, features = the template's FEAT (bit 0: introduced strains; bits 1..: vaccination-tier lanes); for a SEIP kernel
 * (launch_seip of csrc/seip_kernel.hpp) pass n_strain = L, n_wane = M1, spl = 1 and features = 0x100 | tiers.  Registered entries are consulted after the built-in ones.
 */
int dyn_register_instance(int32_t dtype, int32_t method, int32_t ga, int32_t n_strain, int32_t has_e,
                          int32_t has_wane, int32_t has_c, int32_t n_wane, int32_t n_dir, int32_t spl,
                          int32_t features, void *launch_fn);
/* 1 if a tangent kernel for (model shape, method, dtype, n_dir) is compiled in */
int32_t dyn_is_supported_jvp(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir);
/* Does that tangent kernel have the twin that also runs the sampler's side of a NUTS iteration (dyn_solver_opts::nuts_tail)?
 * 1: yes.  0: there is no tangent kernel for the shape at all.  -spl (< 0): the kernel exists without the twin; a builder adds
 * it as dyn_register_instance(..., n_dir, spl, features | 0x1000, launch<..., spl, features | 0x1000>) -- dynode_amd/jit.py does
 * on first use. */
int32_t dyn_fused_twin(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir);
/* Is there a LEAN twin of that tangent kernel for a likelihood on `obs_compartment` in `obs_mode` (the arguments of
 * dyn_solve_batch_loglik) -- the instance with the scored compartment, the likelihood mode, the normalised force of infection, "no
 * seasonal forcing, no discontinuity points, adaptive steps" as compile-time facts, which dyn_solve_batch_loglik dispatches to when
 * the call is that (the reference's inference example: 62.8 -> 49.1 us per gradient-solve; the 2-age x 3-strain model: 229 -> 151)?
 * 1: yes.  0: not applicable (no tangent kernel; a model with seasonal forcing, introduced strains or an un-normalised force of
 * infection; the SEIP family).  -1: the tangent kernel exists without this twin; *spl and *features then hold what a builder
 * registers: dyn_register_instance(..., n_dir, *spl, *features, launch<..., *spl, *features>) -- and the same with | 0x1000 for
 * the twin that carries the sampler (dyn_fused_twin).  dynode_amd/jit.py does on first use. */
int32_t dyn_lean_twin(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir, int32_t obs_compartment, int32_t obs_mode,
                      int32_t *spl, int32_t *features);

/*
 * One NUTS sampler iteration for n_chains independent chains (one GPU thread per chain).
 * Under numpyro the transition kernel is traced together with the model
 * (src/dynode/infer/inference.py:149-163: MCMC(NUTS(model, dense_mass=True, max_tree_depth=5,
 * init_strategy=init_to_median), ...)); here the model's potential and gradient are evaluated by
 * the caller (any program that reads z_eval and writes u_new / g_new, e.g. a HIP graph around
 * dyn_solve_batch_jvp) and this call does everything else a chain does between two potential
 * evaluations: second half of the leapfrog, energy error, multinomial / biased-progressive
 * proposal choice, checkpointed U-turn test, tree doubling, dual-averaging step size, windowed
 * dense mass matrix with its Cholesky factor, recording the draw, momentum refresh, and the first
 * half of the next leapfrog (which produces the next z_eval).  Chains advance asynchronously:
 * every call costs every unfinished chain exactly one gradient.  All arrays are device arrays,
 * row-major with the chain index first; real = float64.
 */
#define DYN_NUTS_MAX_DIM 32 /* up to 8: one compiled instance per dimension, one thread per chain, its vectors in registers; 9..32: a half
                               wave per chain, element l of every vector in lane l (plain u_new / g_new potential or, up to
                               DYN_MAX_SITES, a folded potential's parts; per-chain adaptation: pooled windows and the fused form
                               stop at 8 -- DYN_ERR_UNSUPPORTED beyond) */
#define DYN_NUTS_MAX_DEPTH 10
#define DYN_NUTS_MAX_WINDOWS 16
typedef struct dyn_nuts_state {
    int32_t n_chains, dim, max_depth;  /* dim <= DYN_NUTS_MAX_DIM (32), max_depth <= DYN_NUTS_MAX_DEPTH */
    int32_t num_warmup, num_samples;
    int32_t n_windows;                 /* slow adaptation windows [w_start, w_end) in transitions */
    int32_t pooled;                    /* 0: every chain adapts its own mass matrix (numpyro);
                                          1: window statistics pooled over the chains of this call */
    int32_t w_start[DYN_NUTS_MAX_WINDOWS], w_end[DYN_NUTS_MAX_WINDOWS];
    uint64_t seed;                     /* Philox key; stream = (seed, chain, per-chain counter) */
    double target_accept, max_delta_energy;
    /* exchange with the potential: position to evaluate, value and gradient there */
    double *z_eval;                    /* [C][D]  written by this call */
    const double *u_new, *g_new;       /* [C], [C][D]  read by this call (non-finite = divergent) */
    /* chain state */
    double *z, *u, *g;                 /* current draw, its potential and gradient */
    double *eps, *eps_avg, *da_mu, *da_xbar, *da_gbar, *da_t;      /* dual averaging  [C] */
    double *imm, *mm_sqrt;             /* inverse mass matrix, chol(mass)   [C][D][D] */
    double *wf_n, *wf_mean, *wf_m2;    /* Welford accumulators of the open window */
    /* trajectory (tree) state */
    double *e0, *zl, *rl, *gl, *zr, *rr, *gr, *zp, *up, *gp, *weight, *r_sum, *sum_acc, *sgn;
    /* subtree under construction */
    double *zc, *rc, *gc, *r_half, *s_zp, *s_up, *s_gp, *s_weight, *s_rsum, *s_acc;
    double *r_ck, *rs_ck;              /* U-turn checkpoints  [C][max_depth][D] */
    int32_t *it, *wi, *n_prop, *depth, *right, *leaf, *s_turn, *s_div, *s_n;   /* [C] */
    int64_t *rng_ctr;                  /* [C] */
    /* pooled adaptation (pooled = 1): fixed-point sums per window, zero-initialised by the caller */
    int64_t *pool, *pool_ro;               /* [W + 1][1 + D + D*D]: per window count, sum, sum of outer
                                              products; row W: count and sum of log final step sizes;
                                              pool_ro = copy made after each launch (what chains read) */
    int32_t *pend;                         /* [C] window whose pooled matrix is still to be applied, +1 */
    /* post-warm-up draws */
    double *out_z, *out_acc;           /* [C][num_samples][D], [C][num_samples] */
    int32_t *out_n, *out_div;          /* [C][num_samples] leapfrogs per draw, divergence flag */
    /* ABI 6, optional: the potential at z_eval in parts instead of (u_new, g_new) -- when pot_lp != NULL the kernel forms
     * u = -(pot_lp[c] + pot_ll[c * pot_ll_stride] + pot_offset), g[c][i] = -(pot_dlp[c][i] + pot_dll[c * pot_dll_stride + i])
     * itself (what dyn_potential_combine would write; saves that launch in every sampler iteration).  pot_dll_stride = 0
     * means D (a dense [C][D]); a gradient-solve with one direction per trajectory and padded chains (dyn_latent_param_map,
     * split_directions > n_sites) leaves [C][rows] */
    const double *pot_lp, *pot_dlp, *pot_ll, *pot_dll;
    double pot_offset;
    int32_t pot_ll_stride, pot_dll_stride;
} dyn_nuts_state;
int dyn_nuts_advance(const dyn_nuts_state *st, void *stream);
/* sizeof(dyn_nuts_state), for binding checks */
int32_t dyn_nuts_state_size(void);
/* the sampler's random-number block function (host copy, for known-answer tests):
 * Philox4x32-10, ctr[4], key[2] -> out[4] */
void dyn_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out);

/*
 * Latent (prior) sites of a NUTS model in one launch: for every chain map the unconstrained
 * coordinates z to the constrained values x (numpyro's biject_to of the site's support) and return
 * lp = sum over sites of [log prior(x_i) + log |dx_i/dz_i|] with the derivatives a backward pass
 * needs.  Replaces the per-op evaluation of the reference's priors
 * (examples/sir_infer_parameters.py:47-58) inside the traced potential.
 *   sites   [n_sites]  HOST descriptors;  z, x, dx_dz, dlp_dz [C][n_sites], lp [C]  device, float64
 *   dlp_dz = d lp / d z_i (prior and Jacobian terms), dx_dz = d x_i / d z_i
 */
#define DYN_MAX_SITES 16   /* (8 up to round 3; the one-launch iteration, dyn_nuts_tail_pack, stays at 8) */
enum { DYN_DIST_NORMAL = 0, DYN_DIST_UNIFORM = 1, DYN_DIST_BETA = 2, DYN_DIST_TRUNCNORMAL = 3 };
typedef struct dyn_site_desc {
    int32_t dist;            /* base distribution */
    int32_t reserved;
    double p[4];             /* normal {loc, scale}; uniform {low, high}; beta {a, b, log B(a,b)};
                                truncated normal {loc, scale, log Z} */
    double base_lo, base_hi; /* truncation bounds of the base (truncated normal), else unused */
    double aff_loc, aff_scale; /* site value = aff_loc + aff_scale * base (composed AffineTransforms) */
    double lo, hi;           /* support of the site value (+-INFINITY allowed): selects the bijection */
} dyn_site_desc;
int dyn_latent_sites(const dyn_site_desc *sites, int32_t n_sites, int64_t C, const double *z, double *x,
                     double *lp, double *dx_dz, double *dlp_dz, void *stream);

/*
 * The sampler's potential folded into three launches (dyn_latent_param_map -> dyn_solve_batch_loglik ->
 * dyn_potential_combine), for models whose ODE parameter row is a monomial in the site values:
 *     params[c][j] = coef[j] * prod_i x[c][i] ^ expo[j][i]
 * which is what the reference's get_odeparams family computes (examples/sir.py:87-92, sir_age_stratified.py:112-124,
 * seirs.py:98-104, seirs_multi_strain_age_stratified.py:187-209: beta = r0 / infectious_period, gamma = 1 / infectious_period,
 * sigma = 1 / latent_period, omega = 1 / waning_period).  Replaces the small program numpyro / XLA would trace between the
 * sample sites and diffeqsolve (examples/sir_infer_parameters.py:21-39).
 *   dyn_latent_param_map: dyn_latent_sites + the map.  coef [P], expo [P][n_sites]: DEVICE float64.
 *     params [C][P] and seeds [C][n_sites][P] (= d params / d z, the dparams argument of dyn_solve_batch_loglik) are
 *     written in `dtype` (DYN_F32 / DYN_F64); x, lp, dlp_dz as in dyn_latent_sites.
 *   dyn_potential_combine: u[c] = -(lp[c] + ll[c] + offset), g[c][i] = -(dlp_dz[c][i] + dll[c][i]); all float64, device.
 *   split_directions != 0 (few chains on an otherwise idle GPU): one tangent direction per trajectory instead of n_sites in
 *     one -- params [C][n_sites][P] (chain c repeated n_sites times), seeds [C][n_sites][1][P], to be solved as a batch of
 *     n_sites C rows with n_dir = 1 (a third less work on the serial path of every trajectory at n_sites = 2; same bits); the
 *     combine then reads ll [C n_sites] at c n_sites and dll [C n_sites][1] as [c][i].
 *   split_directions = r >= 2 (n_sites <= r <= 64): the same with r rows per chain -- rows n_sites .. r - 1 of a chain are
 *     padding (its parameters, zero seeds; their outputs are never read): params [C][r][P], seeds [C][r][1][P], ll [C r],
 *     dll [C r][1].  With r a power of two whole chains fall into waves, which dyn_solver_opts::nuts_tail needs.
 */
int dyn_latent_param_map(const dyn_site_desc *sites, int32_t n_sites, int64_t C, const double *z, double *x, double *lp,
                         double *dlp_dz, int32_t P, const double *coef, const double *expo, int32_t dtype,
                         int32_t split_directions, void *params, void *seeds, void *stream);
int dyn_potential_combine(int64_t C, int32_t n, const double *lp, const double *dlp_dz, const double *ll,
                          const double *dll, double offset, int32_t split_directions, double *u, double *g, void *stream);
/* dyn_nuts_advance that also does, for the position it hands out (z_eval), what dyn_latent_param_map (below) would do in a
 * launch of its own: constrained values, log prior and its derivative, the parameter rows and tangent seeds of the solve that
 * follows.  With st->pot_* set, a sampler iteration of the folded potential is TWO launches: dyn_solve_batch_loglik and this.
 * n_sites must equal st->dim (up to DYN_MAX_SITES: beyond 8 the half-wave-per-chain form of the state machine, per-chain
 * adaptation only); the buffers are those of dyn_latent_param_map (lp = st->pot_lp, dlp_dz = st->pot_dlp of the next call). */
int dyn_nuts_advance_mapped(const dyn_nuts_state *st, const dyn_site_desc *sites, int32_t n_sites, int32_t P, const double *coef,
                            const double *expo, int32_t dtype, int32_t split_directions, double *x, double *lp, double *dlp_dz,
                            void *params, void *seeds, void *stream);
/* The blob dyn_solver_opts::nuts_tail points at: the arguments of dyn_nuts_advance_mapped, written to `blob` (HOST memory of
 * dyn_nuts_tail_size() bytes, owned by the caller).  st->pot_lp / pot_dlp / pot_offset must be set (= lp, dlp_dz);
 * st->pot_ll / pot_dll are not used -- the fused launch reads what it wrote itself.  The device buffers named inside must stay
 * valid and unmoved for as long as launches use the blob (a sampler run: the state buffers never move).  DYN_ERR_UNSUPPORTED:
 * more than 8 sampled dimensions (where the site table and the compile-time-dimension state machines stop too). */
int32_t dyn_nuts_tail_size(void);
int dyn_nuts_tail_pack(const dyn_nuts_state *st, const dyn_site_desc *sites, int32_t n_sites, int32_t P, const double *coef,
                       const double *expo, int32_t dtype, int32_t split_directions, double *x, double *lp, double *dlp_dz,
                       void *params, void *seeds, void *blob);

/*
 * Dispatch order.  dyn_solve_batch_ordered is dyn_solve_batch with one more argument: grid slot i integrates trajectory
 * order[i] (DEVICE int32 [B], a permutation of 0..B-1; NULL = identity; an entry outside 0..B-1 makes its slot idle, rows that
 * no entry names stay unwritten).  Outputs are bit-identical for every permutation -- each trajectory is computed from its
 * own inputs and written to its own rows -- but the lane groups of a wave step in lock-step and waves start in index order, so
 * putting trajectories with similar step counts next to each other, the expensive ones first, shortens the launch by 10-15 %
 * (diffrax under vmap / pmap has no counterpart: XLA:CPU runs the samples one after another).  The learned step-count
 * forecast of rounds 2-3 (dyn_cost_order, dynode_amd/schedule.py) bought nothing on top of the hardware's own wave dispatch
 * and left the ABI with version 9; a caller who knows its batch passes its own order.
 */
int dyn_solve_batch_ordered(const dyn_model_desc *model, const dyn_solver_opts *opts, const void *y0, int32_t y0_is_batched,
                            const void *params, const void *contact, int64_t B, double t0, double t1, const void *save_ts,
                            int32_t n_save, const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                            int32_t *n_reject, const int32_t *order, void *stream);


#ifdef __cplusplus
}
#endif
#endif
