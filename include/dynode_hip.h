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
 * keeps no mutable global state and never synchronises: work is enqueued on `stream`.
 * Return value: 0 on success, negative DYN_ERR_* on argument errors (nothing enqueued).
 */
#ifndef DYNODE_HIP_H
#define DYNODE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYN_ABI_VERSION 1
/* the save grid is staged in LDS: n_save * sizeof(real) must not exceed this */
#define DYN_MAX_SAVE_BYTES 49152

/*
 * One member of the compartmental RHS family (the examples of the reference).
 * State layout per trajectory, "compartment-major", each block row-major:
 *     s[A] | e[A,S] (has_e) | i[A,S] | r[A,S,W] | c[A,S] (has_c)
 * Parameter vector per trajectory (length P = dyn_param_dim):
 *     beta[S] gamma[S] sigma[S] (has_e) omega[S] (has_wane) amp phase period (seasonal)
 * RHS (seirs_multi_strain_age_stratified.py:213-243; S=1/no-e/no-wane reduce to
 * sir_age_stratified.py:127-142, seirs.py:88-95, sir.py:78-84):
 *     N_b = s_b + sum_l (e+i+sum_w r)_{b,l};  x_{b,l} = i_{b,l}/N_b (normalize) else i_{b,l}
 *     foi_{a,l} = beta_l * season(t) * sum_b C[a][b] x_{b,l};  flux = foi * s_a
 *     ds = -sum_l flux + sum_l W*omega_l r_{a,l,W-1};  de = flux - sigma e;  di = sigma e - gamma i
 *     dr_0 = gamma i - W omega r_0;  dr_w = W omega (r_{w-1} - r_w);  dc = flux
 *     season(t) = 1 + amp*sin(2*pi*t/period + phase)      (seirs_seasonal_forcing.py:40-55)
 */
typedef struct dyn_model_desc {
    int32_t n_age;     /* A: bins on the contact axis (1..64) */
    int32_t n_strain;  /* S */
    int32_t has_e;
    int32_t has_wane;
    int32_t has_c;
    int32_t n_wane;    /* W >= 1 (W > 1 requires has_wane) */
    int32_t normalize;
    int32_t seasonal;
} dyn_model_desc;

enum { DYN_TSIT5 = 0, DYN_DOPRI5 = 1 };
enum { DYN_F32 = 0, DYN_F64 = 1 };

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
int32_t dyn_state_dim(const dyn_model_desc *m);
int32_t dyn_param_dim(const dyn_model_desc *m);
int32_t dyn_n_compartments(const dyn_model_desc *m);
/* element offsets of each compartment inside the flat state; off has n_comp+1 entries */
int32_t dyn_compartment_offsets(const dyn_model_desc *m, int32_t *off);
/* 1 if a kernel for (model shape, method, dtype) is compiled in, else 0 */
int32_t dyn_is_supported(const dyn_model_desc *m, const dyn_solver_opts *o);
/* trajectories integrated by one 64-lane wavefront for this model (lanes = age bins) */
int32_t dyn_trajectories_per_wave(const dyn_model_desc *m);
/* last launch-failure text of the calling thread ("" if none) */
const char *dyn_last_error(void);

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
/* 1 if a tangent kernel for (model shape, method, dtype, n_dir) is compiled in */
int32_t dyn_is_supported_jvp(const dyn_model_desc *m, const dyn_solver_opts *o, int32_t n_dir);

#ifdef __cplusplus
}
#endif
#endif
