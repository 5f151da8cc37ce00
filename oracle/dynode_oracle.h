/*
 * dynode_oracle.h -- CPU restatement of DynODE's simulate() hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped path is the HIP
 * library declared in include/dynode_hip.h and never routes through here.
 *
 * What is restated (reference paths relative to /root/reference):
 *   - src/dynode/simulation/odes.py:35-145   simulate(): controller choice, save grid
 *   - src/dynode/config/params.py:24-67      SolverParams defaults
 *   - examples/sir.py:78-84, sir_age_stratified.py:127-142, seirs.py:88-95,
 *     seirs_seasonal_forcing.py:40-55, seirs_multi_strain_age_stratified.py:213-243,
 *     sir_age_risk_stratified.py:157-173     the compartmental RHS family
 *   - diffrax 0.7.* (pyproject.toml:12; third party, NOT in /root/reference):
 *     Tsit5 / Dopri5 steppers, PIDController (defaults = I-controller),
 *     ClipStepSizeController, SaveAt(ts) dense output.  Restated from the published
 *     algorithms (Tsitouras 2011; Dormand-Prince 1980 + Shampine 1986 midpoint;
 *     Hairer-Norsett-Wanner II.4 initial step).
 *
 * PARITY STATUS: "parity unpinned" against diffrax itself -- jax/diffrax are not
 * installable in the build container and the reference ships no golden vectors
 * (SURVEY.md section 8c).  The oracle is pinned instead by: RK order conditions of
 * every tableau constant, the reference's own analytic tests (final size, mass
 * conservation, SEIRS equilibrium, seasonal non-stationarity, shape/first-row/
 * save_step/sub-save semantics), closed forms, and fp64 scipy DOP853 ground-truth
 * fixtures under tests/golden/.
 */
#ifndef DYNODE_ORACLE_H
#define DYNODE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order/meaning as dyn_model_desc in include/dynode_hip.h. */
typedef struct dyo_model_desc {
    int32_t n_age;     /* A: bins on the contact axis (age, or age x risk flattened) */
    int32_t n_strain;  /* S */
    int32_t has_e;     /* exposed compartment e[A,S] */
    int32_t has_wane;  /* R -> S waning at rate omega (SEIRS); 0 = SIR-like */
    int32_t has_c;     /* cumulative-incidence compartment c[A,S] */
    int32_t n_wane;    /* W: Erlang stages of r (1 = the reference's single R) */
    int32_t normalize; /* 1: force of infection uses i_b/N_b ; 0: raw i_b */
    int32_t seasonal;  /* beta_t = beta*(1 + amp*sin(2*pi*t/period + phase)) */
    int32_t has_intro; /* externally introduced strains: x_{b,l} += pct_l * NormalPdf(t; time_l, scale_l)
                          for the ages b of intro_age_mask[l] (ode_model.md; config/strains.py:53-109) */
    int32_t n_vax_tiers; /* > 1: the contact axis enumerates (age, tier) groups, tier in the low bits of the
                            group index over KV = 2 or 4 slots; susceptibles move up one tier at the
                            spline rate nu_{group}(t) (include/dynode_hip.h has the full statement) */
    uint64_t intro_age_mask[8];
    int32_t n_vax_knots;
    int32_t family;       /* 0 = the s/e/i/r/c family above; 1 = SEIP with immune histories (ode_model.md), see
                             rhs_seip in dynode_oracle_impl.inc and "SEIP" in include/dynode_hip.h */
    int32_t seasonal_vax; /* SEIP: phi(t) = sin(2 pi (t + tau) / 730)^1000 moves the top vaccination tier down one */
    int32_t reserved;
} dyo_model_desc;

typedef struct dyo_solver_opts {
    int32_t method; /* 0 = Tsit5, 1 = Dopri5 */
    int32_t dtype;  /* 0 = float32, 1 = float64 */
    double rtol;
    double atol;
    int64_t max_steps;
    double constant_dt;    /* > 0: ConstantStepSize (odes.py:115-118) */
    const double *jump_ts; /* sorted discontinuity points (odes.py:120-131) */
    int32_t n_jump;
} dyo_solver_opts;

/* status codes per trajectory */
enum { DYO_OK = 0, DYO_MAX_STEPS = 1, DYO_NONFINITE = 2 };

int32_t dyo_state_dim(const dyo_model_desc *m);
int32_t dyo_param_dim(const dyo_model_desc *m);
int32_t dyo_n_compartments(const dyo_model_desc *m);
/* element offsets of each compartment inside the flat state, length n_comp+1 */
void dyo_compartment_offsets(const dyo_model_desc *m, int32_t *off);

/* RHS alone, fp64, for cross-checks against an einsum twin. */
void dyo_rhs_f64(const dyo_model_desc *m, double t, const double *y, const double *params,
                 const double *contact, double *dydt);

/*
 * Batched solve on the host.  All pointers are host pointers.  `y0`, `params`,
 * `contact`, `save_ts`, `ys_out` are float or double according to opts->dtype.
 *   y0      [D] or [B][D]          params [B][P]      contact [A][A] (lambda_a = sum_b C[a][b] x_b)
 *   save_ts [n_save]               save_mask [n_comp] (NULL = all)
 *   ys_out  [B][n_save][D_saved]   status / n_accept / n_reject [B]
 * Returns 0, or a negative value for an argument error.
 */
int dyo_solve_batch_cpu(const dyo_model_desc *m, const dyo_solver_opts *opts, const void *y0,
                        int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                        double t0, double t1, const void *save_ts, int32_t n_save,
                        const uint8_t *save_mask, void *ys_out, int32_t *status,
                        int32_t *n_accept, int32_t *n_reject, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
