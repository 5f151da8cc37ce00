/*
 * dynode_oracle.c -- CPU restatement of DynODE's simulate() path (TEST INFRASTRUCTURE).
 * See dynode_oracle.h for scope, citations and the "parity unpinned" statement.
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "dynode_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct dyo_tableau {
    double c[7];
    double a[7][7];
    double berr[7]; /* b - bhat */
} dyo_tableau;

/* Tsitouras 2011 5(4) pair, 7 stages, FSAL.  Every constant is checked against the
 * RK order conditions in tests/test_tableau.py (b: order 5, b-berr: order 4). */
static const dyo_tableau TSIT5 = {
    {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0},
    {{0},
     {0.161},
     {-0.008480655492356989, 0.335480655492357},
     {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
     {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
     {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
      -0.028269050394068383},
     {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
      2.324710524099774}},
    {0.00178001105222577714, 0.0008164344596567469, -0.007880878010261995, 0.1447110071732629,
     -0.5823571654525552, 0.45808210592918697, -0.015151515151515152}};

/* Dormand-Prince 1980 5(4) pair, FSAL. */
static const dyo_tableau DOPRI5 = {
    {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0},
    {{0},
     {1.0 / 5},
     {3.0 / 40, 9.0 / 40},
     {44.0 / 45, -56.0 / 15, 32.0 / 9},
     {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
     {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
     {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}},
    {35.0 / 384 - 5179.0 / 57600, 0.0, 500.0 / 1113 - 7571.0 / 16695, 125.0 / 192 - 393.0 / 640,
     -2187.0 / 6784 + 92097.0 / 339200, 11.0 / 84 - 187.0 / 2100, -1.0 / 40}};

/* Shampine 1986 midpoint weights for the Dopri5 quartic dense output. */
static const double DOPRI5_CMID[7] = {6025192743.0 / 30085553152.0 / 2,
                                      0.0,
                                      51252292925.0 / 65400821598.0 / 2,
                                      -2691868925.0 / 45128329728.0 / 2,
                                      187940372067.0 / 1594534317056.0 / 2,
                                      -1776094331.0 / 19743644256.0 / 2,
                                      11237099.0 / 235043384.0 / 2};

int32_t dyo_n_compartments(const dyo_model_desc *m) {
    if (m->family == 1) return 4; /* s e i c */
    return 3 + (m->has_e ? 1 : 0) + (m->has_c ? 1 : 0);
}

/* SEIP: groups = (age, immune history) pairs, per group s[K1][M1] and e, i, c[K1][L] */
static int seip_groups(const dyo_model_desc *m) { return m->n_age << m->n_strain; }
static int seip_tiers(const dyo_model_desc *m) { return m->n_vax_tiers > 1 ? m->n_vax_tiers : 1; }

void dyo_compartment_offsets(const dyo_model_desc *m, int32_t *off) {
    if (m->family == 1) {
        const int G = seip_groups(m), K1 = seip_tiers(m);
        off[0] = 0;
        off[1] = G * K1 * m->n_wane;
        for (int c = 2; c <= 4; ++c) off[c] = off[c - 1] + G * K1 * m->n_strain;
        return;
    }
    const int A = m->n_age, AS = m->n_age * m->n_strain;
    int n = 0, pos = 0;
    off[n++] = pos;
    pos += A; /* s */
    if (m->has_e) {
        off[n++] = pos;
        pos += AS;
    }
    off[n++] = pos;
    pos += AS; /* i */
    off[n++] = pos;
    pos += AS * m->n_wane; /* r */
    if (m->has_c) {
        off[n++] = pos;
        pos += AS;
    }
    off[n] = pos;
}

int32_t dyo_state_dim(const dyo_model_desc *m) {
    if (m->family == 1) return seip_groups(m) * seip_tiers(m) * (m->n_wane + 3 * m->n_strain);
    return m->n_age * (1 + m->n_strain * ((m->has_e ? 1 : 0) + 1 + m->n_wane + (m->has_c ? 1 : 0)));
}

int32_t dyo_param_dim(const dyo_model_desc *m) {
    if (m->family == 1) {
        /* beta gamma sigma [L] | omega [M1] | (intro time scale pct [L]) | (amp phase period) | (tau) | pop [A] | sus [H][K1][M1][L] |
         * spline [A][K1][4 + 2 nk] */
        const int L = m->n_strain, K1 = seip_tiers(m);
        return 3 * L + m->n_wane + (m->has_intro ? 3 * L : 0) + (m->seasonal ? 3 : 0) + (m->seasonal_vax ? 1 : 0) + m->n_age +
               (1 << L) * K1 * m->n_wane * L + m->n_age * K1 * (4 + 2 * m->n_vax_knots);
    }
    return m->n_strain * (2 + (m->has_e ? 1 : 0) + (m->has_wane ? 1 : 0) + (m->has_intro ? 3 : 0)) +
           (m->seasonal ? 3 : 0) +
           (m->n_vax_tiers > 1 ? m->n_age * (m->n_strain + 4 + 2 * m->n_vax_knots) : 0);
}

/* tableau accessors so tests can check the order conditions of what is compiled in */
const double *dyo_tableau_ptr(int32_t method, int32_t which) {
    const dyo_tableau *tb = method == 1 ? &DOPRI5 : &TSIT5;
    switch (which) {
    case 0: return tb->c;
    case 1: return &tb->a[0][0];
    case 2: return tb->berr;
    case 3: return DOPRI5_CMID;
    default: return NULL;
    }
}

#define REAL double
#define SFX f64
#define SIN sin
#define EXP exp
#define SQRT sqrt
#define POW pow
#define FABS fabs
#define NEXTAFTER nextafter
#include "dynode_oracle_impl.inc"
#undef REAL
#undef SFX
#undef SIN
#undef EXP
#undef SQRT
#undef POW
#undef FABS
#undef NEXTAFTER

#define REAL float
#define SFX f32
#define SIN sinf
#define EXP expf
#define SQRT sqrtf
#define POW powf
#define FABS fabsf
#define NEXTAFTER nextafterf
#include "dynode_oracle_impl.inc"
#undef REAL
#undef SFX
#undef SIN
#undef EXP
#undef SQRT
#undef POW
#undef FABS
#undef NEXTAFTER

void dyo_tsit5_dense_weights_f64(double theta, double *b7) { tsit5_bt_f64(theta, b7); }

void dyo_rhs_f64(const dyo_model_desc *m, double t, const double *y, const double *params,
                 const double *contact, double *dydt) {
    double *scratch = (double *)malloc(sizeof(double) * (size_t)m->n_age * (2 * m->n_strain + 2));
    rhs_any_f64(m, t, y, params, contact, dydt, scratch);
    free(scratch);
}

static int check_model(const dyo_model_desc *m) {
    if (!m) return -1;
    if (m->n_age < 1 || m->n_strain < 1 || m->n_wane < 1) return -2;
    if (m->n_wane > 1 && !m->has_wane) return -2;
    if (m->family != 0 && m->family != 1) return -2;
    if (m->family == 1 && (m->n_strain > 4 || m->n_vax_tiers > 4 || m->n_vax_knots < 0 || m->n_vax_knots > 4 ||
                           m->normalize))
        return -2;
    return 0;
}

int dyo_solve_batch_cpu(const dyo_model_desc *m, const dyo_solver_opts *o, const void *y0,
                        int32_t y0_is_batched, const void *params, const void *contact, int64_t B,
                        double t0, double t1, const void *save_ts, int32_t n_save,
                        const uint8_t *save_mask, void *ys_out, int32_t *status, int32_t *n_accept,
                        int32_t *n_reject, int32_t n_threads) {
    int rc = check_model(m);
    if (rc) return rc;
    if (!o || !y0 || !params || !contact || !status || !n_accept || !n_reject) return -1;
    if (B < 0 || n_save < 0 || (n_save > 0 && (!save_ts || !ys_out))) return -3;
    if (o->method != 0 && o->method != 1) return -4;
    if (o->dtype != 0 && o->dtype != 1) return -4;
    if (!(o->constant_dt > 0.0) && (!(o->rtol > 0.0) || !(o->atol > 0.0))) return -5;
    if (o->max_steps < 1 || !(t1 >= t0)) return -5;
    if (o->n_jump < 0 || (o->n_jump > 0 && !o->jump_ts)) return -6;
    if (B == 0) return 0;

    const int ncomp = dyo_n_compartments(m);
    int32_t off[8];
    dyo_compartment_offsets(m, off);
    const int D = off[ncomp];
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(D > 0 ? D : 1));
    if (!sel) return -12;
    int n_sel = 0;
    for (int c = 0; c < ncomp; ++c)
        if (!save_mask || save_mask[c])
            for (int j = off[c]; j < off[c + 1]; ++j) sel[n_sel++] = j;

    if (o->dtype == 1)
        rc = solve_batch_f64(m, o, (const double *)y0, y0_is_batched, (const double *)params,
                             (const double *)contact, B, t0, t1, (const double *)save_ts, n_save,
                             sel, n_sel, (double *)ys_out, status, n_accept, n_reject, n_threads);
    else
        rc = solve_batch_f32(m, o, (const float *)y0, y0_is_batched, (const float *)params,
                             (const float *)contact, B, t0, t1, (const float *)save_ts, n_save, sel,
                             n_sel, (float *)ys_out, status, n_accept, n_reject, n_threads);
    free(sel);
    return rc;
}
