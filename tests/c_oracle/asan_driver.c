/* Small C driver for the oracle built with -fsanitize=address,undefined (make -C oracle asan): the shapes the parity
 * suite leans on -- 2-age SIR (cfg 1), the 8 x 4 multi-strain SEIRS with and without the waning chain and seasonal
 * forcing, sub-saved compartments, discontinuity points, a constant step, a vaccinated model and a small SEIP model --
 * in both precisions and both methods, with ragged batches and an n_save of 1.  Any out-of-bounds access or undefined
 * operation in the restatement aborts the run; the test (tests/test_oracle.py) only checks the exit code and the
 * conservation line printed at the end. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dynode_oracle.h"

static int run(const dyo_model_desc *m, int dtype, int method, int B, double t1, int n_save, const uint8_t *mask,
               double constant_dt, int n_jump, double *sum_out) {
    const int D = dyo_state_dim(m), P = dyo_param_dim(m), A = m->n_age;
    const size_t w = dtype ? 8 : 4;
    double *y0d = calloc((size_t)B * D, 8), *pd = calloc((size_t)B * P, 8), *Cd = calloc((size_t)A * A, 8), *tsd = calloc(n_save, 8);
    int32_t off[8];
    dyo_compartment_offsets(m, off);
    const int nc = dyo_n_compartments(m);
    int d_saved = 0;
    for (int c = 0; c < nc; ++c)
        if (!mask || mask[c]) d_saved += off[c + 1] - off[c];
    for (int b = 0; b < B; ++b) {
        for (int v = 0; v < D; ++v) y0d[(size_t)b * D + v] = v < off[1] ? 100.0 + v : (v < off[2] ? 1.0 : 0.0);
        for (int q = 0; q < P; ++q) pd[(size_t)b * P + q] = 0.05 + 0.01 * ((q * 7 + b) % 13);
        if (m->seasonal) pd[(size_t)b * P + P - 1] = 365.0; /* period */
    }
    if (m->family == 1) { /* SEIP rows carry populations, a table and splines: keep every entry small and positive */
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < P; ++q) pd[(size_t)b * P + q] = 0.02 + 0.001 * ((q * 5 + b) % 17);
    }
    for (int a = 0; a < A * A; ++a) Cd[a] = 0.5 / A + (a % (A + 1) == 0 ? 0.5 : 0.0);
    for (int j = 0; j < n_save; ++j) tsd[j] = n_save > 1 ? t1 * j / (n_save - 1) : t1;
    void *y0 = malloc((size_t)B * D * w), *p = malloc((size_t)B * P * w), *C = malloc((size_t)A * A * w), *ts = malloc((size_t)n_save * w);
    void *out = malloc((size_t)B * n_save * (d_saved ? d_saved : 1) * w);
    if (dtype) {
        memcpy(y0, y0d, (size_t)B * D * 8); memcpy(p, pd, (size_t)B * P * 8); memcpy(C, Cd, (size_t)A * A * 8); memcpy(ts, tsd, (size_t)n_save * 8);
    } else {
        for (size_t i = 0; i < (size_t)B * D; ++i) ((float *)y0)[i] = (float)y0d[i];
        for (size_t i = 0; i < (size_t)B * P; ++i) ((float *)p)[i] = (float)pd[i];
        for (size_t i = 0; i < (size_t)A * A; ++i) ((float *)C)[i] = (float)Cd[i];
        for (int i = 0; i < n_save; ++i) ((float *)ts)[i] = (float)tsd[i];
    }
    double jumps[2] = {0.3 * t1, 0.6 * t1};
    dyo_solver_opts o = {method, dtype, 1e-5, 1e-6, 100000, constant_dt, n_jump ? jumps : NULL, n_jump};
    int32_t *st = calloc(3 * (size_t)B, 4);
    const int rc = dyo_solve_batch_cpu(m, &o, y0, 1, p, C, B, 0.0, t1, ts, n_save, mask, out, st, st + B, st + 2 * B, 3);
    double sum = 0.0;
    for (size_t i = 0; i < (size_t)B * n_save * d_saved; ++i) {
        const double v = dtype ? ((double *)out)[i] : ((float *)out)[i];
        if (isfinite(v)) sum += v;
    }
    int bad = rc;
    for (int b = 0; b < B; ++b) bad |= st[b] > 2 || st[b] < 0;
    *sum_out = sum;
    free(y0d); free(pd); free(Cd); free(tsd); free(y0); free(p); free(C); free(ts); free(out); free(st);
    return bad;
}

int main(void) {
    dyo_model_desc sir2 = {0}, ms = {0}, w8 = {0}, vax = {0}, seip = {0};
    sir2.n_age = 2; sir2.n_strain = 1; sir2.n_wane = 1; sir2.normalize = 1;
    ms.n_age = 8; ms.n_strain = 4; ms.has_e = ms.has_wane = ms.has_c = 1; ms.n_wane = 1; ms.normalize = 1; ms.seasonal = 1;
    w8 = ms; w8.n_wane = 8; w8.seasonal = 0;
    vax.n_age = 4; vax.n_strain = 1; vax.n_wane = 1; vax.n_vax_tiers = 2; vax.n_vax_knots = 2;
    seip.family = 1; seip.n_age = 3; seip.n_strain = 2; seip.has_e = seip.has_wane = seip.has_c = 1; seip.n_wane = 3;
    seip.n_vax_tiers = 2; seip.n_vax_knots = 1; seip.seasonal_vax = 1;
    const uint8_t keep_r[3] = {0, 0, 1}, keep_ic[5] = {0, 0, 1, 0, 1};
    double s, total = 0.0;
    int bad = 0, runs = 0;
    for (int dtype = 0; dtype < 2; ++dtype)
        for (int method = 0; method < 2; ++method) {
            bad |= run(&sir2, dtype, method, 5, 365.0, 366, NULL, 0.0, 0, &s); total += s; ++runs;
            bad |= run(&sir2, dtype, method, 3, 100.0, 101, keep_r, 0.0, 2, &s); total += s; ++runs;
            bad |= run(&sir2, dtype, method, 1, 50.0, 1, NULL, 0.25, 0, &s); total += s; ++runs;
            bad |= run(&ms, dtype, method, 7, 120.0, 25, NULL, 0.0, 0, &s); total += s; ++runs;
            bad |= run(&ms, dtype, method, 2, 120.0, 13, keep_ic, 0.0, 1, &s); total += s; ++runs;
            bad |= run(&w8, dtype, method, 3, 90.0, 10, NULL, 0.0, 0, &s); total += s; ++runs;
            bad |= run(&vax, dtype, method, 4, 60.0, 7, NULL, 0.5, 0, &s); total += s; ++runs;
            bad |= run(&seip, dtype, method, 3, 40.0, 5, NULL, 0.5, 0, &s); total += s; ++runs;
        }
    printf("asan driver: %d runs, bad=%d, checksum %.6e\n", runs, bad, total);
    return bad ? 1 : 0;
}
