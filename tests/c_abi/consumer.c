/* A consumer of libdynode_hip.so written in C: no Python, no torch -- device buffers from the HIP
 * runtime, one dyn_solve_batch call, results printed as text (tests/test_gpu_parity.py compares them
 * with the oracle).  Model: the reference's examples/sir.py literal (1 bin, y0 = (0.9, 0.1, 0),
 * beta = 2/7, gamma = 1/7) for B slightly different betas, float64, daily save over 50 days.
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ consumer.c -I/opt/rocm/include -I<repo>/include
 *        -L<repo>/dynode_amd/lib -ldynode_hip -L/opt/rocm/lib -lamdhip64 -o consumer */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "dynode_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    enum { B = 5, NSAVE = 51, D = 3, P = 2 };
    dyn_model_desc m = {0};
    m.n_age = 1; m.n_strain = 1; m.n_wane = 1; m.normalize = 1;
    dyn_solver_opts o = {0};
    o.method = DYN_TSIT5; o.dtype = DYN_F64; o.rtol = 1e-5; o.atol = 1e-6; o.max_steps = 1000000;
    if (dyn_abi_version() != DYN_ABI_VERSION || dyn_state_dim(&m) != D || dyn_param_dim(&m) != P) return 3;

    double y0[D] = {0.9, 0.1, 0.0}, params[B][P], contact[1] = {1.0}, ts[NSAVE];
    for (int b = 0; b < B; ++b) { params[b][0] = (2.0 + 0.1 * b) / 7.0; params[b][1] = 1.0 / 7.0; }
    for (int j = 0; j < NSAVE; ++j) ts[j] = (double)j;

    double *d_y0, *d_p, *d_c, *d_ts, *d_out;
    int32_t *d_stat;
    CHECK(hipMalloc((void **)&d_y0, sizeof y0));
    CHECK(hipMalloc((void **)&d_p, sizeof params));
    CHECK(hipMalloc((void **)&d_c, sizeof contact));
    CHECK(hipMalloc((void **)&d_ts, sizeof ts));
    CHECK(hipMalloc((void **)&d_out, sizeof(double) * B * NSAVE * D));
    CHECK(hipMalloc((void **)&d_stat, sizeof(int32_t) * 3 * B));
    CHECK(hipMemcpy(d_y0, y0, sizeof y0, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_p, params, sizeof params, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_c, contact, sizeof contact, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ts, ts, sizeof ts, hipMemcpyHostToDevice));

    int rc = dyn_solve_batch(&m, &o, d_y0, 0, d_p, d_c, B, 0.0, 50.0, d_ts, NSAVE, NULL, d_out, d_stat, d_stat + B,
                             d_stat + 2 * B, NULL);
    if (rc) { fprintf(stderr, "dyn_solve_batch: %d (%s)\n", rc, dyn_last_error()); return 4; }
    CHECK(hipDeviceSynchronize());

    static double out[B][NSAVE][D];
    int32_t stat[3][B];
    CHECK(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(stat, d_stat, sizeof stat, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        printf("traj %d status %d accepted %d rejected %d\n", b, stat[0][b], stat[1][b], stat[2][b]);
        for (int j = 0; j < NSAVE; j += 10) printf("%d %d %.17g %.17g %.17g\n", b, j, out[b][j][0], out[b][j][1], out[b][j][2]);
    }
    /* the same batch dispatched in reverse order (dyn_solve_batch_ordered): every byte of the output must be the same */
    {
        int32_t order[B], *d_order;
        void *d_out2;
        static double out2[B][NSAVE][D];
        for (int b = 0; b < B; ++b) order[b] = B - 1 - b;
        CHECK(hipMalloc((void **)&d_order, sizeof order));
        CHECK(hipMalloc(&d_out2, sizeof out2));
        CHECK(hipMemcpy(d_order, order, sizeof order, hipMemcpyHostToDevice));
        rc = dyn_solve_batch_ordered(&m, &o, d_y0, 0, d_p, d_c, B, 0.0, 50.0, d_ts, NSAVE, NULL, d_out2, d_stat, d_stat + B,
                                     d_stat + 2 * B, d_order, NULL);
        if (rc) { fprintf(stderr, "dyn_solve_batch_ordered: %d (%s)\n", rc, dyn_last_error()); return 6; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out2, d_out2, sizeof out2, hipMemcpyDeviceToHost));
        if (memcmp(out, out2, sizeof out) != 0) { fprintf(stderr, "ordered dispatch changed the output\n"); return 7; }
        printf("ordered dispatch identical\n");
    }
    /* work pulling (ABI 7, dyn_solver_opts.work_counter): a batch beyond the resident grid -- forced down to 8 waves here --
     * is integrated by lane groups that draw trajectories from a queue; every byte must equal the static launch, and the
     * two counter words must be zero again afterwards */
    {
        enum { B2 = 1500 };
        static double p2[B2][P], a[B2][NSAVE][D], b2[B2][NSAVE][D];
        double *d_p2, *d_a;
        int32_t *d_stat2, *d_work, work[2] = {-1, -1};
        for (int b = 0; b < B2; ++b) { p2[b][0] = (1.2 + 0.002 * b) / 7.0; p2[b][1] = 1.0 / (5.0 + 0.003 * b); }
        CHECK(hipMalloc((void **)&d_p2, sizeof p2));
        CHECK(hipMalloc((void **)&d_a, sizeof a));
        CHECK(hipMalloc((void **)&d_stat2, sizeof(int32_t) * 3 * B2));
        CHECK(hipMalloc((void **)&d_work, sizeof work));
        CHECK(hipMemset(d_work, 0, sizeof work));
        CHECK(hipMemcpy(d_p2, p2, sizeof p2, hipMemcpyHostToDevice));
        rc = dyn_solve_batch(&m, &o, d_y0, 0, d_p2, d_c, B2, 0.0, 50.0, d_ts, NSAVE, NULL, d_a, d_stat2, d_stat2 + B2, d_stat2 + 2 * B2, NULL);
        if (rc) { fprintf(stderr, "static launch: %d (%s)\n", rc, dyn_last_error()); return 8; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(a, d_a, sizeof a, hipMemcpyDeviceToHost));
        CHECK(hipMemset(d_a, 0xff, sizeof a));
        o.hints.pull_waves = 8;          /* ABI 9: dispatch hints travel in the call, not in the environment */
        o.work_counter = d_work;
        for (int rep = 0; rep < 2; ++rep) {     /* twice: the second launch finds the counters as the first one left them */
            rc = dyn_solve_batch(&m, &o, d_y0, 0, d_p2, d_c, B2, 0.0, 50.0, d_ts, NSAVE, NULL, d_a, d_stat2, d_stat2 + B2, d_stat2 + 2 * B2, NULL);
            if (rc) { fprintf(stderr, "work-pulling launch: %d (%s)\n", rc, dyn_last_error()); return 9; }
            CHECK(hipDeviceSynchronize());
        }
        o.work_counter = NULL;
        o.hints.pull_waves = 0;
        CHECK(hipMemcpy(b2, d_a, sizeof b2, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(work, d_work, sizeof work, hipMemcpyDeviceToHost));
        if (memcmp(a, b2, sizeof a) != 0) { fprintf(stderr, "work pulling changed the output\n"); return 10; }
        if (work[0] != 0 || work[1] != 0) { fprintf(stderr, "work counters left at %d %d\n", work[0], work[1]); return 11; }
        printf("work pulling identical\n");
    }
    /* an unsupported request comes back as an error code and a message, never as a crash */
    m.n_strain = 7;
    rc = dyn_solve_batch(&m, &o, d_y0, 0, d_p, d_c, B, 0.0, 50.0, d_ts, NSAVE, NULL, d_out, d_stat, d_stat + B, d_stat + 2 * B, NULL);
    printf("unsupported rc %d\n", rc);
    return rc == DYN_ERR_UNSUPPORTED ? 0 : 5;
}
