/* The one-process ensemble through the library's EXTENSIONS of the outer boundary (include/pyspeedy_amd_driver.h), from plain C:
 * n containers batched from the start and placed in blocks on all GPUs the process can see (spd_modelstate_init_ensemble_on),
 * the boundary file read ONCE into container 0 and handed to the others device to device (spd_broadcast_boundary: one RCCL
 * broadcast across GPUs, local copies on each), every device model initialised in one pass (spd_init_ensemble), and the time
 * loop in the overlapped form (spd_parallel_step_begin / _end: the range check of step k is collected after step k + 1 has
 * been enqueued) for its first half and as ONE stretch (spd_parallel_steps_begin / _end) for its second.  Same arguments, same input and same output as examples/c_host.c, which does the same run with the
 * reference's own call sequence -- and the same bits in the result.
 *
 *     c_ensemble_host <bc.bin> <out.bin> <nsteps> [n_members = 2]
 *     gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_ensemble_host.c -Lpyspeedy_amd -lpyspeedy_amd \
 *         -Wl,-rpath,$PWD/pyspeedy_amd -o c_ensemble_host
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pyspeedy_amd.h"
#include "pyspeedy_amd_driver.h"

enum { IX = 96, IL = 48, KX = 8, NFIELDS = 12, MAX_MEMBERS = 64 };
static const char *const names[NFIELDS] = {"orog",   "fmask_orig", "alb0",       "veg_high",   "veg_low", "stl12",
                                           "snowd12", "soil_wc_l1", "soil_wc_l2", "soil_wc_l3", "sst12",   "sea_ice_frac12"};
static const int planes[NFIELDS] = {1, 1, 1, 1, 1, 12, 12, 12, 12, 12, 12, 12};

static void check(int rc, const char *what) {
    if (rc == SPD_OK) return;
    fprintf(stderr, "FAILED: %s -> %d %s\n", what, rc, spd_last_error());
    exit(1);
}

static void all_fine(const int32_t *codes, int n, int step) {
    for (int i = 0; i < n; ++i)
        if (codes[i] != 0) {
            fprintf(stderr, "c_ensemble_host: member %d returned code %d at step %d\n", i, (int)codes[i], step);
            exit(1);
        }
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: c_ensemble_host <bc.bin> <out.bin> <nsteps> [n_members]\n");
        return 2;
    }
    const int nsteps = atoi(argv[3]), n = argc > 4 ? atoi(argv[4]) : 2;
    if (n < 1 || n > MAX_MEMBERS || nsteps < 1) return 2;
    int64_t states[MAX_MEMBERS], controls[MAX_MEMBERS], d_start, d_end, token, next_token;
    int32_t codes[MAX_MEMBERS], ndev = 0, peer, local, collective, alive, members;
    check(spd_device_count(&ndev), "device_count");
    check(spd_create_datetime(1982, 1, 1, 0, 0, &d_start), "create_datetime");
    check(spd_create_datetime(1982, 1, 4, 0, 0, &d_end), "create_datetime");
    check(spd_modelstate_init_ensemble_on(states, n, ndev < n ? ndev : n), "modelstate_init_ensemble_on");
    for (int m = 0; m < n; ++m) check(spd_controlparams_init(&controls[m], d_start, d_end), "controlparams_init");
    /* the boundary file goes into container 0 only ... */
    const size_t plane = (size_t)IX * IL;
    double *field = (double *)malloc(12 * plane * sizeof(double)), *sst = (double *)malloc(12 * plane * sizeof(double));
    FILE *f = fopen(argv[1], "rb");
    if (!f || !field || !sst) {
        fprintf(stderr, "c_ensemble_host: cannot read %s\n", argv[1]);
        return 1;
    }
    for (int i = 0; i < NFIELDS; ++i) {
        const size_t count = plane * (size_t)planes[i];
        if (fread(field, sizeof(double), count, f) != count) return 1;
        check(spd_set(states[0], names[i], field, count * sizeof(double)), names[i]);
        if (strcmp(names[i], "sst12") == 0) memcpy(sst, field, count * sizeof(double));
    }
    fclose(f);
    /* ... and reaches the others on the devices */
    check(spd_broadcast_boundary(states, n, 0), "broadcast_boundary");
    check(spd_broadcast_boundary_stats(&peer, &local, &collective), "broadcast_boundary_stats");
    for (int m = 1; m < n; ++m) { /* (member m with its SST raised by 0.25 m K, as in c_host.c) */
        for (size_t k = 0; k < 12 * plane; ++k) field[k] = sst[k] + 0.25 * m;
        check(spd_set(states[m], "sst12", field, 12 * plane * sizeof(double)), "sst12");
    }
    check(spd_init_ensemble(states, controls, codes, n), "init_ensemble");
    all_fine(codes, n, 0);
    /* the time loop: the first half two steps in flight (a host that looks at every step's codes), the second half handed over as
     * ONE stretch (a host that will not look at the state before it is over: every step's range check is recorded on the device) */
    const int single = nsteps / 2, stretch = nsteps - single;
    if (single > 0) {
        check(spd_parallel_step_begin(states, controls, n, &token), "parallel_step_begin");
        for (int it = 1; it < single; ++it) {
            check(spd_parallel_step_begin(states, controls, n, &next_token), "parallel_step_begin");
            check(spd_parallel_step_end(token, codes), "parallel_step_end");
            all_fine(codes, n, it);
            token = next_token;
        }
        check(spd_parallel_step_end(token, codes), "parallel_step_end");
        all_fine(codes, n, single);
    }
    if (stretch > 0) {
        int32_t done[MAX_MEMBERS];
        check(spd_parallel_steps_begin(states, controls, n, stretch, &token), "parallel_steps_begin");
        check(spd_parallel_steps_end(token, codes, done), "parallel_steps_end");
        all_fine(codes, n, nsteps);
        for (int m = 0; m < n; ++m)
            if (done[m] != stretch) return 1;
    }
    int32_t ymdhm[5], month_idx, code;
    check(spd_controlparams_get_model_datetime(controls[n - 1], ymdhm, &month_idx), "get_model_datetime");
    check(spd_driver_stats(states[0], &alive, &members), "driver_stats");
    FILE *out = fopen(argv[2], "wb");
    double *t_grid = (double *)malloc(plane * KX * sizeof(double));
    if (!out || !t_grid) return 1;
    for (int m = 0; m < n; ++m) {
        check(spd_check(states[m], &code), "check");
        if (code != 0) return 1;
        check(spd_transform_spectral2grid(states[m]), "transform_spectral2grid");
        check(spd_get(states[m], "t_grid", t_grid, plane * KX * sizeof(double)), "get_t_grid");
        fwrite(t_grid, sizeof(double), plane * KX, out);
    }
    fclose(out);
    printf("members %d devices %d steps %d model date %04d-%02d-%02d %02d:%02d members in the first device model %d; boundary: "
           "%d GPUs reached collectively, %d peer copies, %d local copies\n", n, (int)ndev, nsteps, (int)ymdhm[0], (int)ymdhm[1],
           (int)ymdhm[2], (int)ymdhm[3], (int)ymdhm[4], (int)members, (int)collective, (int)peer, (int)local);
    for (int m = 0; m < n; ++m) {
        check(spd_modelstate_close(states[m]), "modelstate_close");
        check(spd_controlparams_close(controls[m]), "controlparams_close");
    }
    check(spd_close_datetime(d_start), "close_datetime");
    check(spd_close_datetime(d_end), "close_datetime");
    free(field);
    free(sst);
    free(t_grid);
    return 0;
}
