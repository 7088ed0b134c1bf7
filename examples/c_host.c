/* A plain-C host of the outer boundary (include/pyspeedy_amd_driver.h): the call sequence of the reference's Python layer
 * through the f2py module speedy_driver (pyspeedy/speedy.py) -- modelstate_init, set_<v> of the 12 boundary fields,
 * create_datetime, controlparams_init, init, parallel_step once per model step, check, transform_spectral2grid, get_<v> --
 * for an ensemble of n independent containers, driven from `threads` host threads that each own a share of the containers
 * (the reference's parallel_step is `!f2py threadsafe`: the library gives its lock up while it waits for the GPU).
 * C99, no HIP header, no C++: the boundary is a C ABI.
 *
 *     c_host <bc.bin> <out.bin> <nsteps> [n_members = 2] [threads = 1]
 * bc.bin: the 12 boundary fields as raw doubles in the order of `names`, (96,48) or (96,48,12) each, Fortran order (as
 * examples/fortran_host.f90 reads them).  Member m gets its SST raised by 0.25 m K.  out.bin: t_grid (96,48,8) of every member.
 *
 *     gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_host.c -Lpyspeedy_amd -lpyspeedy_amd -lpthread \
 *         -Wl,-rpath,$PWD/pyspeedy_amd -o c_host
 */
#include <pthread.h>
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

struct share {
    int64_t *states, *controls;
    int32_t *codes;
    int n, nsteps, failed;
    char msg[256]; /* spd_last_error() is per thread: the worker keeps its own message */
};

static void *step_share(void *arg) {
    struct share *s = (struct share *)arg;
    for (int it = 0; it < s->nsteps && !s->failed; ++it) {
        if (spd_parallel_step(s->states, s->controls, s->codes, s->n) != SPD_OK) {
            s->failed = 1;
            snprintf(s->msg, sizeof(s->msg), "%s", spd_last_error());
        }
        for (int i = 0; i < s->n && !s->failed; ++i)
            if (s->codes[i] != 0) {
                s->failed = 1;
                snprintf(s->msg, sizeof(s->msg), "member %d left the accepted range (code %d) at step %d", i, (int)s->codes[i], it);
            }
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: c_host <bc.bin> <out.bin> <nsteps> [n_members] [threads]\n");
        return 2;
    }
    const int nsteps = atoi(argv[3]);
    const int n = argc > 4 ? atoi(argv[4]) : 2, threads = argc > 5 ? atoi(argv[5]) : 1;
    if (n < 1 || n > MAX_MEMBERS || threads < 1 || threads > n) {
        fprintf(stderr, "c_host: 1 <= threads <= n_members <= %d\n", MAX_MEMBERS);
        return 2;
    }
    int64_t states[MAX_MEMBERS], controls[MAX_MEMBERS], d_start, d_end;
    int32_t codes[MAX_MEMBERS], code;
    check(spd_create_datetime(1982, 1, 1, 0, 0, &d_start), "create_datetime");
    check(spd_create_datetime(1982, 1, 4, 0, 0, &d_end), "create_datetime");
    const size_t plane = (size_t)IX * IL;
    double *field = (double *)malloc(12 * plane * sizeof(double));
    for (int m = 0; m < n; ++m) {
        check(spd_modelstate_init(&states[m]), "modelstate_init");
        check(spd_controlparams_init(&controls[m], d_start, d_end), "controlparams_init");
        FILE *f = fopen(argv[1], "rb");
        if (!f || !field) {
            fprintf(stderr, "c_host: cannot read %s\n", argv[1]);
            return 1;
        }
        for (int i = 0; i < NFIELDS; ++i) {
            const size_t count = plane * (size_t)planes[i];
            if (fread(field, sizeof(double), count, f) != count) {
                fprintf(stderr, "c_host: %s is too short\n", argv[1]);
                return 1;
            }
            if (strcmp(names[i], "sst12") == 0)
                for (size_t k = 0; k < count; ++k) field[k] += 0.25 * m;
            check(spd_set(states[m], names[i], field, count * sizeof(double)), names[i]);
        }
        fclose(f);
        check(spd_init(states[m], controls[m], &code), "init");
        if (code != 0) {
            fprintf(stderr, "c_host: init returned %d\n", (int)code);
            return 1;
        }
    }
    /* every thread steps its own block of containers; the blocks never share a container */
    pthread_t tid[MAX_MEMBERS];
    struct share sh[MAX_MEMBERS];
    for (int t = 0, first = 0; t < threads; ++t) {
        const int count = n / threads + (t < n % threads ? 1 : 0);
        sh[t].states = states + first;
        sh[t].controls = controls + first;
        sh[t].codes = codes + first;
        sh[t].n = count;
        sh[t].nsteps = nsteps;
        sh[t].failed = 0;
        sh[t].msg[0] = 0;
        first += count;
        if (threads == 1) step_share(&sh[t]);
        else if (pthread_create(&tid[t], NULL, step_share, &sh[t]) != 0) return 1;
    }
    for (int t = 0; t < threads; ++t) {
        if (threads > 1) pthread_join(tid[t], NULL);
        if (sh[t].failed) {
            fprintf(stderr, "c_host: a step failed (%s)\n", sh[t].msg);
            return 1;
        }
    }
    int32_t y, mo, d, h, mi, month_idx, ymdhm[5], alive, members;
    check(spd_controlparams_get_model_datetime(controls[n - 1], ymdhm, &month_idx), "get_model_datetime");
    check(spd_get_datetime(d_end, &y, &mo, &d, &h, &mi), "get_datetime");
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
    printf("members %d threads %d steps %d model date %04d-%02d-%02d %02d:%02d members in the first device model %d\n", n, threads,
           nsteps, (int)ymdhm[0], (int)ymdhm[1], (int)ymdhm[2], (int)ymdhm[3], (int)ymdhm[4], (int)members);
    for (int m = 0; m < n; ++m) {
        check(spd_modelstate_close(states[m]), "modelstate_close");
        check(spd_controlparams_close(controls[m]), "controlparams_close");
    }
    check(spd_close_datetime(d_start), "close_datetime");
    check(spd_close_datetime(d_end), "close_datetime");
    free(field);
    free(t_grid);
    return 0;
}
