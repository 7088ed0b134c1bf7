// C ABI of pyspeedy_amd (include/pyspeedy_amd.h): context lifecycle, table upload, argument checking and
// kernel dispatch.  No numerics live here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pyspeedy_amd.h"
#include "context.hpp"
#include "device_tables.hpp"
#include "surface_host.hpp"
#include "vertical_consts.hpp"

namespace spd {
hipError_t run_spec2grid(const DeviceTables &T, int stage, const double *src, double *dst, int kcos, int nfields,
                         hipStream_t stream);
hipError_t run_grid2spec(const DeviceTables &T, int stage, const double *src, double *dst, int prescale, int nfields,
                         hipStream_t stream);
hipError_t run_vort2vel(const DeviceTables &T, const double *vor, const double *div, double *ucos, double *vcos,
                        int nfields, hipStream_t s);
hipError_t run_vel2vort(const DeviceTables &T, const double *ucos, const double *vcos, double *vor, double *div,
                        int nfields, hipStream_t s);
hipError_t run_gradient(const DeviceTables &T, const double *psi, double *psdx, double *psdy, int nfields, hipStream_t s);
hipError_t run_scale(const double *in, double *out, const double *table, double sign, int nfields, hipStream_t s);
hipError_t run_physics(const DeviceTables &T, const spd_physics_args &a, int nmembers, int fp32, hipStream_t s);
}  // namespace spd

using namespace spd;

static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int spd_set_error(int code, const std::string &msg) { return fail(code, msg); }

static int hip_fail(hipError_t e, const char *what) {
    return fail(SPD_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define SPD_HIP(call)                                  \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            (void)hipGetLastError(); /* reported here, once: not again by the next launch's hipGetLastError() */ \
            return hip_fail(e_, #call);                \
        }                                              \
    } while (0)

static int upload(spd_context *c, const double *src, size_t n, const double **dst) {
    void *p = nullptr;
    SPD_HIP(hipMalloc(&p, n * sizeof(double)));
    c->allocations.push_back(p);
    SPD_HIP(hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice));
    *dst = static_cast<const double *>(p);
    return SPD_OK;
}

// kernel-friendly polynomial layouts (device_tables.hpp, transforms.hip)
static std::vector<double> make_pinv(const HostTables &h) {
    // [n = 32][lane = m*12 + jq][2]: latitude pairs 2jq, 2jq+1; zero outside the triangle (nsh2, legendre.f90:73)
    std::vector<double> p(static_cast<size_t>(NX) * MX * 12 * 2, 0.0);
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m)
            for (int jq = 0; jq < 12; ++jq)
                for (int q = 0; q < 2; ++q)
                    if (m + n <= TRUNC + 1)
                        p[((static_cast<size_t>(n) * MX + m) * 12 + jq) * 2 + q] = h.poly[m + MX * (n + NX * (2 * jq + q))];
    return p;
}

// Direct-Legendre work list: one lane per (m, parity, two consecutive valid n of that parity), m-major.  Only the
// coefficients the reference fills (n <= trunc, m + n <= trunc + 1: nsh2, legendre.f90:206-217) get a lane, so the
// kernel streams 279 lanes x 24 latitude pairs x 16 B instead of the full 31 x 32 rectangle.
struct DirLane {
    int m, par, na, nb;  // nb = -1: the lane owns a single coefficient
};
static std::vector<DirLane> dir_lanes() {
    std::vector<DirLane> lanes;
    for (int m = 0; m < MX; ++m)
        for (int par = 0; par < 2; ++par) {
            std::vector<int> ns;
            for (int n = par; n <= TRUNC && m + n <= TRUNC + 1; n += 2) ns.push_back(n);
            for (size_t i = 0; i < ns.size(); i += 2) lanes.push_back({m, par, ns[i], i + 1 < ns.size() ? ns[i + 1] : -1});
        }
    return lanes;
}

static std::vector<double> make_pdir(const HostTables &h, const std::vector<DirLane> &lanes, int stride) {
    // [j = 24][lane (stride)][2]: the polynomials of the lane's two coefficients at latitude pair j
    std::vector<double> p(static_cast<size_t>(IY) * stride * 2, 0.0);
    for (int j = 0; j < IY; ++j)
        for (size_t l = 0; l < lanes.size(); ++l) {
            const DirLane &d = lanes[l];
            p[(static_cast<size_t>(j) * stride + l) * 2] = h.poly[d.m + MX * (d.na + NX * j)];
            if (d.nb >= 0) p[(static_cast<size_t>(j) * stride + l) * 2 + 1] = h.poly[d.m + MX * (d.nb + NX * j)];
        }
    return p;
}

// per-lane metadata for the kernel: {LDS positions of re / im in a compact Fourier row, parity, output indices}
static std::vector<int> make_dirmeta(const std::vector<DirLane> &lanes, int stride) {
    std::vector<int> meta(static_cast<size_t>(stride) * 4, -1);
    for (size_t l = 0; l < lanes.size(); ++l) {
        const DirLane &d = lanes[l];
        const int pr = d.m == 0 ? 0 : 2 * d.m - 1, pi = d.m == 0 ? 61 : 2 * d.m;  // pos_re / pos_im of transforms.hip
        meta[4 * l + 0] = pr | (pi << 8);
        meta[4 * l + 1] = d.par;
        meta[4 * l + 2] = d.na * MX + d.m;
        meta[4 * l + 3] = d.nb >= 0 ? d.nb * MX + d.m : -1;
    }
    return meta;
}

// The column kernel carries the vertical-structure tables as compile-time constants (vertical_consts.hpp, generated by
// tools/gen_vertical_consts.py from these very host tables).  Whoever changes the levels, or the arithmetic that makes the tables,
// and forgets to regenerate the header is told here, bit for bit, before a single kernel runs.
static int verify_vertical_consts(const HostTables &h) {
    const DynHostTables d(h);
    struct Row {
        const char *name;
        const double *have, *want;
        int n;
    };
    const Row rows[] = {{"fsg", vc::fsg, h.fsg.data(), 8},       {"dhs", vc::dhs, h.dhs.data(), 8},       {"sigl", vc::sigl, h.sigl.data(), 8},
                        {"sigh", vc::sigh, h.sigh.data(), 9},    {"grdsig", vc::grdsig, h.grdsig.data(), 8}, {"grdscp", vc::grdscp, h.grdscp.data(), 8},
                        {"wvi", vc::wvi, h.wvi.data(), 16},      {"dhsr", vc::dhsr, h.dhsr.data(), 8},    {"fsgr", vc::fsgr, h.fsgr.data(), 8},
                        {"tref", vc::tref, d.tref.data(), 8},    {"tref3", vc::tref3, d.tref3.data(), 8}};
    for (const Row &r : rows)
        if (std::memcmp(r.have, r.want, sizeof(double) * r.n) != 0)
            return fail(SPD_E_ARG, std::string("csrc/vertical_consts.hpp does not hold the table '") + r.name +
                                       "' this library builds: run tools/gen_vertical_consts.py and rebuild");
    return SPD_OK;
}

extern "C" {

const char *spd_version(void) { return "pyspeedy_amd 0.1 (gfx950)"; }
const char *spd_last_error(void) { return g_last_error.c_str(); }

int spd_create(spd_handle *out, int device) {
    if (!out) return fail(SPD_E_ARG, "spd_create: out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(SPD_E_DEVICE, std::string("spd_create: no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e) +
                                      ", " + std::to_string(ndev) + " devices)");
    if (device < 0 || device >= ndev) return fail(SPD_E_ARG, "spd_create: device index out of range");
    SPD_HIP(hipSetDevice(device));
    spd_context *c = new spd_context();
    c->device = device;
    const HostTables &h = c->host;
    DeviceTables &d = c->dev;
    int rc = verify_vertical_consts(h);
    if (rc != SPD_OK) {
        delete c;
        return rc;
    }
    auto up = [&](const double *src, size_t n, const double **dst) {
        if (rc == SPD_OK) rc = upload(c, src, n, dst);
    };
    const std::vector<DirLane> lanes = dir_lanes();
    d.ndir = static_cast<int>(lanes.size());
    d.dir_stride = (d.ndir + 63) / 64 * 64;
    const std::vector<double> pinv = make_pinv(h), pdir = make_pdir(h, lanes, d.dir_stride);
    const std::vector<int> dirmeta = make_dirmeta(lanes, d.dir_stride);
    up(pinv.data(), pinv.size(), &d.pinv);
    up(pdir.data(), pdir.size(), &d.pdir);
    {   // int metadata travels through the same uploader, two ints per double slot
        std::vector<double> raw(dirmeta.size() / 2);
        std::memcpy(raw.data(), dirmeta.data(), dirmeta.size() * sizeof(int));
        const double *dev = nullptr;
        up(raw.data(), raw.size(), &dev);
        d.dirmeta = reinterpret_cast<const int *>(dev);
    }
    up(h.work.data(), 96, &d.work);
    up(h.cosgr.data(), 48, &d.cosgr);
    up(h.cosgr2.data(), 48, &d.cosgr2);
    up(h.wt.data(), 24, &d.wt);
    up(h.el2.data(), NSPEC, &d.el2);
    up(h.elm2.data(), NSPEC, &d.elm2);
    up(h.trfilt.data(), NSPEC, &d.trfilt);
    up(h.gradx.data(), MX, &d.gradx);
    up(h.gradym.data(), NSPEC, &d.gradym);
    up(h.gradyp.data(), NSPEC, &d.gradyp);
    up(h.uvdx.data(), NSPEC, &d.uvdx);
    up(h.uvdym.data(), NSPEC, &d.uvdym);
    up(h.uvdyp.data(), NSPEC, &d.uvdyp);
    up(h.vddym.data(), NSPEC, &d.vddym);
    up(h.vddyp.data(), NSPEC, &d.vddyp);
    up(h.fband.data(), h.fband.size(), &d.fband);
    up(h.coa.data(), 48, &d.coa);
    {   // fp32 copies of the two physics tables (cfg 5), packed two floats per double slot of the same uploader
        std::vector<float> f32(h.fband.size() + 48);
        for (size_t i = 0; i < h.fband.size(); ++i) f32[i] = static_cast<float>(h.fband[i]);
        for (int j = 0; j < 48; ++j) f32[h.fband.size() + j] = static_cast<float>(h.coa[j]);
        std::vector<double> raw((f32.size() + 1) / 2);
        std::memcpy(raw.data(), f32.data(), f32.size() * sizeof(float));
        const double *dev = nullptr;
        up(raw.data(), raw.size(), &dev);
        d.fband32 = reinterpret_cast<const float *>(dev);
        d.coa32 = d.fband32 ? d.fband32 + h.fband.size() : nullptr;
    }
    if (rc != SPD_OK) {
        spd_destroy(c);
        return rc;
    }
    d.fft_scale = static_cast<double>(1.0f / static_cast<float>(IX));
    for (int k = 0; k < 8; ++k) {
        d.fsg[k] = h.fsg[k]; d.dhs[k] = h.dhs[k]; d.sigl[k] = h.sigl[k];
        d.grdsig[k] = h.grdsig[k]; d.grdscp[k] = h.grdscp[k];
    }
    for (int k = 0; k < 9; ++k) d.sigh[k] = h.sigh[k];
    for (int k = 0; k < 16; ++k) d.wvi[k] = h.wvi[k];
    *out = c;
    return SPD_OK;
}

int spd_destroy(spd_handle h) {
    if (!h) return SPD_OK;
    (void)hipSetDevice(h->device);
    for (void *p : h->allocations) (void)hipFree(p);
    for (const spd_context::IdleBlock &b : h->idle_blocks) (void)hipFree(b.base);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->probe_buf) (void)hipFree(h->probe_buf);
    delete h;
    return SPD_OK;
}

int spd_device(spd_handle h) { return h ? h->device : SPD_E_ARG; }

long spd_get_table_host(spd_handle h, const char *name, double *buf, size_t buf_elems) {
    if (!name) return fail(SPD_E_ARG, "spd_get_table_host: null name");
    // h == NULL: device-less query (host table construction only), used by the CPU test tier
    static const HostTables deviceless;
    const HostTables &t = h ? h->host : deviceless;
    std::vector<double> tmp;
    const double *src = nullptr;
    size_t n = 0;
    auto arr = [&](const auto &a) { src = a.data(); n = a.size(); };
    const std::string s(name);
    if (s == "hsg") arr(t.hsg); else if (s == "sigh") arr(t.sigh); else if (s == "dhs") arr(t.dhs);
    else if (s == "fsg") arr(t.fsg); else if (s == "dhsr") arr(t.dhsr); else if (s == "fsgr") arr(t.fsgr);
    else if (s == "sigl") arr(t.sigl); else if (s == "grdsig") arr(t.grdsig); else if (s == "grdscp") arr(t.grdscp);
    else if (s == "wvi") arr(t.wvi); else if (s == "radang") arr(t.radang); else if (s == "coriol") arr(t.coriol);
    else if (s == "sia") arr(t.sia); else if (s == "coa") arr(t.coa); else if (s == "cosgr") arr(t.cosgr);
    else if (s == "cosgr2") arr(t.cosgr2); else if (s == "sia_half") arr(t.sia_half);
    else if (s == "coa_half") arr(t.coa_half); else if (s == "wt") arr(t.wt); else if (s == "epsi") arr(t.epsi);
    else if (s == "repsi") arr(t.repsi); else if (s == "poly") arr(t.poly); else if (s == "work") arr(t.work);
    else if (s == "el2") arr(t.el2); else if (s == "elm2") arr(t.elm2); else if (s == "el4") arr(t.el4);
    else if (s == "trfilt") arr(t.trfilt); else if (s == "gradym") arr(t.gradym); else if (s == "gradyp") arr(t.gradyp);
    else if (s == "uvdx") arr(t.uvdx); else if (s == "uvdym") arr(t.uvdym); else if (s == "uvdyp") arr(t.uvdyp);
    else if (s == "vddym") arr(t.vddym); else if (s == "vddyp") arr(t.vddyp); else if (s == "gradx") arr(t.gradx);
    else if (s == "fband") arr(t.fband);
    else if (s == "tref" || s == "tref3") {  // implicit.f90:71-78 (independent of the time step)
        const DynHostTables d(t);
        const auto &a = s == "tref" ? d.tref : d.tref3;
        tmp.assign(a.begin(), a.end());
        arr(tmp);
    }
    else if (s == "cpol") { tmp = t.cpol(); arr(tmp); }
    else if (s == "nsh2") { tmp.assign(t.nsh2.begin(), t.nsh2.end()); arr(tmp); }
    else if (s == "ifac") { tmp.assign(t.ifac.begin(), t.ifac.end()); arr(tmp); }
    else return fail(SPD_E_ARG, "spd_get_table_host: unknown table '" + s + "'");
    if (buf) {
        if (buf_elems < n) return fail(SPD_E_SIZE, "spd_get_table_host: buffer too small for '" + s + "'");
        std::memcpy(buf, src, n * sizeof(double));
    }
    return static_cast<long>(n);
}

int spd_calendar_walk(int year, int month, int day, int hour, int minute, int nsteps, int32_t *ymdhm, int32_t *month_idx,
                      int32_t *imont1, double *tmonth, double *tyear) {
    if (nsteps < 0 || month < 1 || month > 12 || day < 1 || day > 31) return fail(SPD_E_ARG, "spd_calendar_walk: bad start date or step count");
    Calendar c;
    c.set(year, month, day, hour, minute);
    for (int s = 0;; ++s) {
        if (ymdhm) {
            int32_t *row = ymdhm + 5 * static_cast<size_t>(s);
            row[0] = c.year; row[1] = c.month; row[2] = c.day; row[3] = c.hour; row[4] = c.minute;
        }
        if (month_idx) month_idx[s] = c.month_idx;
        if (imont1) imont1[s] = c.imont1;
        if (tmonth) tmonth[s] = c.tmonth;
        if (tyear) tyear[s] = c.tyear;
        if (s == nsteps) break;
        c.advance();
    }
    return SPD_OK;
}

int spd_daily_forcing_host(double tyear, double *out) {
    if (!out) return fail(SPD_E_ARG, "spd_daily_forcing_host: null buffer");
    static const HostTables deviceless;
    const ZonalForcing z = zonal_average_fields(deviceless, tyear);
    const std::array<double, 48> *rows[5] = {&z.flux_solar_in, &z.flux_ozone_upper, &z.flux_ozone_lower, &z.zenit_correction,
                                              &z.stratospheric_correction};
    for (int r = 0; r < 5; ++r) std::memcpy(out + 48 * r, rows[r]->data(), 48 * sizeof(double));
    return SPD_OK;
}

}  // extern "C"

// ---- dispatch helpers -------------------------------------------------------------------------------
static int check(spd_handle h, int nfields, const char *fn, std::initializer_list<const void *> ptrs) {
    if (!h) return fail(SPD_E_ARG, std::string(fn) + ": null handle");
    if (nfields < 0) return fail(SPD_E_ARG, std::string(fn) + ": negative field count");
    if (nfields > 0)
        for (const void *p : ptrs)
            if (!p) return fail(SPD_E_ARG, std::string(fn) + ": null device pointer");
    return SPD_OK;
}

static int done(hipError_t e, const char *fn) { return e == hipSuccess ? SPD_OK : hip_fail(e, fn); }

static int get_scratch(spd_handle h, size_t bytes, double **out) {
    std::lock_guard<std::mutex> lock(h->scratch_mutex);
    if (bytes > h->scratch_bytes) {
        // growing the scratch is not stream-ordered: callers that capture graphs must size it first by
        // issuing one un-captured call with the largest batch.
        SPD_HIP(hipDeviceSynchronize());
        if (h->scratch) SPD_HIP(hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
        void *p = nullptr;
        SPD_HIP(hipMalloc(&p, bytes));
        h->scratch = static_cast<double *>(p);
        h->scratch_bytes = bytes;
    }
    *out = h->scratch;
    return SPD_OK;
}

extern "C" {

int spd_spec2grid(spd_handle h, const double *spec, double *grid, int kcos, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_spec2grid", {spec, grid})) return rc;
    return done(run_spec2grid(h->dev, 0, spec, grid, kcos, nfields, static_cast<hipStream_t>(stream)), "spd_spec2grid");
}

int spd_grid2spec(spd_handle h, const double *grid, double *spec, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_grid2spec", {grid, spec})) return rc;
    return done(run_grid2spec(h->dev, 0, grid, spec, 0, nfields, static_cast<hipStream_t>(stream)), "spd_grid2spec");
}

int spd_legendre_inv(spd_handle h, const double *spec, double *four, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_legendre_inv", {spec, four})) return rc;
    return done(run_spec2grid(h->dev, 1, spec, four, 1, nfields, static_cast<hipStream_t>(stream)), "spd_legendre_inv");
}

int spd_legendre(spd_handle h, const double *four, double *spec, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_legendre", {four, spec})) return rc;
    return done(run_grid2spec(h->dev, 1, four, spec, 0, nfields, static_cast<hipStream_t>(stream)), "spd_legendre");
}

int spd_fourier_inv(spd_handle h, const double *four, double *grid, int kcos, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_fourier_inv", {four, grid})) return rc;
    return done(run_spec2grid(h->dev, 2, four, grid, kcos, nfields, static_cast<hipStream_t>(stream)), "spd_fourier_inv");
}

int spd_fourier(spd_handle h, const double *grid, double *four, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_fourier", {grid, four})) return rc;
    return done(run_grid2spec(h->dev, 2, grid, four, 0, nfields, static_cast<hipStream_t>(stream)), "spd_fourier");
}

int spd_vort2vel(spd_handle h, const double *vor, const double *div, double *ucos, double *vcos, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_vort2vel", {vor, div, ucos, vcos})) return rc;
    return done(run_vort2vel(h->dev, vor, div, ucos, vcos, nfields, static_cast<hipStream_t>(stream)), "spd_vort2vel");
}

int spd_vel2vort(spd_handle h, const double *ucos, const double *vcos, double *vor, double *div, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_vel2vort", {ucos, vcos, vor, div})) return rc;
    return done(run_vel2vort(h->dev, ucos, vcos, vor, div, nfields, static_cast<hipStream_t>(stream)), "spd_vel2vort");
}

int spd_grid_vel2vort(spd_handle h, const double *ug, const double *vg, double *vor, double *div, int kcos, int nfields,
                      void *stream) {
    if (int rc = check(h, nfields, "spd_grid_vel2vort", {ug, vg, vor, div})) return rc;
    if (nfields == 0) return SPD_OK;
    double *tmp = nullptr;
    const size_t per = static_cast<size_t>(nfields) * 2 * NSPEC;  // doubles per spectral batch
    if (int rc = get_scratch(h, 2 * per * sizeof(double), &tmp)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int prescale = (kcos == 2) ? 1 : 2;  // spectral.f90:229-243
    hipError_t e = run_grid2spec(h->dev, 0, ug, tmp, prescale, nfields, s);
    if (e == hipSuccess) e = run_grid2spec(h->dev, 0, vg, tmp + per, prescale, nfields, s);
    if (e == hipSuccess) e = run_vel2vort(h->dev, tmp, tmp + per, vor, div, nfields, s);
    return done(e, "spd_grid_vel2vort");
}

int spd_gradient(spd_handle h, const double *psi, double *psdx, double *psdy, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_gradient", {psi, psdx, psdy})) return rc;
    return done(run_gradient(h->dev, psi, psdx, psdy, nfields, static_cast<hipStream_t>(stream)), "spd_gradient");
}

int spd_laplacian(spd_handle h, const double *in, double *out, int inverse, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_laplacian", {in, out})) return rc;
    return done(run_scale(in, out, inverse ? h->dev.elm2 : h->dev.el2, -1.0, nfields, static_cast<hipStream_t>(stream)),
                "spd_laplacian");
}

int spd_truncate(spd_handle h, double *field, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_truncate", {field})) return rc;
    return done(run_scale(field, field, h->dev.trfilt, 1.0, nfields, static_cast<hipStream_t>(stream)), "spd_truncate");
}

int spd_grid_filter(spd_handle h, const double *fg1, double *fg2, int nfields, void *stream) {
    if (int rc = check(h, nfields, "spd_grid_filter", {fg1, fg2})) return rc;
    if (nfields == 0) return SPD_OK;
    double *tmp = nullptr;
    const size_t per = static_cast<size_t>(nfields) * 2 * NSPEC;
    if (int rc = get_scratch(h, per * sizeof(double), &tmp)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = run_grid2spec(h->dev, 0, fg1, tmp, 0, nfields, s);
    if (e == hipSuccess) e = run_scale(tmp, tmp, h->dev.trfilt, 1.0, nfields, s);  // spectral.f90:308-313
    if (e == hipSuccess) e = run_spec2grid(h->dev, 0, tmp, fg2, 1, nfields, s);
    return done(e, "spd_grid_filter");
}

int spd_physics(spd_handle h, const spd_physics_args *a, int nmembers, void *stream) {
    if (!h || !a) return fail(SPD_E_ARG, "spd_physics: null argument");
    if (nmembers < 0) return fail(SPD_E_ARG, "spd_physics: negative member count");
    if (nmembers == 0) return SPD_OK;
    const void *required[] = {a->ug, a->vg, a->tg, a->qg, a->phig, a->pslg, a->utend, a->vtend, a->ttend, a->qtend,
                              a->fmask_land, a->phis0, a->forog, a->sst_am, a->alb_land, a->alb_sea, a->snowc,
                              a->land_temp, a->soil_avail_water, a->precnv, a->precls, a->cbmf, a->slrd, a->slr, a->olr,
                              a->slru, a->ustr, a->vstr, a->shf, a->evap, a->hfluxn, a->rad_st4a, a->rad_flux, a->tt_rsw,
                              a->rad_tau2, a->rad_strat_corr, a->tsr, a->ssrd, a->ssr, a->qcloud_equiv};
    for (const void *p : required)
        if (!p) return fail(SPD_E_ARG, "spd_physics: a required device pointer is null");
    if (a->compute_shortwave) {
        const void *sw[] = {a->flux_solar_in, a->flux_ozone_upper, a->flux_ozone_lower, a->zenit_correction,
                            a->stratospheric_correction, a->alb_surface};
        for (const void *p : sw)
            if (!p) return fail(SPD_E_ARG, "spd_physics: shortwave forcing pointer is null on a shortwave step");
    }
    return done(run_physics(h->dev, *a, nmembers, a->fp32, static_cast<hipStream_t>(stream)), "spd_physics");
}

}  // extern "C"
