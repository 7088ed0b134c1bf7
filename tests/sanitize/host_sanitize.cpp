// AddressSanitizer / UndefinedBehaviorSanitizer driver for the host-side C++ of the product (pyspeedy_amd/csrc/tables.cpp,
// surface_host.cpp): builds every table, the dt-dependent implicit tables for the three time steps of a run, walks the
// calendar over two years (interpolation weights, zonal forcing every day) and preprocesses a synthetic set of boundary
// fields.  Compiled by tests/test_sanitizers.py with g++ -fsanitize=address,undefined -fno-sanitize-recover=all.
#include <cmath>
#include <cstdio>
#include <vector>

#include "surface_host.hpp"
#include "tables.hpp"

using namespace spd;

int main() {
    HostTables t;
    double acc = 0.0;
    for (double v : t.cpol()) acc += v;
    for (double v : t.fband) acc += v;
    for (double v : t.work) acc += v;
    DynHostTables d(t);
    const double delt = 86400.0 / 36;
    for (double dt : {0.5 * delt, delt, 2 * delt}) {
        d.set_time_step(t, dt);
        for (double v : d.xj) acc += v;
        acc += d.dmp1[100] + d.elz[991];
    }
    Calendar cal;
    cal.set(1983, 12, 30, 0, 0);  // across a year end and a leap February
    for (int step = 0; step < 36 * 800; ++step) {
        cal.advance();
        const TimeInterp w = time_interp(cal);
        for (int i = 0; i < 5; ++i)
            if (w.m5[i] < 0 || w.m5[i] > 11) return 2;
        if (w.l0 < 0 || w.l0 > 11 || w.l1 < 0 || w.l1 > 11) return 3;
        acc += w.w5[0] + w.wlin + w.wan;
        if (step % 36 == 0) {
            const ZonalForcing z = zonal_average_fields(t, cal.tyear);
            acc += z.flux_solar_in[0] + z.stratospheric_correction[47];
        }
    }
    const int NG = 96 * 48;
    SurfaceFields sf;
    auto fill = [&](std::vector<double> &v, int planes, double lo, double hi) {
        v.resize(static_cast<size_t>(NG) * planes);
        for (size_t i = 0; i < v.size(); ++i) v[i] = lo + (hi - lo) * (0.5 + 0.5 * std::sin(0.37 * i));
    };
    fill(sf.fmask_orig, 1, 0.0, 1.0); fill(sf.alb0, 1, 0.05, 0.6); fill(sf.veg_high, 1, 0.0, 1.0); fill(sf.veg_low, 1, 0.0, 1.0);
    fill(sf.stl12, 12, 230.0, 310.0); fill(sf.snowd12, 12, 0.0, 400.0); fill(sf.soil_wc_l1, 12, 0.0, 0.5);
    fill(sf.soil_wc_l2, 12, 0.0, 0.5); fill(sf.sst12, 12, 270.0, 303.0); fill(sf.sea_ice_frac12, 12, 0.0, 1.0);
    fill(sf.sst_anom, 5, -1.0, 1.0);
    for (size_t i = 0; i < sf.stl12.size(); i += 7) sf.stl12[i] = 9.97e36;  // missing values, as in the boundary files
    for (size_t i = 3; i < sf.sst12.size(); i += 11) sf.sst12[i] = 9.97e36;
    land_sea_init(t, sf);
    for (double v : sf.fmask_land) acc += v;
    for (double v : sf.rhcapl) acc += v;
    for (double v : sf.cdsea) acc += v;
    std::vector<double> phis0(NG, 1500.0), forog;
    orog_land_sfc_drag(phis0, forog);
    acc += forog[17];
    std::printf("host sanitize ok %.6e\n", acc);
    return std::isfinite(acc) ? 0 : 1;
}
