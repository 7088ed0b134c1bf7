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
    const LandSeaConsts k = land_sea_consts(t);
    for (int j = 0; j < 48; ++j) acc += k.rhcaps_row[j] + k.rhcapi_row[j];
    acc += k.rsw + k.swwil2 + k.cdland[1] + k.cdsea[1] + k.cdice[1] + k.rhcapl[0] + k.rhcapl[1];
    if (k.cdland[0] != 0.0 || !(k.one_minus_thrsh < 1.0)) return 4;
    std::vector<double> phis0(NG, 1500.0), forog;
    orog_land_sfc_drag(phis0, forog);
    acc += forog[17];
    std::printf("host sanitize ok %.6e\n", acc);
    return std::isfinite(acc) ? 0 : 1;
}
