// See surface_host.hpp.  Build with -ffp-contract=off.
#include "surface_host.hpp"

#include <algorithm>
#include <cmath>

namespace spd {
namespace {
const int kDaysInMonth[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};  // ncal365, model_control.f90:60
inline int days_before(int month /*1-based*/) {
    int s = 0;
    for (int m = 1; m < month; ++m) s += kDaysInMonth[m - 1];
    return s;
}
}  // namespace

// ------------------------------------------------------------------------------------------- calendar
void Calendar::set(int y, int mo, int d, int h, int mi) {
    year = y; month = mo; day = d; hour = h; minute = mi;
    month_idx = 1;
    update_forcing_params();
}

void Calendar::update_forcing_params() {  // all right-hand sides are default-real expressions in the reference
    imont1 = month;
    tmonth = static_cast<double>((static_cast<float>(day) - 0.5f) / static_cast<float>(kDaysInMonth[month - 1]));
    tyear = static_cast<double>((static_cast<float>(days_before(month) + day) - 0.5f) / static_cast<float>(365));
}

void Calendar::advance() {
    minute += 24 * 60 / 36;
    if (minute >= 60) {
        minute %= 60;
        hour += 1;
    }
    if (hour >= 24) {
        hour %= 24;
        day += 1;
    }
    if (year % 4 == 0 && month == 2) {
        if (day > 29) {
            day = 1;
            month += 1;
            month_idx += 1;
        }
    } else if (day > kDaysInMonth[month - 1]) {
        day = 1;
        month += 1;
        month_idx += 1;
    }
    if (month > 12) {
        month = 1;
        year += 1;
    }
    update_forcing_params();
}

// ------------------------------------------------------------------------------------------- interpolation
TimeInterp time_interp(const Calendar &c) {
    TimeInterp w{};
    const int imon = c.imont1;  // 1-based
    auto wrap = [](int m) { return m < 1 ? m + 12 : (m > 12 ? m - 12 : m); };
    w.m5[0] = wrap(imon - 2) - 1; w.m5[1] = wrap(imon - 1) - 1; w.m5[2] = imon - 1;
    w.m5[3] = wrap(imon + 1) - 1; w.m5[4] = wrap(imon + 2) - 1;
    const double tm = c.tmonth;
    const double c0 = static_cast<double>(1.0f / 12.0f);  // interpolation.f90:82
    const double t0 = c0 * tm, t1 = c0 * (1.0f - tm), t2 = 0.25f * tm * (1 - tm);
    w.w5[0] = -t1 + t2;
    w.w5[1] = -c0 + 8 * t1 - 6 * t2;
    w.w5[2] = 7 * c0 + 10 * t2;
    w.w5[3] = -c0 + 8 * t0 - 6 * t2;
    w.w5[4] = -t0 + t2;
    w.l0 = imon - 1;
    if (tm <= 0.5f) {
        w.l1 = (imon == 1 ? 12 : imon - 1) - 1;
        w.wlin = 0.5f - tm;
        w.a1 = c.month_idx - 1;
        w.wan = 0.5f - tm;
    } else {
        w.l1 = (imon == 12 ? 1 : imon + 1) - 1;
        w.wlin = tm - 0.5f;
        w.a1 = c.month_idx + 1;
        w.wan = tm - 0.5f;
    }
    w.a0 = c.month_idx;  // sst_anom(:, :, 0:n_months+1): plane index = month_idx
    return w;
}

// ------------------------------------------------------------------------------------------- daily solar forcing
// sin and cos of ONE argument: the compiled reference gets both from one `sincos` call (flang and gfortran merge the pair, neither
// keeps errno semantics), and glibc's sincos does not always return its sin() -- one ulp for the rzen of January 2.  The host code
// asks for the pair the same way; tests/golden/calendar.npz holds every day of the year (tests/test_calendar_host.py, bitwise).
namespace {
inline void sin_cos(double x, double &s, double &c) { ::sincos(x, &s, &c); }
}  // namespace
ZonalForcing zonal_average_fields(const HostTables &t, double tyear) {
    ZonalForcing z{};
    const float pih = std::asin(1.0f);  // asin(1.0) in fp32
    const double epssw = 0.020f, solc = 342.0f;
    const double alpha = static_cast<double>(4.0f * pih) * (tyear + static_cast<double>(10.0f / 365.0f));
    const double dalpha = 0.0;
    const double coz1 = 1.0f * std::max(0.0, std::cos(alpha - dalpha));
    const double coz2 = 1.8f, azen = 1.0f;
    const double rzen = -std::cos(alpha) * 23.45f * pih / 90.0f;
    const double fs0 = 6.0f;
    // solar(), shortwave_radiation.f90:277-322
    std::array<double, 48> topsr{};
    {
        const double csol = 4.0f * solc;
        const double pigr = static_cast<double>(2.0f * pih);
        const double al = 2.0f * pigr * tyear;
        double ca1, sa1;
        sin_cos(al, sa1, ca1);
        const double ca2 = ca1 * ca1 - sa1 * sa1, sa2 = 2.f * sa1 * ca1;
        const double ca3 = ca1 * ca2 - sa1 * sa2, sa3 = sa1 * ca2 + sa2 * ca1;
        const double decl = 0.006918f - 0.399912f * ca1 + 0.070257f * sa1 - 0.006758f * ca2 + 0.000907f * sa2 -
                            0.002697f * ca3 + 0.001480f * sa3;
        const double fdis = 1.000110f + 0.034221f * ca1 + 0.001280f * sa1 + 0.000719f * ca2 + 0.000077f * sa2;
        double cdecl, sdecl;
        sin_cos(decl, sdecl, cdecl);
        const double tdecl = sdecl / cdecl;
        const double csolp = csol / pigr;
        for (int j = 0; j < IL; ++j) {
            const double ch0 = std::min(1.0, std::max(-1.0, -tdecl * t.sia[j] / t.coa[j]));
            const double h0 = std::acos(ch0), sh0 = std::sin(h0);
            topsr[j] = csolp * fdis * (h0 * t.sia[j] * sdecl + sh0 * t.coa[j] * cdecl);
        }
    }
    double cos_rzen, sin_rzen;
    sin_cos(rzen, sin_rzen, cos_rzen);
    for (int j = 0; j < IL; ++j) {
        const double flat2 = 1.5f * (t.sia[j] * t.sia[j]) - 0.5f;
        z.flux_solar_in[j] = topsr[j];
        double o3u = 0.5f * epssw;
        double o3l = 0.4f * epssw * (1.0f + coz1 * t.sia[j] + coz2 * flat2);
        const double q = 1.0f - (t.coa[j] * cos_rzen + t.sia[j] * sin_rzen);
        z.zenit_correction[j] = 1.0f + azen * (q * q);
        z.flux_ozone_upper[j] = z.flux_solar_in[j] * o3u * z.zenit_correction[j];
        z.flux_ozone_lower[j] = z.flux_solar_in[j] * o3l * z.zenit_correction[j];
        z.stratospheric_correction[j] = std::max(fs0 - z.flux_solar_in[j], 0.0);
    }
    return z;
}

// ------------------------------------------------------------------------------------------- land / sea model constants
LandSeaConsts land_sea_consts(const HostTables &t) {
    LandSeaConsts k{};
    const double thrsh = 0.1f;  // land_model.f90:38, sea_model.f90:75
    k.thrsh = thrsh;
    k.one_minus_thrsh = 1.0f - thrsh;
    k.one = 1.0f;
    k.alb_thr = 0.4f;
    k.veg_low_weight = 0.8f;
    // soil moisture, land_model.f90:27-31, 89-94
    const double swcap = 0.30f, swwil = 0.17f;
    const int idep2 = 3;
    k.idep2 = idep2;
    k.swwil2 = idep2 * swwil;
    k.rsw = 1.0f / (swcap + idep2 * (swcap - swwil));
    // heat capacities and dissipation times, land_model.f90:119-146
    const double delt = 86400.0f / 36, tdland = 40.f;
    const double hcapl = 1.0f * 2.50e+6f, hcapli = 5.0f * 1.93e+6f;
    k.flandmin = static_cast<double>(1.f / 3.f);
    k.rhcapl[0] = delt / hcapl;
    k.rhcapl[1] = delt / hcapli;
    // sea_model.f90:57-72, 146-187 (global domain: the smoothed domain mask is 1 wherever there is enough sea)
    const float pih = std::asin(1.0f);
    const double crad = static_cast<double>(pih / 90.f);
    const double depth_ml = 60.f, dept0_ml = 40.f, depth_ice = 2.5f, dept0_ice = 1.5f, tdsst = 90.f, tdice = 30.0f;
    k.fseamin = static_cast<double>(1.f / 3.f);
    for (int d = 0; d < 2; ++d) {
        const double dmask = d;
        k.cdland[d] = dmask * tdland / (1.f + dmask * tdland);
        k.cdsea[d] = dmask * tdsst / (1.f + dmask * tdsst);
        k.cdice[d] = dmask * tdice / (1.f + dmask * tdice);
    }
    for (int j = 0; j < IL; ++j) {
        const double deglat_s = t.radang[j] * 90.0f / pih;  // sea_model.f90:108
        const double coslat = std::cos(crad * deglat_s);
        const double hcaps = 4.18e+6f * (depth_ml + (dept0_ml - depth_ml) * ((coslat * coslat) * coslat));
        const double hcapi = 1.93e+6f * (depth_ice + (dept0_ice - depth_ice) * (coslat * coslat));
        k.rhcaps_row[j] = delt / hcaps;
        k.rhcapi_row[j] = delt / hcapi;
    }
    return k;
}

void orog_land_sfc_drag(const std::vector<double> &phis0, std::vector<double> &forog) {
    const double grav = 9.81f, hdrag = 2000.0f;
    const double rhdrag = 1.0f / (grav * hdrag);
    forog.resize(phis0.size());
    for (size_t p = 0; p < phis0.size(); ++p) forog[p] = 1.0f + rhdrag * (1.0f - std::exp(-std::max(phis0[p], 0.0) * rhdrag));
}

}  // namespace spd
