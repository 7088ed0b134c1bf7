// See tables.hpp.  Build with -ffp-contract=off: every operation here must round exactly once.
#include "tables.hpp"

#include <cmath>
#include <utility>

namespace spd {
namespace {

// 1-based (m, n) -> linear index of a Fortran (rows, *) array
inline int at2(int rows, int m, int n) { return (m - 1) + rows * (n - 1); }

void build_geometry(HostTables &t) {
    // geometry.f90:89 -- half levels given as default-real literals
    const float half_levels[9] = {0.000f, 0.050f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f};
    for (int k = 0; k <= KX; ++k) t.hsg[k] = t.sigh[k] = half_levels[k];
    for (int k = 0; k < KX; ++k) {
        const double lo = t.hsg[k], hi = t.hsg[k + 1];
        t.dhs[k] = hi - lo;                 // :94
        t.fsg[k] = 0.5 * (hi + lo);         // :95
        t.dhsr[k] = 0.5 / t.dhs[k];         // :100
        t.fsgr[k] = phc::akap / (2. * t.fsg[k]);
        t.sigl[k] = std::log(t.fsg[k]);     // :138
        t.grdsig[k] = phc::grav / (t.dhs[k] * phc::p0);
        t.grdscp[k] = t.grdsig[k] / phc::cp;
    }
    // geometry.f90:108-130.  Gaussian "latitudes" come from the fp32 asymptotic formula and are NOT
    // Newton-refined; the grid really sits at +-87.2165 deg, not +-87.1591 deg.
    const float pi_f = 3.141592654f, denom = static_cast<float>(IL) + 0.5f;
    for (int j = 0; j < IY; ++j) {
        const int north = IL - 1 - j;
        const float colat = pi_f * (static_cast<float>(j + 1) - 0.25f) / denom;
        const double s = static_cast<double>(std::cos(colat));  // cosf
        const double c = std::sqrt(1.0 - s * s);
        t.sia_half[j] = s;
        t.coa_half[j] = c;
        t.sia[j] = -s;
        t.sia[north] = s;
        t.coa[j] = t.coa[north] = c;
        t.radang[north] = std::asin(s);
        t.radang[j] = -t.radang[north];
        t.cosgr[j] = t.cosgr[north] = 1. / c;
        t.cosgr2[j] = t.cosgr2[north] = 1. / (c * c);
    }
    for (int j = 0; j < IL; ++j) t.coriol[j] = 2.0 * phc::omega * t.sia[j];
    // vertical interpolation weights, geometry.f90:148-154 (log(0.99) in fp32)
    for (int k = 0; k + 1 < KX; ++k) {
        t.wvi[k] = 1. / (t.sigl[k + 1] - t.sigl[k]);
        t.wvi[KX + k] = (std::log(t.sigh[k + 1]) - t.sigl[k]) * t.wvi[k];
    }
    t.wvi[KX - 1] = 0.;
    t.wvi[2 * KX - 1] = (static_cast<double>(std::log(0.99f)) - t.sigl[KX - 1]) * t.wvi[KX - 2];
}

// Gaussian weights (legendre.f90:224-257): Newton iteration on the TRUE nodes, weights halved (sum = 1).
// The iteration carries `prev` across nodes exactly like the reference's z1.
void build_gauss_weights(HostTables &t) {
    const int n = 2 * IY;
    const double eps = 2.220446049250313e-16;
    double prev = 2.0, dpoly = 0.0;
    for (int i = 1; i <= IY; ++i) {
        double z = std::cos(3.141592654 * (static_cast<double>(i) - 0.25) / (static_cast<double>(n) + 0.5));
        while (std::fabs(z - prev) > eps) {
            double pa = 1.0, pb = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double pc = pb;
                pb = pa;
                pa = ((2.0 * static_cast<double>(j) - 1.0) * z * pb - (static_cast<double>(j) - 1.0) * pc) / j;
            }
            dpoly = static_cast<double>(n) * (z * pa - pb) / (z * z - 1.0);
            prev = z;
            z = prev - pa / dpoly;
        }
        t.wt[i - 1] = 2.0 / ((1.0 - z * z) * (dpoly * dpoly));
    }
}

void build_legendre(HostTables &t) {
    build_gauss_weights(t);
    for (int n = 1; n <= NX; ++n) {  // legendre.f90:68-77
        int cnt = 0;
        for (int m = 1; m <= MX; ++m)
            if ((m - 1) + (n - 1) <= TRUNC + 1) cnt += 2;
        t.nsh2[n - 1] = cnt;
    }
    const int R = MX + 1;
    t.epsi.assign(R * (NX + 1), 0.0);
    t.repsi.assign(R * (NX + 1), 0.0);
    for (int m = 1; m <= MX + 1; ++m)
        for (int n = 1; n <= NX + 1; ++n) {  // legendre.f90:79-96: the squares are fp32
            const float fm = static_cast<float>(m - 1), fl = static_cast<float>(n + m - 2);
            const double emm2 = static_cast<double>(fm * fm), ell2 = static_cast<double>(fl * fl);
            double e = 0.0;
            if (n != NX + 1 && !(n == 1 && m == 1)) e = std::sqrt((ell2 - emm2) / (4.0 * ell2 - 1.0));
            t.epsi[at2(R, m, n)] = e;
            t.repsi[at2(R, m, n)] = (e > 0.) ? 1.0 / e : 0.0;
        }
    // associated Legendre functions by the three-term recursion, legendre.f90:260-307
    t.poly.assign(static_cast<size_t>(MX) * NX * IY, 0.0);
    std::vector<double> alp(R * NX);
    std::array<double, MX> diag_factor{};
    for (int m = 1; m <= MX; ++m) {
        const float fm = static_cast<float>(m);
        diag_factor[m - 1] = static_cast<double>(std::sqrt(0.5f * (2.0f * fm + 1.0f) / fm));  // sqrtf
    }
    const double tiny = 1.e-30f;
    for (int j = 1; j <= IY; ++j) {
        const double y = t.coa_half[j - 1], x = t.sia_half[j - 1];
        alp[at2(R, 1, 1)] = static_cast<double>(std::sqrt(0.5f));
        for (int m = 2; m <= MX + 1; ++m) alp[at2(R, m, 1)] = diag_factor[m - 2] * y * alp[at2(R, m - 1, 1)];
        for (int m = 1; m <= MX + 1; ++m) alp[at2(R, m, 2)] = (x * alp[at2(R, m, 1)]) * t.repsi[at2(R, m, 2)];
        for (int n = 3; n <= NX; ++n)
            for (int m = 1; m <= MX + 1; ++m)
                alp[at2(R, m, n)] = (x * alp[at2(R, m, n - 1)] - t.epsi[at2(R, m, n - 1)] * alp[at2(R, m, n - 2)]) *
                                    t.repsi[at2(R, m, n)];
        for (int n = 1; n <= NX; ++n)
            for (int m = 1; m <= MX; ++m) {
                double v = alp[at2(R, m, n)];
                if (std::fabs(v) <= tiny) v = 0.0;
                t.poly[(m - 1) + MX * ((n - 1) + NX * (j - 1))] = v;
            }
    }
}

// FFTPACK real-transform setup for N = 96 (fftpack.f90:1-67).  Factor search order 4,2,3,5 with the factor 2
// moved to the front gives ifac = [96, 4, 2, 4, 4, 3]; twiddles use the fp32 value of 2*pi.
void build_fft(HostTables &t) {
    const int n = IX;
    const int tryout[4] = {4, 2, 3, 5};
    int rest = n, nf = 0;
    for (int c = 0; c < 4 && rest > 1; ++c) {
        const int f = tryout[c];
        while (rest % f == 0) {
            ++nf;
            t.ifac[nf + 1] = f;
            rest /= f;
            if (f == 2 && nf != 1) {
                for (int q = nf; q >= 2; --q) t.ifac[q + 1] = t.ifac[q];
                t.ifac[2] = 2;
            }
        }
    }
    t.ifac[0] = n;
    t.ifac[1] = nf;
    const double two_pi = static_cast<double>(8.f * std::atan(1.f));  // atanf
    const double step = two_pi / n;
    int base = 0, l1 = 1;
    for (int pass = 0; pass + 1 < nf; ++pass) {
        const int ip = t.ifac[pass + 2], l2 = l1 * ip, ido = n / l2;
        int ld = 0;
        for (int j = 1; j < ip; ++j) {
            ld += l1;
            const double ang0 = ld * step;
            double fi = 0.;
            for (int i = 0; 2 * i + 3 <= ido; ++i) {
                fi += 1.;
                const double a = fi * ang0;
                t.work[base + 2 * i] = std::cos(a);
                t.work[base + 2 * i + 1] = std::sin(a);
            }
            base += ido;
        }
        l1 = l2;
    }
}

void build_spectral(HostTables &t) {
    auto mk = [] { return std::vector<double>(NSPEC, 0.0); };
    t.el2 = mk(); t.elm2 = mk(); t.el4 = mk(); t.trfilt = mk(); t.gradym = mk(); t.gradyp = mk();
    t.uvdx = mk(); t.uvdym = mk(); t.uvdyp = mk(); t.vddym = mk(); t.vddyp = mk();
    const double a = phc::rearth;
    const int R = MX + 1;
    for (int n = 1; n <= NX; ++n)
        for (int m = 1; m <= MX; ++m) {
            const int l = (m - 1) + (n - 1), k = at2(MX, m, n);
            t.el2[k] = static_cast<double>(static_cast<float>(l * (l + 1))) / (a * a);  // spectral.f90:76
            t.el4[k] = t.el2[k] * t.el2[k];
            t.trfilt[k] = (l <= TRUNC) ? 1.0 : 0.0;
            t.elm2[k] = (l == 0) ? 0.0 : 1.0 / t.el2[k];  // :85-87
            const double el1 = static_cast<double>(static_cast<float>(l));
            const double fm1 = static_cast<double>(static_cast<float>(m - 1));
            const double e_lo = t.epsi[at2(R, m, n)], e_hi = t.epsi[at2(R, m, n + 1)];
            if (n == 1) {  // :96-100
                t.gradx[m - 1] = fm1 / a;
                t.uvdx[k] = -a / static_cast<double>(static_cast<float>(m));
            } else {  // :102-105 (gradym(:,1) is never assigned by the reference; kept 0)
                t.uvdx[k] = -a * fm1 / (el1 * (el1 + 1));
                t.gradym[k] = (el1 - 1.0) * e_lo / a;
                t.uvdym[k] = -a * e_lo / el1;
                t.vddym[k] = (el1 + 1) * e_lo / a;
            }
            t.gradyp[k] = (el1 + 2.0) * e_hi / a;  // :107-109
            t.uvdyp[k] = -a * e_hi / (el1 + 1.0);
            t.vddyp[k] = el1 * e_hi / a;
        }
}

void build_fband(HostTables &t) {  // longwave_radiation.f90:208-232
    t.fband.assign(301 * 4, 0.0);
    auto fb = [&](int temp, int band) -> double & { return t.fband[(temp - 100) + 301 * (band - 1)]; };
    const double eps1 = 1.0 - static_cast<double>(0.05f);
    for (int T = 200; T <= 320; ++T) {
        // integer square, then fp32 arithmetic, then widened and scaled in fp64
        const float d2 = static_cast<float>((T - 247) * (T - 247)), d3 = static_cast<float>((T - 282) * (T - 282)),
                    d4 = static_cast<float>((T - 315) * (T - 315));
        fb(T, 2) = static_cast<double>(0.148f - 3.0e-6f * d2) * eps1;
        fb(T, 3) = static_cast<double>(0.356f - 5.2e-6f * d3) * eps1;
        fb(T, 4) = static_cast<double>(0.314f + 1.0e-5f * d4) * eps1;
        fb(T, 1) = eps1 - (fb(T, 2) + fb(T, 3) + fb(T, 4));
    }
    for (int b = 1; b <= 4; ++b) {
        for (int T = 100; T < 200; ++T) fb(T, b) = fb(200, b);
        for (int T = 321; T <= 400; ++T) fb(T, b) = fb(320, b);
    }
}

}  // namespace

HostTables::HostTables() {
    build_geometry(*this);
    build_legendre(*this);
    build_fft(*this);
    build_spectral(*this);
    build_fband(*this);
}

// ---------------------------------------------------------------------------------------------------------
// dynamics tables
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr double kGamma = 6.0f, kHscale = 7.5f, kHshum = 2.5f, kThd = 2.4f, kThdd = 2.4f, kThds = 12.0f, kAlph = 0.5f;

// LU decomposition with implicit scaling and partial pivoting, then column-by-column back substitution:
// the algorithm of matrix_inversion.f90 (Numerical Recipes ludcmp/lubksb), n = 8.
struct Lu8 {
    static constexpr int N = KX;
    double a[N * N];
    int piv[N];
    double &at(int i, int j) { return a[i + N * j]; }
    void factor() {
        const double tiny = 1.0e-20f;
        double scale[N];
        for (int i = 0; i < N; ++i) {
            double big = 0.;
            for (int j = 0; j < N; ++j) big = std::fabs(at(i, j)) > big ? std::fabs(at(i, j)) : big;
            scale[i] = 1. / big;
        }
        int imax = 0;
        for (int j = 0; j < N; ++j) {
            for (int i = 0; i < j; ++i) {
                double sum = at(i, j);
                if (i > 0) {
                    for (int k = 0; k < i; ++k) sum = sum - at(i, k) * at(k, j);
                    at(i, j) = sum;
                }
            }
            double big = 0.;
            for (int i = j; i < N; ++i) {
                double sum = at(i, j);
                if (j > 0) {
                    for (int k = 0; k < j; ++k) sum = sum - at(i, k) * at(k, j);
                    at(i, j) = sum;
                }
                const double merit = scale[i] * std::fabs(sum);
                if (merit >= big) {
                    imax = i;
                    big = merit;
                }
            }
            if (j != imax) {
                for (int k = 0; k < N; ++k) std::swap(at(imax, k), at(j, k));
                scale[imax] = scale[j];
            }
            piv[j] = imax;
            if (j != N - 1) {
                if (at(j, j) == 0) at(j, j) = tiny;
                const double inv = 1. / at(j, j);
                for (int i = j + 1; i < N; ++i) at(i, j) = at(i, j) * inv;
            }
        }
        if (at(N - 1, N - 1) == 0.) at(N - 1, N - 1) = tiny;
    }
    void solve(double *b) {
        int first = -1;
        for (int i = 0; i < N; ++i) {
            const int ll = piv[i];
            double sum = b[ll];
            b[ll] = b[i];
            if (first >= 0) {
                for (int j = first; j < i; ++j) sum = sum - at(i, j) * b[j];
            } else if (sum != 0) {
                first = i;
            }
            b[i] = sum;
        }
        for (int i = N - 1; i >= 0; --i) {
            double sum = b[i];
            for (int j = i + 1; j < N; ++j) sum = sum - at(i, j) * b[j];
            b[i] = sum / at(i, i);
        }
    }
};
}  // namespace

DynHostTables::DynHostTables(const HostTables &t) {
    dmp.assign(NSPEC, 0.0); dmpd.assign(NSPEC, 0.0); dmps.assign(NSPEC, 0.0);
    dmp1.assign(NSPEC, 0.0); dmp1d.assign(NSPEC, 0.0); dmp1s.assign(NSPEC, 0.0); elz.assign(NSPEC, 0.0);
    xj.assign(static_cast<size_t>(KX) * KX * (MX + NX + 1), 0.0);
    // horizontal_diffusion.f90:80-107
    const double hdiff = 1.f / (kThd * 3600.f), hdifd = 1.f / (kThdd * 3600.f), hdifs = 1.f / (kThds * 3600.f);
    const double rlap = static_cast<double>(1.f / static_cast<float>(TRUNC * (TRUNC + 1)));
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m) {
            const double twn = static_cast<double>(static_cast<float>(m + n));
            const double elap = (twn * (twn + 1.f) * rlap);
            const double elap4 = ((elap * elap) * elap) * elap;  // elap**npowhd, expanded sequentially like flang does
            dmp[m + MX * n] = hdiff * elap4;
            dmpd[m + MX * n] = hdifd * elap4;
            dmps[m + MX * n] = hdifs * elap;
        }
    const double rgam = phc::rgas * kGamma / (1000.f * phc::grav);
    const double qexp = kHscale / kHshum;
    for (int k = 1; k < KX; ++k) {
        tcorv[k] = std::pow(t.fsg[k], rgam);
        if (k > 1) qcorv[k] = std::pow(t.fsg[k], qexp);
    }
    // implicit.f90:71-78
    for (int k = 0; k < KX; ++k) {
        const double f = t.fsg[k] > 0.2f ? t.fsg[k] : static_cast<double>(0.2f);
        tref[k] = 288.f * std::pow(f, rgam);
        tref2[k] = phc::akap * tref[k];
        tref3[k] = t.fsgr[k] * tref[k];
    }
    // geopotential.f90:25-28 and :72-73
    for (int k = 0; k < KX; ++k) {
        xgeop1[k] = phc::rgas * std::log(t.hsg[k + 1] / t.fsg[k]);
        if (k != KX - 1) xgeop2[k + 1] = phc::rgas * std::log(t.fsg[k + 1] / t.hsg[k + 1]);
    }
    for (int k = 1; k < KX - 1; ++k)
        geo_corf[k] = xgeop1[k] * 0.5f * std::log(t.hsg[k + 1] / t.fsg[k]) / std::log(t.fsg[k + 1] / t.fsg[k - 1]);
}

void DynHostTables::set_time_step(const HostTables &t, double step) {  // implicit.f90:83-218
    dt = step;
    for (int i = 0; i < NSPEC; ++i) {
        dmp1[i] = 1.f / (1.f + dmp[i] * dt);
        dmp1d[i] = 1.f / (1.f + dmpd[i] * dt);
        dmp1s[i] = 1.f / (1.f + dmps[i] * dt);
    }
    const double xi = dt * kAlph, a2 = phc::rearth * phc::rearth;
    const double xxi = xi / a2;
    for (int k = 0; k < KX; ++k) dhsx[k] = xi * t.dhs[k];
    for (int n = 0; n < NX; ++n)
        for (int m = 0; m < MX; ++m)
            elz[m + MX * n] = static_cast<double>(static_cast<float>(m + n) * static_cast<float>(m + n + 1)) * xxi;
    auto M = [](std::array<double, 64> &a, int k, int k1) -> double & { return a[k + KX * k1]; };
    std::array<double, 64> xa{}, xb{}, xe{}, ya{};
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1) M(ya, k, k1) = -phc::akap * tref[k] * t.dhs[k1];
    for (int k = 1; k < KX; ++k)
        M(xa, k, k - 1) = 0.5f * (phc::akap * tref[k] / t.fsg[k] - (tref[k] - tref[k - 1]) / t.dhs[k]);
    for (int k = 0; k < KX - 1; ++k)
        M(xa, k, k) = 0.5f * (phc::akap * tref[k] / t.fsg[k] - (tref[k + 1] - tref[k]) / t.dhs[k]);
    std::array<double, 8> dsum{};
    dsum[0] = t.dhs[0];
    for (int k = 1; k < KX; ++k) dsum[k] = dsum[k - 1] + t.dhs[k];
    for (int k = 0; k < KX - 1; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            M(xb, k, k1) = t.dhs[k1] * dsum[k];
            if (k1 <= k) M(xb, k, k1) = M(xb, k, k1) - t.dhs[k1];
        }
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            M(xc, k, k1) = M(ya, k, k1);
            for (int k2 = 0; k2 < KX - 1; ++k2) M(xc, k, k1) = M(xc, k, k1) + M(xa, k, k2) * M(xb, k2, k1);
        }
    xd.fill(0.0);
    for (int k = 0; k < KX; ++k)
        for (int k1 = k + 1; k1 < KX; ++k1) M(xd, k, k1) = phc::rgas * std::log(t.hsg[k1 + 1] / t.hsg[k1]);
    for (int k = 0; k < KX; ++k) M(xd, k, k) = phc::rgas * std::log(t.hsg[k + 1] / t.fsg[k]);
    for (int k = 0; k < KX; ++k)
        for (int k1 = 0; k1 < KX; ++k1) {
            M(xe, k, k1) = 0.;
            for (int k2 = 0; k2 < KX; ++k2) M(xe, k, k1) = M(xe, k, k1) + M(xd, k, k2) * M(xc, k2, k1);
        }
    for (int l = 1; l <= MX + NX + 1; ++l) {
        const double xxx = static_cast<double>(static_cast<float>(l) * static_cast<float>(l + 1)) / a2;
        Lu8 lu;
        for (int k = 0; k < KX; ++k)
            for (int k1 = 0; k1 < KX; ++k1) lu.at(k, k1) = xi * xi * xxx * (phc::rgas * tref[k] * t.dhs[k1] - M(xe, k, k1));
        for (int k = 0; k < KX; ++k) lu.at(k, k) = lu.at(k, k) + 1.f;
        double *y = xj.data() + static_cast<size_t>(KX) * KX * (l - 1);
        for (int i = 0; i < KX * KX; ++i) y[i] = 0.0;
        for (int i = 0; i < KX; ++i) y[i + KX * i] = 1.;
        lu.factor();
        for (int i = 0; i < KX; ++i) lu.solve(y + KX * i);
    }
    for (auto &v : xc) v = v * xi;
}

std::vector<double> HostTables::cpol() const {
    std::vector<double> c(static_cast<size_t>(2 * MX) * NX * IY);
    for (int j = 0; j < IY; ++j)
        for (int n = 0; n < NX; ++n)
            for (int m = 0; m < MX; ++m) {
                const double v = poly[m + MX * (n + NX * j)];
                c[(2 * m) + 2 * MX * (n + NX * j)] = v;
                c[(2 * m + 1) + 2 * MX * (n + NX * j)] = v;
            }
    return c;
}

}  // namespace spd
