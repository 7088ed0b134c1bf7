// Host-side construction of every constant table the device kernels consume.
//
// The reference builds these at run time in ModGeometry_initialize (speedy.f90/geometry.f90:67-170),
// ModLegendre_initialize (legendre.f90:38-112, 224-307), rffti1 (fftpack.f90:1-67),
// ModSpectral_initialize (spectral.f90:39-116) and radset (longwave_radiation.f90:208-232).  Many of its
// constants are default-real (fp32) literals or fp32 expressions widened to fp64; parity at fp64 tolerance
// requires reproducing those roundings exactly, so this file is compiled with
// floating-point contraction off and evaluates each such sub-expression in `float`.
#pragma once
#include <array>
#include <vector>

namespace spd {

constexpr int IX = 96, IL = 48, IY = 24, KX = 8, MX = 31, NX = 32, TRUNC = 30;
constexpr int NSPEC = MX * NX;      // complex coefficients per spectral field
constexpr int NFOUR = 2 * MX * IL;  // doubles per Fourier plane
constexpr int NGRID = IX * IL;      // doubles per grid field

// default-real parameters of physical_constants.f90:16-30 / params.f90:33-36, widened
namespace phc {
constexpr double rearth = 6.371e+6f, omega = 7.292e-05f, grav = 9.81f, p0 = 1.e+5f, cp = 1004.0f;
constexpr double akap = 2.0f / 7.0f;  // evaluated in fp32
constexpr double rgas = akap * cp;
constexpr double alhc = 2501.0f, alhs = 2801.0f, sbc = 5.67e-8f;
}  // namespace phc

struct HostTables {
    // geometry
    std::array<double, 9> hsg{}, sigh{};
    std::array<double, 8> dhs{}, fsg{}, dhsr{}, fsgr{}, sigl{}, grdsig{}, grdscp{};
    std::array<double, 16> wvi{};  // (kx,2) column-major
    std::array<double, 48> radang{}, coriol{}, sia{}, coa{}, cosgr{}, cosgr2{};
    std::array<double, 24> sia_half{}, coa_half{}, wt{};
    // legendre
    std::vector<double> epsi, repsi;  // (mx+1, nx+1)
    std::vector<double> poly;         // unique polynomials (mx, nx, iy); cpol(2m-1|2m, n, j) = poly(m, n, j)
    std::array<int, 32> nsh2{};
    // fft
    std::array<double, 96> work{};
    std::array<int, 15> ifac{};
    // spectral operators, (mx, nx) column-major
    std::vector<double> el2, elm2, el4, trfilt, gradym, gradyp, uvdx, uvdym, uvdyp, vddym, vddyp;
    std::array<double, 31> gradx{};
    // longwave band fractions fband(100:400, 4)
    std::vector<double> fband;

    HostTables();
    // expanded cpol(2*mx, nx, iy) as the reference stores it
    std::vector<double> cpol() const;
};

// Tables of the callers of the hot path: horizontal diffusion (horizontal_diffusion.f90:50-110), semi-implicit
// gravity-wave correction (implicit.f90:44-218, matrix_inversion.f90), geopotential (geopotential.f90:16-31).
struct DynHostTables {
    std::vector<double> dmp, dmpd, dmps, dmp1, dmp1d, dmp1s, elz;  // (mx,nx)
    std::array<double, 8> tcorv{}, qcorv{}, tref{}, tref2{}, tref3{}, dhsx{}, xgeop1{}, xgeop2{};
    std::array<double, 8> geo_corf{};                              // lapse-rate correction factors, geopotential.f90:73
    std::array<double, 64> xc{}, xd{};                             // (kx,kx) column-major
    std::vector<double> xj;                                        // (kx,kx,mx+nx+1)
    double dt = 0.0;

    explicit DynHostTables(const HostTables &t);
    void set_time_step(const HostTables &t, double dt);            // ModImplicit_set_time_step
};

}  // namespace spd
