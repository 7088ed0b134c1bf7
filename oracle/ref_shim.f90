!> TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
!
!  bind(C) access shim for the *reference* Fortran (pySPEEDY's speedy.f90/*.f90),
!  compiled together with the reference sources where they lie under /root/reference by
!  oracle/build_ref.sh into oracle/_ref/libspeedy_ref.so.  It contains no numerics of its own:
!  every routine only forwards to a reference procedure so that golden vectors can be
!  captured at operator level (the f2py-facing speedy_driver module only exposes whole-model
!  calls).  State objects are addressed through the same opaque integer(8) containers the
!  reference's speedy_driver uses (registry/templates/speedy_driver.f90.j2:38-40).
module ref_shim
    use iso_c_binding
    use params
    use model_state, only : ModelState_t, ModelState_Ptr_t
    implicit none

contains

    function state_of(cnt) result(s)
        integer(c_int64_t), intent(in) :: cnt
        type(ModelState_t), pointer :: s
        type(ModelState_Ptr_t) :: ptr
        ptr = transfer(cnt, ptr)
        s => ptr%p
    end function

    ! ------------------------------------------------------------------ tables
    subroutine shim_geometry(cnt, hsg, dhs, fsg, dhsr, fsgr, radang, coriol, sia, coa, sia_half, &
            coa_half, cosgr, cosgr2, sigl, sigh, grdsig, grdscp, wvi) bind(C, name = "shim_geometry")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(out) :: hsg(kx + 1), dhs(kx), fsg(kx), dhsr(kx), fsgr(kx)
        real(c_double), intent(out) :: radang(il), coriol(il), sia(il), coa(il), sia_half(iy), coa_half(iy)
        real(c_double), intent(out) :: cosgr(il), cosgr2(il), sigl(kx), sigh(0:kx), grdsig(kx), grdscp(kx), wvi(kx, 2)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        hsg = s%mod_geometry%hsg; dhs = s%mod_geometry%dhs; fsg = s%mod_geometry%fsg
        dhsr = s%mod_geometry%dhsr; fsgr = s%mod_geometry%fsgr
        radang = s%mod_geometry%radang; coriol = s%mod_geometry%coriol
        sia = s%mod_geometry%sia; coa = s%mod_geometry%coa
        sia_half = s%mod_geometry%sia_half; coa_half = s%mod_geometry%coa_half(1:iy)
        cosgr = s%mod_geometry%cosgr; cosgr2 = s%mod_geometry%cosgr2
        sigl = s%mod_geometry%sigl; sigh = s%mod_geometry%sigh
        grdsig = s%mod_geometry%grdsig; grdscp = s%mod_geometry%grdscp; wvi = s%mod_geometry%wvi
    end subroutine

    subroutine shim_legendre_tables(cnt, epsi, repsi, cpol, nsh2, wt) bind(C, name = "shim_legendre_tables")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(out) :: epsi(mx + 1, nx + 1), repsi(mx + 1, nx + 1), cpol(2 * mx, nx, iy), wt(iy)
        integer(c_int), intent(out) :: nsh2(nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        epsi = s%mod_spectral%epsi; repsi = s%mod_spectral%repsi; cpol = s%mod_spectral%cpol
        nsh2 = s%mod_spectral%nsh2; wt = s%mod_spectral%wt
    end subroutine

    subroutine shim_fft_tables(cnt, work, ifac) bind(C, name = "shim_fft_tables")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(out) :: work(ix)
        integer(c_int), intent(out) :: ifac(15)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        work = s%mod_spectral%work; ifac = s%mod_spectral%ifac
    end subroutine

    subroutine shim_spectral_tables(cnt, el2, elm2, el4, trfilt, gradx, gradym, gradyp, uvdx, uvdym, uvdyp, &
            vddym, vddyp) bind(C, name = "shim_spectral_tables")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(out), dimension(mx, nx) :: el2, elm2, el4, trfilt, gradym, gradyp
        real(c_double), intent(out), dimension(mx, nx) :: uvdx, uvdym, uvdyp, vddym, vddyp
        real(c_double), intent(out) :: gradx(mx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        el2 = s%mod_spectral%el2; elm2 = s%mod_spectral%elm2; el4 = s%mod_spectral%el4
        trfilt = s%mod_spectral%trfilt; gradx = s%mod_spectral%gradx
        gradym = s%mod_spectral%gradym; gradyp = s%mod_spectral%gradyp
        uvdx = s%mod_spectral%uvdx; uvdym = s%mod_spectral%uvdym; uvdyp = s%mod_spectral%uvdyp
        vddym = s%mod_spectral%vddym; vddyp = s%mod_spectral%vddyp
    end subroutine

    ! --------------------------------------------------------------- operators
    subroutine shim_legendre_inv(cnt, inp, outp) bind(C, name = "shim_legendre_inv")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: inp(2 * mx, nx)
        real(c_double), intent(out) :: outp(2 * mx, il)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        outp = s%mod_spectral%legendre_inv(inp)
    end subroutine

    subroutine shim_legendre(cnt, inp, outp) bind(C, name = "shim_legendre")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: inp(2 * mx, il)
        real(c_double), intent(out) :: outp(2 * mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        outp = s%mod_spectral%legendre(inp)
    end subroutine

    subroutine shim_fourier_inv(cnt, inp, outp, kcos) bind(C, name = "shim_fourier_inv")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: inp(2 * mx, il)
        real(c_double), intent(out) :: outp(ix, il)
        integer(c_int), value :: kcos
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        outp = s%mod_spectral%fourier_inv(inp, kcos)
    end subroutine

    subroutine shim_fourier(cnt, inp, outp) bind(C, name = "shim_fourier")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: inp(ix, il)
        real(c_double), intent(out) :: outp(2 * mx, il)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        outp = s%mod_spectral%fourier(inp)
    end subroutine

    subroutine shim_spec2grid(cnt, spec, grid, kcos) bind(C, name = "shim_spec2grid")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(in) :: spec(mx, nx)
        real(c_double), intent(out) :: grid(ix, il)
        integer(c_int), value :: kcos
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        grid = s%mod_spectral%spec2grid(spec, kcos)
    end subroutine

    subroutine shim_grid2spec(cnt, grid, spec) bind(C, name = "shim_grid2spec")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: grid(ix, il)
        complex(c_double_complex), intent(out) :: spec(mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        spec = s%mod_spectral%grid2spec(grid)
    end subroutine

    subroutine shim_vort2vel(cnt, vor, div, ucos, vcos) bind(C, name = "shim_vort2vel")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(in) :: vor(mx, nx), div(mx, nx)
        complex(c_double_complex), intent(inout) :: ucos(mx, nx), vcos(mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%vort2vel(vor, div, ucos, vcos)
    end subroutine

    subroutine shim_vel2vort(cnt, ucos, vcos, vor, div) bind(C, name = "shim_vel2vort")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex) :: ucos(mx, nx), vcos(mx, nx)
        complex(c_double_complex), intent(inout) :: vor(mx, nx), div(mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%vel2vort(ucos, vcos, vor, div)
    end subroutine

    subroutine shim_grid_vel2vort(cnt, ug, vg, vor, div, kcos) bind(C, name = "shim_grid_vel2vort")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: ug(ix, il), vg(ix, il)
        complex(c_double_complex), intent(out) :: vor(mx, nx), div(mx, nx)
        integer(c_int), value :: kcos
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%grid_vel2vort(ug, vg, vor, div, kcos)
    end subroutine

    subroutine shim_gradient(cnt, psi, psdx, psdy) bind(C, name = "shim_gradient")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(inout) :: psi(mx, nx), psdx(mx, nx), psdy(mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%gradient(psi, psdx, psdy)
    end subroutine

    subroutine shim_laplacian(cnt, inp, outp, inverse) bind(C, name = "shim_laplacian")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(in) :: inp(mx, nx)
        complex(c_double_complex), intent(out) :: outp(mx, nx)
        integer(c_int), value :: inverse
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        if (inverse /= 0) then
            outp = s%mod_spectral%laplacian_inv(inp)
        else
            outp = s%mod_spectral%laplacian(inp)
        end if
    end subroutine

    subroutine shim_truncate(cnt, f) bind(C, name = "shim_truncate")
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(inout) :: f(mx, nx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%truncate(f)
    end subroutine

    subroutine shim_grid_filter(cnt, fg1, fg2) bind(C, name = "shim_grid_filter")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(inout) :: fg1(ix, il), fg2(ix, il)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call s%mod_spectral%grid_filter(fg1, fg2)
    end subroutine

    ! ----------------------------------------------------------------- physics
    !> The column-physics driver exactly as tendencies.f90:231 calls it.
    subroutine shim_physics(cnt, j1, utend, vtend, ttend, qtend) bind(C, name = "shim_physics")
        use physics, only : get_physical_tendencies
        integer(c_int64_t), value :: cnt
        integer(c_int), value :: j1
        real(c_double), intent(inout), dimension(ix, il, kx) :: utend, vtend, ttend, qtend
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_physical_tendencies(s, j1, utend, vtend, ttend, qtend)
    end subroutine

    subroutine shim_qsat(ta, ps, sig, qsat) bind(C, name = "shim_qsat")
        use humidity, only : get_qsat
        real(c_double), intent(in) :: ta(ix, il), ps(ix, il)
        real(c_double), value :: sig
        real(c_double), intent(out) :: qsat(ix, il)
        qsat = get_qsat(ta, ps, sig)
    end subroutine

    subroutine shim_convection(cnt, psa, se, qa, qsat, itop, cbmf, precnv, dfse, dfqa) bind(C, name = "shim_convection")
        use convection, only : get_convection_tendencies
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: psa(ix, il), se(ix, il, kx), qa(ix, il, kx), qsat(ix, il, kx)
        integer(c_int), intent(out) :: itop(ix, il)
        real(c_double), intent(out) :: cbmf(ix, il), precnv(ix, il), dfse(ix, il, kx), dfqa(ix, il, kx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_convection_tendencies(psa, se, qa, qsat, itop, cbmf, precnv, dfse, dfqa, &
                s%mod_geometry%fsg, s%mod_geometry%dhs, s%mod_geometry%wvi)
    end subroutine

    subroutine shim_lsc(cnt, psa, qa, qsat, itop, precls, dtlsc, dqlsc) bind(C, name = "shim_lsc")
        use large_scale_condensation, only : get_large_scale_condensation_tendencies
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: psa(ix, il), qa(ix, il, kx), qsat(ix, il, kx)
        integer(c_int), intent(inout) :: itop(ix, il)
        real(c_double), intent(out) :: precls(ix, il), dtlsc(ix, il, kx), dqlsc(ix, il, kx)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_large_scale_condensation_tendencies(psa, qa, qsat, itop, precls, dtlsc, dqlsc, &
                s%mod_geometry%fsg, s%mod_geometry%dhs)
    end subroutine

    subroutine shim_clouds(qa, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr, qcloud_equiv) &
            bind(C, name = "shim_clouds")
        use shortwave_radiation, only : clouds
        real(c_double), intent(in) :: qa(ix, il, kx), rh(ix, il, kx), precnv(ix, il), precls(ix, il), gse(ix, il), fmask(ix, il)
        integer(c_int) :: iptop(ix, il)
        integer(c_int), intent(out) :: icltop(ix, il)
        real(c_double), intent(out) :: cloudc(ix, il), clstr(ix, il), qcloud_equiv(ix, il)
        call clouds(qa, rh, precnv, precls, iptop, gse, fmask, icltop, cloudc, clstr, qcloud_equiv)
    end subroutine

    !> Shortwave: reads/writes the state's radiation fields (tsr, ssrd, ssr, tt_rsw, rad_tau2, rad_flux, rad_strat_corr).
    subroutine shim_shortwave(cnt, psa, qa, icltop, cloudc, clstr) bind(C, name = "shim_shortwave")
        use shortwave_radiation, only : get_shortwave_rad_fluxes
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: psa(ix, il), qa(ix, il, kx), cloudc(ix, il), clstr(ix, il)
        integer(c_int), intent(in) :: icltop(ix, il)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_shortwave_rad_fluxes(s, psa, qa, icltop, cloudc, clstr)
    end subroutine

    subroutine shim_lw_down(cnt, ta, fsfcd, dfabs, rad_flux, rad_tau2, rad_st4a) bind(C, name = "shim_lw_down")
        use longwave_radiation, only : get_downward_longwave_rad_fluxes
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: ta(ix, il, kx), rad_tau2(ix, il, kx, 4)
        real(c_double), intent(out) :: fsfcd(ix, il), dfabs(ix, il, kx)
        real(c_double), intent(inout) :: rad_flux(ix, il, 4), rad_st4a(ix, il, kx, 2)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_downward_longwave_rad_fluxes(ta, fsfcd, dfabs, s%fband, rad_flux, rad_tau2, rad_st4a, s%mod_geometry%wvi)
    end subroutine

    subroutine shim_lw_up(cnt, ta, ts, fsfcd, fsfcu, fsfc, ftop, dfabs, rad_flux, rad_tau2, rad_st4a, rad_strat_corr) &
            bind(C, name = "shim_lw_up")
        use longwave_radiation, only : get_upward_longwave_rad_fluxes
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: ta(ix, il, kx), ts(ix, il), fsfcd(ix, il), fsfcu(ix, il), rad_st4a(ix, il, kx, 2)
        real(c_double), intent(out) :: fsfc(ix, il), ftop(ix, il)
        real(c_double), intent(inout) :: dfabs(ix, il, kx), rad_flux(ix, il, 4), rad_tau2(ix, il, kx, 4), rad_strat_corr(ix, il, 2)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_upward_longwave_rad_fluxes(ta, ts, fsfcd, fsfcu, fsfc, ftop, dfabs, s%fband, &
                rad_flux, rad_tau2, rad_st4a, rad_strat_corr, s%mod_geometry%dhs)
    end subroutine

    subroutine shim_surface_fluxes(cnt, psa, ua, va, ta, qa, rh, phi, phi0, fmask, forog, tsea, ssrd, slrd, &
            ustr, vstr, shf, evap, slru, hfluxn, tsfc, tskin, u0, v0, t0, &
            alb_land, alb_sea, snowc, land_temp, soil_avail_water) bind(C, name = "shim_surface_fluxes")
        use surface_fluxes, only : get_surface_fluxes
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in) :: psa(ix, il), ua(ix, il, kx), va(ix, il, kx), ta(ix, il, kx), qa(ix, il, kx)
        real(c_double), intent(in) :: rh(ix, il, kx), phi(ix, il, kx), phi0(ix, il), fmask(ix, il), forog(ix, il)
        real(c_double), intent(in) :: tsea(ix, il), ssrd(ix, il), slrd(ix, il)
        real(c_double), intent(out) :: ustr(ix, il, 3), vstr(ix, il, 3), shf(ix, il, 3), evap(ix, il, 3), slru(ix, il, 3)
        real(c_double), intent(out) :: hfluxn(ix, il, 2), tsfc(ix, il), tskin(ix, il), u0(ix, il), v0(ix, il), t0(ix, il)
        real(c_double), intent(in) :: alb_land(ix, il), alb_sea(ix, il), snowc(ix, il), land_temp(ix, il), soil_avail_water(ix, il)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_surface_fluxes(psa, ua, va, ta, qa, rh, phi, phi0, fmask, forog, tsea, ssrd, slrd, &
                ustr, vstr, shf, evap, slru, hfluxn, tsfc, tskin, u0, v0, t0, .true., &
                alb_land, alb_sea, snowc, land_temp, soil_avail_water, &
                s%mod_geometry%coa, s%mod_geometry%sigl, s%mod_geometry%wvi)
    end subroutine

    subroutine shim_vdiff(cnt, se, rh, qa, qsat, phi, icnv, ut, vt, tt, qt) bind(C, name = "shim_vdiff")
        use vertical_diffusion, only : get_vertical_diffusion_tend
        integer(c_int64_t), value :: cnt
        real(c_double), intent(in), dimension(ix, il, kx) :: se, rh, qa, qsat, phi
        integer(c_int), intent(in) :: icnv(ix, il)
        real(c_double), intent(out), dimension(ix, il, kx) :: ut, vt, tt, qt
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_vertical_diffusion_tend(se, rh, qa, qsat, phi, icnv, ut, vt, tt, qt, &
                s%mod_geometry%fsg, s%mod_geometry%dhs, s%mod_geometry%sigh)
    end subroutine

    ! ------------------------------------------------- dynamics (callers, "next")
    !> Grid-point + spectral tendencies exactly as time_stepping.f90:71 calls them.
    subroutine shim_get_tendencies(cnt, vordt, divdt, tdt, psdt, trdt, j2) bind(C, name = "shim_get_tendencies")
        use tendencies, only : get_tendencies
        integer(c_int64_t), value :: cnt
        complex(c_double_complex), intent(inout) :: vordt(mx, nx, kx), divdt(mx, nx, kx), tdt(mx, nx, kx)
        complex(c_double_complex), intent(inout) :: psdt(mx, nx), trdt(mx, nx, kx, ntr)
        integer(c_int), value :: j2
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_tendencies(s, vordt, divdt, tdt, psdt, trdt, j2)
    end subroutine

    subroutine shim_timestep(cnt, j1, j2, dt) bind(C, name = "shim_timestep")
        use time_stepping, only : step
        integer(c_int64_t), value :: cnt
        integer(c_int), value :: j1, j2
        real(c_double), value :: dt
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call step(s, j1, j2, dt)
    end subroutine


    !> Tables of ModImplicit_t / ModHorizontalDiffusion_t (implicit.f90:17-22, horizontal_diffusion.f90:16-30).
    subroutine shim_implicit_tables(cnt, dmp, dmpd, dmps, dmp1, dmp1d, dmp1s, tcorv, qcorv, tcorh, qcorh, &
            tref, tref2, tref3, dhsx, xc, xd, xj, elz) bind(C, name = "shim_implicit_tables")
        integer(c_int64_t), value :: cnt
        real(c_double), intent(out), dimension(mx, nx) :: dmp, dmpd, dmps, dmp1, dmp1d, dmp1s, elz
        real(c_double), intent(out) :: tcorv(kx), qcorv(kx), tref(kx), tref2(kx), tref3(kx), dhsx(kx)
        complex(c_double_complex), intent(out) :: tcorh(mx, nx), qcorh(mx, nx)
        real(c_double), intent(out) :: xc(kx, kx), xd(kx, kx), xj(kx, kx, mx + nx + 1)
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        dmp = s%mod_implicit%dmp; dmpd = s%mod_implicit%dmpd; dmps = s%mod_implicit%dmps
        dmp1 = s%mod_implicit%dmp1; dmp1d = s%mod_implicit%dmp1d; dmp1s = s%mod_implicit%dmp1s
        tcorv = s%mod_implicit%tcorv; qcorv = s%mod_implicit%qcorv
        tcorh = s%mod_implicit%tcorh; qcorh = s%mod_implicit%qcorh
        tref = s%mod_implicit%tref; tref2 = s%mod_implicit%tref2; tref3 = s%mod_implicit%tref3
        dhsx = s%mod_implicit%dhsx; xc = s%mod_implicit%xc; xd = s%mod_implicit%xd; xj = s%mod_implicit%xj
        elz = s%mod_implicit%elz
    end subroutine

    !> The calendar a ControlParams_t carries (model_control.f90:37-47): the reference's driver has no getter for it.
    subroutine shim_control_params(ctl, ymdhm, month_idx, imont1, tmonth, tyear) bind(C, name = "shim_control_params")
        use model_control, only : ControlParams_t, ControlParams_Ptr_t
        integer(c_int64_t), value :: ctl
        integer(c_int), intent(out) :: ymdhm(5), month_idx, imont1
        real(c_double), intent(out) :: tmonth, tyear
        type(ControlParams_Ptr_t) :: ptr
        ptr = transfer(ctl, ptr)
        ymdhm(1) = ptr%p%model_datetime%year
        ymdhm(2) = ptr%p%model_datetime%month
        ymdhm(3) = ptr%p%model_datetime%day
        ymdhm(4) = ptr%p%model_datetime%hour
        ymdhm(5) = ptr%p%model_datetime%minute
        month_idx = ptr%p%month_idx
        imont1 = ptr%p%imont1
        tmonth = ptr%p%tmonth
        tyear = ptr%p%tyear
    end subroutine

    !> One 40-minute step of the calendar alone (model_control.f90:114-160), without stepping a model.
    subroutine shim_advance_date(ctl) bind(C, name = "shim_advance_date")
        use model_control, only : ControlParams_Ptr_t, advance_date
        integer(c_int64_t), value :: ctl
        type(ControlParams_Ptr_t) :: ptr
        ptr = transfer(ctl, ptr)
        call advance_date(ptr%p)
    end subroutine

    !> The zonally uniform daily forcing for an arbitrary fraction of the year (shortwave_radiation.f90:218-275), as
    !  set_forcing calls it once per simulated day (forcing.f90:84).
    subroutine shim_zonal_average_fields(cnt, tyear) bind(C, name = "shim_zonal_average_fields")
        use shortwave_radiation, only : get_zonal_average_fields
        integer(c_int64_t), value :: cnt
        real(c_double), value :: tyear
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call get_zonal_average_fields(s, tyear)
    end subroutine

    subroutine shim_set_geopotential(cnt, time_level) bind(C, name = "shim_set_geopotential")
        use geopotential, only : set_geopotential
        integer(c_int64_t), value :: cnt
        integer(c_int), value :: time_level
        type(ModelState_t), pointer :: s
        s => state_of(cnt)
        call set_geopotential(s, time_level)
    end subroutine

end module
