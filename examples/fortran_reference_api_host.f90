!> A Fortran host written against the REFERENCE's own generated module -- `use speedy_driver` and its procedure names exactly as
!! registry/templates/speedy_driver.f90.j2 generates them (modelstate_init, set_orog ... set_sea_ice_frac12, create_datetime,
!! controlparams_init, init, parallel_step, check, transform_spectral2grid, get_t_grid, get_current_step, get_sst_anom_shape,
!! is_array_t_grid) -- linked against include/speedy_driver_amd.f90, the module of the same name that forwards to the MI355X
!! library.  This is the file the reference's f2py build would wrap instead of its generated speedy_driver.f90; not one
!! call site above it changes.
!!
!!   fortran_reference_api_host <bc.bin> <out.bin> <nsteps> [overlapped]
!! (`overlapped`: the time loop uses parallel_step_begin / parallel_step_end, the library's extension of parallel_step)
!! bc.bin as for fortran_host; two members, member 2 with its SST raised by 0.5 K.  out.bin: t_grid (96,48,8) of both members.
program fortran_reference_api_host
    use speedy_driver
    implicit none
    integer, parameter :: ix = 96, il = 48, kx = 8, n = 2
    integer(8) :: states(n), controls(n), d_start, d_end, token, next_token
    logical :: overlapped
    integer :: codes(n), code, istep, nsteps, m, u, uo, shp(3), y, mo, d, h, mi
    logical :: flag
    real(8) :: f2(ix, il), f12(ix, il, 12), bc2(ix, il, 5), bc12(ix, il, 12, 7), t_grid(ix, il, kx)
    character(len=512) :: arg

    call get_command_argument(3, arg)
    read (arg, *) nsteps
    call get_command_argument(4, arg)
    overlapped = trim(arg) == "overlapped"
    call get_command_argument(1, arg)
    open (newunit=u, file=trim(arg), access="stream", form="unformatted", status="old")
    read (u) bc2      ! orog, fmask_orig, alb0, veg_high, veg_low
    read (u) bc12     ! stl12, snowd12, soil_wc_l1, soil_wc_l2, soil_wc_l3, sst12, sea_ice_frac12
    close (u)

    call create_datetime(1982, 1, 1, 0, 0, d_start)
    call create_datetime(1982, 1, 4, 0, 0, d_end)
    do m = 1, n
        call modelstate_init(states(m))
        call controlparams_init(controls(m), d_start, d_end)
        f2 = bc2(:, :, 1); call set_orog(states(m), f2)
        f2 = bc2(:, :, 2); call set_fmask_orig(states(m), f2)
        f2 = bc2(:, :, 3); call set_alb0(states(m), f2)
        f2 = bc2(:, :, 4); call set_veg_high(states(m), f2)
        f2 = bc2(:, :, 5); call set_veg_low(states(m), f2)
        f12 = bc12(:, :, :, 1); call set_stl12(states(m), f12)
        f12 = bc12(:, :, :, 2); call set_snowd12(states(m), f12)
        f12 = bc12(:, :, :, 3); call set_soil_wc_l1(states(m), f12)
        f12 = bc12(:, :, :, 4); call set_soil_wc_l2(states(m), f12)
        f12 = bc12(:, :, :, 5); call set_soil_wc_l3(states(m), f12)
        f12 = bc12(:, :, :, 6) + 0.5d0 * (m - 1); call set_sst12(states(m), f12)
        f12 = bc12(:, :, :, 7); call set_sea_ice_frac12(states(m), f12)
        call get_sst_anom_shape(states(m), shp)
        if (any(shp /= 0)) stop "sst_anom should not be allocated yet"
        call init(states(m), controls(m), code)
        if (code /= 0) stop "init failed"
    end do

    if (overlapped) then
        ! the library's extension of the same call: the check of step k is collected after step k + 1 has been enqueued
        call parallel_step_begin(states, controls, n, token)
        do istep = 2, nsteps
            call parallel_step_begin(states, controls, n, next_token)
            call parallel_step_end(token, codes, n)
            if (any(codes /= 0)) stop "model variables out of range"
            token = next_token
        end do
        call parallel_step_end(token, codes, n)
        if (any(codes /= 0)) stop "model variables out of range"
    else
        do istep = 1, nsteps
            call parallel_step(states, controls, codes, n)
            if (any(codes /= 0)) stop "model variables out of range"
        end do
    end if
    call get_current_step(states(2), istep)
    call get_land_coupling_flag(states(1), flag)
    call is_array_t_grid(flag)
    call get_datetime(d_end, y, mo, d, h, mi)

    call get_command_argument(2, arg)
    open (newunit=uo, file=trim(arg), access="stream", form="unformatted", status="replace")
    do m = 1, n
        call check(states(m), code)
        if (code /= 0) stop "check failed"
        call transform_spectral2grid(states(m))
        call get_t_grid(states(m), t_grid)
        write (uo) t_grid
    end do
    close (uo)
    print "(a, i0, a, l1, a, i0)", "steps ", istep, "  t_grid is an array ", flag, "  end year ", y
    do m = 1, n
        call modelstate_close(states(m))
        call controlparams_close(controls(m))
    end do
    call close_datetime(d_start)
    call close_datetime(d_end)
end program fortran_reference_api_host
