!> A Fortran host that keeps the reference's call sites: the procedures of speedy_driver (registry/templates/
!! speedy_driver.f90.j2) under their own names with an spd_ prefix -- modelstate_init, set_<v>, create_datetime,
!! controlparams_init, init, parallel_step, check, transform_spectral2grid, get_<v> -- driving a 3-member ensemble whose
!! members are independent containers, exactly as pyspeedy's SpeedyEns holds them.  parallel_step gathers them into one
!! batched device model per GPU on its first call (one set of kernel launches per step and device for the whole ensemble).
!!
!! One process, all GPUs of the node: spd_set_device_placement(spd_device_count) spreads the containers over the devices
!! round-robin; member 1 reads the boundary file, spd_broadcast_boundary hands its fields to the other members device to
!! device (xGMI); a single parallel_step then drives every device, enqueueing all of them before it waits for any.  The last
!! twelve steps are handed over as ONE stretch (spd_parallel_steps_begin / _end: a time loop in which nothing looks at the state
!! in between; the range check of every step is recorded on the device).
!!
!!   fortran_ensemble_host <bc.bin> <out.bin> <nsteps>
!! bc.bin as for fortran_host; member m gets its SST raised by 0.25 (m - 1) K.  out.bin: t_grid (96,48,8) of member 1,
!! then of member 3, as real(8).
program fortran_ensemble_host
    use iso_c_binding
    use pyspeedy_amd_c
    implicit none
    integer, parameter :: ix = 96, il = 48, kx = 8, n = 3
    character(len=16), parameter :: names(12) = [character(len=16) :: "orog", "fmask_orig", "alb0", "veg_high", "veg_low", &
            "stl12", "snowd12", "soil_wc_l1", "soil_wc_l2", "soil_wc_l3", "sst12", "sea_ice_frac12"]
    integer, parameter :: planes(12) = [1, 1, 1, 1, 1, 12, 12, 12, 12, 12, 12, 12]
    integer(c_int64_t) :: states(n), controls(n), d_start, d_end, token
    integer(c_int32_t) :: codes(n), done(n), code, alive, members, y, mo, d, h, mi, ndev, dev
    integer, parameter :: stretch = 12
    real(c_double), allocatable :: field(:, :, :), t_grid(:, :, :)
    character(len=512) :: arg
    integer :: i, m, istep, nsteps, u, uo

    call get_command_argument(3, arg)
    read (arg, *) nsteps
    call check(spd_create_datetime(1982, 1, 1, 0, 0, d_start), "create_datetime")
    call check(spd_create_datetime(1982, 1, 4, 0, 0, d_end), "create_datetime")
    call get_command_argument(1, arg)
    call check(spd_device_count(ndev), "device_count")
    call check(spd_set_device_placement(min(ndev, int(n, c_int32_t))), "set_device_placement")
    do m = 1, n
        call check(spd_modelstate_init(states(m)), "modelstate_init")
        call check(spd_controlparams_init(controls(m), d_start, d_end), "controlparams_init")
    end do
    ! member 1 reads the file; the others get the fields device to device and then their own SST
    open (newunit=u, file=trim(arg), access="stream", form="unformatted", status="old")
    do i = 1, 12
        allocate (field(ix, il, planes(i)))
        read (u) field
        call check(spd_set(states(1), trim(names(i))//c_null_char, field, int(8 * size(field), c_size_t)), &
                   "set_"//trim(names(i)))
        deallocate (field)
    end do
    close (u)
    call check(spd_broadcast_boundary(states, int(n, c_int32_t), 0_c_int32_t), "broadcast_boundary")
    allocate (field(ix, il, 12))
    do m = 2, n
        call check(spd_get(states(m), "sst12"//c_null_char, field, int(8 * size(field), c_size_t)), "get_sst12")
        field = field + 0.25d0 * (m - 1)
        call check(spd_set(states(m), "sst12"//c_null_char, field, int(8 * size(field), c_size_t)), "set_sst12")
    end do
    deallocate (field)
    do m = 1, n
        call check(spd_init(states(m), controls(m), code), "init")
        if (code /= 0) stop "init failed"
    end do

    do istep = 1, max(nsteps - stretch, 0)
        call check(spd_parallel_step(states, controls, codes, int(n, c_int32_t)), "parallel_step")
        if (any(codes /= 0)) stop "model variables out of range"
    end do
    if (nsteps > 0) then
        call check(spd_parallel_steps_begin(states, controls, int(n, c_int32_t), int(min(nsteps, stretch), c_int32_t), token), &
                   "parallel_steps_begin")
        call check(spd_parallel_steps_end(token, codes, done), "parallel_steps_end")
        if (any(codes /= 0) .or. any(done /= min(nsteps, stretch))) stop "model variables out of range inside the stretch"
    end if
    call check(spd_driver_stats(states(2), alive, members), "driver_stats")
    call check(spd_get_datetime(d_start, y, mo, d, h, mi), "get_datetime")

    allocate (t_grid(ix, il, kx))
    call get_command_argument(2, arg)
    open (newunit=uo, file=trim(arg), access="stream", form="unformatted", status="replace")
    do m = 1, n, 2
        call check(spd_check(states(m), code), "check")
        if (code /= 0) stop "check failed"
        call check(spd_transform_spectral2grid(states(m)), "transform_spectral2grid")
        call check(spd_get(states(m), "t_grid"//c_null_char, t_grid, int(8 * size(t_grid), c_size_t)), "get_t_grid")
        write (uo) t_grid
    end do
    close (uo)
    call check(spd_modelstate_device(states(n), dev), "modelstate_device")
    print "(a, i0, a, i0, a, i0, a, i0, a, i0)", "members in one device model ", members, "  device models alive ", alive, &
        "  start year ", y, "  devices ", ndev, "  device of the last member ", dev
    do m = 1, n
        call check(spd_modelstate_close(states(m)), "modelstate_close")
        call check(spd_controlparams_close(controls(m)), "controlparams_close")
    end do
    call check(spd_close_datetime(d_start), "close_datetime")
    call check(spd_close_datetime(d_end), "close_datetime")

contains
    subroutine check(rc, what)
        integer(c_int), intent(in) :: rc
        character(len=*), intent(in) :: what
        character(kind=c_char), pointer :: msg(:)
        integer :: k
        if (rc == SPD_OK) return
        call c_f_pointer(spd_last_error(), msg, [256])
        k = 1
        do while (k < 256 .and. msg(k) /= c_null_char)
            k = k + 1
        end do
        print *, "FAILED: ", what, " -> ", rc, " ", msg(1:k - 1)
        stop 1
    end subroutine
end program fortran_ensemble_host
