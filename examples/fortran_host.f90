!> A Fortran host driving the MI355X model through the C ABI (include/pyspeedy_amd_c.f90): the same sequence the
!! reference's Python layer runs through speedy_driver -- allocate the state, set the 12 boundary fields, init, step, check,
!! transform to the grid, get a variable.
!!
!!   fortran_host <bc.bin> <out.bin> <nsteps>
!! bc.bin: the boundary fields as raw real(8), in the order of `names` below, (96,48) or (96,48,12) each (written by
!! tests/test_fortran_host_gpu.py from the packaged example_bc); out.bin: t_grid (96,48,8) then ps_grid (96,48) as real(8).
program fortran_host
    use iso_c_binding
    use pyspeedy_amd_c
    implicit none
    integer, parameter :: ix = 96, il = 48, kx = 8
    character(len=16), parameter :: names(12) = [character(len=16) :: "orog", "fmask_orig", "alb0", "veg_high", "veg_low", &
            "stl12", "snowd12", "soil_wc_l1", "soil_wc_l2", "soil_wc_l3", "sst12", "sea_ice_frac12"]
    integer, parameter :: planes(12) = [1, 1, 1, 1, 1, 12, 12, 12, 12, 12, 12, 12]
    type(c_ptr) :: ctx, model
    real(c_double), allocatable :: field(:, :, :), t_grid(:, :, :), ps_grid(:, :)
    integer(c_int32_t) :: codes(1)
    character(len=512) :: arg
    integer :: i, nsteps, u
    integer(c_int) :: rc

    call get_command_argument(3, arg)
    read (arg, *) nsteps
    call check(spd_create(ctx, 0_c_int), "spd_create")
    call check(spd_model_create(ctx, 1_c_int, model), "spd_model_create")

    call get_command_argument(1, arg)
    open (newunit=u, file=trim(arg), access="stream", form="unformatted", status="old")
    do i = 1, 12
        allocate (field(ix, il, planes(i)))
        read (u) field
        call check(spd_model_set(model, trim(names(i))//c_null_char, 0_c_int, field, &
                                 int(8 * size(field), c_size_t)), "spd_model_set "//trim(names(i)))
        deallocate (field)
    end do
    close (u)

    call check(spd_model_init(model, 1982_c_int, 1_c_int, 1_c_int, 0_c_int, 0_c_int, c_null_ptr), "spd_model_init")
    call check(spd_model_step(model, int(nsteps, c_int), c_null_ptr), "spd_model_step")
    call check(spd_model_check(model, 2_c_int, codes, c_null_ptr, c_null_ptr), "spd_model_check")
    if (codes(1) /= 0) stop "model variables out of range"
    call check(spd_model_spectral2grid(model, 0_c_int, 1_c_int, c_null_ptr), "spd_model_spectral2grid")

    allocate (t_grid(ix, il, kx), ps_grid(ix, il))
    call check(spd_model_get(model, "t_grid"//c_null_char, 0_c_int, t_grid, int(8 * size(t_grid), c_size_t)), "get t_grid")
    call check(spd_model_get(model, "ps_grid"//c_null_char, 0_c_int, ps_grid, int(8 * size(ps_grid), c_size_t)), "get ps_grid")
    call get_command_argument(2, arg)
    open (newunit=u, file=trim(arg), access="stream", form="unformatted", status="replace")
    write (u) t_grid
    write (u) ps_grid
    close (u)
    print "(a, i0, a, f10.4, a, f12.2)", "steps ", spd_model_current_step(model), "  mean T(lowest level) ", &
        sum(t_grid(:, :, kx)) / (ix * il), "  mean ps ", sum(ps_grid) / (ix * il)
    rc = spd_model_destroy(model)
    rc = spd_destroy(ctx)

contains
    subroutine check(code, what)
        integer(c_int), intent(in) :: code
        character(len=*), intent(in) :: what
        character(kind=c_char), pointer :: msg(:)
        if (code == SPD_OK) return
        call c_f_pointer(spd_last_error(), msg, [256])
        print *, "FAILED: ", what, " -> ", code, " ", msg(1:index_of_nul(msg) - 1)
        stop 1
    end subroutine
    integer function index_of_nul(s)
        character(kind=c_char), intent(in) :: s(:)
        do index_of_nul = 1, size(s)
            if (s(index_of_nul) == c_null_char) return
        end do
    end function
end program fortran_host
