!> ISO_C_BINDING interface to libpyspeedy_amd.so (include/pyspeedy_amd.h) for a Fortran host: what a maintainer of
!! speedy.f90 adds to call the MI355X backend.  The outer boundary (spd_model_*) takes HOST arrays in the reference's own
!! shapes and order, so a Fortran program drives whole model runs without any HIP call of its own; the operator-level entry
!! points (spd_spec2grid ...) take device pointers (type(c_ptr) from hipMalloc) and a stream.
!!
!! Replaces, call for call, the procedures of registry/templates/speedy_driver.f90.j2:
!!   modelstate_init -> spd_create + spd_model_create     set_<v> / get_<v> -> spd_model_set / spd_model_get
!!   init            -> spd_model_init                    step / parallel_step -> spd_model_step
!!   check           -> spd_model_check                   transform_spectral2grid ... -> spd_model_spectral2grid ...
module pyspeedy_amd_c
    use iso_c_binding
    implicit none

    integer(c_int), parameter :: SPD_OK = 0, SPD_E_ARG = -1, SPD_E_DEVICE = -2, SPD_E_SIZE = -3

    interface
        ! ---- context ------------------------------------------------------------------------------------------
        integer(c_int) function spd_create(handle, device) bind(C, name="spd_create")
            import :: c_ptr, c_int
            type(c_ptr), intent(out) :: handle
            integer(c_int), value :: device
        end function
        integer(c_int) function spd_destroy(handle) bind(C, name="spd_destroy")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle
        end function
        type(c_ptr) function spd_last_error() bind(C, name="spd_last_error")
            import :: c_ptr
        end function
        integer(c_long) function spd_get_table_host(handle, name, buf, buf_elems) bind(C, name="spd_get_table_host")
            import :: c_ptr, c_long, c_char, c_double, c_size_t
            type(c_ptr), value :: handle
            character(kind=c_char), intent(in) :: name(*)
            real(c_double), intent(out) :: buf(*)
            integer(c_size_t), value :: buf_elems
        end function
        ! host only (no device): initialize_control + nsteps x advance_date / update_forcing_params, model_control.f90:79-185;
        ! row 1 of every output = after initialize_control, row s + 1 = after s steps; ymdhm(5, nsteps + 1)
        integer(c_int) function spd_calendar_walk(year, month, day, hour, minute, nsteps, ymdhm, month_idx, imont1, tmonth, tyear) &
                bind(C, name="spd_calendar_walk")
            import :: c_int, c_int32_t, c_double
            integer(c_int), value :: year, month, day, hour, minute, nsteps
            integer(c_int32_t), intent(out) :: ymdhm(5, *), month_idx(*), imont1(*)
            real(c_double), intent(out) :: tmonth(*), tyear(*)
        end function
        ! host only: get_zonal_average_fields (shortwave_radiation.f90:218-322) for a fraction of the year; out(48, 5) =
        ! flux_solar_in, flux_ozone_upper, flux_ozone_lower, zenit_correction, stratospheric_correction by latitude
        integer(c_int) function spd_daily_forcing_host(tyear, out) bind(C, name="spd_daily_forcing_host")
            import :: c_int, c_double
            real(c_double), value :: tyear
            real(c_double), intent(out) :: out(48, 5)
        end function

        ! ---- operator level: device pointers, explicit batch count, stream ------------------------------------------
        integer(c_int) function spd_spec2grid(handle, spec, grid, kcos, nfields, stream) bind(C, name="spd_spec2grid")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle, spec, grid, stream
            integer(c_int), value :: kcos, nfields
        end function
        integer(c_int) function spd_grid2spec(handle, grid, spec, nfields, stream) bind(C, name="spd_grid2spec")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle, grid, spec, stream
            integer(c_int), value :: nfields
        end function
        integer(c_int) function spd_vort2vel(handle, vor, div, ucos, vcos, nfields, stream) bind(C, name="spd_vort2vel")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle, vor, div, ucos, vcos, stream
            integer(c_int), value :: nfields
        end function
        integer(c_int) function spd_grid_vel2vort(handle, ug, vg, vor, div, kcos, nfields, stream) &
                bind(C, name="spd_grid_vel2vort")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle, ug, vg, vor, div, stream
            integer(c_int), value :: kcos, nfields
        end function

        ! ---- ensemble model: host arrays in the reference's shapes ---------------------------------------------------
        integer(c_int) function spd_model_create(handle, nmembers, model) bind(C, name="spd_model_create")
            import :: c_ptr, c_int
            type(c_ptr), value :: handle
            integer(c_int), value :: nmembers
            type(c_ptr), intent(out) :: model
        end function
        integer(c_int) function spd_model_destroy(model) bind(C, name="spd_model_destroy")
            import :: c_ptr, c_int
            type(c_ptr), value :: model
        end function
        integer(c_int) function spd_model_set(model, name, member, host_buf, bytes) bind(C, name="spd_model_set")
            import :: c_ptr, c_int, c_char, c_size_t
            type(c_ptr), value :: model
            character(kind=c_char), intent(in) :: name(*)
            integer(c_int), value :: member          ! 0-based; -1 = every member
            type(*), intent(in) :: host_buf(*)
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function spd_model_get(model, name, member, host_buf, bytes) bind(C, name="spd_model_get")
            import :: c_ptr, c_int, c_char, c_size_t
            type(c_ptr), value :: model
            character(kind=c_char), intent(in) :: name(*)
            integer(c_int), value :: member
            type(*) :: host_buf(*)
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function spd_model_init_sst_anom(model, n_months) bind(C, name="spd_model_init_sst_anom")
            import :: c_ptr, c_int
            type(c_ptr), value :: model
            integer(c_int), value :: n_months
        end function
        integer(c_int) function spd_model_init(model, year, month, day, hour, minute, stream) bind(C, name="spd_model_init")
            import :: c_ptr, c_int
            type(c_ptr), value :: model, stream
            integer(c_int), value :: year, month, day, hour, minute
        end function
        integer(c_int) function spd_model_step(model, nsteps, stream) bind(C, name="spd_model_step")
            import :: c_ptr, c_int
            type(c_ptr), value :: model, stream
            integer(c_int), value :: nsteps
        end function
        integer(c_int) function spd_model_check(model, time_level, error_codes, diag, stream) bind(C, name="spd_model_check")
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: model, diag, stream   ! diag: c_null_ptr or c_loc of real(c_double) (kx, 3, nmembers)
            integer(c_int), value :: time_level
            integer(c_int32_t), intent(out) :: error_codes(*)
        end function
        integer(c_int) function spd_model_spectral2grid(model, first, count, stream) bind(C, name="spd_model_spectral2grid")
            import :: c_ptr, c_int
            type(c_ptr), value :: model, stream
            integer(c_int), value :: first, count
        end function
        integer(c_int) function spd_model_grid2spectral(model, first, count, stream) bind(C, name="spd_model_grid2spectral")
            import :: c_ptr, c_int
            type(c_ptr), value :: model, stream
            integer(c_int), value :: first, count
        end function
        integer(c_int) function spd_model_grid_filter(model, first, count, stream) bind(C, name="spd_model_grid_filter")
            import :: c_ptr, c_int
            type(c_ptr), value :: model, stream
            integer(c_int), value :: first, count
        end function
        integer(c_int) function spd_model_current_step(model) bind(C, name="spd_model_current_step")
            import :: c_ptr, c_int
            type(c_ptr), value :: model
        end function
        integer(c_int) function spd_model_set_flags(model, land_coupling, sst_anomaly_coupling, increase_co2) &
                bind(C, name="spd_model_set_flags")
            import :: c_ptr, c_int
            type(c_ptr), value :: model
            integer(c_int), value :: land_coupling, sst_anomaly_coupling, increase_co2
        end function
        ! cfg 5: fp32 /= 0 runs the arithmetic of the column physics in single precision (state and dynamics stay fp64)
        integer(c_int) function spd_model_set_physics_precision(model, fp32) bind(C, name="spd_model_set_physics_precision")
            import :: c_ptr, c_int
            type(c_ptr), value :: model
            integer(c_int), value :: fp32
        end function
        ! launch-plan switches by name ("diag_every_step", "coupler_in_spectral", "spectral_early", "split_dyn"); name is
        ! a C string: pass "diag_every_step"//c_null_char
        integer(c_int) function spd_model_set_option(model, name, value) bind(C, name="spd_model_set_option")
            import :: c_ptr, c_int, c_char, c_int32_t
            type(c_ptr), value :: model
            character(kind=c_char), intent(in) :: name(*)
            integer(c_int32_t), value :: value
        end function
        integer(c_int) function spd_model_set_sppt(model, on, seed, first_member_id) bind(C, name="spd_model_set_sppt")
            import :: c_ptr, c_int, c_int64_t
            type(c_ptr), value :: model
            integer(c_int), value :: on
            integer(c_int64_t), value :: seed, first_member_id
        end function

        ! ---- outer boundary with the reference's own procedures (include/pyspeedy_amd_driver.h) -----------------------
        ! registry/templates/speedy_driver.f90.j2: modelstate_init :216, modelstate_init_sst_anom :225, modelstate_close :240,
        ! create_datetime :163, get_datetime :189, close_datetime :203, controlparams_init :131, controlparams_close :151,
        ! init :29, step :43, parallel_step :58, check :81, transform_* :94-125, get_<v> / set_<v> / get_<v>_shape :250-334
        integer(c_int) function spd_modelstate_init(state_cnt) bind(C, name="spd_modelstate_init")
            import :: c_int, c_int64_t
            integer(c_int64_t), intent(out) :: state_cnt
        end function
        integer(c_int) function spd_modelstate_init_sst_anom(state_cnt, n_months) bind(C, name="spd_modelstate_init_sst_anom")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt
            integer(c_int32_t), value :: n_months
        end function
        integer(c_int) function spd_modelstate_close(state_cnt) bind(C, name="spd_modelstate_close")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: state_cnt
        end function
        integer(c_int) function spd_create_datetime(year, month, day, hour, minute, datetime_cnt) &
                bind(C, name="spd_create_datetime")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int32_t), value :: year, month, day, hour, minute
            integer(c_int64_t), intent(out) :: datetime_cnt
        end function
        integer(c_int) function spd_get_datetime(datetime_cnt, year, month, day, hour, minute) bind(C, name="spd_get_datetime")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: datetime_cnt
            integer(c_int32_t), intent(out) :: year, month, day, hour, minute
        end function
        integer(c_int) function spd_close_datetime(datetime_cnt) bind(C, name="spd_close_datetime")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: datetime_cnt
        end function
        integer(c_int) function spd_controlparams_init(control_cnt, start_datetime_cnt, end_datetime_cnt) &
                bind(C, name="spd_controlparams_init")
            import :: c_int, c_int64_t
            integer(c_int64_t), intent(out) :: control_cnt
            integer(c_int64_t), value :: start_datetime_cnt, end_datetime_cnt
        end function
        integer(c_int) function spd_controlparams_close(control_cnt) bind(C, name="spd_controlparams_close")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: control_cnt
        end function
        integer(c_int) function spd_init(state_cnt, control_cnt, error_code) bind(C, name="spd_init")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt, control_cnt
            integer(c_int32_t), intent(out) :: error_code
        end function
        integer(c_int) function spd_step(state_cnt, control_cnt, error_code) bind(C, name="spd_step")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt, control_cnt
            integer(c_int32_t), intent(out) :: error_code
        end function
        integer(c_int) function spd_parallel_step(state_cnts, control_cnts, error_codes, n_members) &
                bind(C, name="spd_parallel_step")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(in) :: state_cnts(*), control_cnts(*)
            integer(c_int32_t), intent(out) :: error_codes(*)
            integer(c_int32_t), value :: n_members
        end function
        ! extension: the same step with its range check overlapped with the next step
        integer(c_int) function spd_parallel_step_begin(state_cnts, control_cnts, n_members, token) &
                bind(C, name="spd_parallel_step_begin")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(in) :: state_cnts(*), control_cnts(*)
            integer(c_int32_t), value :: n_members
            integer(c_int64_t), intent(out) :: token
        end function
        integer(c_int) function spd_parallel_step_end(token, error_codes) bind(C, name="spd_parallel_step_end")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: token
            integer(c_int32_t), intent(out) :: error_codes(*)
        end function
        ! extension: n_steps steps as ONE call, the range check of every step recorded on the device (the stretch of a time loop in
        ! which nothing looks at the state); steps_done: the steps a member completed before its first failing one
        integer(c_int) function spd_parallel_steps_begin(state_cnts, control_cnts, n_members, n_steps, token) &
                bind(C, name="spd_parallel_steps_begin")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(in) :: state_cnts(*), control_cnts(*)
            integer(c_int32_t), value :: n_members, n_steps
            integer(c_int64_t), intent(out) :: token
        end function
        integer(c_int) function spd_parallel_steps_end(token, error_codes, steps_done) bind(C, name="spd_parallel_steps_end")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: token
            integer(c_int32_t), intent(out) :: error_codes(*), steps_done(*)
        end function
        integer(c_int) function spd_check(state_cnt, error_code) bind(C, name="spd_check")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt
            integer(c_int32_t), intent(out) :: error_code
        end function
        integer(c_int) function spd_transform_spectral2grid(state_cnt) bind(C, name="spd_transform_spectral2grid")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: state_cnt
        end function
        integer(c_int) function spd_transform_grid2spectral(state_cnt) bind(C, name="spd_transform_grid2spectral")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: state_cnt
        end function
        integer(c_int) function spd_apply_grid_filter(state_cnt) bind(C, name="spd_apply_grid_filter")
            import :: c_int, c_int64_t
            integer(c_int64_t), value :: state_cnt
        end function
        integer(c_int) function spd_get(state_cnt, name, buf, bytes) bind(C, name="spd_get")
            import :: c_int, c_int64_t, c_char, c_size_t
            integer(c_int64_t), value :: state_cnt
            character(kind=c_char), intent(in) :: name(*)
            type(*) :: buf(*)
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function spd_set(state_cnt, name, buf, bytes) bind(C, name="spd_set")
            import :: c_int, c_int64_t, c_char, c_size_t
            integer(c_int64_t), value :: state_cnt
            character(kind=c_char), intent(in) :: name(*)
            type(*), intent(in) :: buf(*)
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function spd_get_shape(state_cnt, name, array_shape, ndim) bind(C, name="spd_get_shape")
            import :: c_int, c_int64_t, c_char, c_int32_t
            integer(c_int64_t), value :: state_cnt
            character(kind=c_char), intent(in) :: name(*)
            integer(c_int32_t), intent(out) :: array_shape(5), ndim
        end function
        integer(c_int) function spd_driver_model(state_cnt, model, member, members_in_model) bind(C, name="spd_driver_model")
            import :: c_int, c_int64_t, c_int32_t, c_ptr
            integer(c_int64_t), value :: state_cnt
            type(c_ptr), intent(out) :: model          ! spd_model_handle owned by the driver
            integer(c_int32_t), intent(out) :: member, members_in_model
        end function
        integer(c_int) function spd_modelstate_init_ensemble(state_cnts, n_members) bind(C, name="spd_modelstate_init_ensemble")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(out) :: state_cnts(*)
            integer(c_int32_t), value :: n_members
        end function
        integer(c_int) function spd_modelstate_init_ensemble_on(state_cnts, n_members, n_devices) &
                bind(C, name="spd_modelstate_init_ensemble_on")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(out) :: state_cnts(*)
            integer(c_int32_t), value :: n_members, n_devices   ! n_devices 0: the current device; k: blocks on devices 0 .. k-1
        end function
        integer(c_int) function spd_modelstate_init_ensemble_whole(state_cnts, n_members, n_devices) &
                bind(C, name="spd_modelstate_init_ensemble_whole")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(out) :: state_cnts(*)
            integer(c_int32_t), value :: n_members, n_devices   ! ONE device model per device; n_devices < 0: the process-wide placement
        end function
        integer(c_int) function spd_driver_stats(state_cnt, models_alive, members_in_model) bind(C, name="spd_driver_stats")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt
            integer(c_int32_t), intent(out) :: models_alive, members_in_model
        end function
        ! ---- one process, several GPUs (extension; the reference's ensemble is one process, speedy_driver.f90.j2:58-79) ----
        integer(c_int) function spd_device_count(n_devices) bind(C, name="spd_device_count")
            import :: c_int, c_int32_t
            integer(c_int32_t), intent(out) :: n_devices
        end function
        integer(c_int) function spd_set_device_placement(n_devices) bind(C, name="spd_set_device_placement")
            import :: c_int, c_int32_t
            integer(c_int32_t), value :: n_devices   ! 0: current device; k: containers spread over devices 0 .. k-1
        end function
        integer(c_int) function spd_modelstate_init_on(state_cnt, device) bind(C, name="spd_modelstate_init_on")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(out) :: state_cnt
            integer(c_int32_t), value :: device
        end function
        integer(c_int) function spd_modelstate_device(state_cnt, device) bind(C, name="spd_modelstate_device")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), value :: state_cnt
            integer(c_int32_t), intent(out) :: device
        end function
        integer(c_int) function spd_broadcast_boundary(state_cnts, n, root) bind(C, name="spd_broadcast_boundary")
            import :: c_int, c_int64_t, c_int32_t
            integer(c_int64_t), intent(in) :: state_cnts(*)
            integer(c_int32_t), value :: n, root     ! root: 0-based index into state_cnts
        end function
        integer(c_int) function spd_broadcast_boundary_stats(peer_copies, local_copies, collective_devices) &
                bind(C, name="spd_broadcast_boundary_stats")
            import :: c_int, c_int32_t
            integer(c_int32_t), intent(out) :: peer_copies, local_copies, collective_devices
        end function
        function spd_broadcast_boundary_note() bind(C, name="spd_broadcast_boundary_note") result(text)
            import :: c_ptr
            type(c_ptr) :: text   ! NUL-terminated, valid until this thread's next call
        end function
    end interface
end module pyspeedy_amd_c
