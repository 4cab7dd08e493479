!########################################################################
! ISO_C_BINDING interfaces of the C ABI in include/tlab_amd.h.
! One interface per entry point; names and argument order are those of the header.
!########################################################################
module TLab_AMD_C
    use, intrinsic :: iso_c_binding
    implicit none
    public

    ! struct tlab_slab_transport (include/tlab_amd.h): filled by tlab_comm_slab_transport / tlab_slab_transport_loopback, or by the host with the
    ! five entry points of its own (GPU-aware MPI)
    type, bind(C) :: tlab_slab_transport
        type(c_ptr) :: ctx = c_null_ptr
        integer(c_int) :: nranks = 1, nlocal = 1, first = 0
        type(c_funptr) :: ring_start = c_null_funptr, alltoallv_start = c_null_funptr, wait = c_null_funptr, allreduce = c_null_funptr, &
                          destroy = c_null_funptr
    end type tlab_slab_transport

    ! struct tlab_pencil_transport: MPI_Alltoallv inside the world / ims_comm_x / ims_comm_z communicators for the x/z pencil driver (tlab_pencil_dns_*)
    type, bind(C) :: tlab_pencil_transport
        type(c_ptr) :: ctx = c_null_ptr
        integer(c_int) :: npro_i = 1, npro_k = 1, nlocal = 1, first = 0
        type(c_funptr) :: alltoallv_start = c_null_funptr, wait = c_null_funptr, allreduce = c_null_funptr, destroy = c_null_funptr
    end type tlab_pencil_transport

    interface
        integer(c_int) function tlab_finalize() bind(C, name='tlab_finalize')
            import :: c_int
        end function
        integer(c_int) function tlab_fdm_plan_create(plan, n, nodes, periodic, uniform, scheme1, scheme2, hyper_bc1_ext) bind(C, name='tlab_fdm_plan_create')
            import :: c_int, c_ptr, c_double
            type(c_ptr), intent(out) :: plan
            integer(c_int), value :: n, periodic, uniform, scheme1, scheme2
            real(c_double), intent(in) :: nodes(*)
            real(c_double), value :: hyper_bc1_ext
        end function
        integer(c_int) function tlab_device_count() bind(C, name='tlab_device_count')
            import :: c_int
        end function
        integer(c_int) function tlab_init(device) bind(C, name='tlab_init')
            import :: c_int
            integer(c_int), value :: device
        end function
        integer(c_int) function tlab_sync() bind(C, name='tlab_sync')
            import :: c_int
        end function
        function tlab_last_error() bind(C, name='tlab_last_error') result(p)
            import :: c_ptr
            type(c_ptr) :: p
        end function
        integer(c_int) function tlab_malloc(p, bytes) bind(C, name='tlab_malloc')
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), intent(out) :: p
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function tlab_free(p) bind(C, name='tlab_free')
            import :: c_int, c_ptr
            type(c_ptr), value :: p
        end function
        integer(c_int) function tlab_memcpy_h2d(dst, src, bytes) bind(C, name='tlab_memcpy_h2d')
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: dst, src
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function tlab_memcpy_d2h(dst, src, bytes) bind(C, name='tlab_memcpy_d2h')
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: dst, src
            integer(c_size_t), value :: bytes
        end function
        integer(c_int) function tlab_fdm_plan_create_from_arrays(plan, n, periodic, need_1der, ndl1, ndr1, lhs1, rhs1, &
                                                                 ndl2, ndr2, lhs2, rhs2) bind(C, name='tlab_fdm_plan_create_from_arrays')
            import :: c_int, c_ptr, c_double
            type(c_ptr), intent(out) :: plan
            integer(c_int), value :: n, periodic, need_1der, ndl1, ndr1, ndl2, ndr2
            real(c_double), intent(in) :: lhs1(*), rhs1(*), lhs2(*), rhs2(*)
        end function
        integer(c_int) function tlab_fdm_plan_set_aux(plan, mwn1, mwn2, jac, nodes) bind(C, name='tlab_fdm_plan_set_aux')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, mwn1, mwn2, jac, nodes
        end function
        integer(c_int) function tlab_fdm_plan_set_scheme(plan, mode1, mode2) bind(C, name='tlab_fdm_plan_set_scheme')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: mode1, mode2
        end function
        integer(c_int) function tlab_fdm_plan_set_stagger(plan, mode) bind(C, name='tlab_fdm_plan_set_stagger')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: mode
        end function
        integer(c_int) function tlab_fdm_plan_destroy(plan) bind(C, name='tlab_fdm_plan_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
        end function
        integer(c_int) function tlab_opr_partial(dir, plan, itype, nx, ny, nz, ibc, u, res, tmp1) bind(C, name='tlab_opr_partial')
            import :: c_int, c_ptr
            integer(c_int), value :: dir, itype, nx, ny, nz, ibc
            type(c_ptr), value :: plan, u, res, tmp1
        end function
        integer(c_int) function tlab_opr_burgers(dir, plan, ivel, nx, ny, nz, ibc, nu, s, u, res, tmp1, write_transposed) &
            bind(C, name='tlab_opr_burgers')
            import :: c_int, c_ptr, c_double
            integer(c_int), value :: dir, ivel, nx, ny, nz, ibc, write_transposed
            real(c_double), value :: nu
            type(c_ptr), value :: plan, s, u, res, tmp1
        end function
        integer(c_int) function tlab_opr_burgers_set_anelastic(ny, rbackground, ribackground) bind(C, name='tlab_opr_burgers_set_anelastic')
            import :: c_int, c_ptr
            integer(c_int), value :: ny
            type(c_ptr), value :: rbackground, ribackground      ! host arrays of ny doubles (c_null_ptr: incompressible)
        end function
        integer(c_int) function tlab_filter_create(f, itype, n, periodic, bcsmin, bcsmax, inb_filter, coeffs) bind(C, name='tlab_filter_create')
            import :: c_int, c_ptr
            type(c_ptr), intent(out) :: f
            integer(c_int), value :: itype, n, periodic, bcsmin, bcsmax, inb_filter
            type(c_ptr), value :: coeffs                          ! host array f%coeffs(n, inb_filter)
        end function
        integer(c_int) function tlab_filter_destroy(f) bind(C, name='tlab_filter_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: f
        end function
        integer(c_int) function tlab_opr_filter_1d(dir, f, nx, ny, nz, u, res) bind(C, name='tlab_opr_filter_1d')
            import :: c_int, c_ptr
            integer(c_int), value :: dir, nx, ny, nz
            type(c_ptr), value :: f, u, res
        end function
        integer(c_int) function tlab_opr_burgers_set_dealiasing(dir, f) bind(C, name='tlab_opr_burgers_set_dealiasing')
            import :: c_int, c_ptr
            integer(c_int), value :: dir
            type(c_ptr), value :: f
        end function
        integer(c_int) function tlab_dns_set_anelastic(dns, rbackground, ribackground) bind(C, name='tlab_dns_set_anelastic')
            import :: c_int, c_ptr
            type(c_ptr), value :: dns, rbackground, ribackground
        end function
        integer(c_int) function tlab_poisson_plan_create(plan, gx, gy, gz, nx, ny, nz) bind(C, name='tlab_poisson_plan_create')
            import :: c_int, c_ptr
            type(c_ptr), intent(out) :: plan
            type(c_ptr), value :: gx, gy, gz
            integer(c_int), value :: nx, ny, nz
        end function
        integer(c_int) function tlab_poisson_plan_create_direct(plan, gx, gy, gz, nx, ny, nz, gy_elliptic) &
            bind(C, name='tlab_poisson_plan_create_direct')
            import :: c_int, c_ptr
            type(c_ptr), intent(out) :: plan
            type(c_ptr), value :: gx, gy, gz, gy_elliptic
            integer(c_int), value :: nx, ny, nz
        end function
        integer(c_int) function tlab_opr_poisson(plan, nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy) bind(C, name='tlab_opr_poisson')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy
            integer(c_int), value :: nx, ny, nz, ibc
        end function
        integer(c_int) function tlab_opr_helmholtz(plan, nx, ny, nz, ibc, alpha, a, tmp1, tmp2, bcs_hb, bcs_ht) bind(C, name='tlab_opr_helmholtz')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: plan, a, tmp1, tmp2, bcs_hb, bcs_ht
            integer(c_int), value :: nx, ny, nz, ibc
            real(c_double), value :: alpha
        end function
        integer(c_int) function tlab_transpose(a, nra, nca, b) bind(C, name='tlab_transpose')
            import :: c_int, c_ptr
            type(c_ptr), value :: a, b
            integer(c_int), value :: nra, nca
        end function
        ! ---- RHS assembly / Runge-Kutta substep / wall boundary conditions (include/tlab_amd.h, "next" row n1) ----
        integer(c_int) function tlab_dns_create(dns, gx, gy, gz, poisson, nx, ny, nz, nscal, visc, schmidt) bind(C, name='tlab_dns_create')
            import :: c_int, c_ptr, c_double
            type(c_ptr), intent(out) :: dns
            type(c_ptr), value :: gx, gy, gz, poisson
            integer(c_int), value :: nx, ny, nz, nscal
            real(c_double), value :: visc
            real(c_double), intent(in) :: schmidt(*)
        end function
        integer(c_int) function tlab_dns_destroy(dns) bind(C, name='tlab_dns_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: dns
        end function
        integer(c_int) function tlab_dns_set_bcs(dns, flow_jmin, flow_jmax, scal_jmin, scal_jmax) bind(C, name='tlab_dns_set_bcs')
            import :: c_int, c_ptr
            type(c_ptr), value :: dns
            integer(c_int), intent(in) :: flow_jmin(3), flow_jmax(3), scal_jmin(*), scal_jmax(*)
        end function
        integer(c_int) function tlab_dns_set_remove_divergence(dns, on) bind(C, name='tlab_dns_set_remove_divergence')
            import :: c_int, c_ptr
            type(c_ptr), value :: dns
            integer(c_int), value :: on
        end function
        integer(c_int) function tlab_dns_set_surface_bcs(dns, sfc_jmin, sfc_jmax, cpl_jmin, cpl_jmax) bind(C, name='tlab_dns_set_surface_bcs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: dns
            integer(c_int), intent(in) :: sfc_jmin(*), sfc_jmax(*)
            real(c_double), intent(in) :: cpl_jmin(*), cpl_jmax(*)
        end function
        integer(c_int) function tlab_dns_begin_step(dns) bind(C, name='tlab_dns_begin_step')
            import :: c_int, c_ptr
            type(c_ptr), value :: dns
        end function
        integer(c_int) function tlab_dns_place_blocks(dns, ncand, cand_q, cand_s, cand_hq, cand_hs, cand_txc, txc_stride, dtime, random_trials, seed, &
                                                      choice, report) bind(C, name='tlab_dns_place_blocks')
            import :: c_int, c_ptr, c_double, c_long_long
            type(c_ptr), value :: dns
            integer(c_int), value :: ncand, random_trials, seed
            type(c_ptr), intent(in) :: cand_q(*), cand_s(*), cand_hq(*), cand_hs(*), cand_txc(*)
            integer(c_long_long), value :: txc_stride
            real(c_double), value :: dtime
            integer(c_int), intent(out) :: choice(5)
            real(c_double), intent(out) :: report(5)
        end function
        ! ---- the substep's tail for an unpatched time loop (include/tlab_amd.h: tlab_deferred_*, csrc/deferred.cpp) ----
        integer(c_int) function tlab_deferred_enable(on) bind(C, name='tlab_deferred_enable')
            import :: c_int
            integer(c_int), value :: on
        end function
        integer(c_int) function tlab_deferred_rhs(dns, dte, q, s, hq, hs, txc) bind(C, name='tlab_deferred_rhs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: dns
            real(c_double), value :: dte
            type(c_ptr), intent(in) :: q(*), s(*), hq(*), hs(*), txc(*)
        end function
        integer(c_int) function tlab_deferred_slab_rhs(slab, dte) bind(C, name='tlab_deferred_slab_rhs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: slab
            real(c_double), value :: dte
        end function
        integer(c_int) function tlab_deferred_pencil_rhs(pencil, dte) bind(C, name='tlab_deferred_pencil_rhs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: pencil
            real(c_double), value :: dte
        end function
        integer(c_int) function tlab_deferred_axpy(n, a, x, y) bind(C, name='tlab_deferred_axpy')
            import :: c_int, c_ptr, c_double, c_long_long
            integer(c_long_long), value :: n
            real(c_double), value :: a
            type(c_ptr), value :: x, y
        end function
        integer(c_int) function tlab_deferred_scal(n, a, x) bind(C, name='tlab_deferred_scal')
            import :: c_int, c_ptr, c_double, c_long_long
            integer(c_long_long), value :: n
            real(c_double), value :: a
            type(c_ptr), value :: x
        end function
        integer(c_int) function tlab_deferred_zero(a, n) bind(C, name='tlab_deferred_zero')
            import :: c_int, c_ptr, c_long_long
            type(c_ptr), value :: a
            integer(c_long_long), value :: n
        end function
        integer(c_int) function tlab_deferred_flush() bind(C, name='tlab_deferred_flush')
            import :: c_int
        end function
        integer(c_int) function tlab_deferred_stats(counts) bind(C, name='tlab_deferred_stats')
            import :: c_int, c_long_long
            integer(c_long_long), intent(out) :: counts(6)
        end function
        integer(c_int) function tlab_rhs_global_incompressible_1(dns, dte, q, s, hq, hs, txc) bind(C, name='tlab_rhs_global_incompressible_1')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: dns
            real(c_double), value :: dte
            type(c_ptr), intent(in) :: q(*), s(*), hq(*), hs(*), txc(*)        ! host arrays of device pointers
        end function
        integer(c_int) function tlab_time_substep_incompressible_explicit(dns, dte, kco, scale_tendencies, q, s, hq, hs, txc) &
            bind(C, name='tlab_time_substep_incompressible_explicit')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: dns
            real(c_double), value :: dte, kco
            integer(c_int), value :: scale_tendencies
            type(c_ptr), intent(in) :: q(*), s(*), hq(*), hs(*), txc(*)
        end function
        integer(c_int) function tlab_time_courant(dns, q, cfla, cfld, pmax, dtime) bind(C, name='tlab_time_courant')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: dns
            type(c_ptr), intent(in) :: q(*)
            real(c_double), value :: cfla, cfld
            real(c_double), intent(out) :: pmax(2), dtime
        end function
        integer(c_int) function tlab_pw_rk_update(q, h, dte, kco, scale, n) bind(C, name='tlab_pw_rk_update')
            import :: c_int, c_ptr, c_double, c_long_long
            type(c_ptr), value :: q, h
            real(c_double), value :: dte, kco
            integer(c_int), value :: scale
            integer(c_long_long), value :: n
        end function
        integer(c_int) function tlab_pw_fill(a, val, n) bind(C, name='tlab_pw_fill')
            import :: c_int, c_ptr, c_double, c_long_long
            type(c_ptr), value :: a
            real(c_double), value :: val
            integer(c_long_long), value :: n
        end function
        integer(c_int) function tlab_pw_scale(a, alpha, n) bind(C, name='tlab_pw_scale')
            import :: c_int, c_ptr, c_double, c_long_long
            type(c_ptr), value :: a
            real(c_double), value :: alpha
            integer(c_long_long), value :: n
        end function
        ! ---- x/z pencils (ims_npro_i > 1): tlab_pencil_dns_* of include/tlab_amd.h ----
        integer(c_int) function tlab_pencil_transport_loopback(tr, npro_i, npro_k) bind(C, name='tlab_pencil_transport_loopback')
            import :: c_int, tlab_pencil_transport
            type(tlab_pencil_transport), intent(out) :: tr
            integer(c_int), value :: npro_i, npro_k
        end function
        ! the partitioned z systems of a slab (include/tlab_amd.h): used here as the test whether slabs of kmax planes are thick enough (TLAB_EUNSUPPORTED: no)
        integer(c_int) function tlab_zslab_plan_create(plan, gz, kmax, koffset, chunk) bind(C, name='tlab_zslab_plan_create')
            import :: c_int, c_ptr
            type(c_ptr), intent(out) :: plan
            type(c_ptr), value :: gz
            integer(c_int), value :: kmax, koffset, chunk
        end function
        integer(c_int) function tlab_zslab_plan_destroy(plan) bind(C, name='tlab_zslab_plan_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
        end function
        integer(c_int) function tlab_pencil_dns_create(d, tr, gx, gy, gz, nx, ny, nz_total, nscal, visc, schmidt) bind(C, name='tlab_pencil_dns_create')
            import :: c_int, c_ptr, c_double, tlab_pencil_transport
            type(c_ptr), intent(out) :: d
            type(tlab_pencil_transport), intent(in) :: tr
            type(c_ptr), value :: gx, gy, gz
            integer(c_int), value :: nx, ny, nz_total, nscal
            real(c_double), value :: visc
            real(c_double), intent(in) :: schmidt(*)
        end function
        integer(c_int) function tlab_pencil_dns_destroy(d) bind(C, name='tlab_pencil_dns_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
        end function
        integer(c_int) function tlab_pencil_dns_bind(d, l, q, s, hq, hs, txc) bind(C, name='tlab_pencil_dns_bind')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
            integer(c_int), value :: l
            type(c_ptr), intent(in) :: q(*), s(*), hq(*), hs(*), txc(*)
        end function
        integer(c_long_long) function tlab_pencil_dns_info(d, what) bind(C, name='tlab_pencil_dns_info')
            import :: c_int, c_ptr, c_long_long
            type(c_ptr), value :: d
            integer(c_int), value :: what
        end function
        integer(c_int) function tlab_pencil_dns_set_bcs(d, flow_jmin, flow_jmax, scal_jmin, scal_jmax) bind(C, name='tlab_pencil_dns_set_bcs')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
            integer(c_int), intent(in) :: flow_jmin(*), flow_jmax(*), scal_jmin(*), scal_jmax(*)
        end function
        integer(c_int) function tlab_pencil_dns_begin_step(d) bind(C, name='tlab_pencil_dns_begin_step')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
        end function
        integer(c_int) function tlab_pencil_dns_rhs(d, dte) bind(C, name='tlab_pencil_dns_rhs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), value :: dte
        end function
        integer(c_int) function tlab_pencil_dns_substep(d, dte, kco, scale) bind(C, name='tlab_pencil_dns_substep')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), value :: dte, kco
            integer(c_int), value :: scale
        end function
        ! ---- the decomposed substep (ims_npro_k > 1): tlab_slab_dns_* of include/tlab_amd.h ----
        integer(c_int) function tlab_slab_transport_loopback(tr, nranks) bind(C, name='tlab_slab_transport_loopback')
            import :: c_int, tlab_slab_transport
            type(tlab_slab_transport), intent(out) :: tr
            integer(c_int), value :: nranks
        end function
        integer(c_int) function tlab_slab_dns_create(d, tr, gx, gy, gz, nx, ny, nz_total, nscal, visc, schmidt, gy_elliptic) bind(C, name='tlab_slab_dns_create')
            import :: c_int, c_ptr, c_double, tlab_slab_transport
            type(c_ptr), intent(out) :: d
            type(tlab_slab_transport), intent(in) :: tr
            type(c_ptr), value :: gx, gy, gz, gy_elliptic
            integer(c_int), value :: nx, ny, nz_total, nscal
            real(c_double), value :: visc
            real(c_double), intent(in) :: schmidt(*)
        end function
        integer(c_int) function tlab_slab_dns_destroy(d) bind(C, name='tlab_slab_dns_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
        end function
        integer(c_int) function tlab_slab_dns_bind(d, l, q, s, hq, hs, txc) bind(C, name='tlab_slab_dns_bind')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
            integer(c_int), value :: l
            type(c_ptr), intent(in) :: q(*), s(*), hq(*), hs(*), txc(*)
        end function
        integer(c_long_long) function tlab_slab_dns_info(d, what) bind(C, name='tlab_slab_dns_info')
            import :: c_int, c_ptr, c_long_long
            type(c_ptr), value :: d
            integer(c_int), value :: what
        end function
        integer(c_int) function tlab_slab_dns_set_bcs(d, flow_jmin, flow_jmax, scal_jmin, scal_jmax) bind(C, name='tlab_slab_dns_set_bcs')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
            integer(c_int), intent(in) :: flow_jmin(*), flow_jmax(*), scal_jmin(*), scal_jmax(*)
        end function
        integer(c_int) function tlab_slab_dns_begin_step(d) bind(C, name='tlab_slab_dns_begin_step')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
        end function
        integer(c_int) function tlab_slab_dns_set_surface_bcs(d, sfc_jmin, sfc_jmax, cpl_jmin, cpl_jmax) bind(C, name='tlab_slab_dns_set_surface_bcs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            integer(c_int), intent(in) :: sfc_jmin(*), sfc_jmax(*)
            real(c_double), intent(in) :: cpl_jmin(*), cpl_jmax(*)
        end function
        integer(c_int) function tlab_slab_dns_set_remove_divergence(d, on) bind(C, name='tlab_slab_dns_set_remove_divergence')
            import :: c_int, c_ptr
            type(c_ptr), value :: d
            integer(c_int), value :: on
        end function
        integer(c_int) function tlab_slab_dns_rhs(d, dte) bind(C, name='tlab_slab_dns_rhs')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), value :: dte
        end function
        integer(c_int) function tlab_slab_dns_substep(d, dte, kco, scale_tendencies) bind(C, name='tlab_slab_dns_substep')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), value :: dte, kco
            integer(c_int), value :: scale_tendencies
        end function
        integer(c_int) function tlab_slab_dns_time_courant(d, cfla, cfld, pmax, dtime) bind(C, name='tlab_slab_dns_time_courant')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), value :: cfla, cfld
            real(c_double), intent(out) :: pmax(2), dtime
        end function
        integer(c_int) function tlab_slab_dns_dilatation_bounds(d, dil_min, dil_max) bind(C, name='tlab_slab_dns_dilatation_bounds')
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: d
            real(c_double), intent(out) :: dil_min, dil_max
        end function
        integer(c_int) function tlab_boundary_bcs_neumann_y(plan, ibc, nx, ny, nz, u, bcs_hb, bcs_ht, tmp1) bind(C, name='tlab_boundary_bcs_neumann_y')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, u, bcs_hb, bcs_ht, tmp1
            integer(c_int), value :: ibc, nx, ny, nz
        end function
    end interface

contains
    ! Error convention of the reference: a line in tlab.err and TLab_Stop (base/tlab_workflow.f90:105-207); codes from include/dns_error.h.
    subroutine TLab_AMD_Check(rc, what)
        use TLab_Constants, only: efile
        use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
        integer(c_int), intent(in) :: rc
        character(len=*), intent(in) :: what
        character(kind=c_char), pointer :: msg(:)
        character(len=512) :: text
        integer i
        integer, parameter :: DNS_ERROR_UNDEVELOP = 104          ! include/dns_error.h:99
        if (rc == 0) return
        call c_f_pointer(tlab_last_error(), msg, [512])
        text = ''
        do i = 1, 512
            if (msg(i) == c_null_char) exit
            text(i:i) = msg(i)
        end do
        call TLab_Write_ASCII(efile, 'tlab_amd. '//what//' failed: '//trim(text))
        write (*, '(a)') 'tlab_amd: '//what//' failed: '//trim(text)
        call TLab_Stop(DNS_ERROR_UNDEVELOP)
    end subroutine TLab_AMD_Check

end module TLab_AMD_C
