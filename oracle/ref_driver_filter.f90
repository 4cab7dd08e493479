!########################################################################
! TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
!
! C-callable driver around the reference's own 1-D filter kernels (src/filters/flt_compact.f90, flt_explitic.f90, compiled where they lie) and
! its own solvers (utils/linear3.f90, linear5.f90).  operators/opr_filter.f90 itself cannot be compiled in this image (it uses OPR_Fourier, which
! needs fftw3.f03), so the two routines below transcribe the case lists of OPR_FILTER_INITIALIZE (opr_filter.f90:236-275) and OPR_FILTER_1D
! (:393-460) for the filter types COMPACT (1), 6E (2), 4E (3), COMPACT_CUTOFF (9) -- every arithmetic statement runs in the reference's modules.
!########################################################################
subroutine ref_filter_init(itype, n, periodic, bcsmin, bcsmax, alpha, jac, scale, nodes, ncols, coeffs) bind(C, name='ref_filter_init')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use Filters_Compact
    use Filters_Explicit
    implicit none
    integer(c_int), value :: itype, n, periodic, bcsmin, bcsmax, ncols
    real(c_double), value :: alpha, scale
    real(c_double), intent(in) :: jac(n), nodes(n)
    real(c_double), intent(inout) :: coeffs(n, ncols)
    logical per
    per = periodic /= 0
    coeffs = 0.0_wp
    select case (itype)
    case (3)                                                             ! DNS_FILTER_4E
        call FLT_E4_COEFFS(n, per, scale, nodes, coeffs)
    case (1)                                                             ! DNS_FILTER_COMPACT
        call FLT_C4_LHS(n, bcsmin, bcsmax, alpha, coeffs(1, 6), coeffs(1, 7), coeffs(1, 8))
        if (per) then
            call TRIDPFS(n, coeffs(1, 6), coeffs(1, 7), coeffs(1, 8), coeffs(1, 9), coeffs(1, 10))
        else
            call TRIDFS(n, coeffs(1, 6), coeffs(1, 7), coeffs(1, 8))
        end if
        call FLT_C4_RHS_COEFFS(n, alpha, per, jac, coeffs(1, 1))
    case (9)                                                             ! DNS_FILTER_COMPACT_CUTOFF
        if (per) then
            call FLT_C4P_CUTOFF_LHS(n, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5))
            call PENTADPFS(n, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5), coeffs(1, 6), coeffs(1, 7))
        else
            call FLT_C4_CUTOFF_LHS(n, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5))
            call PENTADFS2(n, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5))
        end if
    end select
end subroutine ref_filter_init

! u, res: (nlines, n) Fortran order, i.e. C arrays [n][nlines]
subroutine ref_filter_1d(itype, n, nlines, periodic, bcsmin, bcsmax, ncols, coeffs, u, res) bind(C, name='ref_filter_1d')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use TLab_Arrays, only: wrk2d
    use Filters_Compact
    use Filters_Explicit
    implicit none
    integer(c_int), value :: itype, n, nlines, periodic, bcsmin, bcsmax, ncols
    real(c_double), intent(in) :: coeffs(n, ncols)
    real(c_double), intent(in) :: u(nlines, n)
    real(c_double), intent(out) :: res(nlines, n)
    logical per
    per = periodic /= 0
    select case (itype)
    case (1)
        call FLT_C4_RHS(n, nlines, per, bcsmin, bcsmax, coeffs, u, res)
        if (per) then
            call TRIDPSS(n, nlines, coeffs(1, 6), coeffs(1, 7), coeffs(1, 8), coeffs(1, 9), coeffs(1, 10), res, wrk2d)
        else
            call TRIDSS(n, nlines, coeffs(1, 6), coeffs(1, 7), coeffs(1, 8), res)
        end if
    case (9)
        if (per) then
            call FLT_C4P_CUTOFF_RHS(n, nlines, u, res)
            call PENTADPSS(n, nlines, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5), coeffs(1, 6), coeffs(1, 7), res)
        else
            call FLT_C4_CUTOFF_RHS(n, nlines, u, res)
            call PENTADSS2(n, nlines, coeffs(1, 1), coeffs(1, 2), coeffs(1, 3), coeffs(1, 4), coeffs(1, 5), res)
        end if
    case (2)
        call FLT_E6(n, nlines, per, bcsmin, bcsmax, u, res)
    case (3)
        call FLT_E4(n, nlines, per, coeffs, u, res)
    end select
end subroutine ref_filter_1d

!########################################################################
! horizontal staggering of the pressure ([Staggering] StaggerHorizontalPressure = yes): the reference keeps it in the module variable
! TLab_WorkFlow::stagger_on, read by FDM_CreatePlan (fdm/fdm.f90:236-248: g%intl, and g%der1%mwn becomes the interpolatory one)
subroutine ref_set_stagger(on) bind(C, name='ref_set_stagger')
    use iso_c_binding
    use TLab_WorkFlow, only: stagger_on
    implicit none
    integer(c_int), value :: on
    stagger_on = on /= 0
end subroutine ref_set_stagger

! which = 0: g%intl%lu0i(n, 5), 1: g%intl%lu1i(n, 5)   (fdm/fdm_interpolate.f90:14-21)
subroutine ref_intl_get(idir, which, n, buf) bind(C, name='ref_intl_get')
    use iso_c_binding
    use TLab_Constants, only: wp
    use ref_state
    implicit none
    integer(c_int), value :: idir, which, n
    real(c_double), intent(out) :: buf(n*5)
    if (which == 0) then
        buf = reshape(gp(idir)%intl%lu0i, [n*5])
    else
        buf = reshape(gp(idir)%intl%lu1i, [n*5])
    end if
end subroutine ref_intl_get
