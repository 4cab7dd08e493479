!########################################################################
! TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
!
! C-callable entry points around the reference's first-order integral operators and second-order ODE
! solvers, i.e. the per-Fourier-mode arithmetic of OPR_Poisson_FourierXZ_Factorize
! (src/operators/opr_elliptic.f90:308-333).  The top-level OPR_Poisson itself cannot be compiled in this image
! (opr_fourier.f90 needs the absent FFTW header fftw3.f03), so FFT parity is pinned separately (~1e-15) and
! these entry points pin everything else:
!
!   FDM_Int1_CreateSystem / FDM_Int1_Initialize / FDM_Int1_Solve   src/fdm/fdm_integral.f90:91,58,219
!   OPR_ODE2_Factorize_NN / _NN_Sing / _DD / _DD_Sing              src/operators/opr_odes.f90:265,165,391,188
!
! The pair of integral plans is built exactly as OPR_Elliptic_Initialize does (opr_elliptic.f90:205-209):
!   fdm_int1(BCS_MIN) with +lambda, fdm_int1(BCS_MAX) with -lambda, both from the y plan gp(2)%der1.
!########################################################################

! lambda here is sqrt(kx^2+kz^2), i.e. the constant of the first-order equations
subroutine ref_int1_create(lambda, ibc, factorize) bind(C, name='ref_int1_create')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM_Integral
    use ref_state
    implicit none
    real(c_double), value :: lambda
    integer(c_int), value :: ibc, factorize

    if (factorize /= 0) then
        call FDM_Int1_Initialize(gp(2)%nodes(:), gp(2)%der1, lambda, ibc, fint(ibc))
    else
        call FDM_Int1_CreateSystem(gp(2)%nodes(:), gp(2)%der1, lambda, ibc, fint(ibc))
    end if
end subroutine ref_int1_create

! which: 1 lhs(n,5)  2 rhs(n,3)  3 rhs_b(5,0:7)  4 rhs_t(0:4,8)
subroutine ref_int1_get(ibc, which, buf, nbuf) bind(C, name='ref_int1_get')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use ref_state
    implicit none
    integer(c_int), value :: ibc, which, nbuf
    real(c_double), intent(out) :: buf(nbuf)
    integer m
    buf(:) = 0.0_wp
    select case (which)
    case (1); m = size(fint(ibc)%lhs); buf(1:m) = reshape(fint(ibc)%lhs, [m])
    case (2); m = size(fint(ibc)%rhs); buf(1:m) = reshape(fint(ibc)%rhs, [m])
    case (3); m = size(fint(ibc)%rhs_b); buf(1:m) = reshape(fint(ibc)%rhs_b, [m])
    case (4); m = size(fint(ibc)%rhs_t); buf(1:m) = reshape(fint(ibc)%rhs_t, [m])
    end select
end subroutine ref_int1_get

! f(nlines, n) in; res(nlines, n) inout (carries the boundary value); du(nlines) out
subroutine ref_int1_solve(ibc, nlines, f, res, du) bind(C, name='ref_int1_solve')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM_Integral
    use ref_state
    implicit none
    integer(c_int), value :: ibc, nlines
    real(c_double), intent(in) :: f(nlines, gp(2)%size)
    real(c_double), intent(inout) :: res(nlines, gp(2)%size)
    real(c_double), intent(out) :: du(nlines)
    real(wp), allocatable :: w(:, :)
    allocate (w(nlines, 2))
    call FDM_Int1_Solve(nlines, fint(ibc), fint(ibc)%rhs, f, res, w, du)
    deallocate (w)
end subroutine ref_int1_solve

! itype: 1 NN, 2 NN_Sing (lambda ignored = 0), 3 DD, 4 DD_Sing.  f(nlines,n) is modified like in the reference.
subroutine ref_ode2(itype, nlines, lambda, f, bcs, u, v) bind(C, name='ref_ode2')
    use iso_c_binding
    use TLab_Constants, only: wp, wi, BCS_MIN, BCS_MAX
    use FDM_Integral
    use OPR_ODES
    use ref_state
    implicit none
    integer(c_int), value :: itype, nlines
    real(c_double), value :: lambda
    real(c_double), intent(inout) :: f(nlines, gp(2)%size), bcs(nlines, 2)
    real(c_double), intent(out) :: u(nlines, gp(2)%size), v(nlines, gp(2)%size)
    real(wp), allocatable :: w1(:, :), w2(:, :)
    real(wp) lam
    integer(wi) n

    n = gp(2)%size
    lam = lambda
    if (itype == 2 .or. itype == 4) lam = 0.0_wp
    call FDM_Int1_Initialize(gp(2)%nodes(:), gp(2)%der1, lam, BCS_MIN, fint(BCS_MIN))
    call FDM_Int1_Initialize(gp(2)%nodes(:), gp(2)%der1, -lam, BCS_MAX, fint(BCS_MAX))
    allocate (w1(n, 12), w2(max(nlines, 3), 3))      ! wrk1d(3,n,2) / wrk1d(n,4); wrk2d(max(nlines,3),3)
    w1 = 0.0_wp; w2 = 0.0_wp
    select case (itype)
    case (1); call OPR_ODE2_Factorize_NN(nlines, fint, fint(BCS_MIN)%rhs, fint(BCS_MAX)%rhs, u, f, bcs, v, w1, w2)
    case (2); call OPR_ODE2_Factorize_NN_Sing(nlines, fint, u, f, bcs, v, w1, w2)
    case (3); call OPR_ODE2_Factorize_DD(nlines, fint, fint(BCS_MIN)%rhs, fint(BCS_MAX)%rhs, u, f, bcs, v, w1, w2)
    case (4); call OPR_ODE2_Factorize_DD_Sing(nlines, fint, u, f, bcs, v, w1, w2)
    end select
    deallocate (w1, w2)
end subroutine ref_ode2


!########################################################################
! Second-order integral operators of the DIRECT elliptic solver (EllipticOrder = CompactDirect*,
! OPR_Poisson_FourierXZ_Direct, src/operators/opr_elliptic.f90:368-455):
!   FDM_Int2_CreateSystem / FDM_Int2_Initialize / FDM_Int2_Solve     src/fdm/fdm_integral.f90:366,334,626
! built from gp(2)%der2 exactly as OPR_Elliptic_Initialize does (opr_elliptic.f90:236-242); lambda2 = kx^2 + kz^2 here.
!########################################################################
subroutine ref_int2_create(lambda2, ibc, factorize) bind(C, name='ref_int2_create')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM_Integral
    use ref_state
    implicit none
    real(c_double), value :: lambda2
    integer(c_int), value :: ibc, factorize

    ! (FDM_Int2_CreateSystem is private to the module: only the factorized system can be inspected)
    call FDM_Int2_Initialize(gp(2)%nodes(:), gp(2)%der2, lambda2, ibc, fint2)
end subroutine ref_int2_create

! which: 1 lhs(n,ndr)  2 rhs(n,ndl)  3 rhs_b(5,0:7)  4 rhs_t(0:4,8)
subroutine ref_int2_get(which, buf, nbuf) bind(C, name='ref_int2_get')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use ref_state
    implicit none
    integer(c_int), value :: which, nbuf
    real(c_double), intent(out) :: buf(nbuf)
    integer m
    buf(:) = 0.0_wp
    select case (which)
    case (1); m = size(fint2%lhs); buf(1:m) = reshape(fint2%lhs, [m])
    case (2); m = size(fint2%rhs); buf(1:m) = reshape(fint2%rhs, [m])
    case (3); m = size(fint2%rhs_b); buf(1:m) = reshape(fint2%rhs_b, [m])
    case (4); m = size(fint2%rhs_t); buf(1:m) = reshape(fint2%rhs_t, [m])
    end select
end subroutine ref_int2_get

! f(nlines, n) in; res(nlines, n) inout (carries both boundary values)
subroutine ref_int2_solve(nlines, f, res) bind(C, name='ref_int2_solve')
    use iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM_Integral
    use ref_state
    implicit none
    integer(c_int), value :: nlines
    real(c_double), intent(in) :: f(nlines, gp(2)%size)
    real(c_double), intent(inout) :: res(nlines, gp(2)%size)
    real(wp), allocatable :: w(:, :)
    allocate (w(nlines, 2))
    call FDM_Int2_Solve(nlines, fint2, fint2%rhs, f, res, w)
    deallocate (w)
end subroutine ref_int2_solve
