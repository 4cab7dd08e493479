!########################################################################
! Drop-in replacement of the Poisson entry of module OPR_Elliptic (operators/opr_elliptic.f90): the procedure pointer OPR_Poisson with
! the abstract interface of :33-46, bound to the device solver: OPR_Poisson_FourierXZ_Factorize (:263-364; BCS_NN and BCS_DD), or -- when the host hands
! over the elliptic plan fdm_loc it made for EllipticOrder = CompactDirect6 (:107-124) -- OPR_Poisson_FourierXZ_Direct (:368-455; all four BCs).
! OPR_Elliptic_Initialize(inifile) of the reference reads [Main] EllipticOrder and builds lambda / fdm_int1 (:86-250); here the plan is
! built from the host plans g(1:3) the unchanged FDM_Initialize made (their coefficient tables and modified wavenumbers).
! p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy must be device arrays (allocation hook, INTEGRATION.md section 3).
!########################################################################
#ifndef TLAB_AMD_ELLIPTIC_MODULE
#define TLAB_AMD_ELLIPTIC_MODULE OPR_Elliptic
#endif
#ifndef TLAB_AMD_PARTIAL_MODULE
#define TLAB_AMD_PARTIAL_MODULE OPR_Partial
#endif
module TLAB_AMD_ELLIPTIC_MODULE
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM, only: fdm_dt
    use TLab_AMD_C
    use TLAB_AMD_PARTIAL_MODULE, only: OPR_Partial_AMD_Plan
    implicit none
    private

    public :: OPR_Elliptic_Initialize_AMD       ! (g, nx, ny, nz [, fdm_loc]): what OPR_Elliptic_Initialize takes from modules FDM / TLab_Memory
#ifdef TLAB_AMD_FULL_HOST
    public :: OPR_Elliptic_Initialize           ! (inifile): operators/opr_elliptic.f90:86, in a complete host (modules TLab_Memory, TLab_Grid present)
#endif
    public :: OPR_Elliptic_AMD_Plan             ! the device plan (for the RHS driver, tlab_dns_create)
    public :: OPR_Elliptic_AMD_PlanY            ! the y plan of EllipticOrder = CompactDirect6 (fdm_loc), c_null_ptr for the factorized solver: tlab_slab_dns_create
    public :: OPR_Poisson
    public :: OPR_Helmholtz                     ! _FourierXZ_Direct (:562-628) or _FourierXZ_Factorize (:466-557), by the plan's type

    abstract interface
        subroutine OPR_Poisson_interface(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)      ! opr_elliptic.f90:33-46
            use TLab_Constants, only: wi, wp
            integer(wi), intent(in) :: nx, ny, nz
            integer, intent(in) :: ibc
            real(wp), intent(inout) :: p(nx, ny, nz)
            real(wp), intent(inout), target :: tmp1(2*ny, nz, nx/2 + 1)
            real(wp), intent(inout), target :: tmp2(2*ny, nz, nx/2 + 1)
            real(wp), intent(in) :: bcs_hb(nx, nz), bcs_ht(nx, nz)
            real(wp), intent(out), optional :: dpdy(nx, ny, nz)
        end subroutine
    end interface
    procedure(OPR_Poisson_interface), pointer :: OPR_Poisson => OPR_Poisson_AMD

    abstract interface
        subroutine OPR_Helmholtz_interface(nx, ny, nz, ibc, alpha, p, tmp1, tmp2, bcs_hb, bcs_ht)   ! opr_elliptic.f90:48-62
            use TLab_Constants, only: wi, wp
            integer(wi), intent(in) :: nx, ny, nz
            integer, intent(in) :: ibc
            real(wp), intent(in) :: alpha
            real(wp), intent(inout) :: p(nx, ny, nz)
            real(wp), intent(inout), target :: tmp1(2*ny, nz, nx/2 + 1)
            real(wp), intent(inout), target :: tmp2(2*ny, nz, nx/2 + 1)
            real(wp), intent(in) :: bcs_hb(nx, nz), bcs_ht(nx, nz)
        end subroutine
    end interface
    procedure(OPR_Helmholtz_interface), pointer :: OPR_Helmholtz => OPR_Helmholtz_AMD

    type(c_ptr), save :: plan = c_null_ptr
    type(c_ptr), save :: plan_elliptic_y = c_null_ptr      ! device copy of fdm_loc (direct solver only)
    type(fdm_dt), save, target :: fdm_loc                  ! opr_elliptic.f90:65

contains
    function OPR_Elliptic_AMD_Plan() result(p)
        type(c_ptr) :: p
        p = plan
    end function OPR_Elliptic_AMD_Plan

    function OPR_Elliptic_AMD_PlanY() result(p)
        type(c_ptr) :: p
        p = plan_elliptic_y
    end function OPR_Elliptic_AMD_PlanY

#ifdef TLAB_AMD_FULL_HOST
    ! OPR_Elliptic_Initialize(inifile)   operators/opr_elliptic.f90:86-250: [Main] EllipticOrder selects the factorized (default: the scheme of
    ! the derivatives) or the direct solver; the direct one differentiates with its own plan fdm_loc = FDM_CreatePlan(y, ...) (:107-124).
    subroutine OPR_Elliptic_Initialize(inifile)
        use FDM, only: g, FDM_CreatePlan
        use FDM_Derivative, only: FDM_COM4_DIRECT, FDM_COM6_DIRECT
        use TLab_Grid, only: y
        use TLab_Memory, only: imax, jmax, kmax
        use TLab_Constants, only: efile
        use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
        character(len=*), intent(in) :: inifile
        character(len=512) sRes
        character(len=32) bakfile
        integer, parameter :: DNS_ERROR_UNDEVELOP = 104
        bakfile = trim(adjustl(inifile))//'.bak'
        call ScanFile_Char(bakfile, inifile, 'Main', 'EllipticOrder', 'void', sRes)
        if (trim(adjustl(sRes)) == 'compactdirect6') then
            fdm_loc%der1%mode_fdm = FDM_COM6_DIRECT
            fdm_loc%der2%mode_fdm = FDM_COM6_DIRECT
            call FDM_CreatePlan(y, fdm_loc)
            call OPR_Elliptic_Initialize_AMD(g, imax, jmax, kmax, fdm_loc)
        else if (trim(adjustl(sRes)) == 'compactdirect4') then
            call TLab_Write_ASCII(efile, __FILE__//'. EllipticOrder = CompactDirect4 is not built on the device path.')
            call TLab_Stop(DNS_ERROR_UNDEVELOP)
        else
            call OPR_Elliptic_Initialize_AMD(g, imax, jmax, kmax)
        end if
    end subroutine OPR_Elliptic_Initialize
#endif

    subroutine OPR_Elliptic_Initialize_AMD(g, nx, ny, nz, fdm_elliptic)
        type(fdm_dt), intent(in) :: g(3)
        integer(wi), intent(in) :: nx, ny, nz
        type(fdm_dt), intent(in), target, optional :: fdm_elliptic     ! fdm_loc after FDM_CreatePlan(y, fdm_loc) with der2%mode_fdm = FDM_COM6_DIRECT
        integer(c_int) rc
        if (present(fdm_elliptic)) then
            ! only the second derivative and the nodes of fdm_loc are used (FDM_Int2_*); its first derivative is handed over as it is
            rc = tlab_fdm_plan_create_from_arrays(plan_elliptic_y, int(fdm_elliptic%size, c_int), 0_c_int, 0_c_int, &
                                                  int(fdm_elliptic%der1%nb_diag(1), c_int), int(fdm_elliptic%der1%nb_diag(2), c_int), &
                                                  fdm_elliptic%der1%lhs, fdm_elliptic%der1%rhs, &
                                                  int(fdm_elliptic%der2%nb_diag(1), c_int), int(fdm_elliptic%der2%nb_diag(2), c_int), &
                                                  fdm_elliptic%der2%lhs, fdm_elliptic%der2%rhs)
            call TLab_AMD_Check(rc, 'tlab_fdm_plan_create_from_arrays')
            rc = tlab_fdm_plan_set_aux(plan_elliptic_y, c_null_ptr, c_null_ptr, c_loc(fdm_elliptic%jac), c_loc(fdm_elliptic%nodes))
            call TLab_AMD_Check(rc, 'tlab_fdm_plan_set_aux')
            rc = tlab_fdm_plan_set_scheme(plan_elliptic_y, 6_c_int, int(fdm_elliptic%der2%mode_fdm, c_int))
            call TLab_AMD_Check(rc, 'tlab_fdm_plan_set_scheme')
            rc = tlab_poisson_plan_create_direct(plan, OPR_Partial_AMD_Plan(1, g(1)), OPR_Partial_AMD_Plan(2, g(2)), OPR_Partial_AMD_Plan(3, g(3)), &
                                                 int(nx, c_int), int(ny, c_int), int(nz, c_int), plan_elliptic_y)
            call TLab_AMD_Check(rc, 'tlab_poisson_plan_create_direct')
        else
            rc = tlab_poisson_plan_create(plan, OPR_Partial_AMD_Plan(1, g(1)), OPR_Partial_AMD_Plan(2, g(2)), OPR_Partial_AMD_Plan(3, g(3)), &
                                          int(nx, c_int), int(ny, c_int), int(nz, c_int))
            call TLab_AMD_Check(rc, 'tlab_poisson_plan_create')
        end if
        OPR_Poisson => OPR_Poisson_AMD
    end subroutine OPR_Elliptic_Initialize_AMD

    subroutine OPR_Poisson_AMD(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)      ! exactly the abstract interface
        integer(wi), intent(in) :: nx, ny, nz
        integer, intent(in) :: ibc
        real(wp), intent(inout) :: p(nx, ny, nz)
        real(wp), intent(inout), target :: tmp1(2*ny, nz, nx/2 + 1)
        real(wp), intent(inout), target :: tmp2(2*ny, nz, nx/2 + 1)
        real(wp), intent(in) :: bcs_hb(nx, nz), bcs_ht(nx, nz)
        real(wp), intent(out), optional :: dpdy(nx, ny, nz)
        call poisson_any(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)
    end subroutine OPR_Poisson_AMD

    subroutine OPR_Helmholtz_AMD(nx, ny, nz, ibc, alpha, p, tmp1, tmp2, bcs_hb, bcs_ht)    ! exactly the abstract interface
        integer(wi), intent(in) :: nx, ny, nz
        integer, intent(in) :: ibc
        real(wp), intent(in) :: alpha
        real(wp), intent(inout) :: p(nx, ny, nz)
        real(wp), intent(inout), target :: tmp1(2*ny, nz, nx/2 + 1)
        real(wp), intent(inout), target :: tmp2(2*ny, nz, nx/2 + 1)
        real(wp), intent(in) :: bcs_hb(nx, nz), bcs_ht(nx, nz)
        call helmholtz_any(nx, ny, nz, ibc, alpha, p, tmp1, tmp2, bcs_hb, bcs_ht)
    end subroutine OPR_Helmholtz_AMD

    subroutine helmholtz_any(nx, ny, nz, ibc, alpha, p, tmp1, tmp2, bcs_hb, bcs_ht)
        integer(wi), intent(in) :: nx, ny, nz
        integer, intent(in) :: ibc
        real(wp), intent(in) :: alpha
        real(wp), intent(inout), target :: p(*), tmp1(*), tmp2(*)
        real(wp), intent(in), target :: bcs_hb(*), bcs_ht(*)
        integer(c_int) rc
        rc = tlab_opr_helmholtz(plan, int(nx, c_int), int(ny, c_int), int(nz, c_int), int(ibc, c_int), real(alpha, c_double), c_loc(p), &
                                c_loc(tmp1), c_loc(tmp2), c_loc(bcs_hb), c_loc(bcs_ht))
        call TLab_AMD_Check(rc, 'tlab_opr_helmholtz')        ! factorized plans: BCS_NN and BCS_DD only, like the reference
    end subroutine helmholtz_any

    subroutine poisson_any(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)           ! c_loc needs the TARGET attribute
        integer(wi), intent(in) :: nx, ny, nz
        integer, intent(in) :: ibc
        real(wp), intent(inout), target :: p(*), tmp1(*), tmp2(*)
        real(wp), intent(in), target :: bcs_hb(*), bcs_ht(*)
        real(wp), intent(inout), target, optional :: dpdy(*)
        type(c_ptr) :: pd
        integer(c_int) rc
        pd = c_null_ptr
        if (present(dpdy)) pd = c_loc(dpdy)
        rc = tlab_opr_poisson(plan, int(nx, c_int), int(ny, c_int), int(nz, c_int), int(ibc, c_int), c_loc(p), c_loc(tmp1), c_loc(tmp2), &
                              c_loc(bcs_hb), c_loc(bcs_ht), pd)
        call TLab_AMD_Check(rc, 'tlab_opr_poisson')          ! factorized plan: BCS_NN / BCS_DD like the reference; the mixed types need the direct plan
    end subroutine poisson_any

end module TLAB_AMD_ELLIPTIC_MODULE
