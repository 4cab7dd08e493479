!########################################################################
! Drop-in replacement of module OPR_Burgers (physics/opr_burgers.f90:23-30,190-431): same names and signatures.
! The reference folds the diffusivity of field `is` into a copy of the LU (OPR_Burgers_Initialize, :92-112); the device
! kernels take it as a number, so OPR_Burgers_Initialize only records visc and schmidt(:).
!
! With -DTLAB_AMD_FULL_HOST (a complete Tlab host: modules NavierStokes, TLab_Memory present) the module also carries the reference's own
! entry OPR_Burgers_Initialize(inifile), which takes visc, schmidt from module NavierStokes, inb_scal from TLab_Memory and g from FDM exactly
! like physics/opr_burgers.f90:52-114, so that dns_main.f90:129 compiles unchanged.
!########################################################################
#ifndef TLAB_AMD_BURGERS_MODULE
#define TLAB_AMD_BURGERS_MODULE OPR_Burgers
#endif
#ifndef TLAB_AMD_PARTIAL_MODULE
#define TLAB_AMD_PARTIAL_MODULE OPR_Partial
#endif
module TLAB_AMD_BURGERS_MODULE
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use FDM, only: fdm_dt
    use TLab_AMD_C
    use TLAB_AMD_PARTIAL_MODULE, only: OPR_Partial_AMD_Plan
    implicit none
    private

    public :: OPR_Burgers_Initialize_AMD     ! (visc, schmidt): what OPR_Burgers_Initialize(inifile) takes from module NavierStokes
#ifdef TLAB_AMD_FULL_HOST
    public :: OPR_Burgers_Initialize         ! (inifile): physics/opr_burgers.f90:52
#endif
    logical, public :: OPR_Burgers_AMD_write_transposed = .true.   ! OPR_B_SELF leaves the transposed operand in tmp1 (opr_burgers.f90:236-240); .false.
    !                                                                saves that write when no caller consumes it (the device kernels ignore u_t)
    public :: OPR_Burgers_X
    public :: OPR_Burgers_Y
    public :: OPR_Burgers_Z
    public :: OPR_Burgers_SetPlans
    public :: OPR_Burgers_AMD_Anelastic      ! (rbackground, ribackground): what OPR_Burgers_Initialize does for nse_eqns == DNS_EQNS_ANELASTIC (:128-183)
    public :: OPR_Burgers_AMD_Dealiasing     ! (idir, type, periodic, bcsmin, bcsmax, coeffs): Dealiasing(idir) after OPR_FILTER_INITIALIZE (:118-121)

    integer, parameter, public :: OPR_B_SELF = 0
    integer, parameter, public :: OPR_B_U_IN = 1

    real(wp), save :: diffusivity(0:16) = 0.0_wp
    type(fdm_dt), pointer, save :: gp(:) => null()      ! the host plans g(1:3) of module FDM

contains
    subroutine OPR_Burgers_Initialize_AMD(visc, schmidt)
        real(wp), intent(in) :: visc, schmidt(:)
        integer is
        diffusivity(0) = visc                             ! opr_burgers.f90:94-98
        do is = 1, size(schmidt)
            diffusivity(is) = visc/schmidt(is)
        end do
    end subroutine OPR_Burgers_Initialize_AMD

#ifdef TLAB_AMD_FULL_HOST
    ! OPR_Burgers_Initialize(inifile)   physics/opr_burgers.f90:52-186
    subroutine OPR_Burgers_Initialize(inifile)
        use FDM, only: g
        use NavierStokes, only: visc, schmidt
        use TLab_Memory, only: inb_scal
        use TLab_Constants, only: efile
        use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
        character(len=*), intent(in) :: inifile
        character(len=32) bakfile
        character(len=512) sRes
        integer ig
        integer, parameter :: DNS_ERROR_OPTION = 85, DNS_ERROR_UNDEVELOP = 104      ! include/dns_error.h
        bakfile = trim(adjustl(inifile))//'.bak'
#ifdef TLAB_AMD_HAVE_OPR_FILTER
        ! a complete host (module OPR_Filter present; operators/opr_filter.f90 needs OPR_Fourier / FFTW and is therefore not part of the test build
        ! of this repository): the reference's own block (opr_burgers.f90:71, 118-125), plus the hand-over of every active filter to the device
        block
            use OPR_Filter, only: filter_dt, FILTER_READBLOCK, OPR_FILTER_INITIALIZE, DNS_FILTER_NONE
            type(filter_dt), save :: Dealiasing(3)
            call FILTER_READBLOCK(bakfile, inifile, 'Dealiasing', Dealiasing)
            do ig = 1, 3
                if (Dealiasing(ig)%type == DNS_FILTER_NONE) cycle
                call OPR_FILTER_INITIALIZE(g(ig), Dealiasing(ig))
                call OPR_Burgers_AMD_Dealiasing(ig, Dealiasing(ig)%type, Dealiasing(ig)%periodic, Dealiasing(ig)%BcsMin, Dealiasing(ig)%BcsMax, &
                                                Dealiasing(ig)%coeffs)      ! tophat: TLAB_EUNSUPPORTED -> TLab_Stop through TLab_AMD_Check
            end do
        end block
#else
        call ScanFile_Char(bakfile, inifile, 'Dealiasing', 'Type', 'none', sRes)    ! FILTER_READBLOCK(.., 'Dealiasing', ..), opr_burgers.f90:71
        if (trim(adjustl(sRes)) /= 'none') then
            call TLab_Write_ASCII(efile, __FILE__//'. [Dealiasing] needs module OPR_Filter of the host: build with -DTLAB_AMD_HAVE_OPR_FILTER.')
            call TLab_Stop(DNS_ERROR_UNDEVELOP)
        end if
#endif
        do ig = 1, 3                                                               ! :75-83
            if (g(ig)%size == 1) cycle
            if (g(ig)%der2%nb_diag(1) /= 3) then
                call TLab_Write_ASCII(efile, __FILE__//'. Undeveloped for more than 3 LHS diagonals in 2. order derivatives.')
                call TLab_Stop(DNS_ERROR_OPTION)
            end if
        end do
        call OPR_Burgers_Initialize_AMD(visc, schmidt(1:inb_scal))                  ! :92-98
        call OPR_Burgers_SetPlans(g)
    end subroutine OPR_Burgers_Initialize
#endif

    subroutine OPR_Burgers_SetPlans(g)
        type(fdm_dt), intent(in), target :: g(3)
        gp => g
    end subroutine OPR_Burgers_SetPlans

    ! Anelastic density correction of the diffusion term (physics/opr_burgers.f90:128-183): rhoinv(1), rhoinv(3) and the scaled U factors of
    ! fdmDiffusion(2) all amount to the factor ribackground(j), which the device applies in the epilogue of the operator.  A complete host calls this
    ! from where the reference has the block, with the profiles of module Thermo_Anelastic; size 0 switches back.
    subroutine OPR_Burgers_AMD_Anelastic(rbackground, ribackground)
        real(wp), intent(in), target :: rbackground(:), ribackground(:)
        integer(c_int) rc
        if (size(rbackground) == 0) then
            rc = tlab_opr_burgers_set_anelastic(0_c_int, c_null_ptr, c_null_ptr)
        else
            rc = tlab_opr_burgers_set_anelastic(int(size(rbackground), c_int), c_loc(rbackground), c_loc(ribackground))
        end if
        call TLab_AMD_Check(rc, 'tlab_opr_burgers_set_anelastic')
    end subroutine OPR_Burgers_AMD_Anelastic

    ! Dealiasing(idir) (physics/opr_burgers.f90:33, 118-121): the host's filter_dt after OPR_FILTER_INITIALIZE; the table f%coeffs goes to the device.
    ! type = 0 (DNS_FILTER_NONE) switches the direction off.
    subroutine OPR_Burgers_AMD_Dealiasing(idir, ftype, periodic, bcsmin, bcsmax, coeffs)
        integer, intent(in) :: idir, ftype, bcsmin, bcsmax
        logical, intent(in) :: periodic
        real(wp), intent(in), target, contiguous :: coeffs(:, :)
        type(c_ptr), save :: handle(3) = c_null_ptr
        integer(c_int) rc, per
        if (c_associated(handle(idir))) then
            rc = tlab_opr_burgers_set_dealiasing(int(idir, c_int), c_null_ptr)
            rc = tlab_filter_destroy(handle(idir))
            handle(idir) = c_null_ptr
        end if
        if (ftype == 0) return
        per = 0_c_int; if (periodic) per = 1_c_int
        if (size(coeffs) > 0) then
            rc = tlab_filter_create(handle(idir), int(ftype, c_int), int(size(coeffs, 1), c_int), per, int(bcsmin, c_int), int(bcsmax, c_int), &
                                    int(size(coeffs, 2), c_int), c_loc(coeffs))
        else
            rc = tlab_filter_create(handle(idir), int(ftype, c_int), int(gp(idir)%size, c_int), per, int(bcsmin, c_int), int(bcsmax, c_int), &
                                    0_c_int, c_null_ptr)
        end if
        call TLab_AMD_Check(rc, 'tlab_filter_create')
        rc = tlab_opr_burgers_set_dealiasing(int(idir, c_int), handle(idir))
        call TLab_AMD_Check(rc, 'tlab_opr_burgers_set_dealiasing')
    end subroutine OPR_Burgers_AMD_Dealiasing

    subroutine burgers_any(idir, ivel, is, nx, ny, nz, bcs, s, u, result, tmp1)
        integer, intent(in) :: idir, ivel, is
        integer(wi), intent(in) :: nx, ny, nz, bcs(2, 2)
        real(wp), intent(in), target :: s(*), u(*)
        real(wp), intent(out), target :: result(*)
        real(wp), intent(inout), target :: tmp1(*)
        integer(c_int) rc, wt
        if (bcs(1, 2) + bcs(2, 2) > 0) error stop 'OPR_Burgers: only developed for biased BCs'     ! opr_burgers.f90:460-463
        wt = 0_c_int
        if (ivel == OPR_B_SELF .and. OPR_Burgers_AMD_write_transposed) wt = 1_c_int               ! the contract of :236-240 (X, Y; Z has no local transpose)
        rc = tlab_opr_burgers(int(idir, c_int), OPR_Partial_AMD_Plan(idir, gp(idir)), int(ivel, c_int), int(nx, c_int), int(ny, c_int), &
                              int(nz, c_int), int(bcs(1, 1) + bcs(2, 1)*2, c_int), diffusivity(is), c_loc(s), c_loc(u), c_loc(result), &
                              c_loc(tmp1), wt)
        call TLab_AMD_Check(rc, 'tlab_opr_burgers')
    end subroutine burgers_any

    subroutine OPR_Burgers_X(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t)
        integer, intent(in) :: ivel, is
        integer(wi), intent(in) :: nx, ny, nz, bcs(2, 2)
        real(wp), intent(in) :: s(nx*ny*nz), u(nx*ny*nz)
        real(wp), intent(out) :: result(nx*ny*nz)
        real(wp), intent(inout) :: tmp1(nx*ny*nz)
        real(wp), intent(in), optional :: u_t(nx*ny*nz)   ! ignored: the device kernels read u in its natural layout
        call burgers_any(1, ivel, is, nx, ny, nz, bcs, s, u, result, tmp1)
    end subroutine OPR_Burgers_X

    subroutine OPR_Burgers_Y(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t)
        integer, intent(in) :: ivel, is
        integer(wi), intent(in) :: nx, ny, nz, bcs(2, 2)
        real(wp), intent(in) :: s(nx*ny*nz), u(nx*ny*nz)
        real(wp), intent(out) :: result(nx*ny*nz)
        real(wp), intent(inout) :: tmp1(nx*ny*nz)
        real(wp), intent(in), optional :: u_t(nx*ny*nz)
        call burgers_any(2, ivel, is, nx, ny, nz, bcs, s, u, result, tmp1)
    end subroutine OPR_Burgers_Y

    subroutine OPR_Burgers_Z(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t)
        integer, intent(in) :: ivel, is
        integer(wi), intent(in) :: nx, ny, nz, bcs(2, 2)
        real(wp), intent(in) :: s(nx*ny*nz), u(nx*ny*nz)
        real(wp), intent(out) :: result(nx*ny*nz)
        real(wp), intent(inout) :: tmp1(nx*ny*nz)
        real(wp), intent(in), optional :: u_t(nx*ny*nz)
        call burgers_any(3, ivel, is, nx, ny, nz, bcs, s, u, result, tmp1)
    end subroutine OPR_Burgers_Z

end module TLAB_AMD_BURGERS_MODULE
