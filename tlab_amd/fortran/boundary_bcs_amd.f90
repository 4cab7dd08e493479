!########################################################################
! Drop-in replacement of the incompressible part of module BOUNDARY_BCS (tools/dns/boundary_bcs.f90): the wall-boundary-condition
! types the RHS reads (BcsFlowJmin%type(1:3), ..., :14-27), their readers for the [BoundaryConditions] block of tlab.ini (:54-121) and
! BOUNDARY_BCS_NEUMANN_Y (:368-473), which marshals to the C ABI.  The reference planes %ref are host arrays in the reference; the
! device RHS keeps its wall planes itself, so BOUNDARY_BCS_INITIALIZE only sizes them.
!########################################################################
module BOUNDARY_BCS
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi, MAX_VARS, efile
    use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
    use FDM, only: fdm_dt
    use TLab_AMD_C
    use OPR_Partial, only: OPR_Partial_AMD_Plan
    implicit none
    save
    private

    type bcs_dt                                              ! boundary_bcs.f90:14-21
        sequence
        integer type(MAX_VARS)                              ! dirichlet, neumann for incompressible
        integer SfcType(MAX_VARS)                           ! Type of Surface Model
        real(wp) cpl(MAX_VARS)                              ! Coupling parameter for surface model
        real(wp) cinf, cout, ctan                           ! characteristic formulation for compressible
        real(wp), allocatable, dimension(:, :, :) :: ref    ! reference fields
    end type bcs_dt

    type(bcs_dt), public :: BcsFlowImin, BcsFlowImax, BcsFlowJmin, BcsFlowJmax, BcsFlowKmin, BcsFlowKmax
    type(bcs_dt), public :: BcsScalImin, BcsScalImax, BcsScalJmin, BcsScalJmax, BcsScalKmin, BcsScalKmax

    public :: BOUNDARY_BCS_NEUMANN_Y
    public :: BOUNDARY_BCS_SCAL_READBLOCK, BOUNDARY_BCS_FLOW_READBLOCK
    public :: BOUNDARY_BCS_INITIALIZE

    integer, parameter, public :: DNS_BCS_NONE = 0           ! :41-46
    integer, parameter, public :: DNS_BCS_NR = 1
    integer, parameter, public :: DNS_BCS_INFLOW = 2
    integer, parameter, public :: DNS_BCS_DIRICHLET = 3
    integer, parameter, public :: DNS_BCS_NEUMANN = 4
    integer, parameter, public :: DNS_SFC_STATIC = 0
    integer, parameter, public :: DNS_SFC_LINEAR = 1

    integer, parameter :: DNS_ERROR_IBC = 20, DNS_ERROR_JBC = 21, DNS_ERROR_UNDEVELOP = 104      ! include/dns_error.h

contains
    ! BOUNDARY_BCS_SCAL_READBLOCK   boundary_bcs.f90:54-93
    subroutine BOUNDARY_BCS_SCAL_READBLOCK(bakfile, inifile, tag, var)
        use TLab_Memory, only: inb_scal
        character(len=*), intent(in) :: bakfile, inifile, tag
        type(bcs_dt), intent(out) :: var
        character(len=512) sRes
        character(len=20) lstr
        integer is
        do is = 1, inb_scal
            write (lstr, *) is; lstr = 'Scalar'//trim(adjustl(lstr))
            call ScanFile_Char(bakfile, inifile, 'BoundaryConditions', trim(adjustl(lstr))//trim(adjustl(tag)), 'void', sRes)
            if (trim(adjustl(sRes)) == 'none') then; var%type(is) = DNS_BCS_NONE
            else if (trim(adjustl(sRes)) == 'dirichlet') then; var%type(is) = DNS_BCS_DIRICHLET
            else if (trim(adjustl(sRes)) == 'neumann') then; var%type(is) = DNS_BCS_NEUMANN
            else
                call TLab_Write_ASCII(efile, __FILE__//'. BoundaryConditions.'//trim(adjustl(lstr)))
                call TLab_Stop(DNS_ERROR_JBC)
            end if
            call ScanFile_Char(bakfile, inifile, 'BoundaryConditions', trim(adjustl(lstr))//'SfcType'//trim(adjustl(tag)), 'static', sRes)
            if (trim(adjustl(sRes)) == 'static') then
                var%SfcType(is) = DNS_SFC_STATIC
            else if (trim(adjustl(sRes)) == 'linear') then   ! dynamic surface model: BOUNDARY_BCS_SURFACE_Y runs inside tlab_rhs_global_incompressible_1
                var%SfcType(is) = DNS_SFC_LINEAR
            else
                call TLab_Write_ASCII(efile, __FILE__//'. BoundaryConditions.'//trim(adjustl(lstr))//'SfcType'//trim(adjustl(tag)))
                call TLab_Stop(DNS_ERROR_JBC)
            end if
            call ScanFile_Real(bakfile, inifile, 'BoundaryConditions', trim(adjustl(lstr))//'Coupling'//trim(adjustl(tag)), '0.0', var%cpl(is))
        end do
    end subroutine BOUNDARY_BCS_SCAL_READBLOCK

    ! BOUNDARY_BCS_FLOW_READBLOCK   boundary_bcs.f90:97-121
    subroutine BOUNDARY_BCS_FLOW_READBLOCK(bakfile, inifile, tag, var)
        character(len=*), intent(in) :: bakfile, inifile, tag
        type(bcs_dt), intent(out) :: var
        character(len=512) sRes
        integer inormal, itangential(2)
        select case (trim(adjustl(tag)))
        case ('Imin', 'Imax')
            inormal = 1
            itangential = [2, 3]
        case ('Jmin', 'Jmax')
            inormal = 2
            itangential = [1, 3]
        case ('Kmin', 'Kmax')
            inormal = 3
            itangential = [1, 2]
        end select
        call ScanFile_Char(bakfile, inifile, 'BoundaryConditions', 'Velocity'//trim(adjustl(tag)), 'freeslip', sRes)
        if (trim(adjustl(sRes)) == 'none') then; var%type(1:3) = DNS_BCS_NONE
        else if (trim(adjustl(sRes)) == 'noslip') then; var%type(1:3) = DNS_BCS_DIRICHLET
        else if (trim(adjustl(sRes)) == 'freeslip') then; var%type(inormal) = DNS_BCS_DIRICHLET
            var%type(itangential) = DNS_BCS_NEUMANN
        else
            call TLab_Write_ASCII(efile, __FILE__//'. BoundaryConditions.Velocity'//trim(adjustl(tag)))
            call TLab_Stop(DNS_ERROR_IBC)
        end if
    end subroutine BOUNDARY_BCS_FLOW_READBLOCK

    ! BOUNDARY_BCS_INITIALIZE   boundary_bcs.f90:125-364: reference planes of the incompressible mode
    subroutine BOUNDARY_BCS_INITIALIZE()
        use TLab_Memory, only: imax, kmax, inb_flow_array, inb_scal_array
        if (.not. allocated(BcsFlowJmin%ref)) allocate (BcsFlowJmin%ref(imax, kmax, inb_flow_array + 1))
        if (.not. allocated(BcsFlowJmax%ref)) allocate (BcsFlowJmax%ref(imax, kmax, inb_flow_array + 1))
        if (.not. allocated(BcsScalJmin%ref)) allocate (BcsScalJmin%ref(imax, kmax, inb_scal_array + 1))
        if (.not. allocated(BcsScalJmax%ref)) allocate (BcsScalJmax%ref(imax, kmax, inb_scal_array + 1))
        BcsFlowJmin%ref = 0.0_wp; BcsFlowJmax%ref = 0.0_wp; BcsScalJmin%ref = 0.0_wp; BcsScalJmax%ref = 0.0_wp
    end subroutine BOUNDARY_BCS_INITIALIZE

    ! BOUNDARY_BCS_NEUMANN_Y(ibc, nx, ny, nz, g, u, bcs_hb, bcs_ht, tmp1)   boundary_bcs.f90:368-473; all arrays on the device
    subroutine BOUNDARY_BCS_NEUMANN_Y(ibc, nx, ny, nz, g, u, bcs_hb, bcs_ht, tmp1)
        integer(wi), intent(in) :: ibc
        integer(wi) nx, ny, nz
        type(fdm_dt), intent(in) :: g
        real(wp), intent(in), target :: u(nx*nz, ny)
        real(wp), intent(inout), target :: tmp1(nx*nz, ny)
        real(wp), intent(out), target :: bcs_hb(nx*nz), bcs_ht(nx*nz)
        integer(c_int) rc
        rc = tlab_boundary_bcs_neumann_y(OPR_Partial_AMD_Plan(2, g), int(ibc, c_int), int(nx, c_int), int(ny, c_int), int(nz, c_int), &
                                         c_loc(u), c_loc(bcs_hb), c_loc(bcs_ht), c_loc(tmp1))
        call TLab_AMD_Check(rc, 'tlab_boundary_bcs_neumann_y')
    end subroutine BOUNDARY_BCS_NEUMANN_Y

end module BOUNDARY_BCS
