#include "dns_error.h"
!########################################################################
! Drop-in replacement of the incompressible part of module BOUNDARY_BCS (tools/dns/boundary_bcs.f90).  Only the delta lives here:
!   BOUNDARY_BCS_NEUMANN_Y (:368-473)   marshals to the C ABI (the walls' Neumann-reduced first-derivative system runs on the device);
!   BOUNDARY_BCS_INITIALIZE (:125-364)  only sizes the reference planes %ref: the device RHS keeps its wall planes itself, and the reference's
!                                       routine drags in the thermodynamics, buffer zones and background profiles that are outside the path.
! The declarations (type bcs_dt, Bcs{Flow,Scal}{I,J,K}{min,max}, DNS_BCS_*, DNS_SFC_*; :15-50) and the two readers of the [BoundaryConditions] block
! of tlab.ini (BOUNDARY_BCS_SCAL_READBLOCK / _FLOW_READBLOCK, :55-124) ARE the reference's: the build extracts them from
! $(REF)/src/tools/dns/boundary_bcs.f90 where it lies into two include files (tlab_amd/fortran/Makefile: sed ranges) -- nothing of them is kept here.
! BOUNDARY_BCS_SURFACE_Y has no counterpart on the host: the dynamic surface model runs inside tlab_rhs_global_incompressible_1.
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

#include "boundary_bcs_ref_decls.inc"

contains
#include "boundary_bcs_ref_readblocks.inc"

    subroutine BOUNDARY_BCS_INITIALIZE()
        use TLab_Memory, only: imax, kmax, inb_flow_array, inb_scal_array
        if (.not. allocated(BcsFlowJmin%ref)) allocate (BcsFlowJmin%ref(imax, kmax, inb_flow_array + 1))
        if (.not. allocated(BcsFlowJmax%ref)) allocate (BcsFlowJmax%ref(imax, kmax, inb_flow_array + 1))
        if (.not. allocated(BcsScalJmin%ref)) allocate (BcsScalJmin%ref(imax, kmax, inb_scal_array + 1))
        if (.not. allocated(BcsScalJmax%ref)) allocate (BcsScalJmax%ref(imax, kmax, inb_scal_array + 1))
        BcsFlowJmin%ref = 0.0_wp; BcsFlowJmax%ref = 0.0_wp; BcsScalJmin%ref = 0.0_wp; BcsScalJmax%ref = 0.0_wp
    end subroutine BOUNDARY_BCS_INITIALIZE

    ! all arrays on the device
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
