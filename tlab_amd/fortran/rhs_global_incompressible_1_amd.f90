!########################################################################
! Link-time replacement of the external subroutine RHS_GLOBAL_INCOMPRESSIBLE_1() (tools/dns/rhs_global_incompressible_1.f90:15-405),
! called by TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT (tools/dns/time.f90:612).  Argument-less like the reference's: it works on the module arrays
! q, s, txc (TLab_Arrays), hq, hs (DNS_ARRAYS) and on dte (TIME), all of which live on the device through the allocation hook, and runs
! the whole assembly -- 12 + 3 ns OPR_Burgers, the pressure forcing, OPR_Poisson, the pressure gradient and the wall boundary conditions --
! in tlab_rhs_global_incompressible_1 without a field ever visiting the host.
!########################################################################
subroutine RHS_GLOBAL_INCOMPRESSIBLE_1()
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use TLab_Memory, only: inb_flow, inb_scal, inb_txc
    use TLab_Arrays
    use DNS_ARRAYS
    use TIME, only: dte
    use TLab_AMD_C
    use TLab_AMD_DNS, only: TLab_AMD_DNS_Handle, TLab_AMD_Slab_Active, TLab_AMD_Slab_Handle, TLab_AMD_Pencil_Active, TLab_AMD_Pencil_Handle
    implicit none

    type(c_ptr) :: pq(3), ps(16), phq(3), phs(16), ptxc(16)
    integer is
    integer(c_int) rc

    ! ims_npro_i > 1: the x/z pencil driver (tlab_amd/csrc/pencil.cpp): I- / K-transpositions around the x / z operators, exactly the MPI branches of
    ! OPR_Partial_X/Z and OPR_Burgers_X/Z (opr_partial.f90:66-147, :185-253; opr_burgers.f90:216-262, :386-426)
    if (TLab_AMD_Pencil_Active()) then
        ! (recorded when the deferred tail is on: the DAXPY / DSCAL calls of time.f90 complete it to tlab_pencil_dns_substep, csrc/deferred.cpp)
        call TLab_AMD_Check(tlab_deferred_pencil_rhs(TLab_AMD_Pencil_Handle(), real(dte, c_double)), 'tlab_deferred_pencil_rhs')
        return
    end if
    ! ims_npro_k > 1: the z-slab driver (tlab_amd/csrc/slab.cpp) on the module arrays it was bound to -- the MPI branches of OPR_Partial_Z,
    ! OPR_Burgers_Z and OPR_Fourier_Z_* (opr_partial.f90:185-195, opr_burgers.f90:386-426, opr_fourier.f90:343-428) without a transposition per operator
    if (TLab_AMD_Slab_Active()) then
        call TLab_AMD_Check(tlab_deferred_slab_rhs(TLab_AMD_Slab_Handle(), real(dte, c_double)), 'tlab_deferred_slab_rhs')       ! (likewise: tlab_slab_dns_substep)
        return
    end if
    if (inb_scal > 16 .or. inb_txc < 9) call TLab_AMD_Check(-1_c_int, 'RHS_GLOBAL_INCOMPRESSIBLE_1: needs inb_scal <= 16 and inb_txc >= 9')
    do is = 1, 3
        pq(is) = c_loc(q(1, is)); phq(is) = c_loc(hq(1, is))
    end do
    ps = c_null_ptr; phs = c_null_ptr; ptxc = c_null_ptr
    do is = 1, inb_scal
        ps(is) = c_loc(s(1, is)); phs(is) = c_loc(hs(1, is))
    end do
    do is = 1, min(int(inb_txc), 16)
        ptxc(is) = c_loc(txc(1, is))
    end do
    ! = tlab_rhs_global_incompressible_1, or its description when the deferred tail is on (TLab_AMD_DNS_Handle): the DAXPY / DSCAL calls that follow
    ! in time.f90 then complete it to ONE tlab_time_substep_incompressible_explicit (csrc/deferred.cpp)
    rc = tlab_deferred_rhs(TLab_AMD_DNS_Handle(), real(dte, c_double), pq, ps, phq, phs, ptxc)
    call TLab_AMD_Check(rc, 'tlab_deferred_rhs')

end subroutine RHS_GLOBAL_INCOMPRESSIBLE_1

!########################################################################
! The whole of TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT (tools/dns/time.f90:559-664) AND the tendency scaling that follows it in TIME_RUNGEKUTTA
! (:272-297) in the fused device driver: RHS, q += dte hq, s += dte hs, hq *= kco, hs *= kco (the last only if scale).  For a host that accepts a
! six-line patch of time.f90 (test_rk_driver.f90, -DTLAB_AMD_FUSED_SUBSTEP): the update loops ride on the last kernels that touch each field.  Since
! round 6 the UNPATCHED loop reaches the same call by itself (the deferred tail, csrc/deferred.cpp: 15.96 ms per substep at 512^3 either way,
! 19.5 with the BLAS calls executed one by one; profiles/r06/fortran_host.json); the patch remains for hosts that prefer to say it in their source.
!########################################################################
subroutine TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD(kco_loc, scale_loc)
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use TLab_Memory, only: inb_flow, inb_scal, inb_txc
    use TLab_Arrays
    use DNS_ARRAYS
    use TIME, only: dte
    use TLab_AMD_C
    use TLab_AMD_DNS, only: TLab_AMD_DNS_Handle, TLab_AMD_Slab_Active, TLab_AMD_Slab_Handle, TLab_AMD_Pencil_Active, TLab_AMD_Pencil_Handle
    implicit none
    real(wp), intent(in) :: kco_loc
    logical, intent(in) :: scale_loc

    type(c_ptr) :: pq(3), ps(16), phq(3), phs(16), ptxc(16)
    integer is
    integer(c_int) rc

    if (TLab_AMD_Pencil_Active()) then
        call TLab_AMD_Check(tlab_pencil_dns_substep(TLab_AMD_Pencil_Handle(), real(dte, c_double), real(kco_loc, c_double), &
                                                    merge(1_c_int, 0_c_int, scale_loc)), 'tlab_pencil_dns_substep')
        return
    end if
    if (TLab_AMD_Slab_Active()) then
        call TLab_AMD_Check(tlab_slab_dns_substep(TLab_AMD_Slab_Handle(), real(dte, c_double), real(kco_loc, c_double), &
                                                  merge(1_c_int, 0_c_int, scale_loc)), 'tlab_slab_dns_substep')
        return
    end if
    if (inb_scal > 16 .or. inb_txc < 9) call TLab_AMD_Check(-1_c_int, 'TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD: needs inb_scal <= 16 and inb_txc >= 9')
    do is = 1, 3
        pq(is) = c_loc(q(1, is)); phq(is) = c_loc(hq(1, is))
    end do
    ps = c_null_ptr; phs = c_null_ptr; ptxc = c_null_ptr
    do is = 1, inb_scal
        ps(is) = c_loc(s(1, is)); phs(is) = c_loc(hs(1, is))
    end do
    do is = 1, min(int(inb_txc), 16)
        ptxc(is) = c_loc(txc(1, is))
    end do
    rc = tlab_time_substep_incompressible_explicit(TLab_AMD_DNS_Handle(), real(dte, c_double), real(kco_loc, c_double), &
                                                   merge(1_c_int, 0_c_int, scale_loc), pq, ps, phq, phs, ptxc)
    call TLab_AMD_Check(rc, 'tlab_time_substep_incompressible_explicit')

end subroutine TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD
