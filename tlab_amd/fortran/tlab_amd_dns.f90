!########################################################################
! Device side of the Runge-Kutta driver, for a Fortran host that keeps tools/dns/time.f90 and dns_main.f90:
!   module TLab_AMD_DNS            the handle of tlab_dns_create, built lazily from what the reference's own modules hold
!                                  (FDM: g ; TLab_Memory: imax, jmax, kmax, inb_scal ; NavierStokes: visc, schmidt ;
!                                  BOUNDARY_BCS: Bcs{Flow,Scal}J{min,max}%type ; OPR_Elliptic: the Poisson plan)
!   subroutine DAXPY, DSCAL        the two BLAS-1 routines TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT and TIME_RUNGEKUTTA call in their -DUSE_BLAS
!                                  branches (tools/dns/time.f90:649, :281, :291) on q, hq, s, hs -- here for DEVICE arrays, so that those
!                                  statements stay textually what they are
!   subroutine TLab_AMD_Zero       hq = 0.0_wp (time.f90:213-214) for a device array: the one statement of the loop with no BLAS branch
!   module IO_Fields_AMD           IO_Read_Fields / IO_Write_Fields (base/io_fields.f90:150, :346) for device arrays: the reference's own
!                                  routines on a host buffer + tlab_memcpy_h2d / d2h
!########################################################################
module TLab_AMD_DNS
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use TLab_AMD_C
    implicit none
    private

    public :: TLab_AMD_DNS_Handle          ! type(c_ptr): creates the device driver state on first use
    public :: TLab_AMD_Slab_Active         ! ims_npro_k > 1 (or TLAB_AMD_FORCE_SLAB=1, one rank as its own neighbour: tests): the z-slab driver runs the RHS
    public :: TLab_AMD_Slab_Handle         ! type(c_ptr): tlab_slab_dns_create over the communicator of TLabMPI_Transpose, module arrays bound
    public :: TLab_AMD_Pencil_Active       ! ims_npro_i > 1 (or TLAB_AMD_FORCE_PENCIL=1): the RHS goes to the native x/z pencil driver (tlab_pencil_dns_*)
    public :: TLab_AMD_Pencil_Handle
    public :: TLab_AMD_DNS_Begin_Step      ! optional: tells the device RHS that hq, hs are zero (no fill, no read of the old tendencies)
    public :: TLab_AMD_Zero
    public :: TLab_AMD_DNS_Finalize
    public :: TLab_AMD_Place_Arrays        ! optional, once, before the fields are read: which device allocations play q, s, hq, hs, txc (tlab_dns_place_blocks)

    type(c_ptr), save :: dns = c_null_ptr
    type(c_ptr), save :: slab = c_null_ptr
    type(c_ptr), save :: pencil = c_null_ptr

contains
    ! the UNPATCHED time loop (RHS, then DAXPY / DSCAL per field: time.f90:612-664, :272-297) as one fused substep of whichever driver runs the RHS
    ! (csrc/deferred.cpp): on unless TLAB_AMD_DEFER=0
    subroutine enable_deferred_tail()
        integer defer_stat
        character(len=8) defer_env
        call get_environment_variable('TLAB_AMD_DEFER', defer_env, status=defer_stat)
        if (.not. (defer_stat == 0 .and. trim(defer_env) == '0')) call TLab_AMD_Check(tlab_deferred_enable(1_c_int), 'tlab_deferred_enable')
    end subroutine enable_deferred_tail

    logical function TLab_AMD_Slab_Active()
        use TLabMPI_VARS, only: ims_npro_k
        character(len=8) val
        integer stat
        TLab_AMD_Slab_Active = ims_npro_k > 1
        if (.not. TLab_AMD_Slab_Active) then
            call get_environment_variable('TLAB_AMD_FORCE_SLAB', val, status=stat)
            TLab_AMD_Slab_Active = (stat == 0 .and. trim(val) == '1')
        end if
    end function TLab_AMD_Slab_Active

    logical function TLab_AMD_Pencil_Active()
        use TLabMPI_VARS, only: ims_npro_i
        character(len=8) val
        integer stat
        TLab_AMD_Pencil_Active = ims_npro_i > 1
        if (.not. TLab_AMD_Pencil_Active) then
            call get_environment_variable('TLAB_AMD_FORCE_PENCIL', val, status=stat)
            TLab_AMD_Pencil_Active = (stat == 0 .and. trim(val) == '1')
        end if
        ! z-slabs too thin for the partitioned z systems (tlab_zslab_plan_create refuses them: the coupling between slab separators is not negligible
        ! below ~50 planes): the reference's own scheme -- K-transpositions around the z operators -- through the pencil driver with npro_i = 1
        if (.not. TLab_AMD_Pencil_Active) TLab_AMD_Pencil_Active = TLab_AMD_Slab_Active() .and. thin_slabs()
    end function TLab_AMD_Pencil_Active

    logical function thin_slabs()
        use FDM, only: g
        use TLab_Memory, only: kmax
        use OPR_Partial, only: OPR_Partial_AMD_Plan
        logical, save :: known = .false., thin = .false.
        type(c_ptr) :: zp
        integer(c_int) :: rc
        if (.not. known) then
            zp = c_null_ptr
            rc = tlab_zslab_plan_create(zp, OPR_Partial_AMD_Plan(3, g(3)), int(kmax, c_int), 0_c_int, 0_c_int)
            thin = (rc == -2_c_int)      ! TLAB_EUNSUPPORTED (include/tlab_amd.h)
            if (rc == 0_c_int) rc = tlab_zslab_plan_destroy(zp)
            known = .true.
        end if
        thin_slabs = thin
    end function thin_slabs

    ! The x/z pencil driver (include/tlab_amd.h: tlab_pencil_dns_*): imax, kmax are the LOCAL sizes, g(1) and g(3) the plans of the GLOBAL x and z
    ! directions, as FDM_Initialize leaves them in an MPI run.  Without a communicator (one rank: tests) the loopback transport serves.
    function TLab_AMD_Pencil_Handle() result(h)
        use FDM, only: g
        use TLab_Memory, only: imax, jmax, kmax, inb_scal, inb_txc
        use TLab_Arrays, only: q, s, txc
        use DNS_ARRAYS, only: hq, hs
        use NavierStokes, only: visc, schmidt
        use BOUNDARY_BCS, only: BcsFlowJmin, BcsFlowJmax, BcsScalJmin, BcsScalJmax
        use OPR_Partial, only: OPR_Partial_AMD_Plan
        use DNS_LOCAL, only: remove_divergence
        use OPR_Elliptic, only: OPR_Elliptic_AMD_PlanY
        use TLabMPI_VARS, only: ims_npro_i, ims_npro_k
        use TLabMPI_Transpose, only: TLabMPI_Trp_AMD_Pencil_Transport
        type(c_ptr) :: h
        type(tlab_pencil_transport) :: tr
        type(c_ptr) :: pq(3), ps(16), phq(3), phs(16), ptxc(16)
        integer(c_int) :: rc, fj0(3), fj1(3), sj0(16), sj1(16)
        real(c_double) :: sc(16)
        integer is
        if (.not. c_associated(pencil)) then
            if (inb_scal > 16 .or. inb_txc < 9) call TLab_AMD_Check(-1_c_int, 'TLab_AMD_Pencil_Handle: needs inb_scal <= 16 and inb_txc >= 9')
            ! the supported subset, refused rather than dropped (as TLab_AMD_Slab_Handle)
            if (.not. remove_divergence) call TLab_AMD_Check(-1_c_int, 'TLab_AMD_Pencil_Handle: TermDivergence = none is not built into the pencil driver')
            if (c_associated(OPR_Elliptic_AMD_PlanY())) &
                call TLab_AMD_Check(-1_c_int, 'TLab_AMD_Pencil_Handle: EllipticOrder = CompactDirect* is not built into the pencil driver (factorized solver only)')
            if (inb_scal > 0) then
                if (any(BcsScalJmin%SfcType(1:inb_scal) /= 0) .or. any(BcsScalJmax%SfcType(1:inb_scal) /= 0)) &
                    call TLab_AMD_Check(-1_c_int, 'TLab_AMD_Pencil_Handle: the dynamic surface model (BcsScal%SfcType) runs on one rank only')
            end if
            if (max(ims_npro_i, 1)*max(ims_npro_k, 1) > 1) then
                call TLabMPI_Trp_AMD_Pencil_Transport(tr)
            else
                call TLab_AMD_Check(tlab_pencil_transport_loopback(tr, 1_c_int, 1_c_int), 'tlab_pencil_transport_loopback')
            end if
            sc = 1.0_c_double
            sc(1:inb_scal) = schmidt(1:inb_scal)
            rc = tlab_pencil_dns_create(pencil, tr, OPR_Partial_AMD_Plan(1, g(1)), OPR_Partial_AMD_Plan(2, g(2)), OPR_Partial_AMD_Plan(3, g(3)), &
                                        int(imax*max(ims_npro_i, 1), c_int), int(jmax, c_int), int(kmax*max(ims_npro_k, 1), c_int), int(inb_scal, c_int), &
                                        real(visc, c_double), sc)
            call TLab_AMD_Check(rc, 'tlab_pencil_dns_create')
            fj0 = BcsFlowJmin%type(1:3); fj1 = BcsFlowJmax%type(1:3)
            sj0 = 3; sj1 = 3
            sj0(1:inb_scal) = BcsScalJmin%type(1:inb_scal); sj1(1:inb_scal) = BcsScalJmax%type(1:inb_scal)
            call TLab_AMD_Check(tlab_pencil_dns_set_bcs(pencil, fj0, fj1, sj0, sj1), 'tlab_pencil_dns_set_bcs')
            ps = c_null_ptr; phs = c_null_ptr; ptxc = c_null_ptr
            do is = 1, 3
                pq(is) = c_loc(q(1, is)); phq(is) = c_loc(hq(1, is))
            end do
            do is = 1, inb_scal
                ps(is) = c_loc(s(1, is)); phs(is) = c_loc(hs(1, is))
            end do
            do is = 1, 9
                ptxc(is) = c_loc(txc(1, is))
            end do
            call TLab_AMD_Check(tlab_pencil_dns_bind(pencil, 0_c_int, pq, ps, phq, phs, ptxc), 'tlab_pencil_dns_bind')
            call enable_deferred_tail()
        end if
        h = pencil
    end function TLab_AMD_Pencil_Handle

    ! The decomposed driver (include/tlab_amd.h: tlab_slab_dns_*): kmax is the LOCAL number of planes, g(3) the plan of the GLOBAL z direction, as
    ! FDM_Initialize leaves it in an MPI run (g(3)%size = kmax*ims_npro_k).  The module arrays are bound once; they never move.
    function TLab_AMD_Slab_Handle() result(h)
        use FDM, only: g
        use TLab_Memory, only: imax, jmax, kmax, inb_scal, inb_txc
        use TLab_Arrays, only: q, s, txc
        use DNS_ARRAYS, only: hq, hs
        use NavierStokes, only: visc, schmidt
        use BOUNDARY_BCS, only: BcsFlowJmin, BcsFlowJmax, BcsScalJmin, BcsScalJmax
        use OPR_Partial, only: OPR_Partial_AMD_Plan
        use OPR_Elliptic, only: OPR_Elliptic_AMD_PlanY
        use DNS_LOCAL, only: remove_divergence
        use TLabMPI_VARS, only: ims_npro_k
        use TLabMPI_Transpose, only: TLabMPI_Trp_AMD_Slab_Transport
        type(c_ptr) :: h
        type(tlab_slab_transport) :: tr
        type(c_ptr) :: pq(3), ps(16), phq(3), phs(16), ptxc(16)
        integer(c_int) :: rc, fj0(3), fj1(3), sj0(16), sj1(16)
        real(c_double) :: sc(16), cp0(16), cp1(16)
        integer is
        if (.not. c_associated(slab)) then
            if (inb_scal > 16 .or. inb_txc < 9) call TLab_AMD_Check(-1_c_int, 'TLab_AMD_Slab_Handle: needs inb_scal <= 16 and inb_txc >= 9')
            ! what the z-slab driver does not build is REFUSED here, not dropped: the same tlab.ini must integrate the same equations on 1 and on N ranks
            ! (the anelastic formulation and dealiasing filters are refused inside tlab_slab_dns_create / _rhs from the operator state)
            call TLabMPI_Trp_AMD_Slab_Transport(tr)
            sc = 1.0_c_double
            sc(1:inb_scal) = schmidt(1:inb_scal)
            rc = tlab_slab_dns_create(slab, tr, OPR_Partial_AMD_Plan(1, g(1)), OPR_Partial_AMD_Plan(2, g(2)), OPR_Partial_AMD_Plan(3, g(3)), &
                                      int(imax, c_int), int(jmax, c_int), int(kmax*max(ims_npro_k, 1), c_int), int(inb_scal, c_int), &
                                      real(visc, c_double), sc, OPR_Elliptic_AMD_PlanY())      ! the host's elliptic choice: factorized, or CompactDirect6
            call TLab_AMD_Check(rc, 'tlab_slab_dns_create')
            call TLab_AMD_Check(tlab_slab_dns_set_remove_divergence(slab, merge(1_c_int, 0_c_int, remove_divergence)), &     ! dns.ini [Main] TermDivergence
                                'tlab_slab_dns_set_remove_divergence')
            if (inb_scal > 0) then          ! dynamic surface model of the scalars (BcsScalJmin%SfcType, %cpl): its plane average is an all-reduce
                sj0 = 0; sj1 = 0; cp0 = 0.0_c_double; cp1 = 0.0_c_double
                sj0(1:inb_scal) = BcsScalJmin%SfcType(1:inb_scal); sj1(1:inb_scal) = BcsScalJmax%SfcType(1:inb_scal)
                cp0(1:inb_scal) = BcsScalJmin%cpl(1:inb_scal); cp1(1:inb_scal) = BcsScalJmax%cpl(1:inb_scal)
                if (any(sj0(1:inb_scal) /= 0) .or. any(sj1(1:inb_scal) /= 0)) &
                    call TLab_AMD_Check(tlab_slab_dns_set_surface_bcs(slab, sj0, sj1, cp0, cp1), 'tlab_slab_dns_set_surface_bcs')
            end if
            fj0 = BcsFlowJmin%type(1:3); fj1 = BcsFlowJmax%type(1:3)
            sj0 = 3; sj1 = 3
            sj0(1:inb_scal) = BcsScalJmin%type(1:inb_scal); sj1(1:inb_scal) = BcsScalJmax%type(1:inb_scal)
            call TLab_AMD_Check(tlab_slab_dns_set_bcs(slab, fj0, fj1, sj0, sj1), 'tlab_slab_dns_set_bcs')
            ps = c_null_ptr; phs = c_null_ptr; ptxc = c_null_ptr
            do is = 1, 3
                pq(is) = c_loc(q(1, is)); phq(is) = c_loc(hq(1, is))
            end do
            do is = 1, inb_scal
                ps(is) = c_loc(s(1, is)); phs(is) = c_loc(hs(1, is))
            end do
            do is = 1, 9
                ptxc(is) = c_loc(txc(1, is))
            end do
            call TLab_AMD_Check(tlab_slab_dns_bind(slab, 0_c_int, pq, ps, phq, phs, ptxc), 'tlab_slab_dns_bind')
            call enable_deferred_tail()
        end if
        h = slab
    end function TLab_AMD_Slab_Handle

    function TLab_AMD_DNS_Handle() result(h)
        use FDM, only: g
        use TLab_Memory, only: imax, jmax, kmax, inb_scal
        use NavierStokes, only: visc, schmidt
        use BOUNDARY_BCS, only: BcsFlowJmin, BcsFlowJmax, BcsScalJmin, BcsScalJmax
        use OPR_Partial, only: OPR_Partial_AMD_Plan
        use OPR_Elliptic, only: OPR_Elliptic_AMD_Plan
        use DNS_LOCAL, only: remove_divergence
        type(c_ptr) :: h
        integer(c_int) :: rc, fj0(3), fj1(3), sj0(16), sj1(16)
        real(c_double) :: cp0(16), cp1(16)
        real(c_double) :: sc(16)
        integer ns
        if (.not. c_associated(dns)) then
            if (inb_scal > 16) call TLab_AMD_Check(-1_c_int, 'TLab_AMD_DNS_Handle: at most 16 scalars')
            ns = max(1, int(inb_scal))
            sc = 1.0_c_double
            sc(1:inb_scal) = schmidt(1:inb_scal)
            rc = tlab_dns_create(dns, OPR_Partial_AMD_Plan(1, g(1)), OPR_Partial_AMD_Plan(2, g(2)), OPR_Partial_AMD_Plan(3, g(3)), &
                                 OPR_Elliptic_AMD_Plan(), int(imax, c_int), int(jmax, c_int), int(kmax, c_int), int(inb_scal, c_int), &
                                 real(visc, c_double), sc)
            call TLab_AMD_Check(rc, 'tlab_dns_create')
            fj0 = BcsFlowJmin%type(1:3); fj1 = BcsFlowJmax%type(1:3)
            sj0 = 3; sj1 = 3
            sj0(1:inb_scal) = BcsScalJmin%type(1:inb_scal); sj1(1:inb_scal) = BcsScalJmax%type(1:inb_scal)
            rc = tlab_dns_set_bcs(dns, fj0, fj1, sj0, sj1)
            call TLab_AMD_Check(rc, 'tlab_dns_set_bcs')
            rc = tlab_dns_set_remove_divergence(dns, merge(1_c_int, 0_c_int, remove_divergence))      ! dns.ini [Main] TermDivergence
            call TLab_AMD_Check(rc, 'tlab_dns_set_remove_divergence')
            call enable_deferred_tail()
            if (inb_scal > 0) then          ! dynamic surface model of the scalars (BcsScalJmin%SfcType, %cpl)
                sj0 = 0; sj1 = 0; cp0 = 0.0_c_double; cp1 = 0.0_c_double
                sj0(1:inb_scal) = BcsScalJmin%SfcType(1:inb_scal); sj1(1:inb_scal) = BcsScalJmax%SfcType(1:inb_scal)
                cp0(1:inb_scal) = BcsScalJmin%cpl(1:inb_scal); cp1(1:inb_scal) = BcsScalJmax%cpl(1:inb_scal)
                if (any(sj0(1:inb_scal) /= 0) .or. any(sj1(1:inb_scal) /= 0)) then
                    rc = tlab_dns_set_surface_bcs(dns, sj0, sj1, cp0, cp1)
                    call TLab_AMD_Check(rc, 'tlab_dns_set_surface_bcs')
                end if
            end if
        end if
        h = dns
    end function TLab_AMD_DNS_Handle

    ! Where the arrays live decides 16.0 .. 17.0 ms per substep of the 512^3 box from process to process (DESIGN.md section 4; INTEGRATION.md section
    ! 3c).  The host's arrays are two-dimensional -- components a fixed stride apart -- so whole blocks are placed: ncand - 1 more allocations per
    ! block through the allocation hook's tlab_malloc, the substep timed on combinations (tlab_dns_place_blocks), the module arrays q, s, txc
    ! (TLab_Arrays), hq, hs (DNS_ARRAYS) re-associated with the winners, the aliases of TLab_Pointers* reset, the losers freed.  ONE added call in
    ! dns_main.f90, after the operators are initialised and before IO_Read_Fields (the arrays hold nothing yet; every candidate is overwritten).
    ! TLAB_AMD_PLACE = number of candidates per block (default 6, as many as the memory gives; 0 or 1: nothing happens); single-domain runs only.
    subroutine TLab_AMD_Place_Arrays()
        use TLab_Constants, only: lfile
        use TLab_WorkFlow, only: TLab_Write_ASCII
        use TLab_Memory, only: isize_field, isize_txc_field, inb_scal, inb_txc, TLab_AMD_Reset_Pointers
        use TLab_Arrays, only: q, s, txc
        use DNS_ARRAYS, only: hq, hs
        integer, parameter :: maxcand = 8
        type(c_ptr) :: cq(maxcand), cs(maxcand), chq(maxcand), chs(maxcand), ctxc(maxcand), p
        integer(c_int) :: choice(5), rc
        real(c_double) :: report(5)
        integer(c_size_t) :: nb(5)
        integer :: ncand, ic, ib, stat, nq, ns2, ntxc
        character(len=16) :: val
        character(len=256) :: line

        ncand = 6                       ! (four of four processes at 15.8 ms with 8 candidates, three of four with 4: profiles/r06/host_placement_candidates.txt)
        call get_environment_variable('TLAB_AMD_PLACE', val, status=stat)
        if (stat == 0) then
            read (val, *, iostat=stat) ncand
            if (stat /= 0) ncand = 6
        end if
        ncand = min(ncand, maxcand)
        if (ncand < 2 .or. TLab_AMD_Pencil_Active() .or. TLab_AMD_Slab_Active()) return
        if (.not. (associated(q) .and. associated(hq) .and. associated(txc))) return
        if (inb_scal > 0) then
            if (.not. (associated(s) .and. associated(hs))) return
        end if
        nq = size(q, 2); ns2 = 0; ntxc = size(txc, 2)
        if (inb_scal > 0) ns2 = size(s, 2)
        if (nq /= 3 .or. size(hq, 2) /= 3 .or. ntxc < 9 .or. ns2 /= inb_scal) return             ! (incompressible layout only)
        if (inb_scal > 0) then
            if (size(hs, 2) /= inb_scal) return
        end if
        nb = [int(isize_field, c_size_t)*3_c_size_t, int(isize_field, c_size_t)*int(max(ns2, 1), c_size_t), int(isize_field, c_size_t)*3_c_size_t, &
              int(isize_field, c_size_t)*int(max(ns2, 1), c_size_t), int(isize_txc_field, c_size_t)*int(ntxc, c_size_t)]*8_c_size_t
        cq = c_null_ptr; cs = c_null_ptr; chq = c_null_ptr; chs = c_null_ptr; ctxc = c_null_ptr
        cq(1) = c_loc(q(1, 1)); chq(1) = c_loc(hq(1, 1)); ctxc(1) = c_loc(txc(1, 1))
        if (inb_scal > 0) then
            cs(1) = c_loc(s(1, 1)); chs(1) = c_loc(hs(1, 1))
        end if
        alloc: do ic = 2, ncand                       ! as many complete sets of candidates as the memory gives
            do ib = 1, 5
                if ((ib == 2 .or. ib == 4) .and. inb_scal == 0) cycle
                if (tlab_malloc(p, nb(ib)) /= 0) then
                    if (ib > 1) rc = tlab_free(cq(ic))
                    if (ib > 2 .and. inb_scal > 0) rc = tlab_free(cs(ic))
                    if (ib > 3) rc = tlab_free(chq(ic))
                    if (ib > 4 .and. inb_scal > 0) rc = tlab_free(chs(ic))
                    ncand = ic - 1
                    exit alloc
                end if
                select case (ib)
                case (1); cq(ic) = p
                case (2); cs(ic) = p
                case (3); chq(ic) = p
                case (4); chs(ic) = p
                case (5); ctxc(ic) = p
                end select
            end do
        end do alloc
        if (ncand < 2) return
        rc = tlab_dns_place_blocks(TLab_AMD_DNS_Handle(), int(ncand, c_int), cq, cs, chq, chs, ctxc, int(isize_txc_field, c_long_long), 1.0e-3_c_double, &
                                   int(4*ncand, c_int), 1_c_int, choice, report)
        call TLab_AMD_Check(rc, 'tlab_dns_place_blocks')
        call TLab_AMD_Check(tlab_sync(), 'tlab_sync')
        ! the winners become the host's arrays; everything else goes back
        call c_f_pointer(cq(choice(1) + 1), q, [isize_field, 3])
        call c_f_pointer(chq(choice(3) + 1), hq, [isize_field, 3])
        call c_f_pointer(ctxc(choice(5) + 1), txc, [isize_txc_field, ntxc])
        if (inb_scal > 0) then
            call c_f_pointer(cs(choice(2) + 1), s, [isize_field, ns2])
            call c_f_pointer(chs(choice(4) + 1), hs, [isize_field, ns2])
        end if
        do ic = 1, ncand
            if (ic /= choice(1) + 1) rc = tlab_free(cq(ic))
            if (ic /= choice(3) + 1) rc = tlab_free(chq(ic))
            if (ic /= choice(5) + 1) rc = tlab_free(ctxc(ic))
            if (inb_scal > 0) then
                if (ic /= choice(2) + 1) rc = tlab_free(cs(ic))
                if (ic /= choice(4) + 1) rc = tlab_free(chs(ic))
            end if
        end do
        call TLab_AMD_Reset_Pointers()
        write (line, '(a,i0,a,i0,a,f8.3,a,f8.3,a,f8.3,a,f8.3,a,5(1x,i0))') 'PLACEMENT: candidates ', ncand, ' trials ', int(report(5)), &
            ' ms_first ', report(1), ' ms_kept ', report(2), ' ms_median ', report(3), ' ms_worst ', report(4), ' choice', choice
        call TLab_Write_ASCII(lfile, trim(line))
    end subroutine TLab_AMD_Place_Arrays

    subroutine TLab_AMD_DNS_Begin_Step()
        if (TLab_AMD_Pencil_Active()) then
            ! the pencil driver ADDS to the tendencies like the reference: the start of a step zeroes them (hq = hs = 0, time.f90:212-216)
            call TLab_AMD_Check(tlab_pencil_dns_begin_step(TLab_AMD_Pencil_Handle()), 'tlab_pencil_dns_begin_step')
        else if (TLab_AMD_Slab_Active()) then
            call TLab_AMD_Check(tlab_slab_dns_begin_step(TLab_AMD_Slab_Handle()), 'tlab_slab_dns_begin_step')
        else
            call TLab_AMD_Check(tlab_dns_begin_step(TLab_AMD_DNS_Handle()), 'tlab_dns_begin_step')
        end if
    end subroutine TLab_AMD_DNS_Begin_Step

    ! a = 0.0_wp for a device array of any rank (sequence association: pass the array, its size)
    subroutine TLab_AMD_Zero(a, n)
        real(wp), intent(inout), target :: a(*)
        integer(wi), intent(in) :: n
        ! recorded like DAXPY / DSCAL when the deferred tail is on: `hq = 0 ; hs = 0` in front of a substep becomes tlab_dns_begin_step
        call TLab_AMD_Check(tlab_deferred_zero(c_loc(a), int(n, c_long_long)), 'tlab_deferred_zero')
    end subroutine TLab_AMD_Zero

    subroutine TLab_AMD_DNS_Finalize()
        use OPR_Partial, only: OPR_Partial_AMD_Release_Stale
        integer(c_int) rc
        rc = tlab_deferred_enable(0_c_int)            ! (runs what is still recorded)
        if (c_associated(slab)) rc = tlab_slab_dns_destroy(slab)
        slab = c_null_ptr
        if (c_associated(pencil)) rc = tlab_pencil_dns_destroy(pencil)
        pencil = c_null_ptr
        if (c_associated(dns)) rc = tlab_dns_destroy(dns)
        dns = c_null_ptr
        call OPR_Partial_AMD_Release_Stale()
        rc = tlab_sync()
    end subroutine TLab_AMD_DNS_Finalize

end module TLab_AMD_DNS

! ###################################################################
! DAXPY / DSCAL on device arrays (the -DUSE_BLAS branches of tools/dns/time.f90:649-660, :279-293)
subroutine DAXPY(n, da, dx, incx, dy, incy)
    use, intrinsic :: iso_c_binding
    use TLab_AMD_C
    implicit none
    integer, intent(in) :: n, incx, incy
    real(c_double), intent(in) :: da
    real(c_double), intent(in), target :: dx(*)
    real(c_double), intent(inout), target :: dy(*)
    if (incx /= 1 .or. incy /= 1) call TLab_AMD_Check(-1_c_int, 'DAXPY on device arrays (unit strides only)')
    ! y = y + a x -- executed at once, or (tlab_deferred_enable, the default of TLab_AMD_DNS_Handle) recorded: the update loops of
    ! TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT then ride on the last kernels of the RHS they follow (csrc/deferred.cpp)
    call TLab_AMD_Check(tlab_deferred_axpy(int(n, c_long_long), da, c_loc(dx), c_loc(dy)), 'tlab_deferred_axpy')
end subroutine DAXPY

subroutine DSCAL(n, da, dx, incx)
    use, intrinsic :: iso_c_binding
    use TLab_AMD_C
    implicit none
    integer, intent(in) :: n, incx
    real(c_double), intent(in) :: da
    real(c_double), intent(inout), target :: dx(*)
    if (incx /= 1) call TLab_AMD_Check(-1_c_int, 'DSCAL on device arrays (unit stride only)')
    call TLab_AMD_Check(tlab_deferred_scal(int(n, c_long_long), da, c_loc(dx)), 'tlab_deferred_scal')       ! (as DAXPY above)
end subroutine DSCAL

! ###################################################################
module IO_Fields_AMD
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi
    use TLab_AMD_C
    use IO_Fields, only: IO_Read_Fields, IO_Write_Fields          ! the reference's own, base/io_fields.f90, compiled where it lies
    implicit none
    private
    public :: IO_Read_Fields_AMD, IO_Write_Fields_AMD

contains
    ! IO_Read_Fields(fname, nx, ny, nz, nt, nfield, iread, a, params) with a(nx*ny*nz, *) on the device
    subroutine IO_Read_Fields_AMD(fname, nx, ny, nz, nt, nfield, iread, a, params)
        character(len=*) fname
        integer, intent(in) :: nfield, iread
        integer(wi), intent(in) :: nx, ny, nz, nt
        real(wp), intent(out), target :: a(nx*ny*nz, *)
        real(wp), intent(inout) :: params(:)
        real(wp), allocatable, target :: host(:, :)
        integer nloc
        nloc = merge(nfield, 1, iread == 0)
        allocate (host(nx*ny*nz, nloc))
        call IO_Read_Fields(fname, nx, ny, nz, nt, nfield, iread, host, params)
        call TLab_AMD_Check(tlab_memcpy_h2d(c_loc(a), c_loc(host), int(nx, c_size_t)*ny*nz*nloc*8_c_size_t), 'tlab_memcpy_h2d')
        deallocate (host)
    end subroutine IO_Read_Fields_AMD

    subroutine IO_Write_Fields_AMD(fname, nx, ny, nz, nt, nfield, a)
        character(len=*), intent(in) :: fname
        integer, intent(in) :: nfield
        integer(wi), intent(in) :: nx, ny, nz, nt
        real(wp), intent(in), target :: a(nx*ny*nz, nfield)
        real(wp), allocatable, target :: host(:, :)
        allocate (host(nx*ny*nz, nfield))
        call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(host), c_loc(a), int(nx, c_size_t)*ny*nz*nfield*8_c_size_t), 'tlab_memcpy_d2h')
        call IO_Write_Fields(fname, nx, ny, nz, nt, nfield, host)
        deallocate (host)
    end subroutine IO_Write_Fields_AMD

end module IO_Fields_AMD
