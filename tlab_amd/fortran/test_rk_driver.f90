!########################################################################
! Fortran mini-driver: the start-up sequence of program DNS (tools/dns/dns_main.f90:62-157) and the Runge-Kutta loop of module TIME
! (tools/dns/time.f90:185-330, :559-664) on DEVICE memory, with the drop-in modules under the REFERENCE'S OWN NAMES:
!
!   from the reference, compiled where it lies (oracle/_ref): TLab_Constants, TLab_WorkFlow (TLab_Start/Stop/Write_ASCII), TLab_Grid
!       (TLab_Grid_Read), FDM (FDM_Initialize: the plans g(1:3) are the host's), IO_Fields (IO_Read/Write_Fields), ScanFile_* (io_ascii.f90)
!   drop-in (tlab_amd/fortran/): TLab_Memory + TLab_Arrays (allocation hook), OPR_Partial, OPR_Burgers (OPR_Burgers_Initialize(inifile)),
!       OPR_Elliptic (OPR_Elliptic_Initialize(inifile), OPR_Poisson), OPR_Fourier (OPR_Fourier_Initialize()), BOUNDARY_BCS,
!       external RHS_GLOBAL_INCOMPRESSIBLE_1(), DAXPY / DSCAL for device arrays
!   host stand-ins defined in test_host_modules.f90 and in this file (the reference's versions drag in thermodynamics, particles, statistics ... and opr_fourier.f90, which
!       needs fftw3.f03): NavierStokes (visc, schmidt: physics/navierstokes.f90:25), DNS_ARRAYS (hq, hs: tools/dns/dns_local.f90:297-309) and
!       TIME -- the latter keeps the statements of time.f90 (its -DUSE_BLAS branches) line by line, minus particles / compressible / implicit.
!
! Reads tlab.ini, grid, flow.0.{1,2,3}, scal.0.{1..} from the working directory (written by tests/test_gpu_fortran_dropin.py in the
! reference's formats), advances [Time] Iterations steps, writes flow.<n>.*, scal.<n>.*.
!########################################################################
module TIME
    use TLab_Constants, only: efile, wp, wi
    use TLab_WorkFlow, only: flow_on, scal_on
    use TLab_Memory, only: imax, jmax, kmax, isize_field
    use TLab_Memory, only: inb_flow, inb_scal
    use NavierStokes
    use TLab_AMD_DNS, only: TLab_AMD_Zero, TLab_AMD_DNS_Begin_Step
    implicit none
    private

    integer(wi), public :: rkm_mode             ! Type of Runge-Kutta scheme
    integer(wi), public :: rkm_endstep          ! number of substeps
    integer(wi), public :: rkm_substep          ! substep counter
    real(wp), public :: dtime                   ! time step
    real(wp), public :: dte                     ! time step of each substep
    real(wp), public :: rtime = 0.0_wp, etime
    real(wp) kdt(5), kco(4), ktime(5)           ! explicit scheme coefficients

    integer, parameter, public :: RKM_EXP3 = 3, RKM_EXP4 = 4

    public :: TIME_INITIALIZE
    public :: TIME_RUNGEKUTTA

    integer(wi) is

contains
    ! TIME_INITIALIZE   tools/dns/time.f90:72-180 (coefficients :86-108)
    subroutine TIME_INITIALIZE()
        kdt = 0.0_wp; kco = 0.0_wp; ktime = 0.0_wp
        select case (rkm_mode)
        case (RKM_EXP3)             ! Runge-Kutta explicit 3th order from Williamson 1980
            rkm_endstep = 3
            kdt(1:3) = [1.0_wp/3.0_wp, 15.0_wp/16.0_wp, 8.0_wp/15.0_wp]
            ktime(1:3) = [0.0_wp, 1.0_wp/3.0_wp, 3.0_wp/4.0_wp]
            kco(1:2) = [-5.0_wp/9.0_wp, -153.0_wp/128.0_wp]
        case (RKM_EXP4)             ! Runge-Kutta explicit 4th order 5 stages from Carpenter & Kennedy 1994
            rkm_endstep = 5
            kdt(1) = 1432997174477.0_wp/9575080441755.0_wp
            kdt(2) = 5161836677717.0_wp/13612068292357.0_wp
            kdt(3) = 1720146321549.0_wp/2090206949498.0_wp
            kdt(4) = 3134564353537.0_wp/4481467310338.0_wp
            kdt(5) = 2277821191437.0_wp/14882151754819.0_wp
            ktime(1) = 0.0_wp
            ktime(2) = kdt(1)
            ktime(3) = 2526269341429.0_wp/6820363962896.0_wp
            ktime(4) = 2006345519317.0_wp/3224310063776.0_wp
            ktime(5) = 2802321613138.0_wp/2924317926251.0_wp
            kco(1) = -567301805773.0_wp/1357537059087.0_wp
            kco(2) = -2404267990393.0_wp/2016746695238.0_wp
            kco(3) = -3550918686646.0_wp/2091501179385.0_wp
            kco(4) = -1275806237668.0_wp/842570457699.0_wp
        end select
    end subroutine TIME_INITIALIZE

    ! TIME_RUNGEKUTTA   tools/dns/time.f90:185-330
    subroutine TIME_RUNGEKUTTA()
        use TLab_Arrays
        use DNS_ARRAYS

        ! -------------------------------------------------------------------
        real(wp) alpha
        integer ij_len

        !########################################################################
        ij_len = isize_field

        ! -------------------------------------------------------------------
        ! Initialize arrays to zero for the explcit low-storage algorithm
        ! -------------------------------------------------------------------
        if (rkm_mode == RKM_EXP3 .or. rkm_mode == RKM_EXP4) then
#ifdef TLAB_AMD_FUSED_SUBSTEP
            call TLab_AMD_DNS_Begin_Step()                                      ! hq = hs = 0 as a flag: the first launch of each field overwrites
#else
            if (flow_on) call TLab_AMD_Zero(hq, isize_field*inb_flow)           ! hq = 0.0_wp   (time.f90:213; no BLAS branch: the one patched line)
            if (scal_on) call TLab_AMD_Zero(hs, isize_field*inb_scal)           ! hs = 0.0_wp   (time.f90:214)
#endif
        end if
        !########################################################################
        ! Loop over the sub-stages
        !########################################################################
        do rkm_substep = 1, rkm_endstep

            ! -------------------------------------------------------------------
            ! Update transported (or prognostic) variables q and s
            ! -------------------------------------------------------------------
            dte = dtime*kdt(rkm_substep)
            etime = rtime + dtime*ktime(rkm_substep)

            select case (nse_eqns)
            case (DNS_EQNS_INCOMPRESSIBLE, DNS_EQNS_ANELASTIC)
                if (rkm_mode == RKM_EXP3 .or. rkm_mode == RKM_EXP4) then
                    call TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT()
                end if
            end select

            ! -------------------------------------------------------------------
            ! Update RHS hq and hs in the explicit low-storage algorithm
            ! -------------------------------------------------------------------
#ifdef TLAB_AMD_FUSED_SUBSTEP
            if (.false.) then                   ! (the scaling rides on the substep: TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD)
#else
            if ((rkm_mode == RKM_EXP3 .or. rkm_mode == RKM_EXP4) .and. &
                rkm_substep < rkm_endstep) then
#endif

                alpha = kco(rkm_substep)

                if (flow_on) then
                    do is = 1, inb_flow
                        call DSCAL(ij_len, alpha, hq(1, is), 1)
                    end do
                end if

                if (scal_on) then
                    do is = 1, inb_scal
                        call DSCAL(ij_len, alpha, hs(1, is), 1)
                    end do
                end if
            end if

        end do

        return
    end subroutine TIME_RUNGEKUTTA

    ! TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT   tools/dns/time.f90:559-664 (EQNS_CONVECTIVE, EQNS_RHS_COMBINED; no sources, no buffer zone)
    subroutine TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT()
        use TLab_Arrays, only: q, s, txc
        use DNS_ARRAYS, only: hq, hs

        integer ij_len
        external RHS_GLOBAL_INCOMPRESSIBLE_1
#ifdef TLAB_AMD_FUSED_SUBSTEP
        external TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD

        ! the patch: RHS, the two update loops below and the DSCAL loops of TIME_RUNGEKUTTA in one call of the fused device driver
        if (rkm_substep < rkm_endstep) then
            call TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD(kco(rkm_substep), .true.)
        else
            call TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT_AMD(1.0_wp, .false.)
        end if
        return
#endif

        ij_len = isize_field

        select case (nse_advection)
        case (EQNS_CONVECTIVE)
            call RHS_GLOBAL_INCOMPRESSIBLE_1()
        end select

        ! #######################################################################
        ! Perform the time stepping for incompressible equations
        ! #######################################################################
        do is = 1, inb_flow
            call DAXPY(ij_len, dte, hq(1, is), 1, q(1, is), 1)
        end do

        do is = 1, inb_scal                                             ! (time.f90:658 asks `#ifdef BLAS` for this loop, not USE_BLAS: build the host with both)
            call DAXPY(ij_len, dte, hs(1, is), 1, s(1, is), 1)
        end do

        return
    end subroutine TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT

end module TIME

! ###################################################################
program test_rk_driver
    use TLab_Constants, only: wp, wi, ifile, gfile, lfile, tag_flow, tag_scal
    use TLab_WorkFlow, only: TLab_Start, TLab_Stop, TLab_Write_ASCII, fourier_on, flow_on, scal_on, stagger_on
    use DNS_LOCAL, only: remove_divergence
    use TLab_Memory, only: imax, jmax, kmax, isize_field, inb_flow, inb_flow_array, inb_scal, inb_scal_array, inb_txc, inb_wrk1d, inb_wrk2d
    use TLab_Memory, only: TLab_Initialize_Memory, TLab_Allocate_Real
    use TLab_Arrays
    use TLab_Grid, only: TLab_Grid_Read, x, y, z
    use FDM, only: g, FDM_Initialize
    use NavierStokes, only: visc, schmidt
    use IO_Fields, only: io_fileformat, io_datatype, IO_MPIIO, IO_TYPE_DOUBLE
    use IO_Fields_AMD
    use OPR_Burgers
    use OPR_Elliptic
    use OPR_Fourier
    use BOUNDARY_BCS
    use DNS_ARRAYS
    use TIME
    use TLab_AMD_DNS, only: TLab_AMD_DNS_Finalize, TLab_AMD_DNS_Handle, TLab_AMD_Place_Arrays
    use TLabMPI_VARS
    use TLabMPI_Transpose
    use TLab_AMD_C, only: tlab_sync, tlab_memcpy_d2h, TLab_AMD_Check, tlab_time_courant, tlab_deferred_stats
    use, intrinsic :: iso_c_binding
    implicit none

    interface
        integer(c_int) function tlab_comm_get_unique_id(id) bind(C, name='tlab_comm_get_unique_id')
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: id(128)
        end function
    end interface
    character(kind=c_char) :: comm_id(128)
    type(tmpi_transpose_dt) :: plan_i, plan_k
    real(wp), allocatable, target :: h1(:), h2(:)
    real(wp) :: resid
    character(len=64) line
    character(len=32) fname, bakfile
    character(len=512) sRes
    integer(wi) nitera_first, nitera_last, itime
    real(wp) params(2), reynolds, cfla, cfld
    real(c_double) pmax(2), dt_c
    type(c_ptr) pq(3)
    logical step_from_cfl
    ! timing of the loop (TLAB_AMD_TIMING = number of untimed iterations in front): wall clock between two tlab_sync, per substep
    integer(wi) warm_iters
    integer(8) clock0, clock1, clock_rate
    integer timing_stat
    integer(c_long_long) dstat(6)
    character(len=16) timing_env

    ! ###################################################################
    call TLab_Start()                                                          ! dns_main.f90:62

    bakfile = trim(adjustl(ifile))//'.bak'
    call ScanFile_Int(bakfile, ifile, 'Grid', 'Imax', '0', imax)               ! TLab_Initialize_Parameters (base/tlab_initialize_parameters.f90)
    call ScanFile_Int(bakfile, ifile, 'Grid', 'Jmax', '0', jmax)
    call ScanFile_Int(bakfile, ifile, 'Grid', 'Kmax', '0', kmax)
    fourier_on = .true.; flow_on = .true.; scal_on = .true.
    inb_flow = 3; inb_flow_array = 3
    call ScanFile_Int(bakfile, ifile, 'Main', 'Scalars', '1', inb_scal)
    inb_scal_array = inb_scal
    inb_txc = 9                                                                ! tools/dns/dns_read_local.f90:711
    inb_wrk1d = 20; inb_wrk2d = 6
    io_fileformat = IO_MPIIO; io_datatype = IO_TYPE_DOUBLE

    call ScanFile_Char(bakfile, ifile, 'Staggering', 'StaggerHorizontalPressure', 'no', sRes)     ! tlab_initialize_parameters.f90:114-116
    stagger_on = (trim(adjustl(sRes)) == 'yes')
    call ScanFile_Char(bakfile, ifile, 'Main', 'TermDivergence', 'remove', sRes)                  ! dns_read_local.f90:79-81
    remove_divergence = (trim(adjustl(sRes)) /= 'none')

    call TLab_Grid_Read(gfile, x, y, z)                                        ! :75
    call FDM_Initialize(ifile)                                                 ! :76  (the reference's own: tables of g(1:3) stay the host's)

    call ScanFile_Real(bakfile, ifile, 'Parameters', 'Reynolds', '5000', reynolds)      ! NavierStokes_Initialize_Parameters, navierstokes.f90:150-195
    visc = 1.0_wp/reynolds
    schmidt = 1.0_wp
    call ScanFile_Real(bakfile, ifile, 'Parameters', 'Schmidt', '1.0', schmidt(1))

    ! Time marching: the reference's keys ([Main] TimeOrder, TimeStep, TimeCFL: dns_read_local.f90:100-141; [Iteration] Start, End: :170-175), so that
    ! an example's own tlab.ini runs as it is (examples/Case01); the [Time] block of the earlier fixtures of this driver is still read first
    call ScanFile_Char(bakfile, ifile, 'Time', 'Scheme', 'void', sRes)
    if (trim(adjustl(sRes)) == 'void') call ScanFile_Char(bakfile, ifile, 'Main', 'TimeOrder', 'RungeKuttaExplicit3', sRes)
    if (trim(adjustl(sRes)) == 'rungekuttaexplicit4') then; rkm_mode = RKM_EXP4
    else; rkm_mode = RKM_EXP3; end if
    call ScanFile_Real(bakfile, ifile, 'Time', 'TimeStep', '0.0', dtime)
    if (dtime /= 0.0_wp) then                   ! the earlier fixtures of this driver: a fixed step
        cfla = -1.0_wp; cfld = -1.0_wp
    else                                        ! dns_read_local.f90:60-76: the CFL numbers rule whenever TimeCFL > 0, TimeStep otherwise
        if (rkm_mode == RKM_EXP4) then; sRes = '1.2'; else; sRes = '0.6'; end if
        call ScanFile_Real(bakfile, ifile, 'Main', 'TimeCFL', trim(adjustl(sRes)), cfla)
        write (sRes, *) 0.25_wp*cfla
        call ScanFile_Real(bakfile, ifile, 'Main', 'TimeDiffusiveCFL', trim(adjustl(sRes)), cfld)
        call ScanFile_Real(bakfile, ifile, 'Main', 'TimeStep', '0.05', dtime)
    end if
    call ScanFile_Int(bakfile, ifile, 'Time', 'Start', '-1', nitera_first)
    if (nitera_first < 0) call ScanFile_Int(bakfile, ifile, 'Iteration', 'Start', '0', nitera_first)
    call ScanFile_Int(bakfile, ifile, 'Time', 'End', '-1', nitera_last)
    if (nitera_last < 0) call ScanFile_Int(bakfile, ifile, 'Iteration', 'End', '1', nitera_last)
    call BOUNDARY_BCS_SCAL_READBLOCK(bakfile, ifile, 'Jmin', BcsScalJmin)      ! dns_read_local.f90:242-243, :257-258
    call BOUNDARY_BCS_SCAL_READBLOCK(bakfile, ifile, 'Jmax', BcsScalJmax)
    call BOUNDARY_BCS_FLOW_READBLOCK(bakfile, ifile, 'Jmin', BcsFlowJmin)
    call BOUNDARY_BCS_FLOW_READBLOCK(bakfile, ifile, 'Jmax', BcsFlowJmax)

    ! #######################################################################
    ! Initialize memory space and grid data
    ! #######################################################################
    call TLab_Initialize_Memory(__FILE__)                                      ! :97

    call TLab_Allocate_Real(__FILE__, hq, [isize_field, inb_flow], 'flow-rhs') ! :103-104
    call TLab_Allocate_Real(__FILE__, hs, [isize_field, inb_scal], 'scal-rhs')

    ! ###################################################################
    ! Initialize operators
    ! ###################################################################
    call OPR_Burgers_Initialize(ifile)                                         ! :129

    call OPR_Elliptic_Initialize(ifile)                                        ! :131

    if (fourier_on) call OPR_Fourier_Initialize()                              ! :139

    call TLab_AMD_Place_Arrays()                                               ! the ONE added call (INTEGRATION.md section 3c): which allocations play q, s, hq, hs, txc

    ! ###################################################################
    ! Initialize fields
    ! ###################################################################
    itime = nitera_first

    if (scal_on) then
        write (fname, *) nitera_first; fname = trim(adjustl(tag_scal))//trim(adjustl(fname))
        call IO_Read_Fields_AMD(fname, imax, jmax, kmax, itime, inb_scal, 0, s, params(1:1))       ! :150
    end if

    write (fname, *) nitera_first; fname = trim(adjustl(tag_flow))//trim(adjustl(fname))
    call IO_Read_Fields_AMD(fname, imax, jmax, kmax, itime, inb_flow, 0, q, params(1:2))           ! :154
    rtime = params(1)

    ! ###################################################################
    ! Transposition layer on one rank (ims_npro = 1): TLabMPI_Trp_Initialize + the round trip of OPR_CHECK (operators/opr_check.f90:46-91),
    ! forward then backward = identity, through the RCCL communicator and the same-named module procedures (real and complex plans)
    ! ###################################################################
    call TLab_AMD_Check(tlab_comm_get_unique_id(comm_id), 'tlab_comm_get_unique_id')      ! rank 0 + MPI_Bcast in a parallel host
    call TLabMPI_Trp_AMD_Comm(comm_id)
    call TLabMPI_Trp_Initialize(ifile)                                         ! dns_main.f90:67
    plan_i = TLabMPI_Trp_PlanI(imax, jmax*kmax, message='check Ox')
    plan_k = TLabMPI_Trp_PlanK(kmax, imax*jmax, message='check Oz')
    allocate (h1(isize_field), h2(isize_field))
    call TLabMPI_Trp_ExecI_Forward(q(:, 1), txc(:, 1), plan_i)
    call TLabMPI_Trp_ExecI_Backward(txc(:, 1), txc(:, 2), plan_i)
    call TLabMPI_Trp_ExecK_Forward(txc(:, 2), txc(:, 3), plan_k)
    call TLabMPI_Trp_ExecK_Backward(txc(:, 3), txc(:, 4), plan_k)
    call TLab_AMD_Check(tlab_sync(), 'tlab_sync')
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(h1), c_loc(q(1, 1)), int(isize_field, c_size_t)*8_c_size_t), 'd2h')
    call TLab_AMD_Check(tlab_memcpy_d2h(c_loc(h2), c_loc(txc(1, 4)), int(isize_field, c_size_t)*8_c_size_t), 'd2h')
    resid = maxval(abs(h1 - h2))
    write (line, '(a,es10.3)') 'Checking transposition round trip: residual ', resid
    call TLab_Write_ASCII(lfile, line)
    deallocate (h1, h2)

    call BOUNDARY_BCS_INITIALIZE()                                             ! :193
    call TIME_INITIALIZE()                                                     ! :224

    ! ###################################################################
    ! Do simulation: Integrate equations
    ! ###################################################################
    step_from_cfl = cfla > 0.0_wp               ! time.f90:530: the step follows the Courant numbers whenever TimeCFL > 0
    warm_iters = -1
    call get_environment_variable('TLAB_AMD_TIMING', timing_env, status=timing_stat)
    if (timing_stat == 0) read (timing_env, *, iostat=timing_stat) warm_iters
    if (timing_stat /= 0) warm_iters = -1
    do while (itime < nitera_last)                                             ! :246-250
        if (warm_iters >= 0 .and. itime == nitera_first + warm_iters) then
            call TLab_AMD_Check(tlab_sync(), 'tlab_sync')
            call system_clock(clock0, clock_rate)
        end if
        if (step_from_cfl) then                                                ! TIME_COURANT() at the start of the iteration (dns_main.f90:243, time.f90:365-548)
            pq = [c_loc(q(1, 1)), c_loc(q(1, 2)), c_loc(q(1, 3))]
            call TLab_AMD_Check(tlab_time_courant(TLab_AMD_DNS_Handle(), pq, real(cfla, c_double), real(cfld, c_double), pmax, dt_c), 'tlab_time_courant')
            dtime = dt_c
            write (line, '(a,es23.16)') 'TIME_COURANT: dtime = ', dtime
            call TLab_Write_ASCII(lfile, line)
        end if
        call TIME_RUNGEKUTTA()
        itime = itime + 1
        rtime = rtime + dtime
    end do
    if (warm_iters >= 0 .and. nitera_last > nitera_first + warm_iters) then
        call TLab_AMD_Check(tlab_sync(), 'tlab_sync')
        call system_clock(clock1)
        write (sRes, '(a,i0,a,es14.7)') 'TIMING: substeps ', (nitera_last - nitera_first - warm_iters)*rkm_endstep, ' ms_per_substep ', &
            1.0e3_wp*real(clock1 - clock0, wp)/real(clock_rate, wp)/real((nitera_last - nitera_first - warm_iters)*rkm_endstep, wp)
        call TLab_Write_ASCII(lfile, trim(sRes))
        call TLab_AMD_Check(tlab_deferred_stats(dstat), 'tlab_deferred_stats')
        write (sRes, '(a,6(1x,i0))') 'DEFERRED: fused literal begin_steps eager_axpy eager_scal eager_zero', dstat
        call TLab_Write_ASCII(lfile, trim(sRes))
    end if

    write (fname, *) itime; fname = trim(adjustl(tag_flow))//trim(adjustl(fname))
    call IO_Write_Fields_AMD(fname, imax, jmax, kmax, itime, inb_flow, q)      ! DNS control: IO_Write_Fields, dns_main.f90:344
    if (scal_on) then
        write (fname, *) itime; fname = trim(adjustl(tag_scal))//trim(adjustl(fname))
        call IO_Write_Fields_AMD(fname, imax, jmax, kmax, itime, inb_scal, s)
    end if

    call TLabMPI_Trp_AMD_Finalize()
    call TLab_AMD_DNS_Finalize()
    call TLab_Write_ASCII(lfile, 'test_rk_driver finished.')
    call TLab_Stop(0)
end program test_rk_driver
