!########################################################################
! Drop-in replacement of module TLabMPI_Transpose (base/tlab_mpi_transpose.f90): the same public names -- TLabMPI_Trp_Initialize,
! TLabMPI_Trp_PlanI / PlanK, the generic TLabMPI_Trp_Exec{I,K}_{Forward,Backward} (real and complex), tmpi_plan_dx / tmpi_plan_dz -- on DEVICE
! arrays: pack / unpack in HIP, the exchange an RCCL grouped send/recv inside the direction's communicator (include/tlab_amd_comm.h,
! libtlab_amd_comm.so).  The decomposition itself (ims_pro, ims_npro_i, ims_npro_k, ...) stays in TLabMPI_VARS, set by the host's own
! TLabMPI_Initialize; the RCCL communicators are created from it once:
!
!     if (ims_pro == 0) rc = tlab_comm_get_unique_id(id)          ! 128 bytes
!     call MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD)        ! the host's MPI stays responsible for start-up
!     call TLabMPI_Trp_AMD_Comm(id)                               ! tlab_comm_init(.., ims_npro, ims_pro, ims_npro_i, ims_npro_k)
!########################################################################
module TLabMPI_Transpose
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: lfile, efile, wp, wi
    use TLab_Memory, only: imax, jmax, kmax
    use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
    use TLabMPI_VARS, only: ims_pro, ims_npro, ims_npro_i, ims_npro_k
    use TLab_AMD_C, only: TLab_AMD_Check, tlab_slab_transport, tlab_pencil_transport
    implicit none
    private

    public :: TLabMPI_Trp_Initialize
    public :: TLabMPI_Trp_PlanI, TLabMPI_Trp_PlanK
    public :: TLabMPI_Trp_ExecK_Forward, TLabMPI_Trp_ExecK_Backward
    public :: TLabMPI_Trp_ExecI_Forward, TLabMPI_Trp_ExecI_Backward
    public :: TLabMPI_Trp_AMD_Comm, TLabMPI_Trp_AMD_Finalize
    public :: TLabMPI_Trp_AMD_Slab_Transport        ! the exchanges of the z-slab driver over the same communicator (tlab_comm_slab_transport)
    public :: TLabMPI_Trp_AMD_Pencil_Transport      ! ... and of the x/z pencil driver (tlab_comm_pencil_transport)

    type, public :: tmpi_transpose_dt
        integer(wi) :: nlines                                       ! as in the reference (tlab_mpi_transpose.f90:19-25)
        integer(wi) :: size3d
        integer :: dir = 0, nmax = 0, npage = 0                     ! what the device plan is made from, on first use
    end type tmpi_transpose_dt
    type(tmpi_transpose_dt), public :: tmpi_plan_dx         ! general plans used in derivatives and other operators
    type(tmpi_transpose_dt), public :: tmpi_plan_dz

    type(c_ptr), save :: comm = c_null_ptr
    ! the [Parallel] options of TLabMPI_Trp_Initialize (tlab_mpi_transpose.f90:29-47)
    integer, public, save :: trp_mode_i = 1, trp_mode_k = 1               ! TLAB_MPI_TRP_ASYNCHRONOUS: validated and kept, no effect on the device path
    logical, public, save :: trp_single_i = .false., trp_single_k = .false.   ! TransposeTypeI / TransposeTypeK = single: fp32 on the wire (real plans)

    type :: cache_dt                                                ! device plans (tlab_trp_plan_t) by (dir, nmax, npage, element size)
        integer :: dir = 0, nmax = 0, npage = 0, e = 0
        type(c_ptr) :: h = c_null_ptr
    end type cache_dt
    type(cache_dt), save :: cache(32)
    integer, save :: ncache = 0

    interface
        integer(c_int) function tlab_comm_init(comm, id, nranks, rank, npro_i, npro_k) bind(C, name='tlab_comm_init')
            import :: c_int, c_ptr, c_char
            type(c_ptr), intent(out) :: comm
            character(kind=c_char), intent(in) :: id(128)
            integer(c_int), value :: nranks, rank, npro_i, npro_k
        end function
        integer(c_int) function tlab_comm_slab_transport(comm, tr) bind(C, name='tlab_comm_slab_transport')
            import :: c_int, c_ptr, tlab_slab_transport
            type(c_ptr), value :: comm
            type(tlab_slab_transport), intent(out) :: tr
        end function
        integer(c_int) function tlab_comm_pencil_transport(comm, tr) bind(C, name='tlab_comm_pencil_transport')
            import :: c_int, c_ptr, tlab_pencil_transport
            type(c_ptr), value :: comm
            type(tlab_pencil_transport), intent(out) :: tr
        end function
        integer(c_int) function tlab_comm_destroy(comm) bind(C, name='tlab_comm_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
        end function
        integer(c_int) function tlab_trp_plan_create(plan, comm, dir, nmax, npage, elem_doubles, rank_dir, npro_dir) bind(C, name='tlab_trp_plan_create')
            import :: c_int, c_ptr
            type(c_ptr), intent(out) :: plan
            type(c_ptr), value :: comm
            integer(c_int), value :: dir, nmax, npage, elem_doubles, rank_dir, npro_dir
        end function
        integer(c_int) function tlab_trp_plan_destroy(plan) bind(C, name='tlab_trp_plan_destroy')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
        end function
        integer(c_int) function tlab_trp_plan_set_wire(plan, single) bind(C, name='tlab_trp_plan_set_wire')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: single
        end function
        integer(c_int) function tlab_trp_exec(plan, forward, a, b) bind(C, name='tlab_trp_exec')
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, a, b
            integer(c_int), value :: forward
        end function
    end interface

    interface TLabMPI_Trp_ExecK_Forward
        module procedure TLabMPI_Trp_ExecK_Forward_Real, TLabMPI_Trp_ExecK_Forward_Complex
    end interface TLabMPI_Trp_ExecK_Forward
    interface TLabMPI_Trp_ExecK_Backward
        module procedure TLabMPI_Trp_ExecK_Backward_Real, TLabMPI_Trp_ExecK_Backward_Complex
    end interface TLabMPI_Trp_ExecK_Backward
    interface TLabMPI_Trp_ExecI_Forward
        module procedure TLabMPI_Trp_ExecI_Forward_Real, TLabMPI_Trp_ExecI_Forward_Complex
    end interface TLabMPI_Trp_ExecI_Forward
    interface TLabMPI_Trp_ExecI_Backward
        module procedure TLabMPI_Trp_ExecI_Backward_Real, TLabMPI_Trp_ExecI_Backward_Complex
    end interface TLabMPI_Trp_ExecI_Backward

    integer, parameter :: DNS_ERROR_PARPARTITION = 45           ! include/dns_error.h

contains
    ! the RCCL communicators (world, ims_comm_x, ims_comm_z) from the decomposition of TLabMPI_VARS
    subroutine TLabMPI_Trp_AMD_Comm(id)
        character(kind=c_char), intent(in) :: id(128)
        call TLab_AMD_Check(tlab_comm_init(comm, id, int(ims_npro, c_int), int(ims_pro, c_int), int(ims_npro_i, c_int), int(ims_npro_k, c_int)), &
                            'tlab_comm_init')
    end subroutine TLabMPI_Trp_AMD_Comm

    subroutine TLabMPI_Trp_AMD_Slab_Transport(tr)
        type(tlab_slab_transport), intent(out) :: tr
        if (.not. c_associated(comm)) call TLab_AMD_Check(-1_c_int, 'TLabMPI_Transpose: no communicator yet (call TLabMPI_Trp_AMD_Comm first)')
        call TLab_AMD_Check(tlab_comm_slab_transport(comm, tr), 'tlab_comm_slab_transport')
    end subroutine TLabMPI_Trp_AMD_Slab_Transport

    subroutine TLabMPI_Trp_AMD_Pencil_Transport(tr)
        type(tlab_pencil_transport), intent(out) :: tr
        if (.not. c_associated(comm)) call TLab_AMD_Check(-1_c_int, 'TLabMPI_Transpose: no communicator yet (call TLabMPI_Trp_AMD_Comm first)')
        call TLab_AMD_Check(tlab_comm_pencil_transport(comm, tr), 'tlab_comm_pencil_transport')
    end subroutine TLabMPI_Trp_AMD_Pencil_Transport

    subroutine TLabMPI_Trp_AMD_Finalize()
        integer(c_int) rc
        integer i
        do i = 1, ncache
            rc = tlab_trp_plan_destroy(cache(i)%h)
        end do
        ncache = 0
        if (c_associated(comm)) rc = tlab_comm_destroy(comm)
        comm = c_null_ptr
    end subroutine TLabMPI_Trp_AMD_Finalize

    ! TLabMPI_Trp_Initialize(inifile)   tlab_mpi_transpose.f90:66-201.  The [Parallel] block is read as the reference reads it (:82-122):
    !   TransposeModeI / TransposeModeK (fallback [Main] ComModeITranspose / ComModeKTranspose) = none | asynchronous | sendrecv | alltoall select among
    !     MPI strategies; here every transposition is one grouped ncclSend / ncclRecv, so the value is validated (same error as the reference) and kept;
    !   TransposeTypeI / TransposeTypeK = double | single: the wire precision of the REAL transpositions (tlab_trp_plan_set_wire; complex ones always
    !     travel in double precision, :386-399).
    subroutine TLabMPI_Trp_Initialize(inifile)
        character(len=*), intent(in) :: inifile
        character(len=32) bakfile
        character(len=512) sRes
        integer, parameter :: DNS_ERROR_OPTION = 15, DNS_ERROR_UNDEVELOP = 104
        bakfile = trim(adjustl(inifile))//'.bak'
        call ScanFile_Char(bakfile, inifile, 'Parallel', 'TransposeModeI', 'void', sRes)
        if (trim(adjustl(sRes)) == 'void') call ScanFile_Char(bakfile, inifile, 'Main', 'ComModeITranspose', 'asynchronous', sRes)
        trp_mode_i = mode_code(sRes, 'TransposeModeI')
        call ScanFile_Char(bakfile, inifile, 'Parallel', 'TransposeModeK', 'void', sRes)
        if (trim(adjustl(sRes)) == 'void') call ScanFile_Char(bakfile, inifile, 'Main', 'ComModeKTranspose', 'asynchronous', sRes)
        trp_mode_k = mode_code(sRes, 'TransposeModeK')
        call ScanFile_Char(bakfile, inifile, 'Parallel', 'TransposeTypeK', 'Double', sRes)
        if (trim(adjustl(sRes)) == 'double') then; trp_single_k = .false.
        elseif (trim(adjustl(sRes)) == 'single') then; trp_single_k = .true.
        else
            call TLab_Write_ASCII(efile, __FILE__//'. Wrong TransposeTypeK.')
            call TLab_Stop(DNS_ERROR_UNDEVELOP)
        end if
        call ScanFile_Char(bakfile, inifile, 'Parallel', 'TransposeTypeI', 'Double', sRes)
        if (trim(adjustl(sRes)) == 'double') then; trp_single_i = .false.
        elseif (trim(adjustl(sRes)) == 'single') then; trp_single_i = .true.
        else
            call TLab_Write_ASCII(efile, __FILE__//'. Wrong TransposeTypeI.')
            call TLab_Stop(DNS_ERROR_UNDEVELOP)
        end if
        if (ims_npro_i > 1) tmpi_plan_dx = TLabMPI_Trp_PlanI(imax, kmax*jmax, message='Ox derivatives.')     ! :189-192
        if (ims_npro_k > 1) tmpi_plan_dz = TLabMPI_Trp_PlanK(kmax, imax*jmax, message='Oz derivatives.')     ! :194-197
    end subroutine TLabMPI_Trp_Initialize

    integer function mode_code(sRes, key) result(m)      ! TLAB_MPI_TRP_NONE .. _ALLTOALL (tlab_mpi_transpose.f90:29-33)
        character(len=*), intent(in) :: sRes, key
        integer, parameter :: DNS_ERROR_OPTION = 15
        select case (trim(adjustl(sRes)))
        case ('none'); m = 0
        case ('asynchronous'); m = 1
        case ('sendrecv'); m = 2
        case ('alltoall'); m = 3
        case default
            m = -1
            call TLab_Write_ASCII(efile, __FILE__//'. Wrong '//key//' option.')
            call TLab_Stop(DNS_ERROR_OPTION)
        end select
    end function mode_code

    ! TLabMPI_Trp_PlanI(nmax, npage, ...)   tlab_mpi_transpose.f90:205-286
    function TLabMPI_Trp_PlanI(nmax, npage, message) result(trp_plan)
        integer(wi), intent(in) :: npage, nmax
        character(len=*), intent(in), optional :: message
        type(tmpi_transpose_dt) :: trp_plan
        if (present(message)) call TLab_Write_ASCII(lfile, 'Creating device transposition plan for '//trim(adjustl(message)))
        if (mod(npage, ims_npro_i) /= 0) then
            call TLab_Write_ASCII(efile, 'TLabMPI_TypeI_Create. Ratio npage/npro not an integer.')
            call TLab_Stop(DNS_ERROR_PARPARTITION)
        end if
        trp_plan%nlines = npage/ims_npro_i
        trp_plan%size3d = npage*nmax
        trp_plan%dir = 1; trp_plan%nmax = nmax; trp_plan%npage = npage
    end function TLabMPI_Trp_PlanI

    ! TLabMPI_Trp_PlanK(nmax, npage, ...)   tlab_mpi_transpose.f90:290-339
    function TLabMPI_Trp_PlanK(nmax, npage, message) result(trp_plan)
        integer(wi), intent(in) :: npage, nmax
        character(len=*), intent(in), optional :: message
        type(tmpi_transpose_dt) :: trp_plan
        if (present(message)) call TLab_Write_ASCII(lfile, 'Creating device transposition plan for '//trim(adjustl(message)))
        if (mod(npage, ims_npro_k) /= 0) then
            call TLab_Write_ASCII(efile, 'TLabMPI_TypeK_Create. Ratio npage/npro not an integer.')
            call TLab_Stop(DNS_ERROR_PARPARTITION)
        end if
        trp_plan%nlines = npage/ims_npro_k
        trp_plan%size3d = npage*nmax
        trp_plan%dir = 3; trp_plan%nmax = nmax; trp_plan%npage = npage
    end function TLabMPI_Trp_PlanK

    ! device plan for 8-byte (e = 1) or 16-byte (e = 2) elements, created on first use (the reference's plans are plain values, passed intent(in))
    function handle(trp_plan, e) result(h)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        integer, intent(in) :: e
        type(c_ptr) :: h
        integer i
        do i = 1, ncache
            if (cache(i)%dir == trp_plan%dir .and. cache(i)%nmax == trp_plan%nmax .and. cache(i)%npage == trp_plan%npage .and. cache(i)%e == e) then
                h = cache(i)%h
                return
            end if
        end do
        if (ncache == size(cache)) call TLab_AMD_Check(-1_c_int, 'TLabMPI_Transpose: more than 32 distinct transposition plans')
        ! without a communicator the device plan is a one-rank plan: with a decomposed direction that would be a local copy instead of an exchange
        if (.not. c_associated(comm) .and. ((trp_plan%dir == 1 .and. ims_npro_i > 1) .or. (trp_plan%dir == 3 .and. ims_npro_k > 1))) &
            call TLab_AMD_Check(-1_c_int, 'TLabMPI_Transpose: decomposed direction but no RCCL communicator (call TLabMPI_Trp_AMD_Comm after TLabMPI_Initialize)')
        call TLab_AMD_Check(tlab_trp_plan_create(h, comm, int(trp_plan%dir, c_int), int(trp_plan%nmax, c_int), int(trp_plan%npage, c_int), &
                                                 int(e, c_int), 0_c_int, 1_c_int), 'tlab_trp_plan_create')
        if (e == 1 .and. ((trp_plan%dir == 1 .and. trp_single_i) .or. (trp_plan%dir == 3 .and. trp_single_k))) &      ! [Parallel] TransposeType = single
            call TLab_AMD_Check(tlab_trp_plan_set_wire(h, 1_c_int), 'tlab_trp_plan_set_wire')
        ncache = ncache + 1
        cache(ncache)%dir = trp_plan%dir; cache(ncache)%nmax = trp_plan%nmax; cache(ncache)%npage = trp_plan%npage; cache(ncache)%e = e
        cache(ncache)%h = h
    end function handle

    subroutine exec_any(trp_plan, e, forward, a, b)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        integer, intent(in) :: e, forward
        type(c_ptr), intent(in) :: a, b
        call TLab_AMD_Check(tlab_trp_exec(handle(trp_plan, e), int(forward, c_int), a, b), 'tlab_trp_exec')
    end subroutine exec_any

    ! TLabMPI_Trp_ExecK_Forward / _Backward   tlab_mpi_transpose.f90:343-458 ;  ExecI :462-553
    subroutine TLabMPI_Trp_ExecK_Forward_Real(a, b, trp_plan)
        real(wp), intent(in), target :: a(*)
        real(wp), intent(out), target :: b(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 1, 1, c_loc(a), c_loc(b))
    end subroutine
    subroutine TLabMPI_Trp_ExecK_Forward_Complex(a, b, trp_plan)
        complex(wp), intent(in), target :: a(*)
        complex(wp), intent(out), target :: b(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 2, 1, c_loc(a), c_loc(b))
    end subroutine
    subroutine TLabMPI_Trp_ExecK_Backward_Real(b, a, trp_plan)
        real(wp), intent(in), target :: b(*)
        real(wp), intent(out), target :: a(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 1, 0, c_loc(b), c_loc(a))
    end subroutine
    subroutine TLabMPI_Trp_ExecK_Backward_Complex(b, a, trp_plan)
        complex(wp), intent(in), target :: b(*)
        complex(wp), intent(out), target :: a(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 2, 0, c_loc(b), c_loc(a))
    end subroutine
    subroutine TLabMPI_Trp_ExecI_Forward_Real(a, b, trp_plan)
        real(wp), intent(in), target :: a(*)
        real(wp), intent(out), target :: b(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 1, 1, c_loc(a), c_loc(b))
    end subroutine
    subroutine TLabMPI_Trp_ExecI_Forward_Complex(a, b, trp_plan)
        complex(wp), intent(in), target :: a(*)
        complex(wp), intent(out), target :: b(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 2, 1, c_loc(a), c_loc(b))
    end subroutine
    subroutine TLabMPI_Trp_ExecI_Backward_Real(b, a, trp_plan)
        real(wp), intent(in), target :: b(*)
        real(wp), intent(out), target :: a(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 1, 0, c_loc(b), c_loc(a))
    end subroutine
    subroutine TLabMPI_Trp_ExecI_Backward_Complex(b, a, trp_plan)
        complex(wp), intent(in), target :: b(*)
        complex(wp), intent(out), target :: a(*)
        type(tmpi_transpose_dt), intent(in) :: trp_plan
        call exec_any(trp_plan, 2, 0, c_loc(b), c_loc(a))
    end subroutine

end module TLabMPI_Transpose
