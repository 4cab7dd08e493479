!########################################################################
! Drop-in replacement of base/tlab_memory.f90 (modules TLab_Arrays, TLab_Pointers, TLab_Memory): the same public names, but the field-sized
! arrays q, s, txc, wrk3d -- and whatever else the host allocates through TLab_Allocate_Real -- live in HBM.
!
! The reference allocates every big array in one routine, TLab_Allocate_Real(C_FILE_LOC, a, dims, s) (base/tlab_memory.f90:306-331; callers
! :201-207, tools/dns/dns_main.f90:103-104).  An ALLOCATABLE cannot be given foreign memory, so the declarations change from
! `real(wp), allocatable` to `real(wp), pointer, contiguous` (source-compatible for every statement that indexes, slices or passes them)
! and the allocate statement becomes tlab_malloc + c_f_pointer: c_loc(u) inside the operator shims then IS the device address.
! Small host-side scratch (wrk1d, wrk2d: coefficient work of the unchanged FDM / initialisation code) stays in host memory.
!########################################################################
module TLab_Arrays
    use TLab_Constants, only: wp
    implicit none
    save

    real(wp), pointer, contiguous :: q(:, :) => null()      ! Eulerian fields, flow variables          (device)
    real(wp), pointer, contiguous :: s(:, :) => null()      ! Eulerian fields, scalar variables        (device)
    real(wp), pointer, contiguous :: txc(:, :) => null()    ! Temporary space for Eulerian fields      (device)
    real(wp), pointer, contiguous :: wrk1d(:, :) => null()  ! Work arrays (scratch space)              (host)
    real(wp), pointer, contiguous :: wrk2d(:, :) => null()  ! Work arrays (scratch space)              (host)
    real(wp), pointer, contiguous :: wrk3d(:) => null()     ! Work arrays (scratch space)              (device)

end module TLab_Arrays

! ###################################################################
module TLab_Pointers
    use TLab_Constants, only: wp
    implicit none

    real(wp), pointer :: u(:) => null()
    real(wp), pointer :: v(:) => null()
    real(wp), pointer :: w(:) => null()

    real(wp), pointer :: tmp1(:) => null()
    real(wp), pointer :: tmp2(:) => null()
    real(wp), pointer :: tmp3(:) => null()
    real(wp), pointer :: tmp4(:) => null()
    real(wp), pointer :: tmp5(:) => null()
    real(wp), pointer :: tmp6(:) => null()
    real(wp), pointer :: tmp7(:) => null()
    real(wp), pointer :: tmp8(:) => null()
    real(wp), pointer :: tmp9(:) => null()

end module TLab_Pointers

! ###################################################################
module TLab_Memory
    use, intrinsic :: iso_c_binding
    use TLab_Constants, only: wp, wi, lfile, efile
    use TLab_WorkFlow, only: TLab_Write_ASCII, TLab_Stop
    use TLab_AMD_C, only: tlab_malloc, tlab_init, TLab_AMD_Check
    implicit none
    private
    save

    ! Arrays sizes (base/tlab_memory.f90:115-127)
    integer(wi), public :: imax, jmax, kmax     ! number of grid nodes per direction locally per processor
    integer(wi), public :: isize_field = 0      ! =imax*jmax*kmax, 3D fields sizes locally per processor
    integer(wi), public :: inb_flow             ! # of prognostic 3d flow fields (flow evolution equations)
    integer(wi), public :: inb_flow_array       ! >= inb_flow, # of prognostic and diagnostic 3d flow arrays
    integer(wi), public :: inb_scal             ! # of prognostic 3d scal fields (scal evolution equations)
    integer(wi), public :: inb_scal_array       ! >= inb_scal, # of prognostic and diagnostic 3d scal arrays
    integer(wi), public :: isize_wrk1d = 0, inb_wrk1d
    integer(wi), public :: isize_wrk2d = 0, inb_wrk2d
    integer(wi), public :: isize_wrk3d = 0
    integer(wi), public :: isize_txc_field = 0, inb_txc
    integer(wi), public :: isize_txc_dimz

    public :: TLab_Initialize_Memory
    public :: TLab_Allocate_Real

    logical :: device_ready = .false.

contains
    ! ###################################################################
    ! TLab_Initialize_Memory   base/tlab_memory.f90:164-216 (serial branch)
    subroutine TLab_Initialize_Memory(C_FILE_LOC)
        use TLab_Arrays
        use TLab_WorkFlow, only: fourier_on
        character(len=*), intent(in) :: C_FILE_LOC

        isize_field = imax*jmax*kmax
        isize_txc_field = imax*jmax*kmax
        if (fourier_on) then
            isize_txc_dimz = (imax + 2)*jmax            ! Add space for Nyquist frequency
            isize_txc_field = max(isize_txc_field, isize_txc_dimz*kmax)
        end if
        isize_wrk1d = max(imax, max(jmax, kmax))
        isize_wrk2d = max(imax*jmax, max(imax*kmax, jmax*kmax))
        isize_wrk3d = max(isize_wrk3d, isize_field)
        isize_wrk3d = max(isize_wrk3d, isize_txc_field)

        call TLab_Allocate_Real(C_FILE_LOC, q, [isize_field, inb_flow_array], 'flow')
        call TLab_Allocate_Real(C_FILE_LOC, s, [isize_field, inb_scal_array], 'scal')
        call TLab_Allocate_Real(C_FILE_LOC, txc, [isize_txc_field, inb_txc], 'txc')
        allocate (wrk1d(isize_wrk1d, inb_wrk1d))        ! host scratch of the unchanged initialisation code
        allocate (wrk2d(isize_wrk2d, inb_wrk2d))
        call TLab_Allocate_Real(C_FILE_LOC, wrk3d, [isize_wrk3d], 'wrk3d')

        call TLab_Set_Pointers()
    end subroutine TLab_Initialize_Memory

    ! TLab_Set_Pointers   base/tlab_memory.f90:220-251
    subroutine TLab_Set_Pointers()
        use TLab_Arrays
        use TLab_Pointers
        integer(wi) idummy(2)

        idummy = shape(q)
        if (idummy(2) >= 1) u(1:isize_field) => q(1:isize_field, 1)
        if (idummy(2) >= 2) v(1:isize_field) => q(1:isize_field, 2)
        if (idummy(2) >= 3) w(1:isize_field) => q(1:isize_field, 3)
        idummy = shape(txc)
        if (idummy(2) >= 1) tmp1(1:isize_field) => txc(1:isize_field, 1)
        if (idummy(2) >= 2) tmp2(1:isize_field) => txc(1:isize_field, 2)
        if (idummy(2) >= 3) tmp3(1:isize_field) => txc(1:isize_field, 3)
        if (idummy(2) >= 4) tmp4(1:isize_field) => txc(1:isize_field, 4)
        if (idummy(2) >= 5) tmp5(1:isize_field) => txc(1:isize_field, 5)
        if (idummy(2) >= 6) tmp6(1:isize_field) => txc(1:isize_field, 6)
        if (idummy(2) >= 7) tmp7(1:isize_field) => txc(1:isize_field, 7)
        if (idummy(2) >= 8) tmp8(1:isize_field) => txc(1:isize_field, 8)
        if (idummy(2) >= 9) tmp9(1:isize_field) => txc(1:isize_field, 9)
    end subroutine TLab_Set_Pointers

    ! ###################################################################
    ! TLab_Allocate_Real(C_FILE_LOC, a, dims, s)   base/tlab_memory.f90:306-331 : the allocation hook
    subroutine TLab_Allocate_Real(C_FILE_LOC, a, dims, s)
        character(len=*), intent(in) :: C_FILE_LOC
        real(wp), pointer, intent(inout) :: a(..)
        integer(wi), intent(in) :: dims(:)
        character(len=*), intent(in) :: s

        type(c_ptr) :: p
        integer(c_size_t) :: nbytes
        integer(c_int) :: rc
        integer :: id
        character(len=128) :: line
        integer, parameter :: DNS_ERROR_ALLOC = 80               ! include/dns_error.h:75

        if (.not. device_ready) then
            call TLab_AMD_Check(tlab_init(0_c_int), 'tlab_init')
            device_ready = .true.
        end if
        nbytes = 8_c_size_t
        do id = 1, size(dims)
            nbytes = nbytes*int(max(dims(id), 0), c_size_t)
        end do
        write (line, '(a,i0,a)') 'Allocating array '//trim(adjustl(s))//' of ', nbytes, ' bytes in HBM (tlab_malloc)'      ! TLAB_ALLOCATE_LOG
        call TLab_Write_ASCII(lfile, line)
        rc = tlab_malloc(p, max(nbytes, 8_c_size_t))
        if (rc /= 0) then                                        ! TLAB_ALLOCATE_ERR
            call TLab_Write_ASCII(efile, C_FILE_LOC//'. Error while allocating memory space for '//trim(adjustl(s))//'.')
            call TLab_Stop(DNS_ERROR_ALLOC)
        end if
        select rank (a)
        rank (1)
            call c_f_pointer(p, a, [dims(1)])
        rank (2)
            call c_f_pointer(p, a, [dims(1), dims(2)])
        rank (3)
            call c_f_pointer(p, a, [dims(1), dims(2), dims(3)])
        rank (4)
            call c_f_pointer(p, a, [dims(1), dims(2), dims(3), dims(4)])
        rank default
            call TLab_Write_ASCII(efile, C_FILE_LOC//'. Rank too large while allocating memory space for '//trim(adjustl(s))//'.')
            call TLab_Stop(DNS_ERROR_ALLOC)
        end select
    end subroutine TLab_Allocate_Real

end module TLab_Memory
