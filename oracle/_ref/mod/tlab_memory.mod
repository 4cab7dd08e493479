﻿!mod$ v1 sum:d21c265ab3fb0f9d
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_memory
use tlab_constants,only:sp
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:longi
use tlab_constants,only:lfile
use tlab_constants,only:efile
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
private::sp
private::wp
private::wi
private::longi
private::lfile
private::efile
private::tlab_write_ascii
private::tlab_stop
integer(4)::imax
integer(4)::jmax
integer(4)::kmax
integer(4)::isize_field
integer(4)::inb_flow
integer(4)::inb_flow_array
integer(4)::inb_scal
integer(4)::inb_scal_array
integer(4)::isize_wrk1d
integer(4)::inb_wrk1d
integer(4)::isize_wrk2d
integer(4)::inb_wrk2d
integer(4)::isize_wrk3d
integer(4)::isize_txc_field
integer(4)::inb_txc
integer(4)::isize_txc_dimz
character(128_8,1),private::str
character(128_8,1),private::line
integer(4),private::ierr
private::tlab_set_pointers
private::tlab_set_pointers_3d
private::tlab_set_pointers_c
private::tlab_allocate_log_short
private::tlab_allocate_log_long
private::tlab_allocate_err
interface tlab_allocate_log
procedure::tlab_allocate_log_short
procedure::tlab_allocate_log_long
end interface
private::tlab_allocate_log
contains
subroutine tlab_initialize_memory(c_file_loc)
character(*,1),intent(in)::c_file_loc
end
subroutine tlab_set_pointers()
end
subroutine tlab_set_pointers_3d()
end
subroutine tlab_set_pointers_c()
end
subroutine tlab_allocate_real(c_file_loc,a,dims,s)
character(*,1),intent(in)::c_file_loc
real(8),allocatable,intent(inout)::a(..)
integer(4),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_real_long(c_file_loc,a,dims,s)
character(*,1),intent(in)::c_file_loc
real(8),allocatable,intent(inout)::a(:)
integer(8),intent(in)::dims(1_8:1_8)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_single(c_file_loc,a,dims,s)
character(*,1),intent(in)::c_file_loc
real(4),allocatable,intent(inout)::a(..)
integer(4),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_int(c_file_loc,a,dims,s)
character(*,1),intent(in)::c_file_loc
integer(4),allocatable,intent(inout)::a(..)
integer(4),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_long_int(c_file_loc,a,dims,s)
character(*,1),intent(in)::c_file_loc
integer(8),allocatable,intent(inout)::a(..)
integer(4),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_log_short(log_file,dims,s)
character(*,1),intent(in)::log_file
integer(4),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_log_long(log_file,dims,s)
character(*,1),intent(in)::log_file
integer(8),intent(in)::dims(:)
character(*,1),intent(in)::s
end
subroutine tlab_allocate_err(c_file_loc,log_file,s)
character(*,1)::c_file_loc
character(*,1)::log_file
character(*,1)::s
end
end
