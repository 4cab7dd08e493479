﻿!mod$ v1 sum:ff3fca9ebc58e858
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_grid
use tlab_constants,only:efile
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
private::efile
private::wp
private::wi
private::tlab_write_ascii
private::tlab_stop
type::grid_dt
sequence
character(8_8,1)::name
integer(4)::size
logical(4)::periodic=.false._4
real(8)::scale
real(8),allocatable::nodes(:)
end type
type(grid_dt)::x
type(grid_dt)::y
type(grid_dt)::z
contains
subroutine tlab_grid_read(name,x,y,z,sizes)
character(*,1)::name
type(grid_dt),intent(inout)::x
type(grid_dt),intent(inout)::y
type(grid_dt),intent(inout)::z
integer(4),intent(in),optional::sizes(1_8:3_8)
end
subroutine tlab_grid_write(name,x,y,z)
character(*,1)::name
type(grid_dt),intent(in)::x
type(grid_dt),intent(in)::y
type(grid_dt),intent(in)::z
end
end
