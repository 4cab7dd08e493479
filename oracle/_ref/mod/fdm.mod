﻿!mod$ v1 sum:06183c4da53c4dbe
!need$ ff3fca9ebc58e858 n tlab_grid
!need$ 54e7d2d00bf7ac8c n fdm_interpolate
!need$ 8d4bae2479538272 n fdm_integral
!need$ 370470eb4a3adeb1 n tlab_constants
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 9281856c4f7b499b n fdm_derivative
module fdm
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:roundoff_wp
use tlab_constants,only:efile
use tlab_constants,only:wfile
use tlab_constants,only:bcs_dd
use tlab_constants,only:bcs_nd
use tlab_constants,only:bcs_dn
use tlab_constants,only:bcs_nn
use tlab_constants,only:bcs_min
use tlab_constants,only:bcs_max
use tlab_constants,only:bcs_none
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
use tlab_workflow,only:stagger_on
use tlab_grid,only:x
use tlab_grid,only:y
use tlab_grid,only:z
use tlab_grid,only:grid_dt
use fdm_derivative,only:fdm_derivative_dt
use fdm_derivative,only:fdm_com4_jacobian
use fdm_derivative,only:fdm_com6_jacobian_penta
use fdm_derivative,only:fdm_com6_jacobian
use fdm_derivative,only:fdm_com6_jacobian_hyper
use fdm_derivative,only:fdm_com8_jacobian
use fdm_derivative,only:fdm_com6_direct
use fdm_derivative,only:fdm_com4_direct
use fdm_derivative,only:fdm_der1_initialize
use fdm_derivative,only:fdm_der1_solve
use fdm_derivative,only:fdm_der2_initialize
use fdm_derivative,only:fdm_der2_solve
use fdm_interpolate,only:fdm_interpol_dt
use fdm_interpolate,only:fdm_interpol_initialize
use fdm_interpolate,only:fdm_interpol
use fdm_interpolate,only:fdm_interpol_der1
use fdm_integral,only:fdm_integral_dt
use fdm_integral,only:fdm_int1_initialize
use fdm_integral,only:fdm_int1_createsystem
use fdm_integral,only:fdm_int1_solve
use fdm_integral,only:fdm_int2_initialize
use fdm_integral,only:fdm_int2_solve
private::wp
private::wi
private::roundoff_wp
private::efile
private::wfile
private::bcs_dd
private::bcs_nd
private::bcs_dn
private::bcs_nn
private::bcs_min
private::bcs_max
private::bcs_none
private::tlab_write_ascii
private::tlab_stop
private::stagger_on
private::x
private::y
private::z
private::grid_dt
private::fdm_derivative_dt
private::fdm_com4_jacobian
private::fdm_com6_jacobian_penta
private::fdm_com6_jacobian
private::fdm_com6_jacobian_hyper
private::fdm_com8_jacobian
private::fdm_com6_direct
private::fdm_com4_direct
private::fdm_der1_initialize
private::fdm_der1_solve
private::fdm_der2_initialize
private::fdm_der2_solve
private::fdm_interpol_dt
private::fdm_interpol_initialize
private::fdm_interpol
private::fdm_interpol_der1
private::fdm_integral_dt
private::fdm_int1_initialize
private::fdm_int1_createsystem
private::fdm_int1_solve
private::fdm_int2_initialize
private::fdm_int2_solve
type::fdm_dt
sequence
character(8_8,1)::name
integer(4)::size
logical(4)::uniform=.false._4
logical(4)::periodic=.false._4
real(8)::scale
real(8),allocatable::nodes(:)
real(8),allocatable::jac(:,:)
type(fdm_derivative_dt)::der1
type(fdm_derivative_dt)::der2
type(fdm_interpol_dt)::intl
end type
type(fdm_dt),protected::g(1_8:3_8)
type(fdm_integral_dt),protected::fdm_int0(1_8:2_8)
contains
subroutine fdm_initialize(inifile)
character(*,1),intent(in),optional::inifile
end
subroutine fdm_createplan(x,g,locscale)
type(grid_dt)::x
type(fdm_dt),intent(inout)::g
real(8),intent(in),optional::locscale
end
end
