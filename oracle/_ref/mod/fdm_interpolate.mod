﻿!mod$ v1 sum:54e7d2d00bf7ac8c
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 7890a03f87a12397 n fdm_com0_jacobian
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_interpolate
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:pi_wp
use tlab_constants,only:efile
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
use tlab_workflow,only:stagger_on
use fdm_com0_jacobian,only:fdm_c0int6p_lhs
use fdm_com0_jacobian,only:fdm_c0intvp6p_rhs
use fdm_com0_jacobian,only:fdm_c0intpv6p_rhs
use fdm_com0_jacobian,only:fdm_c0intvp6_lhs
use fdm_com0_jacobian,only:fdm_c0intpv6_lhs
use fdm_com0_jacobian,only:fdm_c0intvp6_rhs
use fdm_com0_jacobian,only:fdm_c0intpv6_rhs
use fdm_com0_jacobian,only:fdm_c1int6p_lhs
use fdm_com0_jacobian,only:fdm_c1intvp6p_rhs
use fdm_com0_jacobian,only:fdm_c1intpv6p_rhs
use fdm_com0_jacobian,only:fdm_c1intvp6_lhs
use fdm_com0_jacobian,only:fdm_c1intpv6_lhs
use fdm_com0_jacobian,only:fdm_c1intvp6_rhs
use fdm_com0_jacobian,only:fdm_c1intpv6_rhs
private::wp
private::wi
private::pi_wp
private::efile
private::tlab_write_ascii
private::tlab_stop
private::stagger_on
private::fdm_c0int6p_lhs
private::fdm_c0intvp6p_rhs
private::fdm_c0intpv6p_rhs
private::fdm_c0intvp6_lhs
private::fdm_c0intpv6_lhs
private::fdm_c0intvp6_rhs
private::fdm_c0intpv6_rhs
private::fdm_c1int6p_lhs
private::fdm_c1intvp6p_rhs
private::fdm_c1intpv6p_rhs
private::fdm_c1intvp6_lhs
private::fdm_c1intpv6_lhs
private::fdm_c1intvp6_rhs
private::fdm_c1intpv6_rhs
type::fdm_interpol_dt
sequence
integer(4)::mode_fdm
integer(4)::size
real(8),allocatable::lu0i(:,:)
real(8),allocatable::lu1i(:,:)
end type
contains
subroutine fdm_interpol_initialize(x,dx,var,wn)
real(8),intent(in)::x(:)
real(8),intent(in)::dx(:)
type(fdm_interpol_dt),intent(inout)::var
real(8),intent(inout)::wn(:)
end
subroutine fdm_interpol(dir,nlines,g,u,result,wrk2d)
integer(4),intent(in)::dir
integer(4),intent(in)::nlines
type(fdm_interpol_dt),intent(in)::g
real(8),intent(in)::u(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(inout)::wrk2d(1_8:*)
end
subroutine fdm_interpol_der1(dir,nlines,g,u,result,wrk2d)
integer(4),intent(in)::dir
integer(4),intent(in)::nlines
type(fdm_interpol_dt),intent(in)::g
real(8),intent(in)::u(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(inout)::wrk2d(1_8:*)
end
end
