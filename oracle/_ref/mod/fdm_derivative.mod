﻿!mod$ v1 sum:9281856c4f7b499b
!need$ ee3e4b05f7cc2dad n fdm_matmul
!need$ fe23d2f9afd7370a n fdm_comx_direct
!need$ c314556627c1bfdb n fdm_com1_jacobian
!need$ 8e3ed643e1b51b90 n fdm_com2_jacobian
!need$ 7890a03f87a12397 n fdm_com0_jacobian
!need$ 370470eb4a3adeb1 n tlab_constants
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 7cca51c0634c6b29 n fdm_base
module fdm_derivative
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:pi_wp
use tlab_constants,only:efile
use tlab_constants,only:wfile
use tlab_constants,only:bcs_dd
use tlab_constants,only:bcs_nd
use tlab_constants,only:bcs_dn
use tlab_constants,only:bcs_nn
use tlab_constants,only:bcs_none
use tlab_constants,only:bcs_periodic
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
use fdm_matmul,only:matmul_3d
use fdm_matmul,only:matmul_3d_add
use fdm_matmul,only:matmul_3d_antisym
use fdm_matmul,only:matmul_3d_sym
use fdm_matmul,only:matmul_5d
use fdm_matmul,only:matmul_5d_add
use fdm_matmul,only:matmul_5d_antisym
use fdm_matmul,only:matmul_5d_sym
use fdm_matmul,only:matmul_7d_antisym
use fdm_matmul,only:matmul_7d_sym
use fdm_base,only:pi
use fdm_base,only:pi_p
use fdm_base,only:pi_pp_3
use fdm_base,only:lag
use fdm_base,only:lag_p
use fdm_base,only:lag_pp_3
use fdm_base,only:coef_e1n3_biased
use fdm_base,only:coef_e1n2_biased
use fdm_base,only:fdm_bcs_neumann
use fdm_base,only:fdm_bcs_reduce
use fdm_comx_direct,only:fdm_c1n4_direct
use fdm_comx_direct,only:fdm_c1n6_direct
use fdm_comx_direct,only:fdm_c2n6_direct
use fdm_comx_direct,only:fdm_c2n4_direct
use fdm_com1_jacobian,only:fdm_c1n4_jacobian
use fdm_com1_jacobian,only:fdm_c1n6_jacobian
use fdm_com1_jacobian,only:fdm_c1n6_jacobian_penta
use fdm_com2_jacobian,only:fdm_c2n4_jacobian
use fdm_com2_jacobian,only:fdm_c2n6_jacobian
use fdm_com2_jacobian,only:fdm_c2n6_hyper_jacobian
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
private::wfile
private::bcs_dd
private::bcs_nd
private::bcs_dn
private::bcs_nn
private::bcs_none
private::bcs_periodic
private::tlab_write_ascii
private::tlab_stop
private::matmul_3d
private::matmul_3d_add
private::matmul_3d_antisym
private::matmul_3d_sym
private::matmul_5d
private::matmul_5d_add
private::matmul_5d_antisym
private::matmul_5d_sym
private::matmul_7d_antisym
private::matmul_7d_sym
private::pi
private::pi_p
private::pi_pp_3
private::lag
private::lag_p
private::lag_pp_3
private::coef_e1n3_biased
private::coef_e1n2_biased
private::fdm_bcs_neumann
private::fdm_bcs_reduce
private::fdm_c1n4_direct
private::fdm_c1n6_direct
private::fdm_c2n6_direct
private::fdm_c2n4_direct
private::fdm_c1n4_jacobian
private::fdm_c1n6_jacobian
private::fdm_c1n6_jacobian_penta
private::fdm_c2n4_jacobian
private::fdm_c2n6_jacobian
private::fdm_c2n6_hyper_jacobian
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
type::fdm_derivative_dt
sequence
integer(4)::mode_fdm
integer(4)::size
logical(4)::periodic=.false._4
logical(4)::need_1der=.false._4
integer(4)::nb_diag(1_8:2_8)
real(8)::rhs_b(1_8:4_8,0_8:7_8)
real(8)::rhs_t(0_8:4_8,1_8:7_8)
real(8),allocatable::lhs(:,:)
real(8),allocatable::rhs(:,:)
real(8),allocatable::mwn(:)
real(8),allocatable::lu(:,:)
procedure(matmul_interface),nopass,pointer::matmul
end type
private::matmul_interface
abstract interface
subroutine matmul_interface(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
end interface
integer(4),parameter::fdm_com4_jacobian=4_4
integer(4),parameter::fdm_com6_jacobian_penta=5_4
integer(4),parameter::fdm_com6_jacobian=6_4
integer(4),parameter::fdm_com6_jacobian_hyper=7_4
integer(4),parameter::fdm_com8_jacobian=8_4
integer(4),parameter::fdm_com6_direct=16_4
integer(4),parameter::fdm_com4_direct=17_4
private::fdm_der1_createsystem
private::fdm_der2_createsystem
contains
subroutine fdm_der1_initialize(x,dx,g,periodic,bcs_cases)
real(8),intent(in)::x(:)
real(8),intent(in)::dx(:)
type(fdm_derivative_dt),intent(inout)::g
logical(4),intent(in)::periodic
integer(4),intent(in)::bcs_cases(:)
end
subroutine fdm_der1_createsystem(x,dx,g,periodic)
real(8),intent(in)::x(:)
real(8),intent(in)::dx(:)
type(fdm_derivative_dt),intent(inout)::g
logical(4),intent(in)::periodic
end
subroutine fdm_der1_solve(nlines,ibc,g,lu1,u,result,wrk2d)
integer(4),intent(in)::nlines
integer(4),intent(in)::ibc
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lu1(:,:)
real(8),intent(in)::u(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(inout)::wrk2d(1_8:*)
end
subroutine fdm_der2_initialize(x,dx,g,periodic,uniform)
real(8),intent(in)::x(:)
real(8),intent(inout)::dx(:,:)
type(fdm_derivative_dt),intent(inout)::g
logical(4),intent(in)::periodic
logical(4),intent(in)::uniform
end
subroutine fdm_der2_createsystem(x,dx,g,periodic,uniform)
real(8),intent(in)::x(:)
real(8),intent(inout)::dx(:,:)
type(fdm_derivative_dt),intent(inout)::g
logical(4),intent(in)::periodic
logical(4),intent(in)::uniform
end
subroutine fdm_der2_solve(nlines,g,lu,u,result,du,wrk2d)
integer(4),intent(in)::nlines
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lu(:,:)
real(8),intent(in)::u(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(in)::du(1_8:int(nlines,kind=8),1_8:int(g%size,kind=8))
real(8),intent(out)::wrk2d(1_8:*)
end
end
