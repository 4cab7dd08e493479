﻿!mod$ v1 sum:8d4bae2479538272
!need$ 9281856c4f7b499b n fdm_derivative
!need$ 7cca51c0634c6b29 n fdm_base
!need$ ee3e4b05f7cc2dad n fdm_matmul
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_integral
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:efile
use tlab_constants,only:bcs_dd
use tlab_constants,only:bcs_nd
use tlab_constants,only:bcs_dn
use tlab_constants,only:bcs_nn
use tlab_constants,only:bcs_min
use tlab_constants,only:bcs_max
use tlab_constants,only:bcs_both
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
use fdm_derivative,only:fdm_derivative_dt
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
private::wp
private::wi
private::efile
private::bcs_dd
private::bcs_nd
private::bcs_dn
private::bcs_nn
private::bcs_min
private::bcs_max
private::bcs_both
private::tlab_write_ascii
private::tlab_stop
private::fdm_derivative_dt
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
type::fdm_integral_dt
sequence
integer(4)::mode_fdm
real(8)::lambda
integer(4)::bc
real(8)::rhs_b(1_8:5_8,0_8:7_8)
real(8)::rhs_t(0_8:4_8,1_8:8_8)
real(8),allocatable::lhs(:,:)
real(8),allocatable::rhs(:,:)
end type
private::fdm_int2_createsystem
contains
subroutine fdm_int1_initialize(x,g,lambda,ibc,fdmi)
real(8),intent(in)::x(:)
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lambda
integer(4),intent(in)::ibc
type(fdm_integral_dt),intent(inout)::fdmi
end
subroutine fdm_int1_createsystem(x,g,lambda,ibc,fdmi)
real(8),intent(in)::x(:)
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lambda
integer(4),intent(in)::ibc
type(fdm_integral_dt),intent(inout)::fdmi
end
subroutine fdm_int1_solve(nlines,fdmi,rhsi,f,result,wrk2d,du_boundary)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi
real(8),intent(in)::rhsi(:,:)
real(8),intent(in)::f(1_8:int(nlines,kind=8),1_8:size(fdmi%lhs,dim=1,kind=8))
real(8),intent(inout)::result(1_8:int(nlines,kind=8),1_8:size(fdmi%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out),optional::du_boundary(1_8:int(nlines,kind=8))
end
subroutine fdm_int2_initialize(x,g,lambda2,ibc,fdmi)
real(8),intent(in)::x(:)
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lambda2
integer(4),intent(in)::ibc
type(fdm_integral_dt),intent(inout)::fdmi
end
subroutine fdm_int2_createsystem(x,g,lambda2,ibc,fdmi)
real(8),intent(in)::x(:)
type(fdm_derivative_dt),intent(in)::g
real(8),intent(in)::lambda2
integer(4),intent(in)::ibc
type(fdm_integral_dt),intent(inout)::fdmi
end
subroutine fdm_int2_solve(nlines,fdmi,rhsi,f,result,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi
real(8),intent(in)::rhsi(:,:)
real(8),intent(in)::f(1_8:int(nlines,kind=8),1_8:size(fdmi%lhs,dim=1,kind=8))
real(8),intent(inout)::result(1_8:int(nlines,kind=8),1_8:size(fdmi%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:2_8)
end
end
