﻿!mod$ v1 sum:6ecf0f727b23a71b
!need$ 8d4bae2479538272 n fdm_integral
!need$ 370470eb4a3adeb1 n tlab_constants
module opr_odes
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:bcs_min
use tlab_constants,only:bcs_max
use fdm_integral,only:fdm_integral_dt
use fdm_integral,only:fdm_int1_solve
private::wp
private::wi
private::bcs_min
private::bcs_max
private::fdm_integral_dt
private::fdm_int1_solve
contains
subroutine opr_ode2_factorize_dn_sing(nlines,fdmi,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi(1_8:2_8)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(in)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:int(int(size(fdmi(1_8)%lhs,dim=1,kind=8)*size(fdmi(1_8)%lhs,dim=2,kind=8),kind=4),kind=8),1_8:3_8)
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:3_8)
end
subroutine opr_ode2_factorize_nd_sing(nlines,fdmi,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi(1_8:2_8)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(in)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:int(int(size(fdmi(1_8)%lhs,dim=1,kind=8)*size(fdmi(1_8)%lhs,dim=2,kind=8),kind=4),kind=8),1_8:3_8)
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:3_8)
end
subroutine opr_ode2_factorize_nn_sing(nlines,fdmi,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi(1_8:2_8)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:int(int(size(fdmi(1_8)%lhs,dim=1,kind=8)*size(fdmi(1_8)%lhs,dim=2,kind=8),kind=4),kind=8),1_8:3_8)
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:3_8)
end
subroutine opr_ode2_factorize_dd_sing(nlines,fdmi,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi(1_8:2_8)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(in)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:size(fdmi(1_8)%lhs,dim=1,kind=8),1_8:4_8)
real(8),intent(inout)::wrk2d(1_8:int(nlines,kind=8),1_8:3_8)
end
subroutine opr_ode2_factorize_nn(nlines,fdmi,rhsi_b,rhsi_t,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(inout)::fdmi(1_8:2_8)
real(8),intent(in)::rhsi_b(:,:)
real(8),intent(in)::rhsi_t(:,:)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(in)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:3_8,1_8:size(fdmi(1_8)%lhs,dim=1,kind=8),1_8:2_8)
real(8),intent(inout)::wrk2d(1_8:int(max(nlines,3_4),kind=8),1_8:3_8)
end
subroutine opr_ode2_factorize_dd(nlines,fdmi,rhsi_b,rhsi_t,u,f,bcs,v,wrk1d,wrk2d)
integer(4)::nlines
type(fdm_integral_dt),intent(in)::fdmi(1_8:2_8)
real(8),intent(in)::rhsi_b(:,:)
real(8),intent(in)::rhsi_t(:,:)
real(8),intent(out)::u(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::f(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(in)::bcs(1_8:int(nlines,kind=8),1_8:2_8)
real(8),intent(out)::v(1_8:int(nlines,kind=8),1_8:size(fdmi(1_8)%lhs,dim=1,kind=8))
real(8),intent(inout)::wrk1d(1_8:2_8,1_8:size(fdmi(1_8)%lhs,dim=1,kind=8),1_8:2_8)
real(8),intent(inout)::wrk2d(1_8:int(max(nlines,2_4),kind=8),1_8:3_8)
end
end
