﻿!mod$ v1 sum:fe23d2f9afd7370a
!need$ 7cca51c0634c6b29 n fdm_base
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_comx_direct
use tlab_constants,only:wp
use tlab_constants,only:wi
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
private::coef_c1n4
private::coef_c1n3_biased
private::coef_c1n6
private::coef_c2n4
private::coef_c2n3_biased
private::a2n6_coef
private::b2n6_coef
private::c2n6_coef
private::pip_o_pi
private::pipp_o_pi
private::d_coef
private::a1d
private::a2d
private::b1d
private::b2d
private::c1d
private::c2d
contains
subroutine fdm_c1n4_direct(nmax,x,lhs,rhs,nb_diag)
integer(4),intent(in)::nmax
real(8),intent(in)::x(1_8:int(nmax,kind=8))
real(8),intent(out)::lhs(1_8:int(nmax,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nmax,kind=8),1_8:3_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
end
subroutine fdm_c1n6_direct(nmax,x,lhs,rhs,nb_diag)
integer(4),intent(in)::nmax
real(8),intent(in)::x(1_8:int(nmax,kind=8))
real(8),intent(out)::lhs(1_8:int(nmax,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nmax,kind=8),1_8:5_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
end
function coef_c1n4(x,i) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
real(8)::coef(1_8:6_8)
end
function coef_c1n3_biased(x,i,backwards) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
logical(4),intent(in),optional::backwards
real(8)::coef(1_8:5_8)
end
function coef_c1n6(x,i) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
real(8)::coef(1_8:8_8)
end
subroutine fdm_c2n6_direct(nmax,x,lhs,rhs,nb_diag)
integer(4),intent(in)::nmax
real(8),intent(in)::x(1_8:int(nmax,kind=8))
real(8),intent(out)::lhs(1_8:int(nmax,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nmax,kind=8),1_8:5_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
end
subroutine fdm_c2n4_direct(nmax,x,lhs,rhs,nb_diag)
integer(4),intent(in)::nmax
real(8),intent(in)::x(1_8:int(nmax,kind=8))
real(8),intent(out)::lhs(1_8:int(nmax,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nmax,kind=8),1_8:5_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
end
function coef_c2n4(x,i) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
real(8)::coef(1_8:6_8)
end
function coef_c2n3_biased(x,i,backwards) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
logical(4),intent(in),optional::backwards
real(8)::coef(1_8:6_8)
end
function a2n6_coef(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function b2n6_coef(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function c2n6_coef(x,j,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
real(8)::f
end
function pip_o_pi(x,j,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
real(8)::f
end
function pipp_o_pi(x,j,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
real(8)::f
end
function d_coef(x,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
real(8)::f
end
function a1d(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function a2d(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function b1d(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function b2d(x,im,ip,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::im
integer(4),intent(in)::ip
integer(4),intent(in)::i
real(8)::f
end
function c1d(x,j,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
real(8)::f
end
function c2d(x,j,i) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
real(8)::f
end
end
