﻿!mod$ v1 sum:8e3ed643e1b51b90
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_com2_jacobian
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:pi_wp
private::wp
private::wi
private::pi_wp
logical(4),private::periodic_loc
private::create_system_2der
contains
subroutine fdm_c2n4_jacobian(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8),1_8:2_8)
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:8_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine fdm_c2n6_jacobian(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8),1_8:2_8)
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:8_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine fdm_c2n6_hyper_jacobian(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8),1_8:2_8)
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:10_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine create_system_2der(dx,lhs,rhs,rhs_d1,coef_int,coef_bc1,coef_bc2,coef_bc3)
real(8),intent(in)::dx(:,:)
real(8),intent(out)::lhs(:,:)
real(8),intent(out)::rhs(:,:)
real(8),intent(out)::rhs_d1(:,:)
real(8),intent(in)::coef_int(1_8:5_8)
real(8),intent(in),optional::coef_bc1(1_8:6_8)
real(8),intent(in),optional::coef_bc2(1_8:6_8)
real(8),intent(in),optional::coef_bc3(1_8:8_8)
end
end
