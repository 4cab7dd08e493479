﻿!mod$ v1 sum:c314556627c1bfdb
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_com1_jacobian
use tlab_constants,only:wp
use tlab_constants,only:wi
private::wp
private::wi
logical(4),private::periodic_loc
private::create_system_1der
contains
subroutine fdm_c1n4_jacobian(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8))
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:3_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine fdm_c1n6_jacobian(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8))
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:3_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:5_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine fdm_c1n6_jacobian_penta(nx,dx,lhs,rhs,nb_diag,coef,periodic)
integer(4),intent(in)::nx
real(8),intent(in)::dx(1_8:int(nx,kind=8))
real(8),intent(out)::lhs(1_8:int(nx,kind=8),1_8:5_8)
real(8),intent(out)::rhs(1_8:int(nx,kind=8),1_8:7_8)
integer(4),intent(out)::nb_diag(1_8:2_8)
real(8),intent(out)::coef(1_8:5_8)
logical(4),intent(in),optional::periodic
end
subroutine create_system_1der(dx,lhs,rhs,coef_int,coef_bc1,coef_bc2,coef_bc3)
real(8),intent(in)::dx(:)
real(8),intent(out)::lhs(:,:)
real(8),intent(out)::rhs(:,:)
real(8),intent(in)::coef_int(1_8:5_8)
real(8),intent(in),optional::coef_bc1(1_8:6_8)
real(8),intent(in),optional::coef_bc2(1_8:6_8)
real(8),intent(in),optional::coef_bc3(1_8:8_8)
end
end
