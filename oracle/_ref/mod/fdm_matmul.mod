﻿!mod$ v1 sum:ee3e4b05f7cc2dad
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_matmul
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:bcs_dd
use tlab_constants,only:bcs_dn
use tlab_constants,only:bcs_nd
use tlab_constants,only:bcs_nn
use tlab_constants,only:bcs_none
use tlab_constants,only:bcs_min
use tlab_constants,only:bcs_max
use tlab_constants,only:bcs_both
use tlab_constants,only:bcs_periodic
private::wp
private::wi
private::bcs_dd
private::bcs_dn
private::bcs_nd
private::bcs_nn
private::bcs_none
private::bcs_min
private::bcs_max
private::bcs_both
private::bcs_periodic
contains
subroutine matmul_3d(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_3d_add(rhs,u,f)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
end
subroutine matmul_3d_antisym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_3d_sym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_5d(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_5d_add(rhs,u,f)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
end
subroutine matmul_5d_antisym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_5d_sym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_7d_antisym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
subroutine matmul_7d_sym(rhs,u,f,ibc,rhs_b,rhs_t,bcs_b,bcs_t)
real(8),intent(in)::rhs(:,:)
real(8),intent(in)::u(:,:)
real(8),intent(out)::f(:,:)
integer(4),intent(in)::ibc
real(8),intent(in),optional::rhs_b(1_8:,0_8:)
real(8),intent(in),optional::rhs_t(0_8:,1_8:)
real(8),intent(out),optional::bcs_b(:)
real(8),intent(out),optional::bcs_t(:)
end
end
