﻿!mod$ v1 sum:7cca51c0634c6b29
!need$ aeab807d21fdaebf n tlab_workflow
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_base
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
use tlab_constants,only:efile
use tlab_workflow,only:tlab_write_ascii
use tlab_workflow,only:tlab_stop
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
private::efile
private::tlab_write_ascii
private::tlab_stop
contains
function pi(x,j,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::idx(:)
real(8)::f
end
function pi_p(x,j,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::idx(:)
real(8)::f
end
function pi_pp_3(x,j,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::idx(:)
real(8)::f
end
function lag(x,j,i,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
integer(4),intent(in)::idx(:)
real(8)::f
end
function lag_p(x,j,i,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
integer(4),intent(in)::idx(:)
real(8)::f
end
function lag_pp_3(x,j,i,idx) result(f)
real(8),intent(in)::x(:)
integer(4),intent(in)::j
integer(4),intent(in)::i
integer(4),intent(in)::idx(:)
real(8)::f
end
function coef_e1n3_biased(x,i,backwards) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
logical(4),intent(in),optional::backwards
real(8)::coef(1_8:4_8)
end
function coef_e1n2_biased(x,i,backwards) result(coef)
real(8),intent(in)::x(:)
integer(4),intent(in)::i
logical(4),intent(in),optional::backwards
real(8)::coef(1_8:3_8)
end
subroutine fdm_bcs_neumann(ibc,lhs,rhs,rhs_b,rhs_t)
integer(4),intent(in)::ibc
real(8),intent(inout)::lhs(:,:)
real(8),intent(in)::rhs(:,:)
real(8),intent(inout)::rhs_b(1_8:,0_8:)
real(8),intent(inout)::rhs_t(0_8:,1_8:)
end
subroutine fdm_bcs_reduce(ibc,lhs,rhs,rhs_b,rhs_t)
integer(4),intent(in)::ibc
real(8),intent(inout)::lhs(:,:)
real(8),intent(in),optional::rhs(:,:)
real(8),intent(out),optional::rhs_b(1_8:,0_8:)
real(8),intent(out),optional::rhs_t(0_8:,1_8:)
end
end
