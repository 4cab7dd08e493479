﻿!mod$ v1 sum:1a10e0d3339a6934
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_pointers
use tlab_constants,only:wp
type::pointers_dt
sequence
character(32_4,1)::tag
real(8),pointer::field(:)
end type
real(8),pointer::u(:)
intrinsic::null
real(8),pointer::v(:)
real(8),pointer::w(:)
real(8),pointer::e(:)
real(8),pointer::rho(:)
real(8),pointer::p(:)
real(8),pointer::t(:)
real(8),pointer::vis(:)
real(8),pointer::tmp1(:)
real(8),pointer::tmp2(:)
real(8),pointer::tmp3(:)
real(8),pointer::tmp4(:)
real(8),pointer::tmp5(:)
real(8),pointer::tmp6(:)
real(8),pointer::tmp7(:)
real(8),pointer::tmp8(:)
real(8),pointer::tmp9(:)
end
