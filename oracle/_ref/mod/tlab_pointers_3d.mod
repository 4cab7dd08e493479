﻿!mod$ v1 sum:cc3d8fa3f58cf793
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_pointers_3d
use tlab_constants,only:wp
type::pointers3d_dt
sequence
character(32_4,1)::tag
real(8),pointer::field(:,:,:)
end type
real(8),pointer::u(:,:,:)
intrinsic::null
real(8),pointer::v(:,:,:)
real(8),pointer::w(:,:,:)
real(8),pointer::e(:,:,:)
real(8),pointer::rho(:,:,:)
real(8),pointer::p(:,:,:)
real(8),pointer::t(:,:,:)
real(8),pointer::vis(:,:,:)
real(8),pointer::p_q(:,:,:,:)
real(8),pointer::p_s(:,:,:,:)
real(8),pointer::p_wrk1d(:,:)
real(8),pointer::p_wrk2d(:,:,:)
real(8),pointer::p_wrk3d(:,:,:)
real(8),pointer::tmp1(:,:,:)
real(8),pointer::tmp2(:,:,:)
real(8),pointer::tmp3(:,:,:)
real(8),pointer::tmp4(:,:,:)
real(8),pointer::tmp5(:,:,:)
real(8),pointer::tmp6(:,:,:)
real(8),pointer::tmp7(:,:,:)
real(8),pointer::tmp8(:,:,:)
real(8),pointer::tmp9(:,:,:)
end
