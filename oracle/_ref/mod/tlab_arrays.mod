﻿!mod$ v1 sum:013c577ac3aa622b
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_arrays
use tlab_constants,only:wp
real(8),allocatable,target::q(:,:)
real(8),allocatable,target::s(:,:)
real(8),allocatable,target::txc(:,:)
real(8),allocatable,target::wrk1d(:,:)
real(8),allocatable,target::wrk2d(:,:)
real(8),allocatable,target::wrk3d(:)
end
