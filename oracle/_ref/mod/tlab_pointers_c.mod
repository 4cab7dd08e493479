﻿!mod$ v1 sum:ea11635038e5e66a
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_pointers_c
use tlab_constants,only:wp
complex(8),pointer::c_wrk1d(:,:)
intrinsic::null
complex(8),pointer::c_wrk3d(:,:)
end
