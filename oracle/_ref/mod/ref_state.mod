﻿!mod$ v1 sum:a3162f7a7bbbaa31
!need$ 8d4bae2479538272 n fdm_integral
!need$ ff3fca9ebc58e858 n tlab_grid
!need$ 370470eb4a3adeb1 n tlab_constants
!need$ 06183c4da53c4dbe n fdm
module ref_state
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_grid,only:grid_dt
use fdm,only:fdm_dt
use fdm_integral,only:fdm_integral_dt
type(fdm_dt),target::gp(1_8:3_8)
type(grid_dt)::gr(1_8:3_8)
type(fdm_integral_dt)::fint(1_8:2_8)
type(fdm_integral_dt)::fint2
end
