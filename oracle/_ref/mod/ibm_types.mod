﻿!mod$ v1 sum:227de5ae189e9347
!need$ 370470eb4a3adeb1 n tlab_constants
module ibm_types
use tlab_constants,only:wi
type::ibm_geo_dt
sequence
character(32_4,1)::name
integer(4)::number
integer(4)::height
integer(4)::width
integer(4)::hill_slope
logical(4)::mirrored
end type
end
