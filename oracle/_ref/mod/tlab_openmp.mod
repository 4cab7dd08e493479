﻿!mod$ v1 sum:2bd709c189bd3652
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_openmp
use tlab_constants,only:wi
private::wi
integer(4)::tlab_omp_numthreads
integer(4)::tlab_omp_error
contains
subroutine tlab_omp_partition(len,omp_srt,omp_end,omp_siz)
integer(4),intent(in)::len
integer(4),intent(inout)::omp_srt
integer(4),intent(inout)::omp_end
integer(4),intent(inout)::omp_siz
end
end
