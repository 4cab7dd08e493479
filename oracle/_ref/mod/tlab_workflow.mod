﻿!mod$ v1 sum:aeab807d21fdaebf
!need$ 370470eb4a3adeb1 n tlab_constants
module tlab_workflow
use tlab_constants,only:sp
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_constants,only:longi
use tlab_constants,only:lfile
use tlab_constants,only:efile
private::sp
private::wp
private::wi
private::longi
private::lfile
private::efile
character(128_8,1),private::line
integer(4)::imode_verbosity
integer(4)::imode_sim
logical(4)::flow_on
logical(4)::scal_on
logical(4)::fourier_on
logical(4)::stagger_on
contains
subroutine tlab_start()
end
subroutine tlab_stop(error_code)
integer(4),intent(in)::error_code
end
subroutine tlab_write_ascii(file,lineloc,flag_all)
character(*,1),intent(in)::file
character(*,1),intent(in)::lineloc
logical(4),intent(in),optional::flag_all
end
end
