﻿!mod$ v1 sum:7890a03f87a12397
!need$ 370470eb4a3adeb1 n tlab_constants
module fdm_com0_jacobian
use tlab_constants,only:wp
use tlab_constants,only:wi
contains
subroutine fdm_c0int6p_lhs(imax,a,b,c)
integer(4),intent(in)::imax
real(8),intent(out)::a(1_8:int(imax,kind=8))
real(8),intent(out)::b(1_8:int(imax,kind=8))
real(8),intent(out)::c(1_8:int(imax,kind=8))
end
subroutine fdm_c0intvp6p_rhs(imax,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
subroutine fdm_c0intpv6p_rhs(imax,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
subroutine fdm_c0intvp6_lhs(imaxp,a,b,c)
integer(4),intent(in)::imaxp
real(8),intent(out)::a(1_8:int(imaxp,kind=8))
real(8),intent(out)::b(1_8:int(imaxp,kind=8))
real(8),intent(out)::c(1_8:int(imaxp,kind=8))
end
subroutine fdm_c0intpv6_lhs(imax,a,b,c)
integer(4),intent(in)::imax
real(8),intent(out)::a(1_8:int(imax,kind=8))
real(8),intent(out)::b(1_8:int(imax,kind=8))
real(8),intent(out)::c(1_8:int(imax,kind=8))
end
subroutine fdm_c0intvp6_rhs(imax,imaxp,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::imaxp
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imaxp,kind=8))
end
subroutine fdm_c0intpv6_rhs(imax,imaxp,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::imaxp
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imaxp,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
subroutine fdm_c1int6p_lhs(imax,dx,a,b,c)
integer(4),intent(in)::imax
real(8),intent(in)::dx(1_8:int(imax,kind=8))
real(8),intent(out)::a(1_8:int(imax,kind=8))
real(8),intent(out)::b(1_8:int(imax,kind=8))
real(8),intent(out)::c(1_8:int(imax,kind=8))
end
subroutine fdm_c1intvp6p_rhs(imax,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
subroutine fdm_c1intpv6p_rhs(imax,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
subroutine fdm_c1intvp6_lhs(imaxp,dx,a,b,c)
integer(4),intent(in)::imaxp
real(8),intent(in)::dx(1_8:int(imaxp,kind=8))
real(8),intent(out)::a(1_8:int(imaxp,kind=8))
real(8),intent(out)::b(1_8:int(imaxp,kind=8))
real(8),intent(out)::c(1_8:int(imaxp,kind=8))
end
subroutine fdm_c1intpv6_lhs(imax,dx,a,b,c)
integer(4),intent(in)::imax
real(8),intent(in)::dx(1_8:int(imax,kind=8))
real(8),intent(out)::a(1_8:int(imax,kind=8))
real(8),intent(out)::b(1_8:int(imax,kind=8))
real(8),intent(out)::c(1_8:int(imax,kind=8))
end
subroutine fdm_c1intvp6_rhs(imax,imaxp,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::imaxp
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imaxp,kind=8))
end
subroutine fdm_c1intpv6_rhs(imax,imaxp,jkmax,u,d)
integer(4),intent(in)::imax
integer(4),intent(in)::imaxp
integer(4),intent(in)::jkmax
real(8),intent(in)::u(1_8:int(jkmax,kind=8),1_8:int(imaxp,kind=8))
real(8),intent(out)::d(1_8:int(jkmax,kind=8),1_8:int(imax,kind=8))
end
end
