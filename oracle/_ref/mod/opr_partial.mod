﻿!mod$ v1 sum:4774036a12c61918
!need$ 013c577ac3aa622b n tlab_arrays
!need$ 06183c4da53c4dbe n fdm
!need$ 9281856c4f7b499b n fdm_derivative
!need$ 54e7d2d00bf7ac8c n fdm_interpolate
!need$ 5dae8e9f6e7e08f8 n ibm_vars
!need$ 370470eb4a3adeb1 n tlab_constants
module opr_partial
use tlab_constants,only:wp
use tlab_constants,only:wi
use tlab_arrays,only:wrk2d
use tlab_arrays,only:wrk3d
use fdm,only:fdm_dt
use fdm_derivative,only:fdm_der1_solve
use fdm_derivative,only:fdm_der2_solve
use fdm_interpolate,only:fdm_interpol
use fdm_interpolate,only:fdm_interpol_der1
use ibm_vars,only:ibm_partial
private::wp
private::wi
private::wrk2d
private::wrk3d
private::fdm_dt
private::fdm_der1_solve
private::fdm_der2_solve
private::fdm_interpol
private::fdm_interpol_der1
private::ibm_partial
integer(4),parameter::opr_p1=1_4
integer(4),parameter::opr_p2=2_4
integer(4),parameter::opr_p2_p1=3_4
integer(4),parameter::opr_p1_int_vp=5_4
integer(4),parameter::opr_p1_int_pv=6_4
integer(4),parameter::opr_p0_int_vp=7_4
integer(4),parameter::opr_p0_int_pv=8_4
integer(4),parameter::opr_p0_ibm=9_4
private::opr_partial1_ibm
private::opr_ibm
contains
subroutine opr_partial_x(type,nx,ny,nz,bcs,g,u,result,tmp1)
integer(4),intent(in)::type
integer(4),intent(in)::nx
integer(4),intent(in)::ny
integer(4),intent(in)::nz
integer(4),intent(in)::bcs(:,:)
type(fdm_dt),intent(in)::g
real(8),intent(in),target::u(1_8:int(nx*ny*nz,kind=8))
real(8),intent(out),target::result(1_8:int(nx*ny*nz,kind=8))
real(8),intent(inout),optional,target::tmp1(1_8:int(nx*ny*nz,kind=8))
end
subroutine opr_partial_z(type,nx,ny,nz,bcs,g,u,result,tmp1)
integer(4),intent(in)::type
integer(4),intent(in)::nx
integer(4),intent(in)::ny
integer(4),intent(in)::nz
integer(4),intent(in)::bcs(:,:)
type(fdm_dt),intent(in)::g
real(8),intent(in),target::u(1_8:int(nx*ny*nz,kind=8))
real(8),intent(out),target::result(1_8:int(nx*ny*nz,kind=8))
real(8),intent(inout),optional,target::tmp1(1_8:int(nx*ny*nz,kind=8))
end
subroutine opr_partial_y(type,nx,ny,nz,bcs,g,u,result,tmp1)
integer(4),intent(in)::type
integer(4),intent(in)::nx
integer(4),intent(in)::ny
integer(4),intent(in)::nz
integer(4),intent(in)::bcs(:,:)
type(fdm_dt),intent(in)::g
real(8),intent(in),target::u(1_8:int(nx*ny*nz,kind=8))
real(8),intent(out),target::result(1_8:int(nx*ny*nz,kind=8))
real(8),intent(inout),optional,target::tmp1(1_8:int(nx*ny*nz,kind=8))
end
subroutine opr_partial1_ibm(nlines,ibc,g,u,result)
integer(4),intent(in)::nlines
integer(4),intent(in)::ibc
type(fdm_dt),intent(in)::g
real(8),intent(in)::u(1_8:int(nlines*g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines*g%size,kind=8))
end
subroutine opr_ibm(is,nlines,g,u,result)
integer(4),intent(in)::is
integer(4),intent(in)::nlines
type(fdm_dt),intent(in)::g
real(8),intent(in)::u(1_8:int(nlines*g%size,kind=8))
real(8),intent(out)::result(1_8:int(nlines*g%size,kind=8))
end
end
