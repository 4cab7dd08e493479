﻿!mod$ v1 sum:5dae8e9f6e7e08f8
!need$ 227de5ae189e9347 n ibm_types
!need$ 370470eb4a3adeb1 n tlab_constants
module ibm_vars
use tlab_constants,only:max_vars
use tlab_constants,only:wp
use tlab_constants,only:wi
use ibm_types,only:ibm_geo_dt
integer(4)::imode_ibm
integer(4)::imode_ibm_scal
real(8),allocatable,target::eps(:)
real(8),allocatable,target::epsp(:)
integer(4),allocatable::nobi(:)
integer(4),allocatable::nobj(:)
integer(4),allocatable::nobk(:)
integer(4),allocatable::nobi_b(:)
integer(4),allocatable::nobj_b(:)
integer(4),allocatable::nobk_b(:)
integer(4),allocatable::nobi_e(:)
integer(4),allocatable::nobj_e(:)
integer(4),allocatable::nobk_e(:)
integer(4)::nobi_max
integer(4)::nobj_max
integer(4)::nobk_max
integer(4)::nob_max
real(8)::max_height_objlo
real(8)::max_height_objup
integer(4),allocatable::ibm_case_x(:)
integer(4),allocatable::ibm_case_y(:)
integer(4),allocatable::ibm_case_z(:)
real(8),allocatable,target::fld_ibm(:)
real(8)::ibmscaljmin(1_8:10_8)
real(8)::ibmscaljmax(1_8:10_8)
real(8),allocatable::xa(:)
real(8),allocatable::xb(:)
real(8),allocatable::ya(:)
real(8),allocatable::yb(:)
real(8),allocatable::dy(:)
real(8),allocatable::facu(:)
real(8),allocatable::facl(:)
real(8),allocatable::gamma_0(:)
real(8),allocatable::gamma_1(:)
real(8),allocatable::scal_bcs(:,:)
logical(4)::ibm_burgers
logical(4)::ibm_partial
logical(4)::ibm_objup
logical(4)::ibm_restart
integer(4)::nflu
integer(4)::ibm_io
integer(4)::isize_nobi
integer(4)::isize_nobj
integer(4)::isize_nobk
integer(4)::isize_nobi_be
integer(4)::isize_nobj_be
integer(4)::isize_nobk_be
integer(4)::nspl
integer(4)::isize_wrk1d_ibm
logical(4)::ims_pro_ibm_x
logical(4)::ims_pro_ibm_y
logical(4)::ims_pro_ibm_z
type(ibm_geo_dt)::ibm_geo
character(32_4,1),parameter::eps_name="eps0.1                          "
character(32_4,1),parameter::epsp_name="epsp0.1                         "
character(32_4,1),parameter::eps_name_real="eps0                            "
character(32_4,1),parameter::epsp_name_real="epsp0                           "
integer(4),parameter::ibm_io_real=1_4
integer(4),parameter::ibm_io_int=2_4
integer(4),parameter::ibm_io_bit=3_4
end
