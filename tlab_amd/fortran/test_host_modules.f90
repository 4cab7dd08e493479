!########################################################################
! Host stand-ins for the test mini-driver (test_rk_driver.f90) ONLY: the two host modules of the reference that the drop-in reads from
! and that cannot be compiled from the reference here (their files drag in thermodynamics, particles, statistics ... and, through
! OPR_Fourier, the FFTW header the image lacks).  Same module and variable names as the reference, nothing else; in a Tlab build the
! reference's own modules take their place.
!########################################################################
module NavierStokes
    use TLab_Constants, only: wp, MAX_VARS
    implicit none
    real(wp), public :: visc, schmidt(MAX_VARS)                     ! molecular transport (physics/navierstokes.f90:25)
    integer, parameter :: DNS_EQNS_INCOMPRESSIBLE = 3, DNS_EQNS_ANELASTIC = 4, EQNS_CONVECTIVE = 3
    integer :: nse_eqns = DNS_EQNS_INCOMPRESSIBLE, nse_advection = EQNS_CONVECTIVE
end module NavierStokes

module DNS_LOCAL
    implicit none
    logical :: remove_divergence = .true.                           ! Remove residual divergence every time step (tools/dns/dns_local.f90:36)
end module DNS_LOCAL

module DNS_ARRAYS
    use TLab_Constants, only: wp
    implicit none
    save
    real(wp), pointer, contiguous, public :: hq(:, :) => null()     ! Right-hand sides Eulerian fields (device; allocatable in the reference)
    real(wp), pointer, contiguous, public :: hs(:, :) => null()
end module DNS_ARRAYS


module TLabMPI_VARS                                                  ! base/tlab_mpi_vars.f90:6-15 (the decomposition; set by TLabMPI_Initialize)
    use TLab_Constants, only: wi
    implicit none
    integer :: ims_pro = 0, ims_npro = 1
    integer :: ims_pro_i = 0, ims_npro_i = 1, ims_pro_k = 0, ims_npro_k = 1
    integer(wi) :: ims_offset_i = 0, ims_offset_k = 0
end module TLabMPI_VARS
