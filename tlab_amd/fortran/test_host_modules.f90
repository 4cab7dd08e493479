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

module DNS_ARRAYS
    use TLab_Constants, only: wp
    implicit none
    save
    real(wp), pointer, contiguous, public :: hq(:, :) => null()     ! Right-hand sides Eulerian fields (device; allocatable in the reference)
    real(wp), pointer, contiguous, public :: hs(:, :) => null()
end module DNS_ARRAYS

