!########################################################################
! Drop-in replacement of the initialisation entry of module OPR_Fourier (operators/opr_fourier.f90:54-208).
! The reference creates its FFTW plans there; on the device the transforms belong to the Poisson plan (rocFFT along x, own strided
! transform along z: tlab_poisson_plan_create), so OPR_Fourier_Initialize() has nothing left to do -- it exists so that
! dns_main.f90:139 compiles and links unchanged.  The transforms themselves are reachable through the C ABI
! (tlab_poisson_fft_x / tlab_poisson_fft_z); spectra / convolution helpers (OPR_Fourier_F/B/..., :437-797) are out of scope.
!########################################################################
module OPR_Fourier
    implicit none
    private
    public :: OPR_Fourier_Initialize
contains
    subroutine OPR_Fourier_Initialize()
    end subroutine OPR_Fourier_Initialize
end module OPR_Fourier
