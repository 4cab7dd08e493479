# base/tlab_memory.f90 of the reference -> the same three modules with the field-sized arrays in HBM (INTEGRATION.md section 3).
# Applied at build time to $(REF)/src/base/tlab_memory.f90 where it lies (tlab_amd/fortran/Makefile); nothing of that file is kept in this repo.
#
# 1. module TLab_Arrays: q, s, txc, wrk3d become POINTER, CONTIGUOUS (an ALLOCATABLE cannot be given foreign memory); wrk1d, wrk2d -- coefficient
#    scratch of the unchanged host code -- stay what they are.
/^module TLab_Arrays/,/^end module TLab_Arrays/{
s/real(wp), allocatable :: \(q\|s\|txc\)(:, :)/real(wp), pointer, contiguous :: \1(:, :) => null()/
s/real(wp), allocatable :: wrk3d(:)/real(wp), pointer, contiguous :: wrk3d(:) => null()/
s/target q, s, txc, wrk1d, wrk2d, wrk3d/target wrk1d, wrk2d/
}
# 2. module TLab_Memory: TLab_Allocate_Real becomes generic -- the reference's routine for ALLOCATABLE (host) arrays under a specific name, and the
#    device routine of tlab_allocate_real_device.inc for POINTER arrays (tlab_malloc + c_f_pointer); callers keep their call lines.
/^module TLab_Memory/,/^end module TLab_Memory/{
s/subroutine TLab_Allocate_Real(C_FILE_LOC, a, dims, s)/subroutine TLab_Allocate_Real_Host(C_FILE_LOC, a, dims, s)/
s/end subroutine TLab_Allocate_Real$/end subroutine TLab_Allocate_Real_Host/
/^contains/{
i\
    interface TLab_Allocate_Real\
        module procedure TLab_Allocate_Real_Host, TLab_Allocate_Real_Device\
    end interface TLab_Allocate_Real\
    public :: TLab_AMD_Reset_Pointers
a\
#include "tlab_allocate_real_device.inc"
}
# 3. the pointer-remapping helpers ask associated(), not allocated(), of the arrays that are pointers now
s/allocated(\(q\|s\|wrk3d\))/associated(\1)/g
}
