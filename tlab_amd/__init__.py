"""tlab_amd -- MI355X-native implementation of Tlab's Navier-Stokes RHS hot path.

The compute path is hand-written HIP (tlab_amd/csrc) behind the C ABI of include/tlab_amd.h, loaded here with
ctypes.  There is NO CPU fallback: importing works without a GPU (plan construction is host code), but every
operator raises TlabError unless tlab_init() found an MI355X.
"""
from .lib import TlabError, load, lib_path  # noqa: F401
from .operators import (  # noqa: F401
    FdmPlan, init, sync,
    OPR_P1, OPR_P2, OPR_P2_P1, OPR_B_SELF, OPR_B_U_IN, OPR_P1_INT_VP, OPR_P1_INT_PV, OPR_P0_INT_VP, OPR_P0_INT_PV,
    BCS_DD, BCS_ND, BCS_DN, BCS_NN,
    FDM_COM4_JACOBIAN, FDM_COM6_JACOBIAN_PENTA, FDM_COM6_JACOBIAN, FDM_COM6_JACOBIAN_HYPER, FDM_COM6_DIRECT, FDM_COM4_DIRECT,
    OPR_Partial_X, OPR_Partial_Y, OPR_Partial_Z,
    OPR_Burgers_X, OPR_Burgers_Y, OPR_Burgers_Z, Filter, OPR_FILTER_1D, set_dealiasing,
    TLab_Transpose, PoissonPlan, OPR_Poisson, OPR_Helmholtz, poisson_set_exact, BOUNDARY_BCS_NEUMANN_Y,
)
