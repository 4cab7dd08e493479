// Host-side (init-time) construction of Tlab's compact finite-difference plans.
// C++ restatement of fdm/fdm.f90, fdm_derivative.f90, fdm_com{1,2}_jacobian.f90, fdm_base.f90 (FDM_Bcs_Neumann)
// and utils/linear3.f90 (TRIDFS/TRIDPFS).  Tables keep the reference's column-major (row, diagonal) layout:
// T(i, k) == t[i + n*k] with 0-based i, k, so they can be compared with, or taken from, the Fortran host.
#pragma once
#include <vector>

namespace tlab {

enum { BCS_PERIODIC = -1, BCS_DD = 0, BCS_ND = 1, BCS_DN = 2, BCS_NN = 3 };
enum { FDM_COM4_JACOBIAN = 4, FDM_COM6_JACOBIAN_PENTA = 5, FDM_COM6_JACOBIAN = 6, FDM_COM6_JACOBIAN_HYPER = 7,
       FDM_COM6_DIRECT = 16, FDM_COM4_DIRECT = 17 };

// type(fdm_derivative_dt), fdm/fdm_derivative.f90:16-29
struct DerTables {
    int mode_fdm = 0;
    int n = 0;
    bool periodic = false;
    bool need_1der = false;
    bool direct = false;               // FDM_COM4_DIRECT / FDM_COM6_DIRECT: per-row RHS (MatMul_5d); tables come from the host (from_arrays)
    int ndl = 0, ndr = 0;              // nb_diag(1), nb_diag(2)
    std::vector<double> lhs;           // (n,5)
    std::vector<double> rhs;           // (n,7) first derivative, (n,12) second derivative
    int rhs_cols = 0;
    std::vector<double> mwn;           // (n)
    std::vector<double> lu;            // (n, lu_cols)
    int lu_cols = 0;
    double rhs_b[4 * 8];               // rhs_b(4,0:7): [ (j-1) + 4*c ]
    double rhs_t[5 * 7];               // rhs_t(0:4,7): [ r + 5*(c-1) ]
    DerTables();
};

// type(fdm_dt), fdm/fdm.f90:14-29
struct FdmTables {
    int n = 0;
    bool periodic = false, uniform = false;
    std::vector<double> nodes;         // (n)
    std::vector<double> jac;           // (n,3)
    DerTables der1, der2;
    // horizontal pressure staggering (TLab_WorkFlow::stagger_on; fdm.f90:236-248, fdm_interpolate.f90): g%intl of a periodic direction
    bool stagger = false;
    std::vector<double> lu0i, lu1i;    // (n,5) LU of the interpolation / interpolatory first-derivative systems (TRIDPFS)
};

// FDM_Interpol_Initialize (fdm/fdm_interpolate.f90:33-96): lu0i, lu1i from FDM_C0INT6P_LHS / FDM_C1INT6P_LHS (fdm_com0_jacobian.f90:29-44, 287-320);
// replace_mwn: der1.mwn becomes the interpolatory modified wavenumber (:74-93), as FDM_CreatePlan does with stagger_on
void interpol_initialize(FdmTables &g, bool replace_mwn);

// utils/linear3.f90:29-51, :269-316
void tridfs(int nmax, double *a, double *b, double *c);
void tridpfs(int nmax, double *a, double *b, double *c, double *d, double *e);
// utils/linear3.f90:56-150, :321-442 for a single line (used for the Jacobian at plan creation, fdm.f90:201,224)
void tridss1(int nmax, const double *a, const double *b, const double *c, double *f);
// utils/linear5.f90:156-203 PENTADFS2, :273-347 PENTADPFS (f, g: the two Woodbury vectors), :207-244 PENTADSS2 for a single line
void pentadfs2(int nmax, double *a, double *b, double *c, double *d, double *e);
void pentadpfs(int nmax, double *a, double *b, double *c, double *d, double *e, double *f, double *g);
void pentadss2_1(int nmax, const double *a, const double *b, const double *c, const double *d, const double *e, double *f);

// fdm/fdm_base.f90:194-300
void fdm_bcs_neumann(int ibc, int n, int ndl, double *lhs /*(n,ndl)*/, int ndr, const double *rhs /*(n,>=ndr) ld n*/,
                     double *rhs_b, double *rhs_t);

// fdm/fdm_derivative.f90:63-142, :282-333 (CreateSystem + LU)
// penta_bc1_ext: the out-of-bounds coefficient the reference reads for the wall row of CompactJacobian6Penta (fdm_com1_jacobian.f90:237 with
// icmax = 4: coef_bc1(7) of a 6-element array; 1/6 in the flang-built reference, the twin of hyper_bc1_ext)
void der1_initialize(DerTables &g, int n, const double *dx, bool periodic, const int *bcs_cases, int ncases, double penta_bc1_ext = 1.0 / 6.0);
void der1_factorize(DerTables &g, const int *bcs_cases, int ncases);      // the LU part of der1_initialize, for host-built lhs / rhs tables
void der2_initialize(DerTables &g, int n, const double *dx2 /*(n,2)*/, bool periodic, bool uniform, double hyper_bc1_ext);

// RHS product B*u for one line, reference operation order (fdm/fdm_matmul.f90); used at plan creation only
void der1_matmul1(const DerTables &g, int ibc, const double *u, double *f);
void der2_matmul1(const DerTables &g, int ibc, const double *u, double *f);

// fdm/fdm.f90:143-252
void fdm_create_plan(FdmTables &g, int n, const double *nodes, bool periodic, bool uniform, int mode1, int mode2,
                     double hyper_bc1_ext, double penta_bc1_ext = 1.0 / 6.0);

}  // namespace tlab
