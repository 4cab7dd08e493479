/* tlab_amd.h -- C ABI of the MI355X-native Tlab Navier-Stokes RHS operators.
 *
 * This is the drop-in boundary.  Tlab has no FFI: its boundary is Fortran module procedures
 * (SURVEY.md 8b).  Each entry point below is what an ISO_C_BINDING interface in the same-named
 * Fortran module binds to (tlab_amd/fortran/); the reference interface it replaces
 * is cited as path:line relative to the reference's src/.
 *
 * Conventions
 *  - every function returns 0 on success, a negative TLAB_E* code otherwise; tlab_last_error() gives text.
 *    (reference: operators stop the program through TLab_Stop, base/tlab_workflow.f90:105; the Fortran shim
 *    maps non-zero to TLab_Write_ASCII(efile, ...) + TLab_Stop(DNS_ERROR_UNDEVELOP).)
 *  - all field pointers are DEVICE pointers (hipMalloc / tlab_malloc / torch CUDA tensors), fp64,
 *    x fastest: index = i + nx*(j + ny*k), exactly the host layout of the reference
 *    (operators/opr_partial.f90:40: u(nx*ny*nz)).
 *  - coefficient tables passed IN from a host (tlab_fdm_plan_create_from_arrays) are HOST pointers,
 *    column-major as Fortran stores lhs(n,ndl), rhs(n,ndr).
 *  - kernels are enqueued on the stream set by tlab_set_stream (default: the null stream); nothing
 *    synchronises except tlab_sync() and the *_get / memcpy helpers.
 *  - one in-flight operator per process, like the reference (module state).
 */
#ifndef TLAB_AMD_H
#define TLAB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------- */
#define TLAB_OK 0
#define TLAB_EINVAL (-1)      /* bad argument (size, direction, type, aliasing)                  */
#define TLAB_EUNSUPPORTED (-2)/* valid in the reference, not built here (falls to CPU in Fortran) */
#define TLAB_EHIP (-3)        /* HIP / rocFFT / RCCL runtime error                               */
#define TLAB_ENOMEM (-4)

/* ---- constants mirrored from the reference ------------------------------------------------ */
/* operators/opr_partial.f90:19-26 */
#define TLAB_OPR_P1 1
#define TLAB_OPR_P2 2
#define TLAB_OPR_P2_P1 3
#define TLAB_OPR_P1_INT_VP 5   /* interpolatory first derivative velocity -> pressure grid (staggering; periodic x, z) */
#define TLAB_OPR_P1_INT_PV 6
#define TLAB_OPR_P0_INT_VP 7   /* interpolation velocity -> pressure grid */
#define TLAB_OPR_P0_INT_PV 8
/* physics/opr_burgers.f90:29-30 */
#define TLAB_OPR_B_SELF 0
#define TLAB_OPR_B_U_IN 1
/* base/tlab_constants.f90:63-66 */
#define TLAB_BCS_DD 0
#define TLAB_BCS_ND 1
#define TLAB_BCS_DN 2
#define TLAB_BCS_NN 3
/* fdm/fdm_derivative.f90:51-58 */
#define TLAB_FDM_COM4_JACOBIAN 4
#define TLAB_FDM_COM6_JACOBIAN_PENTA 5
#define TLAB_FDM_COM6_JACOBIAN 6
#define TLAB_FDM_COM6_JACOBIAN_HYPER 7

/* ---- runtime ------------------------------------------------------------------------------ */
int tlab_init(int device);                 /* hipSetDevice + library state; idempotent            */
int tlab_device_count(void);               /* visible GPUs (hipGetDeviceCount; 0 if none): a multi-rank host selects mod(local rank, count) */
int tlab_finalize(void);
const char *tlab_last_error(void);
int tlab_set_stream(void *hip_stream);     /* hipStream_t; NULL = default stream                  */
int tlab_sync(void);                       /* hipStreamSynchronize on the current stream          */
/* allocation hook: replaces the allocate() inside TLab_Allocate_Real (base/tlab_memory.f90:306-331) so that
 * q, s, txc, wrk3d, hq, hs live in HBM; the Fortran side wraps the pointer with c_f_pointer. */
int tlab_malloc(void **p, size_t bytes);
int tlab_free(void *p);
int tlab_memcpy_h2d(void *dst, const void *src, size_t bytes);
int tlab_memcpy_d2h(void *dst, const void *src, size_t bytes);

/* ---- FDM plans: type(fdm_dt) of fdm/fdm.f90:14-29 + type(fdm_derivative_dt) fdm_derivative.f90:16-29 ---- */
typedef struct tlab_fdm_plan *tlab_fdm_plan_t;

/* Replaces FDM_CreatePlan (fdm/fdm.f90:143-252): Jacobians from the node positions, compact schemes
 * scheme1 (first derivative: TLAB_FDM_COM4_JACOBIAN | TLAB_FDM_COM6_JACOBIAN | TLAB_FDM_COM6_JACOBIAN_PENTA) and scheme2 (second derivative:
 * COM4_JACOBIAN | COM6_JACOBIAN | COM6_JACOBIAN_HYPER), Neumann variants (FDM_Bcs_Neumann, fdm_base.f90:194),
 * factorizations.  nodes: HOST pointer, n doubles.
 * hyper_bc1_ext: value of the out-of-bounds coefficient the reference reads for the C2N6-Hyper wall row
 * (fdm_com2_jacobian.f90:224 with icmax=4; DESIGN.md "reference defects"): pass 0.1 to reproduce the flang-built
 * reference bit for bit, 0.0 for the consistent closure. */
int tlab_fdm_plan_create(tlab_fdm_plan_t *out, int n, const double *nodes, int periodic, int uniform,
                         int scheme1, int scheme2, double hyper_bc1_ext);

/* Same plan from coefficient tables the (unchanged) Fortran host already built in FDM_Initialize:
 * lhs1 = g%der1%lhs(n,1:ndl1), rhs1 = g%der1%rhs(n,1:ndr1), lhs2 = g%der2%lhs(n,1:ndl2),
 * rhs2 = g%der2%rhs(n,1:ndr2+ndl2) (the last ndl2 columns are the Jacobian-correction diagonals,
 * fdm_derivative.f90:356,437-440).  ndl2 must be 3; ndl1 = 3, or 5 with ndr1 = 7 (CompactJacobian6Penta: the library factorizes with
 * PENTADFS2 / PENTADPFS as fdm_derivative.f90:90-119 does; the derivative then runs one line per thread, no chunked fast path). */
int tlab_fdm_plan_create_from_arrays(tlab_fdm_plan_t *out, int n, int periodic, int need_1der,
                                     int ndl1, int ndr1, const double *lhs1, const double *rhs1,
                                     int ndl2, int ndr2, const double *lhs2, const double *rhs2);
/* Remaining members of type(fdm_dt) a host-built plan may carry (any pointer may be NULL): der1%mwn(n), der2%mwn(n) (periodic directions:
 * OPR_Poisson needs der1%mwn of x and z, opr_elliptic.f90:199-203), jac(n,3) (TIME_COURANT, time.f90:148), nodes(n). */
int tlab_fdm_plan_set_aux(tlab_fdm_plan_t plan, const double *mwn1, const double *mwn2, const double *jac, const double *nodes);
/* mode_fdm of the two derivatives of a host-built plan (fdm_derivative.f90:52-58).  FDM_COM6_DIRECT (16) / FDM_COM4_DIRECT (17) as mode2
 * make rhs2 a per-row pentadiagonal operator (MatMul_5d) -- [Main] SpaceOrder2 = CompactDirect6 of examples/Case81-93; the coefficient
 * tables of fdm_comx_direct.f90 are the host's (tlab_fdm_plan_create_from_arrays).  As mode1 (SpaceOrder1 = CompactDirect4 / CompactDirect6,
 * FDM_C1N4_Direct / FDM_C1N6_Direct: 3 / 5 per-row RHS diagonals, MatMul_3d / MatMul_5d with the Neumann-reduced rows of FDM_Bcs_Neumann)
 * in non-periodic directions (periodic ones fall back to the Jacobian schemes, fdm.f90:155-158). */
int tlab_fdm_plan_set_scheme(tlab_fdm_plan_t plan, int mode1, int mode2);
/* [Staggering] StaggerHorizontalPressure = yes (TLab_WorkFlow::stagger_on): FDM_CreatePlan then gives a periodic direction the interpolation
 * tables g%intl (FDM_Interpol_Initialize, fdm/fdm_interpolate.f90:33-96) and replaces g%der1%mwn by the interpolatory modified wavenumbers
 * (fdm.f90:236-248).  mode 1: both (plans of tlab_fdm_plan_create); 2: tables only (host-built plans whose mwn1 already is the host's); 0: off.
 * With it tlab_opr_partial takes the types TLAB_OPR_P0/P1_INT_VP/PV along x and z, a Poisson plan built on such x / z plans has the one singular
 * mode (1,1) (opr_elliptic.f90:144-146) and the RHS driver takes the staggered branch (rhs_global_incompressible_1.f90:216-226, 266-273, 307-317). */
int tlab_fdm_plan_set_stagger(tlab_fdm_plan_t plan, int mode);
int tlab_fdm_plan_destroy(tlab_fdm_plan_t p);

/* read back plan tables (HOST buffer, column-major like the reference) for parity tests. which:
 *  1 der1%lhs(n,5) 2 der1%rhs(n,7) 3 der1%lu(n,5|20) 4 der1%rhs_b(4,0:7) 5 der1%rhs_t(0:4,7) 6 der1%mwn(n)
 *  7 der2%lhs(n,5) 8 der2%rhs(n,12) 9 der2%lu(n,5|3) 10 der2%mwn(n) 11 jac(n,3) 12 intl%lu0i(n,5) 13 intl%lu1i(n,5)
 * returns the number of doubles written (or <0). */
int tlab_fdm_plan_get(tlab_fdm_plan_t p, int which, double *buf, int nbuf);
int tlab_fdm_plan_info(tlab_fdm_plan_t p, int what); /* 0 n, 1 ndl1, 2 ndr1, 3 ndl2, 4 ndr2, 5 need_1der, 6 periodic, 7 staggered; after tlab_init:
                                                        * 8 chunks per x line of the wave-per-line kernel (64 one wave, 128 / 256 two / four waves
                                                        * per line, 0 another kernel), 9 whether all those chunks share one set of tables (1: an
                                                        * exactly uniform grid, scalar loads; 0: per-chunk tables staged in LDS) */

/* ---- operators ----------------------------------------------------------------------------- */
/* OPR_Partial_X/Y/Z(type, nx, ny, nz, bcs, g, u, result, tmp1)   operators/opr_partial.f90:31,266,154
 * dir = 1,2,3.  ibc = bcs(1,1) + 2*bcs(2,1) (opr_partial.f90:91).  type = TLAB_OPR_P1 | P2 | P2_P1.
 * tmp1: first derivative on return for P2_P1; scratch (may be NULL when the plan needs no Jacobian
 * correction) for P2; unused for P1.  u, result, tmp1 must not alias.  2-D guard: a direction of size 1
 * returns zeros (opr_partial.f90:175-177,287-289). */
int tlab_opr_partial(int dir, tlab_fdm_plan_t g, int type, int nx, int ny, int nz, int ibc,
                     const double *u, double *result, double *tmp1);

/* OPR_Burgers_X/Y/Z(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t)   physics/opr_burgers.f90:190,277,359
 * result = nu * d2s/dx2 - u * ds/dx along dir; nu = visc (is = 0) or visc/schmidt(is)
 * (OPR_Burgers_Initialize, opr_burgers.f90:92-112, folds it into the LU; here it is an argument).
 * ivel = TLAB_OPR_B_SELF: advecting velocity is s itself; TLAB_OPR_B_U_IN: it is u (natural layout).
 * The reference additionally threads a *transposed* copy of the velocity through tmp1/u_t to save CPU
 * transposes (opr_burgers.f90:236-240); the device kernels read u in its natural layout, so u_t is ignored.
 * If write_transposed != 0 and ivel == SELF, tmp1 receives the transposed operand exactly as the
 * reference leaves it (X: (ny*nz, nx); Y with nz > 1: (nz, nx*ny)), bit-exact; otherwise tmp1 is scratch
 * (first derivative). */
int tlab_opr_burgers(int dir, tlab_fdm_plan_t g, int ivel, int nx, int ny, int nz, int ibc, double nu,
                     const double *s, const double *u, double *result, double *tmp1, int write_transposed);
/* The anelastic branch of OPR_Burgers_1D (physics/opr_burgers.f90:504-507) with the module state OPR_Burgers_Initialize sets up for
 * nse_eqns == DNS_EQNS_ANELASTIC (:128-183): the diffusion term is multiplied by ribackground(j) -- rhoinv(1), rhoinv(3) along x and z, the
 * scaled U factors of fdmDiffusion(2) along y (the same product, rounding aside).  HOST pointers, ny values each; ny = 0 or NULL: off.
 * While it is on, tlab_opr_burgers runs the two derivatives unfused and a weighted epilogue. */
int tlab_opr_burgers_set_anelastic(int ny, const double *rbackground, const double *ribackground);

/* ---- 1-D filters: OPR_FILTER_1D (operators/opr_filter.f90:393-460) and the dealiasing branch of OPR_Burgers_1D ----------------------
 * type(filter_dt) as OPR_FILTER_INITIALIZE leaves it (opr_filter.f90:28-41, 236-275): type = DNS_FILTER_COMPACT 1 | _6E 2 | _4E 3 |
 * _COMPACT_CUTOFF 9 (:56-65; tophat 8 and the 3-D spectral / Helmholtz types: TLAB_EUNSUPPORTED), BcsMin / BcsMax = DNS_FILTER_BCS_* of
 * filters/flt_base.f90:5-11, coeffs = f%coeffs(size, inb_filter), HOST pointer, column-major (NULL for explicit6; the generators
 * FLT_C4_RHS_COEFFS / FLT_E4_COEFFS and the LU of the compact forms stay the host's). */
typedef struct tlab_filter *tlab_filter_t;
#define TLAB_FILTER_COMPACT 1
#define TLAB_FILTER_6E 2
#define TLAB_FILTER_4E 3
#define TLAB_FILTER_COMPACT_CUTOFF 9
int tlab_filter_create(tlab_filter_t *out, int type, int size, int periodic, int bcsmin, int bcsmax, int inb_filter, const double *coeffs);
int tlab_filter_destroy(tlab_filter_t f);
/* result = filter(u) along direction dir of a field (nx, ny, nz), out of place: FLT_C4_RHS + TRIDSS / TRIDPSS, FLT_C4[P]_CUTOFF_RHS + PENTADSS2 /
 * PENTADPSS, FLT_E6, FLT_E4 (src/filters/flt_compact.f90, flt_explitic.f90) -- one line per thread, the reference's operation order. */
int tlab_opr_filter_1d(int dir, tlab_filter_t f, int nx, int ny, int nz, const double *u, double *result);
/* OPR_FILTER(nx, ny, nz, f, u, txc) (opr_filter.f90:283-392), directional branch: the x, y, z filters in that order (NULL = none), each repeat[d]
 * times (NULL = once), in place; tmp: one field of scratch. */
int tlab_opr_filter(int nx, int ny, int nz, tlab_filter_t fx, tlab_filter_t fy, tlab_filter_t fz, const int *repeat, double *u, double *tmp);
/* [Dealiasing] Dealiasing(dir) of OPR_Burgers (physics/opr_burgers.f90:33, 71, 118-125): while a filter is set for a direction, tlab_opr_burgers
 * along it computes nu d2s - filter(u) filter(ds/dx) (:478-500), unfused; NULL: none.  The filter is not owned. */
int tlab_opr_burgers_set_dealiasing(int dir, tlab_filter_t f);

/* ---- Poisson solver --------------------------------------------------------------------------- */
/* OPR_Elliptic_Initialize (operators/opr_elliptic.f90:86-250, TYPE_FACTORIZE) + OPR_Fourier_Initialize
 * (operators/opr_fourier.f90:54-208): eigenvalues lambda(k,i) from the modified wavenumbers of the x and z plans,
 * singular modes, first-order integral operators of the y plan, rocFFT plans, and the homogeneous solutions of
 * every mode.  Serial decomposition (one GPU owns all of x and z). */
typedef struct tlab_poisson_plan *tlab_poisson_plan_t;
int tlab_poisson_plan_create(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                             int nx, int ny, int nz);
int tlab_poisson_plan_destroy(tlab_poisson_plan_t p);

/* OPR_Poisson(nx, ny, nz, ibc, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)   operators/opr_elliptic.f90:33-46, :263-364
 * Solves lap p = f with periodic x, z.  ibc = TLAB_BCS_NN: bcs_hb, bcs_ht (nx*nz each) are dp/dy at the walls (OPR_ODE2_Factorize_NN / _NN_Sing per
 * mode: the call of the RHS); ibc = TLAB_BCS_DD: they are p at the walls (OPR_ODE2_Factorize_DD / _DD_Sing, opr_elliptic.f90:322-329; marching
 * kernels, 2.4x the time of the per-mode stage, work arrays allocated on first use).  The reference has no other type for the factorized solver
 * (:312-331); direct plans (below) take all four.  p: forcing in, solution out (its wall planes are overwritten with the BC
 * data first, like the reference :285-286).  tmp1, tmp2: work arrays of isize_txc_field = (nx+2)*ny*nz doubles
 * (base/tlab_memory.f90:186-187), destroyed.  dpdy (may be NULL): dp/dy. */
int tlab_opr_poisson(tlab_poisson_plan_t plan, int nx, int ny, int nz, int ibc, double *p, double *tmp1, double *tmp2,
                     const double *bcs_hb, const double *bcs_ht, double *dpdy);

/* Which per-mode solver the factorized plans created AFTER this call use (also TLAB_POISSON_EXACT=1 in the environment):
 *   0 (default): k_ode_nn, the register-chunked solver (2.4 ms at 512^3).  Its substitutions run as a parallel scan over 8-row chunks, i.e. in
 *      another order than the reference's serial sweeps: as accurate as the reference, but where the problem amplifies rounding (the
 *      projection of a non-solenoidal field: forcing div(q)/dte ~ 1e4-1e6 for a pressure of 1e2-1e3) the two would differ by what two builds of
 *      the reference itself differ by (with / without fused multiply-adds: 4e-12 in p, 1.4e-11 in dp/dy on 512-point lines; one ulp of
 *      forcing noise -- another FFT library -- gives 6e-13 and 2.5e-12).  That difference sits in the few modes with lambda h^2 << 1, so on a
 *      single device those (at most 128 modes, sqrt(lambda) mean(h) <= 0.06) are solved by a marching sub-plan beside k_ode_nn
 *      (TLAB_POISSON_LOW_MODES=0 turns it off; decomposed plans run it on the rank that owns kx = 0..): 7.6e-13 /
 *      3.1e-12 on that projection, <= 3e-13 / 9e-13 on forcing without the cancellation; costs 0.1-0.2 ms per solve.
 *   1: the marching kernels (k_int1: one thread per mode, 5.8 ms at 512^3), which repeat FDM_Int1_Solve / OPR_ODE2_Factorize_NN operation by
 *      operation, in the reference's order and without fused multiply-adds: 4e-13 / 1.7e-12 on the same case, i.e. at the FFT-noise floor. */
int tlab_poisson_set_exact(int on);

/* EllipticOrder = CompactDirect6: OPR_Elliptic_Initialize with TYPE_DIRECT (operators/opr_elliptic.f90:107-163, 228-245) and
 * OPR_Poisson_FourierXZ_Direct (:368-455).  gy_elliptic is the reference's fdm_loc: a y plan whose second derivative holds the CompactDirect6
 * tables (tlab_fdm_plan_create_from_arrays + tlab_fdm_plan_set_scheme(.., 16) + tlab_fdm_plan_set_aux(.. nodes)); gy stays the plan of the
 * derivatives (dp/dy = OPR_Partial_Y(OPR_P1, p), :447-449; NOT owned, must outlive the Poisson plan).  lambda(k,i) = mwn2_x(i) + mwn2_z(k) from
 * the SECOND-derivative modified wavenumbers of gx, gz; one singular mode (1,1), solved with BCS_DN and p = 0 at the bottom.  Per mode one
 * second-order integral solve (FDM_Int2_Initialize / FDM_Int2_Solve, fdm/fdm_integral.f90:334-673).  With such a plan tlab_opr_poisson accepts
 * ibc = TLAB_BCS_DD / ND / DN / NN (bcs_hb, bcs_ht: function value at a D end, derivative at an N end). */
int tlab_poisson_plan_create_direct(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                                    int nx, int ny, int nz, tlab_fdm_plan_t gy_elliptic);
/* OPR_Helmholtz(nx, ny, nz, ibc, alpha, a, tmp1, tmp2, bcs_hb, bcs_ht)   operators/opr_elliptic.f90:48-62
 * Solves lap a + alpha a = f (the solver of the implicit RK, alpha < 0).  a: forcing in, solution out; tmp1, tmp2 as tlab_opr_poisson.
 * - DIRECT plan: OPR_Helmholtz_FourierXZ_Direct (:562-628), per mode the second-order integral system with constant lambda2 - alpha, any of
 *   the four boundary types, no singular-mode treatment.
 * - factorized plan of tlab_poisson_plan_create: OPR_Helmholtz_FourierXZ_Factorize (:466-557), per mode OPR_ODE2_Factorize_NN / _DD with
 *   sqrt(lambda - alpha) (must be positive for every mode), ibc = TLAB_BCS_NN or TLAB_BCS_DD (others: TLAB_EUNSUPPORTED, like the reference).
 *   Where the reference factorizes 2 systems per mode on every call (:518-522), the tables of an alpha are built on its first call and kept
 *   (the 4 most recent alphas; about 6 field-sized arrays each); the x, y, z plans given to tlab_poisson_plan_create must still be alive. */
int tlab_opr_helmholtz(tlab_poisson_plan_t plan, int nx, int ny, int nz, int ibc, double alpha, double *a, double *tmp1, double *tmp2,
                       const double *bcs_hb, const double *bcs_ht);
/* Decomposed direct plans, driven stage by stage like the factorized ones (set_wall_planes, fft_x, [exchange], fft_z, tlab_poisson_direct_ode,
 * fft_z, [exchange], fft_x) -- with ONE field on the way back: dp/dy is OPR_Partial_Y of p on the slab (y is never split).
 * mode 0: z-slab plan, (a, b) = (koffset, nproc_k) as tlab_poisson_plan_create_slab; mode 1: kx-pencil plan, (a, b) = (ioffset, nxl). */
int tlab_poisson_plan_create_direct_decomposed(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                                               int nx, int ny, int kmax, int nz_total, int mode, int a, int b, tlab_fdm_plan_t gy_elliptic);
/* the per-mode stage of a direct plan on spectral fields (nx/2+1, ny, nz) complex: f_hat -> p_hat (may alias) */
int tlab_poisson_direct_ode(tlab_poisson_plan_t plan, int ibc, double *f_hat, double *p_hat);

/* z-slab variant (ims_npro_k = nproc_k ranks, one GPU each; base/tlab_mpi_procs.f90:76-94): this rank owns planes
 * [koffset, koffset + kmax) of nz_total.  The per-mode tables are built for the rank's own kz range (opr_elliptic.f90:167-194)
 * and the z transform works on the K-transposed layout (nlines = (nx/2+1)*ny/nproc_k lines of nz_total points).  Such a plan is
 * driven stage by stage, with the caller's all-to-all (TLabMPI_Trp_ExecK_*, base/tlab_mpi_transpose.f90:343-458) in between:
 *   set_wall_planes, fft_x(+1), [K-forward], fft_z(+1), [K-backward], ode, [K-forward], fft_z(-1), [K-backward], fft_x(-1).  */
int tlab_poisson_plan_create_slab(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                                  int nx, int ny, int kmax, int nz_total, int koffset, int nproc_k);
/* kx-pencil variant: the physical box is the z-slab (nx, ny, kmax); the spectral box holds the kx range [ioffset, ioffset+nxl) of
 * ALL kz: (nxl, ny, nz_total).  One all-to-all (slab -> pencil) after fft_x(+1) and one before fft_x(-1) replace the four
 * K-transposes per field of the plan above; fft_z and ode then work on the pencil:
 *   set_wall_planes, fft_x(+1), [slab->pencil], fft_z(+1), ode, fft_z(-1), [pencil->slab], fft_x(-1).  */
int tlab_poisson_plan_create_pencil(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                                    int nx, int ny, int kmax, int nz_total, int ioffset, int nxl);
/* Repacking around the slab <-> pencil all-to-all: slab = complex (nxh, ny, kmax); buffer = for every peer p the block [kmax][ny][nxl_p] of
 * its kx range [ioff[p], ioff[p+1]) (ioff[nproc] = nxh implied), blocks back to back.  dir = +1 slab -> buffer, -1 buffer -> slab.  nproc <= 8.
 * (The pencil side needs no repacking: z is its slowest index.)  Bit-exact index work. */
int tlab_pencil_repack(double *slab, double *buffer, int nxh, int ny, int kmax, int nproc, const int *ioff, int dir);
/* The same with up to 16 blocks placed anywhere in the buffer: block b = kx range [start[b], start[b+1]) (the last one ends at nx/2+1), laid out
 * [kmax][ny][width] from complex element base[b] on.  Used by the slab driver to send every peer's kx range in two halves, all first halves
 * ahead of all second halves, so that the solves of one half run while the other half is on the wire (tlab_amd/parallel.py). */
int tlab_pencil_repack_blocks(double *slab, double *buffer, int nxh, int ny, int kmax, int nblocks, const int *start, const long long *base, int dir);
int tlab_poisson_set_wall_planes(tlab_poisson_plan_t plan, double *p, const double *bcs_hb, const double *bcs_ht);
/* OPR_Fourier_X_Forward (dir = +1) / _Backward (dir = -1), opr_fourier.f90:219,277; out of place, unnormalised like FFTW.  The forward transform is
 * the library's one-pass kernel where its lengths apply (nx/2 = 8^a * {1,2,4}, 128 <= nx <= 2048), rocFFT otherwise; the backward one is rocFFT's.
 * dir = -2: the library's own inverse kernel (the one that finishes the v equation inside tlab_opr_poisson when the RHS driver asks for it), for tests. */
int tlab_poisson_fft_x(tlab_poisson_plan_t plan, int dir, double *in, double *out);
/* The same transforms with the complex side directly in the pack layout of tlab_pencil_repack_blocks (same nblocks / start / base), i.e. the
 * repack pass folded into the library's own x-transform kernels: dir = +1 real in -> pack buffer out, dir = -1 pack buffer in -> real out.
 * _final: the inverse of dp^/dy finishes the v equation in its epilogue (h = h - dp/dy, wall planes of h zeroed, q += dte h, h *= kco when
 * scale; rhs_global_incompressible_1.f90:236-241,258-261 + time.f90 RK update) instead of writing dp/dy.  TLAB_EUNSUPPORTED where the own
 * kernels do not apply (see above); the inverse differs from rocFFT's by rounding.  Used by the native slab driver (tlab_slab_dns_*). */
int tlab_poisson_fft_x_packed(tlab_poisson_plan_t plan, int dir, double *in, double *out, int nblocks, const int *start, const long long *base);
int tlab_poisson_fft_x_packed_final(tlab_poisson_plan_t plan, double *in, double *q, double *h, double dte, double kco, int scale, int nblocks,
                                    const int *start, const long long *base);
int tlab_poisson_fft_z(tlab_poisson_plan_t plan, int dir, double *in, double *out);   /* the FFT inside OPR_Fourier_Z_*, :355,422 */
int tlab_poisson_ode(tlab_poisson_plan_t plan, double *f_hat, double *p_hat, double *dp_hat); /* mode loop of opr_elliptic.f90:308-333 */

/* Accumulating forms of the two operators, as the RHS uses them (hq = hq + OPR_Burgers(...), tmp = tmp + OPR_Partial(hq + q/dte)):
 *   tlab_opr_burgers_add : result += nu d2s/dx2 - vel ds/dx                       (rhs_global_incompressible_1.f90:98-136 + :106-112)
 *   tlab_opr_partial_add : result (+)= d/dx (u + scale*ub)   (ub may be NULL; acc = 0 overwrites)        (:197-201, :228-230, :257-259)
 * One fused kernel when the sizes are on a fast path; otherwise the reference's own sequence with the temporaries tmp1, tmp2. */
int tlab_opr_burgers_add(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, double nu, const double *s, const double *vel,
                         double *result, double *tmp1, double *tmp2);
int tlab_opr_partial_add(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, const double *u, const double *ub, double scale,
                         double *result, int acc, double *tmp1, double *tmp2);
/* Last pass over a velocity component folded into the pressure gradient (Dirichlet walls): h -= dp/dx_dir; h = 0 on the wall planes
 * j = 1, ny; q += dte h; h *= kco if scale (rhs_global_incompressible_1.f90:319-320, :348-352, :373-375; time.f90:645-664, :272-297).
 * One kernel on a fast path (dir = 1, 3), otherwise OPR_Partial into tmp1 + the pointwise pass. */
int tlab_opr_gradient_final(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, const double *p, double *q, double *h, double dte,
                            double kco, int scale, double *tmp1);
/* nf (1..4) transported fields advected by the same velocity (the u-, v-, w- and scalar equations all call OPR_Burgers_X with u, etc.):
 * result[f] += nu[f] d2s[f]/dx2 - vel ds[f]/dx in one launch, the velocity being fetched from HBM once.  HOST arrays of DEVICE pointers.
 * overwrite != 0: result[f] = ... (the tendencies are zero at the start of a Runge-Kutta step, time.f90:212-216: no fill, no read). */
int tlab_opr_burgers_add_n(int dir, tlab_fdm_plan_t g, int nx, int ny, int nz, int ibc, int nf, const double *nu, const double *const *s,
                           const double *vel, double *const *result, double *tmp1, double *tmp2, int overwrite);

/* ---- z-derivatives on a z-slab without transposes (multi-GPU) ---------------------------------------------
 * Replaces TLabMPI_Trp_ExecK_Forward + OPR_Partial_Z / OPR_Burgers_Z + TLabMPI_Trp_ExecK_Backward (opr_partial.f90:154-262,
 * opr_burgers.f90:331-440, base/tlab_mpi_transpose.f90:343-458) for a periodic z split into slabs of kmax planes: the compact
 * system is partitioned at the slab boundaries, so that a slab only needs 3 halo planes of the operand and, per line and
 * implicit system, one value from each neighbour (see tlab_amd/csrc/zslab.hip).  Two-phase protocol per operator:
 *   phase 1: head[nsys][nx*ny], tail[nsys][nx*ny] are written          (nsys = 1 for partial_z, 2 for burgers_z)
 *   caller:  tail -> right neighbour (arrives as tail_left), head -> left neighbour (arrives as head_right)
 *   phase 2: result = d/dz (u [+ scale*ub])   resp.   nu d2s/dz2 - vel ds/dz ;  acc != 0: result += ...
 * Every operand pointer addresses the first plane of the slab and must have 3 valid planes before it and after the slab
 * (the neighbours' planes, periodic in z).  plan_create returns TLAB_EUNSUPPORTED when the slab is too thin for the coupling
 * between slab separators to vanish in double precision (kmax >~ 50 for the sixth-order schemes): keep the transpose path then.
 * chunk: rows per wave (0 = automatic, 16 or 32). */
typedef struct tlab_zslab_plan *tlab_zslab_plan_t;
int tlab_zslab_plan_create(tlab_zslab_plan_t *out, tlab_fdm_plan_t gz, int kmax, int koffset, int chunk);
int tlab_zslab_plan_destroy(tlab_zslab_plan_t plan);
int tlab_zslab_partial_z(tlab_zslab_plan_t plan, int phase, int nx, int ny, const double *u, const double *ub, double scale,
                         double *head, double *tail, const double *tail_left, const double *head_right, double *result, int acc);
int tlab_zslab_burgers_z(tlab_zslab_plan_t plan, int phase, int nx, int ny, double nu, const double *s, const double *vel,
                         double *head, double *tail, const double *tail_left, const double *head_right, double *result, int acc);
/* nf (1..4) transported fields per launch; head, tail, tail_left, head_right: [nf][2][nx*ny] */
int tlab_zslab_burgers_z_n(tlab_zslab_plan_t plan, int phase, int nx, int ny, int nf, const double *nu, const double *const *s,
                           const double *vel, double *head, double *tail, const double *tail_left, const double *head_right,
                           double *const *result, int acc);
/* phase 2 of d/dz p with the final update of w as its epilogue (as tlab_opr_gradient_final; phase 1 = tlab_zslab_partial_z(plan, 1, ...)) */
int tlab_zslab_gradient_final_z(tlab_zslab_plan_t plan, int nx, int ny, const double *p, const double *tail_left, const double *head_right,
                                double *q, double *h, double dte, double kco, int scale);

/* ---- RHS assembly and Runge-Kutta substep ("next" row n1 of SURVEY.md 8f) --------------------------------- */
/* Module state the reference spreads over TLab_Memory / NavierStokes / OPR_Burgers / BOUNDARY_BCS: plans, sizes,
 * visc = 1/Reynolds, schmidt(1:nscal) (physics/navierstokes.f90), wall boundary conditions (no-slip, Dirichlet scalars). */
typedef struct tlab_dns *tlab_dns_t;
int tlab_dns_create(tlab_dns_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                    tlab_poisson_plan_t poisson, int nx, int ny, int nz, int nscal, double visc, const double *schmidt);
int tlab_dns_destroy(tlab_dns_t d);
/* on (default): the pointwise sums of the RHS are folded into the operator kernels (same summation order as the reference);
 * off: the reference's literal sequence of temporaries + pointwise loops.  Both give the same result to round-off. */
int tlab_dns_set_fusion(tlab_dns_t d, int on);
/* Start of a Runge-Kutta step: TIME_RUNGEKUTTA sets hq = 0, hs = 0 there (tools/dns/time.f90:212-216).  Instead of filling the arrays,
 * tell the driver: the next tlab_rhs_global_incompressible_1 / tlab_time_substep_incompressible_explicit treats them as zero (its first
 * operator launch overwrites instead of accumulating), whatever they contain. */
int tlab_dns_begin_step(tlab_dns_t d);

/* ---- The tail of the substep for a host whose time loop is NOT patched (csrc/deferred.cpp) ---------------------------------------------------
 * tools/dns/time.f90 calls RHS_GLOBAL_INCOMPRESSIBLE_1() (:612), then DAXPY(n, dte, hq(1,is), 1, q(1,is), 1) per field (:649-660) and, in every
 * substep but the last, DSCAL(n, kco, hq(1,is), 1) per field (:279-293); `hq = 0 ; hs = 0` opens a step (:212-216).  Executed one by one the BLAS
 * calls are 2 (3 + ns) passes over the fields.  With tlab_deferred_enable(1) the entry points below only RECORD: when the recorded sequence is
 * exactly that of time.f90 (the tendencies and states the RHS was given, dte as the factor, one kco for all) the library runs ONE
 * tlab_time_substep_incompressible_explicit(dte, kco, scale) -- the call of the patched host, bit for bit -- and the
 * zero fills become tlab_dns_begin_step.  Anything else is executed literally in the order it came.  A recorded sequence runs before any other
 * launch of the library (every entry point that enqueues work, tlab_sync, the copies, tlab_free flush first).  NOT covered: host statements that
 * read a device array directly without a call into the library -- call tlab_deferred_flush() (or tlab_sync()) before them.  A recorded substep that
 * FAILS when another entry point makes it run has no caller to report to: its code is returned by the next tlab_sync / tlab_deferred_flush (the text
 * stays in tlab_last_error()).  Off by default:
 * the four operations then execute immediately (tlab_rhs_global_incompressible_1, tlab_pw_rk_update, tlab_pw_scale, tlab_pw_fill). */
int tlab_deferred_enable(int on);
int tlab_deferred_rhs(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc);
int tlab_deferred_axpy(long long n, double a, const double *x, double *y);      /* DAXPY with unit strides: y += a x */
int tlab_deferred_scal(long long n, double a, double *x);                       /* DSCAL with unit stride:  x *= a   */
int tlab_deferred_zero(double *a, long long n);                                 /* a(1:n) = 0                        */
int tlab_deferred_flush(void);
/* counts[6]: fused substeps run, sequences executed literally, begin_steps taken from zero fills, eager axpy, eager scal, eager zero fills */
int tlab_deferred_stats(long long *counts);
/* Which device allocations should play q, s, hq, hs, txc?  The rate of a kernel that streams many arrays at once depends on the SET of allocations
 * it streams (not on any one of them): 4.8 .. 5.9 TB/s for one 13-stream kernel over sets of 1-GiB hipMalloc allocations, 16.0 .. 17.2 ms per substep
 * of the 512^3 box from process to process (profiles/r05/placement_*.txt).  No rule predicts it, so it is searched: given a pool
 * of npool >= 2 (3 + nscal) + 9 candidate arrays, each of at least isize_txc_field doubles, the substep itself (one three-stage Runge-Kutta step per
 * trial, timed by events) is run on the assignment "pool in order", on random_trials random assignments and on one pass of single-role exchanges,
 * and the fastest comes back: assignment[r] = index into pool of role r, roles in the order q[0..2], s[0..nscal-1], hq[0..2], hs[0..nscal-1],
 * txc[0..8].  state: 3 + nscal device arrays the trial fields start from (NULL: zeros).  EVERY array of the pool is overwritten; the caller puts its
 * fields into the chosen arrays afterwards.  report (NULL or 5 doubles): ms per substep of the first assignment, the best, the median, the worst,
 * and the number of trials.  A start-up cost of (random_trials + nroles + 1) x 4 substeps; the results of the run do not depend on it. */
int tlab_dns_place_arrays(tlab_dns_t d, int npool, double *const *pool, const double *const *state, double dtime, int random_trials,
                          unsigned seed, int *assignment, double *report);
/* After the search the first assignment and the winner are timed again back to back (three steps each); the winner is kept only if its median is
 * the lower one, and report[0], report[1] come from those repeats.  The begin_step flag of the driver is what it was on entry; the contents of
 * EVERY pool array (state, tendencies, work arrays) are undefined afterwards. */
/* The same search for a host whose arrays are two-dimensional (Tlab: q(isize_field, 3), s(isize_field, ns), hq, hs, txc(isize_txc_field, 9);
 * base/tlab_memory.f90:201-207, dns_main.f90:103-104): components lie a fixed stride apart, so only whole BLOCKS can be placed.  cand_X[0..ncand-1]
 * are candidate allocations for block X (index 0 = what the host holds now; q / hq: 3 n doubles, s / hs: nscal n, txc: 9 txc_stride); the substep
 * is timed on the host's own combination, on random_trials random ones and on one pass of single-block exchanges, the winner confirmed as above.
 * choice[5]: the candidate index per block in the order q, s, hq, hs, txc; report as tlab_dns_place_arrays.  All candidates are overwritten: call
 * it BEFORE the fields are read, then point the host arrays at the chosen allocations and free the others. */
int tlab_dns_place_blocks(tlab_dns_t d, int ncand, double *const *cand_q, double *const *cand_s, double *const *cand_hq, double *const *cand_hs,
                          double *const *cand_txc, long long txc_stride, double dtime, int random_trials, unsigned seed, int *choice, double *report);
/* Wall boundary conditions in y: BcsFlowJmin%type(1:3), BcsFlowJmax%type(1:3), BcsScalJmin%type(1:nscal), BcsScalJmax%type(1:nscal)
 * (tools/dns/boundary_bcs.f90:24-27, read at :102-190).  Values as in the reference: DNS_BCS_DIRICHLET / DNS_BCS_NEUMANN.
 * Default at creation: all Dirichlet ('noslip'; the reference's 'freeslip' is {NEUMANN, DIRICHLET, NEUMANN} for (u,v,w)).
 * The normal velocity (index 1 here, 0-based) must stay Dirichlet: the pressure solver's Neumann data assume v = 0 at the walls. */
#define TLAB_DNS_BCS_DIRICHLET 3
#define TLAB_DNS_BCS_NEUMANN 4
int tlab_dns_set_bcs(tlab_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax);
/* [PressureFilter] of dns.ini: p and dp/dy pass OPR_FILTER after the Poisson solve (rhs_global_incompressible_1.f90:286-290; examples/Case92-93:
 * compact filter in y with zero ends).  Filters not owned; NULL = none. */
int tlab_dns_set_pressure_filter(tlab_dns_t d, tlab_filter_t fx, tlab_filter_t fy, tlab_filter_t fz, const int *repeat);
/* remove_divergence of dns.ini (tools/dns/dns_read_local.f90): on (default): the pressure forcing is div(hq + q/dte), which removes the residual
 * divergence of q (rhs_global_incompressible_1.f90:177-232); off: div(hq) (:234-250). */
int tlab_dns_set_remove_divergence(tlab_dns_t d, int on);
/* Dynamic surface model of the scalars: BcsScalJmin/Jmax%SfcType (0 = DNS_SFC_STATIC, 1 = DNS_SFC_LINEAR) and %cpl per scalar
 * ([BoundaryConditions] Scalar<i>SfcTypeJmin/Jmax, Scalar<i>CouplingJmin/Jmax; tools/dns/boundary_bcs.f90:29-31, 76-87).  With a linear surface the
 * wall plane of the scalar's tendency is its old value plus cpl times the anomaly of the diffusive surface flux: the tendency planes kept at the
 * start of the RHS (rhs_global_incompressible_1.f90:77-87) and BOUNDARY_BCS_SURFACE_Y (boundary_bcs.f90:478-546: d/dy of the scalar, AVG1V2D of its
 * plane j = 1 -- at both ends, as the reference has it) in the boundary section (:393-396).  Single-domain driver (the plane average is an
 * all-reduce in a decomposed run: the slab / pencil drivers do not take it). */
int tlab_dns_set_surface_bcs(tlab_dns_t d, const int *sfc_jmin, const int *sfc_jmax, const double *cpl_jmin, const double *cpl_jmax);
/* nse_eqns == DNS_EQNS_ANELASTIC: the density weights of the RHS (Thermo_Anelastic_WEIGHT_INPLACE / _SUBTRACT with rbackground / ribackground,
 * tools/dns/rhs_global_incompressible_1.f90:211-214, :326-329; the Neumann data of the pressure times rbackground at the walls, :275-277) and of
 * OPR_Burgers (below).  rbackground, ribackground: HOST pointers, ny values each, ribackground = 1 / rbackground -- the profiles the host's
 * thermodynamics made (Thermo_Anelastic is outside the path); NULL switches back to the incompressible equations.  Anelastic substeps take the
 * literal operator sequence (no fused forms). */
int tlab_dns_set_anelastic(tlab_dns_t d, const double *rbackground, const double *ribackground);

/* BOUNDARY_BCS_NEUMANN_Y(ibc, nx, ny, nz, g, u, bcs_hb, bcs_ht, tmp1)   tools/dns/boundary_bcs.f90:368-473
 * Wall values of u (planes of nx*nz, x fastest) such that du/dy = 0 at the bottom (ibc = 1), top (2) or both (3) walls,
 * from the Neumann-reduced first-derivative system of g.  u is read only (its wall planes are not used); a plane not selected
 * by ibc is left untouched; tmp1 is work space of nx*ny*nz. */
int tlab_boundary_bcs_neumann_y(tlab_fdm_plan_t g, int ibc, int nx, int ny, int nz, const double *u, double *bcs_hb,
                                double *bcs_ht, double *tmp1);

/* ---- per-iteration monitors ("next" row n2 of SURVEY.md 8f) -------------------------------------------------------------
 * TIME_COURANT()   tools/dns/time.f90:365-548 (incompressible branch; constants of TIME_INITIALIZE :138-176):
 *   pmax[0] = max |u|/dx + |v|/dy + |w|/dz over the local box, pmax[1] = schmidtfactor * max(1/dx^2 + 1/dy^2 + 1/dz^2);
 *   dtime (may be NULL) = min(cfla / pmax[0], cfld / pmax[1]) if cfla > 0.  With several ranks the caller MPI_MAX-reduces pmax first
 *   (time.f90:522) -- tlab_dns_set_slab tells a z-slab its first global plane (ims_offset_k).  Synchronises the stream.
 * FI_INVARIANT_P(nx,ny,nz,u,v,w,result,tmp1)   mappings/fi_vectorcalculus.f90:111-141 : result = -div(u,v,w)
 * MINMAX(imax,jmax,kmax,a,amn,amx)             utils/minmax.f90:6 (local part; DNS_BOUNDS_CONTROL dns_local.f90:184-187 uses both) */
int tlab_dns_set_slab(tlab_dns_t d, int koffset);
int tlab_time_courant(tlab_dns_t d, double *const *q, double cfla, double cfld, double *pmax, double *dtime);
int tlab_fi_invariant_p(tlab_dns_t d, const double *u, const double *v, const double *w, double *result, double *tmp1);
int tlab_minmax(tlab_dns_t d, const double *a, int nx, int ny, int nz, double *amn, double *amx);

/* RHS_GLOBAL_INCOMPRESSIBLE_1()   tools/dns/rhs_global_incompressible_1.f90:15-405 (argument-less in the reference:
 * it works on the module arrays q, s, hq, hs, txc and on dte).  q[3] = u,v,w; s[nscal]; hq[3], hs[nscal] are
 * accumulated into; txc[9] = tmp1..tmp9, each of isize_txc_field = (nx+2)*ny*nz doubles.  HOST arrays of DEVICE pointers.
 * Convective form, RhsMode = combined, remove_divergence = yes, no buffer zone / IBM / anelastic terms. */
int tlab_rhs_global_incompressible_1(tlab_dns_t d, double dte, double *const *q, double *const *s, double *const *hq,
                                     double *const *hs, double *const *txc);

/* One explicit low-storage Runge-Kutta substep = the unit of the metric:
 * TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT (tools/dns/time.f90:559-664): RHS, then q += dte*hq, s += dte*hs;
 * followed, if scale_tendencies != 0, by hq *= kco, hs *= kco (time.f90:261-298, skipped after the last substep). */
int tlab_time_substep_incompressible_explicit(tlab_dns_t d, double dte, double kco, int scale_tendencies,
                                              double *const *q, double *const *s, double *const *hq, double *const *hs,
                                              double *const *txc);

/* ---- the decomposed substep: RHS_GLOBAL_INCOMPRESSIBLE_1 / TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT with ims_npro_k > 1 ----------------
 * z-slabs (ims_npro_i x ims_npro_k = 1 x P, base/tlab_mpi_procs.f90:76-94): x- and y-operators are local.  What the reference does with a
 * K-transposition around every z-operator and z-FFT (operators/opr_partial.f90:185-195, 248-253; physics/opr_burgers.f90:386-426;
 * operators/opr_fourier.f90:343-428; base/tlab_mpi_transpose.f90:343-553) is done here without transposing a field for a derivative: the compact
 * z-systems are partitioned at the slab boundaries (tlab_zslab_* above: 3 halo planes of the operand + one value per line and system from each ring
 * neighbour), and the Poisson solver goes z-slab -> kx-pencil with ONE all-to-all after the x-FFT and one per output field on the way back, pipelined
 * in two kx halves against the per-mode solves (tlab_amd/csrc/slab.cpp).  The exchanges go through a transport the caller hands in:
 *   - tlab_comm_slab_transport, in libtlab_amd_comm.so: RCCL grouped ncclSend / ncclRecv on the library's communication stream, so that the x / y
 *     operators on the compute stream (tlab_set_stream) run while halo planes, interface values and pencil blocks travel over xGMI;
 *   - tlab_slab_transport_loopback: all P ranks inside one process on one device (exchanges = device copies): the complete decomposed algorithm
 *     against the single-domain result on a single GPU;
 *   - any other struct of these five entry points: a Fortran host's GPU-aware MPI (MPI_Isend / MPI_Irecv / MPI_Alltoallv on device pointers), or the
 *     host-staged gloo processes of the tests.
 * All counts are doubles, all buffers DEVICE pointers; `stream` is the caller's compute stream (hipStream_t).  An exchange must not start before the
 * work enqueued on `stream` up to the call has finished, and must be complete for the work enqueued on `stream` after `wait` of its ticket. */
typedef struct tlab_slab_transport {
    void *ctx;
    int nranks;           /* ims_npro_k                                                                                              */
    int nlocal;           /* ranks this process executes: 1, or nranks for the single-process loopback                                */
    int first;            /* ims_pro_k of the first (or only) local rank                                                              */
    /* Periodic ring in z.  For the local rank l < nlocal and the message i < nmsg (count[i] doubles): to_left[l*nmsg + i] arrives in from_right[..]
     * of rank - 1, to_right[l*nmsg + i] in from_left[..] of rank + 1.  With two ranks both neighbours are the same peer: sends are posted left then
     * right, receives from the right then from the left.  Returns a ticket >= 0, or a negative TLAB_E* code. */
    int (*ring_start)(void *ctx, void *stream, int nmsg, const long long *count, double *const *to_left, double *const *to_right,
                      double *const *from_right, double *const *from_left);
    /* MPI_Alltoallv among the nranks: send[l] holds the blocks for rank 0, 1, .. back to back, scount[l*nranks + p] doubles each; recv[l] / rcount
     * likewise by source.  Returns a ticket. */
    int (*alltoallv_start)(void *ctx, void *stream, double *const *send, const long long *scount, double *const *recv, const long long *rcount);
    int (*wait)(void *ctx, void *stream, int ticket);
    /* MPI_ALLREDUCE of n HOST doubles per local rank (values[l*n + i]), in place; op 0 = MPI_MAX, 1 = MPI_MIN (tools/dns/time.f90:522),
     * 2 = MPI_SUM (the plane average of the dynamic surface model, boundary_bcs.f90:520) */
    int (*allreduce)(void *ctx, double *values, int n, int op);
    void (*destroy)(void *ctx);     /* may be NULL */
} tlab_slab_transport;
int tlab_slab_transport_loopback(tlab_slab_transport *out, int nranks);

/* ---- x/z pencils: ims_npro_i x ims_npro_k blocks (base/tlab_mpi_procs.f90:76-94), the reference's own scheme ---------------------------------
 * Every x operator through an I-transposition inside ims_comm_x (operators/opr_partial.f90:66-147, physics/opr_burgers.f90:216-262), every z operator
 * through a K-transposition inside ims_comm_z (opr_partial.f90:185-253, opr_burgers.f90:386-426), the operator sequence of
 * tools/dns/rhs_global_incompressible_1.f90:98-398 with its transposed-velocity reuse, the Poisson solver on kx-pencils over all ranks.
 * Exchanges = MPI_Alltoallv inside one of three communicators, supplied by the caller like tlab_slab_transport:
 *   which = 0: all ranks (world order); 1: ims_comm_x of the local rank (members by ims_pro_i); 2: ims_comm_z (members by ims_pro_k).
 *   send[l] holds the blocks for the members back to back, scount[l*size + j] doubles each; recv[l] / rcount likewise by source.
 * Implementations: tlab_pencil_transport_loopback (all ranks in this process, exchanges = device copies), tlab_comm_pencil_transport
 * (libtlab_amd_comm.so: grouped ncclSend / ncclRecv inside the RCCL communicators of tlab_comm_init), or the host's GPU-aware MPI. */
typedef struct tlab_pencil_transport {
    void *ctx;
    int npro_i, npro_k;   /* ims_npro_i, ims_npro_k                                                 */
    int nlocal, first;    /* world ranks this process executes: [first, first + nlocal)             */
    int (*alltoallv_start)(void *ctx, void *stream, int which, double *const *send, const long long *scount, double *const *recv,
                           const long long *rcount);                                                /* returns a ticket >= 0 or a TLAB_E* code */
    int (*wait)(void *ctx, void *stream, int ticket);
    int (*allreduce)(void *ctx, double *values, int n, int op);                                     /* as tlab_slab_transport                  */
    void (*destroy)(void *ctx);                                                                     /* may be NULL                             */
} tlab_pencil_transport;
int tlab_pencil_transport_loopback(tlab_pencil_transport *out, int npro_i, int npro_k);

typedef struct tlab_pencil_dns *tlab_pencil_dns_t;
/* gx, gz: the plans of the GLOBAL x and z directions (nx, nz_total nodes), gy the y plan.  Conditions of the reference's decomposition: nx, nz
 * divisible by npro_i, npro_k; imax even; npro_i divides kmax; imax*jmax divisible by npro_k; at least one kx mode per rank.  The transport context
 * passes to the driver on success only.  Same supported subset as tlab_slab_dns_create (anelastic / dealiasing refused); the driver ADDS to the
 * tendencies like the reference: call tlab_pencil_dns_begin_step (hq = hs = 0) at the start of every Runge-Kutta step. */
int tlab_pencil_dns_create(tlab_pencil_dns_t *out, const tlab_pencil_transport *transport, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                           int nx, int ny, int nz_total, int nscal, double visc, const double *schmidt);
int tlab_pencil_dns_destroy(tlab_pencil_dns_t d);
/* q[3], s[nscal], hq[3], hs[nscal] of imax*jmax*kmax doubles, txc[9] of tlab_pencil_dns_info(d, 3) doubles each, for local rank l */
int tlab_pencil_dns_bind(tlab_pencil_dns_t d, int l, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc);
long long tlab_pencil_dns_info(tlab_pencil_dns_t d, int what);   /* 0 imax, 1 kmax, 2 kmax / npro_i, 3 doubles of a txc array, 4 nlocal, 5 first local rank */
int tlab_pencil_dns_set_bcs(tlab_pencil_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax);
int tlab_pencil_dns_begin_step(tlab_pencil_dns_t d);
int tlab_pencil_dns_rhs(tlab_pencil_dns_t d, double dte);                                    /* RHS_GLOBAL_INCOMPRESSIBLE_1 */
/* The transpositions are started AHEAD of independent launches (the reference's analogue: tools/dns/rhs_global_incompressible_nbc.f90:135-382): while one
 * operator is applied to a transposed field, the forward transposition of the next field and the backward transposition of the previous result are in
 * flight on the transport's stream; TLAB_PENCIL_OVERLAP=0 (read at creation) keeps the literal sequence.  tlab_pencil_dns_trace: on != 0 records, per
 * RHS, one line per exchange start ("start forward i" / "start backward i"), per launch ("launch operator i" / "launch local") and per wait ("wait
 * forward i" / "wait backward i"); the text of the last RHS is copied into buf (size bytes). */
int tlab_pencil_dns_trace(tlab_pencil_dns_t d, int on, char *buf, int size);
int tlab_pencil_dns_substep(tlab_pencil_dns_t d, double dte, double kco, int scale_tendencies);   /* + the update loops of time.f90:645-664, :272-297 */

typedef struct tlab_slab_dns *tlab_slab_dns_t;
/* gx, gy: the local plans; gz: the plan of the GLOBAL z direction (nz_total nodes).  kmax = nz_total / nranks planes per rank; returns
 * TLAB_EUNSUPPORTED when the slabs are too thin for the partitioned z-systems (kmax <~ 50: tlab_zslab_plan_create) -- such runs keep the
 * reference's K-transposition scheme.  gy_elliptic: NULL, or the y plan of EllipticOrder = CompactDirect6 (tlab_poisson_plan_create_direct).
 * The plans are not owned; the transport struct is copied (its destroy, if any, runs in tlab_slab_dns_destroy).
 * OWNERSHIP of transport->ctx passes to the driver only when the call returns TLAB_OK; on any refusal the caller still owns it and must run
 * transport->destroy(ctx) itself.
 * SUPPORTED SUBSET (everything else is refused with TLAB_EUNSUPPORTED, never silently dropped): convective form, RhsMode = combined, Dirichlet or
 * Neumann walls with an impermeable wall-normal velocity, static or dynamic (linear) scalar surfaces, factorized or CompactDirect6 elliptic solver,
 * remove_divergence on or off; NOT the anelastic formulation or dealiasing filters (single-domain driver tlab_dns_* only). */
int tlab_slab_dns_create(tlab_slab_dns_t *out, const tlab_slab_transport *transport, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz,
                         int nx, int ny, int nz_total, int nscal, double visc, const double *schmidt, tlab_fdm_plan_t gy_elliptic);
int tlab_slab_dns_destroy(tlab_slab_dns_t d);
/* The module arrays of local rank l (0 for a one-rank process): q[3], s[nscal], hq[3], hs[nscal], txc[9] as tlab_rhs_global_incompressible_1
 * (HOST arrays of DEVICE pointers; txc of (nx+2)*ny*kmax doubles each).  Plain arrays: the neighbours' halo planes land in buffers of the driver
 * (the columns of a Fortran host's q(isize_field, 3) lie back to back: there is no room around a field).  The pointers are kept: call again if
 * the host moves its arrays. */
int tlab_slab_dns_bind(tlab_slab_dns_t d, int l, double *const *q, double *const *s, double *const *hq, double *const *hs, double *const *txc);
/* what: 0 kmax, 1 nranks, 2 nlocal, 3 doubles of one halo (3 planes), 4 pipeline stages of the pencil exchange, 5 first local rank,
 * 6 whether the repack passes are folded into the x-transforms (default where the library's own x kernels apply; TLAB_SLAB_FUSED_X=0 at creation
 * keeps the separate passes and rocFFT's inverse, whose results equal the Python driver's to the bit) */
long long tlab_slab_dns_info(tlab_slab_dns_t d, int what);
int tlab_slab_dns_set_bcs(tlab_slab_dns_t d, const int *flow_jmin, const int *flow_jmax, const int *scal_jmin, const int *scal_jmax);
int tlab_slab_dns_begin_step(tlab_slab_dns_t d);                 /* as tlab_dns_begin_step */
/* The deferred tail (tlab_deferred_enable, above) behind the decomposed drivers: the link-time RHS_GLOBAL_INCOMPRESSIBLE_1 of an unpatched host calls these
 * instead of tlab_slab_dns_rhs / tlab_pencil_dns_rhs; the DAXPY / DSCAL calls of time.f90 that follow (tlab_deferred_axpy / _scal on the BOUND arrays of
 * the one local rank) complete the description to tlab_slab_dns_substep / tlab_pencil_dns_substep, zero fills to *_begin_step.  With several local
 * ranks in one process (loopback runs) or the layer off they execute at once. */
int tlab_deferred_slab_rhs(tlab_slab_dns_t d, double dte);
int tlab_deferred_pencil_rhs(tlab_pencil_dns_t d, double dte);
int tlab_slab_dns_set_remove_divergence(tlab_slab_dns_t d, int on);
/* as tlab_dns_set_surface_bcs: the dynamic surface model of the scalars on z-slabs.  The plane average of BOUNDARY_BCS_SURFACE_Y (AVG1V2D,
 * boundary_bcs.f90:520,535) is an all-reduce: the transport's allreduce is called with op = 2 (MPI_SUM) on the ranks' plane averages. */
int tlab_slab_dns_set_surface_bcs(tlab_slab_dns_t d, const int *sfc_jmin, const int *sfc_jmax, const double *cpl_jmin, const double *cpl_jmax);      /* as tlab_dns_set_remove_divergence ([Main] TermDivergence) */
/* RHS_GLOBAL_INCOMPRESSIBLE_1 on the bound arrays of all local ranks (tools/dns/rhs_global_incompressible_1.f90:98-398) */
int tlab_slab_dns_rhs(tlab_slab_dns_t d, double dte);
/* TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT (tools/dns/time.f90:559-664 + :261-298): RHS with the RK update folded into its last passes */
int tlab_slab_dns_substep(tlab_slab_dns_t d, double dte, double kco, int scale_tendencies);
/* TIME_COURANT with the MPI_MAX of time.f90:522: pmax[2] and dtime (may be NULL) are the same on every rank */
int tlab_slab_dns_time_courant(tlab_slab_dns_t d, double cfla, double cfld, double *pmax, double *dtime);
/* DNS_BOUNDS_CONTROL (tools/dns/dns_local.f90:157-187): extremes of div(q) over the whole box; destroys txc[0], txc[1] */
int tlab_slab_dns_dilatation_bounds(tlab_slab_dns_t d, double *dil_min, double *dil_max);

/* The pointwise loops of the RHS / RK update as separate calls (rhs_global_incompressible_1.f90:106-112, :197-201, :257-259,
 * :348-352, :279-280, :373-375; time.f90:645-664 + :272-297), for drivers that interleave communication. */
int tlab_pw_add3(double *h, const double *a, const double *b, const double *c, long long n);             /* h += a + b + c   */
int tlab_pw_axpy3(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3,
                  const double *q1, const double *q2, const double *q3, double s, long long n);          /* o = h + q*s, x3  */
int tlab_pw_sum3(double *a, const double *b, const double *c, long long n);                              /* a = a + b + c    */
int tlab_pw_sub3(double *h1, double *h2, double *h3, const double *a, const double *b, const double *c, long long n); /* h -= .., x3 */
int tlab_pw_rk_update(double *q, double *h, double dte, double kco, int scale, long long n);             /* q += dte h; h *= kco */
int tlab_pw_fill(double *a, double value, long long n);      /* a = value   (hq = 0.0_wp at the start of a Runge-Kutta step, time.f90:212-216) */
int tlab_pw_scale(double *a, double alpha, long long n);     /* a = alpha a (hq = kco hq between substeps, time.f90:272-297)                    */
/* fused tail of a substep for one field: h -= g (g may be NULL); wall planes of h = pb / pt (NULL = zeros); q += dte h; h *= kco if scale
 * (rhs_global_incompressible_1.f90:348-352, :373-375; time.f90:645-664, :272-297) */
int tlab_pw_final_update(double *q, double *h, const double *g, const double *pb, const double *pt, double dte, double kco, int scale,
                         int nx, int ny, int nz);
int tlab_pw_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz);
int tlab_pw_fill_wall_planes(double *f, double vb, double vt, int nx, int ny, int nz);
int tlab_pw_set_wall_planes(double *f, const double *pb, const double *pt, int nx, int ny, int nz);      /* NULL plane = zeros */

/* TLab_Transpose(a, nra, nca, ma, b, mb)   utils/tlab_transpose.f90:14-82 : b(j,i) = a(i,j), bit-exact */
int tlab_transpose(const double *a, int nra, int nca, double *b);

/* which kernel family the last operator call used: 0 none, 1 generic (any n), 2 wave-per-line (x),
 * 3 register-tile (y/z).  For tests / profiling. */
int tlab_last_kernel_path(void);
/* force a kernel family (0 = automatic). */
int tlab_force_kernel_path(int path);
/* tuning knobs for experiments and tests: key 1 = rows per wave of the register-tile kernel (16 | 32 | 64, 0 = automatic); 2 = policy of the half-wave
 * tile kernel (1 off, 2 forced, 0 automatic); 3 = lines per tile of the fused Burgers tiles (16 | 32); 4 = number of persistent workgroups of k_ptile
 * (0 = one per CU, rounded down to a multiple of 8; tests force counts that are not).  Environment switches read once per process (A/B runs):
 * TLAB_XLINE_OCC (1 = the one-wave-per-SIMD form of the fused x-Burgers kernel at 512 points), TLAB_PENCIL_OVERLAP (0 = literal operator sequence of the
 * pencil driver), TLAB_PENTA_TILE_X (0 = pentadiagonal x derivative through two transposes); the others are listed where they are read (csrc/). */
int tlab_set_tuning(int key, int value);

/* Live kernel timing (HIP events on the library's stream around every kernel launch) for bench.py's roofline object.
 * tlab_profile_report writes one line per kernel: "name<TAB>calls<TAB>total_ms<TAB>total_algorithmic_bytes". */
int tlab_profile_enable(int on);
/* tag != NULL and not empty: only launches with exactly this name are timed (two event records per timed launch cost the stream ~3 us each: with
 * every launch of the 512^3 substep timed, 0.2 ms of its 16); NULL or "": all of them. */
int tlab_profile_filter(const char *tag);
int tlab_profile_reset(void);
int tlab_profile_report(char *buf, int nbuf);

/* Debug aid, never on an operator path: runs the *device algorithm's* precomputed tables (chunked Thomas +
 * separator system) through a scalar host emulation so the tables can be checked without a GPU.
 * which = 1 (first derivative, variant ibc) or 2 (second derivative); chunks = number of chunks;
 * f (HOST, n doubles): right-hand side in, solution out. */
int tlab_debug_host_chunked_solve(tlab_fdm_plan_t p, int which, int ibc, int chunks, double *f);
/* Debug aid, never on an operator path: the HOST-factorized tables of the 3- / 7-diagonal first-order integral operators (FDM_Int1_Initialize for
 * SpaceOrder1 = CompactJacobian4 | CompactDirect4 | CompactJacobian6Penta, tlab_amd/csrc/int1_generic.cpp) of the y plan gy for the nm constants
 * lam (ibc: 1 BCS_MIN, 2 BCS_MAX; the sign convention is the caller's), so that they can be checked against the oracle without a GPU.  HOST buffers:
 * fac [nd][n][nm] (nd = RHS diagonals of the derivative), rb, rt [40][nm] (rhs_b(1:5, 0:7), rhs_t(0:4, 1:8)), R [n][LHS diagonals] row-major. */
int tlab_debug_int1_tables(tlab_fdm_plan_t gy, int ibc, int nm, const double *lam, double *fac, double *rb, double *rt, double *R);
/* ... and one FDM_Int1_Solve with them on the DEVICE (k_int1g), two lines per mode.  HOST buffers: f, res [2][n][nm] (res: the solution, boundary
 * values included), bv [2][nm] (the value given at the bottom, ibc = 1, or at the top, 2; the opposite end takes f's entry), du [2][nm].
 * variant (the kernel instantiations the plan builders use): 0 as described; 1: unit forcing (line 0: f = delta at the row opposite to the given
 * end, value 0; line 1: f = 0, value 1; f and bv ignored); 2: ibc = 2 with three lines (f's two and a zero one; values 0, 0, 1), res / du = lines 1, 2. */
/* Debug aid: the pack-layout map tlab_poisson_fft_x_packed works with (host arithmetic, no GPU): element (line, kx) of the complex slab sits at
 * off[kx] + line * width[kx] complex values of the pack buffer -- the layout tlab_pencil_repack_blocks writes for the same nblocks / start / base. */
int tlab_debug_pack_map(int nxh, int nblocks, const int *start, const long long *base, long long *off, int *width);
int tlab_debug_int1_solve(tlab_fdm_plan_t gy, int ibc, int variant, int nm, const double *lam, const double *f, const double *bv, double *res, double *du);

#ifdef __cplusplus
}
#endif
#endif /* TLAB_AMD_H */
