// Kernel argument blocks and host launchers (implemented in kernels.hip, pointwise.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "device_tables.hpp"

namespace tlab {

struct XLineArgs {          // k_xline: derivative along the contiguous index, n = 64*M
    const double *in0;      // field (u or s)
    const double *in1;      // advecting velocity (MODE_BURGERS)
    double *out0, *out1;
    const double *in0b;     // optional: operand = in0 + in0b_scale * in0b   (tmp = hq + q/dte fused into the divergence, P1 only)
    double in0b_scale;
    int acc;                // 1: out0 += value instead of out0 = value (MODE_P1, MODE_BURGERS); 2: out0 -= value (MODE_P1)
    long long nlines;
    StencilDev s1, s2;      // first / second derivative RHS operators
    SystemDev y1, y2;       // first / second derivative chunked systems (P = 64)
    double nu;
    // MODE_BURGERS with several transported fields sharing the advecting velocity in1 (nf >= 1): field f reads fs[f], adds to (acc) or
    // overwrites fo[f] with fnu[f] d2 - in1 d1.  in0 / out0 / nu are ignored then.
    int nf;
    const double *fs[4];
    double *fo[4];
    double fnu[4];
    // MODE_P1 "final update" epilogue (fq != NULL): instead of storing d = D1 u, finish the substep of one velocity component with it:
    //   h = out0 (tendency) ; hv = h - d ; hv = 0 on the wall planes j = 0, fny-1 ; fq += fdte hv ; h = fscale ? fkco hv : hv
    // (rhs_global_incompressible_1.f90:348-352, :373-375 for Dirichlet walls; time.f90:645-664, :272-297)
    double *fq;
    double fdte, fkco;
    int fscale, fnx, fny;
    const double *fpb, *fpt;      // final-update epilogue: wall tendencies of the planes j = 0 / fny-1, [fnx][nz] each (NULL: zero, Dirichlet walls)
    // MODE_BURGERS, 8 rows per lane at most: ffin[f] != 0 finishes the substep of transported field f (a scalar: no pressure term) in the epilogue
    // of this launch, which must be the LAST one that adds to its tendency:  h = fo[f] (+ this term) ; h = 0 on the wall planes ;
    // fs[f] += fdte h ; fo[f] = fscale ? fkco h : h     (time.f90:645-664, :272-297; Dirichlet walls)
    int ffin[4];
    // ... and fdiv != NULL: for the field whose operand is the advecting velocity itself (u along x) the launch also writes the x term of the
    // pressure forcing, d/dx (h + fidte u) with the finished tendency h (rhs_global_incompressible_1.f90:197-230), into fdiv
    double *fdiv;
    double fidte;
    // MODE_BURGERS, anelastic (opr_burgers.f90:128-183, :504-507): the diffusion term of a line carries ribackground(j), j = line % ari_ny (x lines
    // run over (j, k)); ari == NULL: incompressible
    const double *ari;
    int ari_ny;
    int line_barriers;      // several waves per line and several lines per workgroup: the waves of a line meet at an LDS counter (xline_barrier); 0: workgroup barriers
};

struct RTileArgs {          // k_rtile: derivative along a strided index
    const double *in0;      // field
    const double *in1;      // first derivative (D1IN modes)
    const double *in2;      // advecting velocity (MODE_BURGERS_D1IN)
    double *out0;
    double *out1;           // first derivative (MODE_P2_P1 of k_htile)
    const double *in0b;     // optional second operand term: operand = in0 + in0b_scale * in0b (k_rtile MODE_P1)
    double in0b_scale;
    int acc;                // 1: out0 += value (k_rtile MODE_P1, k_htile MODE_BURGERS); 2: out0 -= value (MODE_P1)
    LineGeom g;
    StencilDev s1, s2;
    SystemDev y1, y2;       // chunked with P = n / rtile_chunk(n)  (k_htile: n / htile_chunk(n, mode))
    JacCorrDev jc;
    double nu;
    int nf;                 // k_htile MODE_BURGERS: as in XLineArgs (velocity = in2)
    const double *fs[4];
    double *fo[4];
    double fnu[4];
    // MODE_P1 "final update" epilogue (fq != NULL): instead of storing d = D1 u, finish the substep of one velocity component with it:
    //   h = out0 (tendency) ; hv = h - d ; hv = 0 on the wall planes j = 0, fny-1 ; fq += fdte hv ; h = fscale ? fkco hv : hv
    // (rhs_global_incompressible_1.f90:348-352, :373-375 for Dirichlet walls; time.f90:645-664, :272-297)
    double *fq;
    double fdte, fkco;
    int fscale, fnx, fny;
    const double *fpb, *fpt;      // final-update epilogue: wall tendencies of the planes j = 0 / fny-1, [fnx][nz] each (NULL: zero, Dirichlet walls)
    // k_htile MODE_BURGERS: bit f of fresh_mask set = field f OVERWRITES its tendency although acc is set (a field whose first term this launch
    // adds, in a launch that accumulates for the others)
    unsigned fresh_mask;
    // k_htile MODE_BURGERS with ONE field that is the advecting velocity itself (v along y, w along z), in the LAST launch that adds to its
    // tendency h: fdiv != NULL also ADDS this direction's term of the pressure forcing, d/dy (h + fidte v) (rhs_global_incompressible_1.f90:197-230),
    // to fdiv -- a third solve on the lines the workgroup holds anyway, instead of a separate pass that re-reads h and v
    double *fdiv;
    double fidte;
    // k_rtile MODE_P1 along y on the FINISHED tendency h (= in0) of a field with Neumann walls (fneu: bit 0 = jmin, bit 1 = jmax): the derivative of
    // the Neumann variant is not stored; its rows 1 / n-2 give the wall tendencies of BOUNDARY_BCS_NEUMANN_Y (boundary_bcs.f90:368-473; fcb / fct =
    // the three stencil coefficients and the LHS coefficient, as k_neumann_planes takes them), a Dirichlet side gets zero, and the final update of
    // k_final_update follows in the same launch: q (= fq) += fdte h, h = fscale ? fkco h : h.
    int fneu;
    double fcb[4], fct[4];
    // k_htile MODE_BURGERS, anelastic: ribackground [ari_ny] on the diffusion term -- ari_mode 1: lines along y, the factor of row j; 2: lines along z,
    // the factor of the tile's y row, j = (first line / ari_nx) % ari_ny (32-line tiles inside one row: ari_nx % 32 == 0)
    const double *ari;
    int ari_mode, ari_nx, ari_ny;
};

struct GenericArgs {        // k_generic: any n
    const double *in0;
    const double *in1;      // first derivative for the Jacobian correction (may be NULL)
    double *out0;
    LineGeom g;
    StencilDev s;
    SystemDev y;            // chunked with P = 1; red[0] = 1 / beta
    JacCorrDev jc;
};

// k_penta1: first derivative of CompactJacobian6Penta (pentadiagonal LHS, 7-diagonal antisymmetric RHS), one thread per line, the
// reference's own operation sequence: MatMul_7d_antisym (fdm_matmul.f90:491-558) + PENTADSS2 / PENTADPSS (utils/linear5.f90:207-411)
struct PentaArgs {
    const double *in0;
    double *out0;
    LineGeom g;
    const double *rhs;      // [7][n] device, column-major g%der1%rhs
    const double *lu;       // [7][n] periodic | [20][n] (4 Neumann variants x 5 columns), exactly g%der1%lu
    int periodic, ibc;
    double rb[4 * 8];       // rhs_b(4,0:7): [(j-1) + 4 c]
    double rt[5 * 7];       // rhs_t(0:4,7): [r + 5 (c-1)]
};
hipError_t launch_penta1(const PentaArgs &a, hipStream_t st);

// k_pentatile (pentatile.hip): the same operator on register tiles; rows / blocks / smw: device copies of what pentatile_build makes
struct PentaTileArgs {
    const double *in0;
    double *out0;
    LineGeom g;
    const double *rhs;      // as PentaArgs
    const double *rows;     // [9][n]
    const double *blocks;   // [2][C][C][4]
    const double *smw;      // [2][n] + 8 (periodic lines)
    double r6, r7;          // interior coefficients rhs(5, 6), rhs(5, 7)
    int periodic, ibc;
    double rb[4 * 8];
    double rt[5 * 7];
};
bool pentatile_ok(const LineGeom &g);
void pentatile_build(int n, int C, bool periodic, int ibc, const double *lu, std::vector<double> &rows, std::vector<double> &blocks, std::vector<double> &smw);
hipError_t launch_pentatile(const PentaTileArgs &a, hipStream_t st);
bool pentatile_x_ok(const LineGeom &g);                                   // lines along x: 16 lines per workgroup through an LDS tile
hipError_t launch_pentatile_x(const PentaTileArgs &a, hipStream_t st);

bool xline_supported(int n);
int rtile_chunk(int n);
void rtile_force_chunk(int m);
hipError_t launch_xline(int mode, int n, int chunks, bool lane_variant, const XLineArgs &a, hipStream_t st);
hipError_t launch_rtile(int mode, const RTileArgs &a, hipStream_t st);
int htile_chunk(int n, int mode);
void ptile_set_grid(int n);       // tests: forced grid of the persistent kernel (0 = default)
void htile_set_lines(int lines);   // tuning: 16 = narrow Burgers tiles (two workgroups per CU)
bool htile_narrow();
hipError_t launch_htile(int mode, const RTileArgs &a, hipStream_t st);
hipError_t launch_generic(bool sym, const GenericArgs &a, hipStream_t st);
hipError_t launch_burgers_epilogue(double *out, const double *vel, const double *d1, double nu, long long ntot, hipStream_t st);
hipError_t launch_fill(double *out, double v, long long ntot, hipStream_t st);
hipError_t launch_transpose(const double *a, double *b, int nra, int nca, hipStream_t st);

// pointwise.hip
hipError_t launch_add3(double *h, const double *a, const double *b, const double *c, long long n, hipStream_t st);
hipError_t launch_axpy3(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3, const double *q1,
                        const double *q2, const double *q3, double s, long long n, hipStream_t st);
hipError_t launch_axpy3w(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3, const double *q1,
                         const double *q2, const double *q3, double s, const double *w, int nx, int ny, long long n, hipStream_t st);
hipError_t launch_minmax_partial(const double *a, const double *v, const double *w, const double *odx, const double *ody, const double *odz,
                                 int mode, int nx, int ny, int nz, int koff, int zon, double *part, int nblocks, hipStream_t st);
hipError_t launch_negate(double *a, long long n, hipStream_t st);
hipError_t launch_scale(double *a, double alpha, long long n, hipStream_t st);
hipError_t launch_surface_flux(double *ref, const double *t, int j, int javg, double sign, double diff, double cpl, double *avg_scratch, int nx, int ny,
                               int nz, hipStream_t st);
hipError_t launch_weight_y(double *out, const double *in, const double *w, int nx, int ny, long long n, int mode, hipStream_t st);
hipError_t launch_burgers_epilogue_anelastic(double *out, const double *vel, const double *d1, double nu, const double *ri, int nx, int ny,
                                             long long n, hipStream_t st);
hipError_t launch_pencil_repack(double *a, double *buf, int nxh, int ny, int kmax, int nproc, const int *ioff, const long long *base, int dir,
                                hipStream_t st);
hipError_t launch_add1(double *h, const double *a, long long n, hipStream_t st);
hipError_t launch_axpy1(double *o, const double *a, const double *b, double s, long long n, hipStream_t st);
hipError_t launch_sum3(double *a, const double *b, const double *c, long long n, hipStream_t st);
hipError_t launch_sub3(double *h1, double *h2, double *h3, const double *a, const double *b, const double *c, long long n, hipStream_t st);
hipError_t launch_rk_update(double *q, double *h, double dte, double kco, int scale, long long n, hipStream_t st);
hipError_t launch_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz, hipStream_t st);
hipError_t launch_fill_wall_planes(double *f, double vb, double vt, int nx, int ny, int nz, hipStream_t st);
hipError_t launch_final_update(double *q, double *h, const double *g, const double *pb, const double *pt, double dte, double kco, int scale,
                               int nx, int ny, int nz, hipStream_t st, const double *gw = nullptr);
hipError_t launch_set_wall_planes(double *f, const double *pb, const double *pt, int nx, int ny, int nz, hipStream_t st);
hipError_t launch_wall_weighted(const double *a1, const double *a2, const double *wb, const double *wt, int K, double *ob1, double *ot1, double *ob2,
                                double *ot2, int nx, int ny, int nz, hipStream_t st);
hipError_t launch_wall_fix(double *q, double *h, const double *sb, const double *st, double dte, double kco, int scale, int nx, int ny, int nz,
                           hipStream_t stream);
hipError_t launch_sub2(double *o, const double *a, const double *b, long long n, hipStream_t st);
hipError_t launch_copy_blocks(int n, const double *const *src, double *const *dst, const long long *cnt, hipStream_t st);      // n device copies, batched launches
hipError_t launch_neumann_planes(const double *u, const double *du, const double *cb, const double *ct, int do_b, int do_t, double *hb,
                                 double *ht, int nx, int ny, int nz, hipStream_t st);

}  // namespace tlab
