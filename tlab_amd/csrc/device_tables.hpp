// Plain-old-data descriptors handed to the kernels by value (they live in SGPRs / kernarg memory).
#pragma once

namespace tlab {

// RHS operator B of "A u' = B u" (fdm/fdm_matmul.f90).
//  antisymmetric (first derivative, MatMul_{3,5}d_antisym):  f = (u[+1]-u[-1]) + c2 (u[+2]-u[-2]) + c3 (u[+3]-u[-3])
//  symmetric     (second derivative, MatMul_{5,7}d_sym):     f = c0 u + (u[+1]+u[-1]) + c2 (u[+2]+u[-2]) + c3 (u[+3]+u[-3])
// Non-periodic: the first/last 3 rows are dense rows over u[0..5] / u[n-6..n-1] (biased or Neumann-reduced closures).
struct StencilDev {
    int sym;
    int periodic;
    double c0, c2, c3;
    double bb[3][6];
    double bt[3][6];
    // Direct schemes (fdm_comx_direct.f90, MatMul_5d fdm_matmul.f90:265-319): per-row pentadiagonal B.  rowc != NULL: [n][5] = r1..r5 of
    // every row, f = r1 u[-2] + r2 u[-1] + r3 u[0] + r4 u[+1] + r5 u[+2] in the interior, with r4 stored as exactly 1.0 where the
    // reference's loop omits the factor (rows 5 .. n-4, 1-based) and as its table value in rows 4 and n-3.  Direct FIRST derivatives use the
    // same form: MatMul_5d with the Neumann-reduced rows 4 / n-3 of the variant written into the table, MatMul_3d (:70-121) as (0, r1, r2, 1, 0).
    const double *rowc;
};

// Chunked tridiagonal system (chunked.hpp) on the device.
struct SystemDev {
    const double *rowtab;   // [5][n]: Lm, Dinv, Cm, V, W
    const double *red;      // wave-per-line kernel: [13][64] = k1[6][64], k2[6][64], dinv[64]; register-tile kernel: ginv[P][P]
    int lane_invariant;     // 1: every chunk has the same tables (circulant, uniform grid) -> scalar loads of chunk 0
    int chunk_invariant;    // 1: every INTERIOR chunk (1 .. P-2) has the tables of chunk 1 to the bit (uniform grid; walls only touch the first / last chunk)
    const double *band;     // register-tile kernels, P <= 32: ginv[c][(c + d) mod P], d = -2 .. 2, as [P][5] -- or NULL when the entries further from the
                            // diagonal are not negligible (they fall off like 0.38^(rows per chunk) per chunk for the compact schemes: 1e-27 at distance 2
                            // with 32-row chunks).  k_ptile keeps this form where the full matrices do not fit beside its tile.
};

// Jacobian correction of the second derivative on non-uniform grids: f += A2 dx2 du (MatMul_3d_add, fdm_matmul.f90:126-153)
struct JacCorrDev {
    const double *j;        // [3][n] : r1, r2, r3 ; NULL when not needed
};

enum KernelMode {
    MODE_P1 = 1,        // out0 = D1 u
    MODE_P2 = 2,        // out0 = D2 u                       (uniform grid)
    MODE_P2_P1 = 3,     // out0 = D2 u, out1 = D1 u          (uniform grid)
    MODE_BURGERS = 4,   // out0 = nu D2 s - vel D1 s         (uniform grid)
    MODE_P2_D1IN = 5,   // out0 = D2 u + Jc d1, d1 read from in1
    MODE_BURGERS_D1IN = 6  // out0 = nu (D2 s + Jc d1) - vel d1, d1 read from in1 (Jc may be NULL), vel from in2
};

// how a 3-D field is cut into lines along direction dir (x fastest: idx = i + nx*(j + ny*k))
struct LineGeom {
    int n;                  // points per line
    long long nlines;
    long long row_stride;   // elements between consecutive points of a line
    int lines_inner;        // number of lines contiguous in memory (X: 1 [lines are rows], Y: nx, Z: nx*ny)
    long long outer_stride; // elements between groups of lines_inner lines (Y: nx*ny; X: nx; Z: unused)
};

}  // namespace tlab
