// OPR_Poisson_FourierXZ_Factorize (operators/opr_elliptic.f90:263-364) on the MI355X.
//
//   p  --rocFFT r2c (x)-->  --rocFFT c2c (z)-->  f^(kx, j, kz)            (OPR_Fourier_X/Z_Forward, opr_fourier.f90:219,333)
//   per Fourier mode: (d/dy + l)(d/dy - l) p^ = f^,  l = sqrt(kx'^2 + kz'^2)  (OPR_ODE2_Factorize_NN, opr_odes.f90:265-386)
//   p^, dp^/dy --c2c (z)--> --c2r (x)--> p, dpdy                          (OPR_Fourier_Z/X_Backward)
//
// The reference transposes the spectral array so that each mode's y-line is contiguous and loops over modes on one
// core.  Here a mode is a THREAD: modes (kx fastest) lie across the lanes, y is the slow index, so every access is a
// coalesced 16-B-per-lane row and no transpose is needed.  Each first-order integral solve (FDM_Int1_Solve,
// fdm_integral.f90:219-314: tridiagonal matmul + pentadiagonal solve whose matrix B + l A depends on the mode) is one
// kernel: the pentadiagonal LU (PENTADFS, linear5.f90:30-71) is recomputed on the fly per mode instead of being
// stored (the reference stores 2 LUs per mode = 5.4 GB at 512^3), the forward-substituted lines and the three U
// factors per row go through a scratch array, the backward sweep reads them back.  The three homogeneous solutions
// the reference recomputes on every call (opr_odes.f90:308-324) depend only on the mode and are computed once at plan
// creation.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tlab_amd.h"
#include "plan.hpp"
#include "fftz.hpp"
#include "poisson_host.hpp"
#include "int1_generic.hpp"
#include "profile.hpp"

namespace tlab {

// ------------------------------------------------------------------------------------------------
// device tables
// ------------------------------------------------------------------------------------------------
struct Int1Dev {
    const double *L0, *L1, *R;   // row-major [n][5], [n][5], [n][3]
    double rb[3][4], rt[3][4];
    int n;
};

enum { FS_FIELD = 0, FS_LINEAR = 1, FS_UNIT = 2 };

struct Int1Args {
    Int1Dev T;
    const double *lam;      // [nm] |lambda| of each mode; the kernel applies the sign of its system
    double lam_sign;        // +1 (BCS_MIN system) or -1 (BCS_MAX system)
    long long nm;           // number of modes handled (threads)
    // f source
    const double *fsrc;     // FS_FIELD: complex field (nxh, ny, nz); FS_LINEAR: SoA [(l*n + j)*nm + t]
    int nlf;                // FS_LINEAR: number of stored lines (lines >= nlf are zero)
    int unit_row;           // FS_UNIT: row of the unit entry of line 0
    double fscale;          // FS_FIELD: normalisation 1/(nx*nz) folded into the load (opr_elliptic.f90:295)
    int nxh, ny;            // FS_FIELD layout
    int zero_bsave;         // 1: the f value saved as "opposite boundary value" is zero (f(:,nx)=0 / f(:,1)=0 in the callers)
    // given boundary value per line: constants, or per-mode array [(l*nm) + t] if bv_ptr != NULL
    double bv[3];
    const double *bv_ptr;
    // outputs
    double *scratch;        // SoA [(k*n + j)*nm + t], k < NL + 3
    double *dst;            // SoA [(l*n + j)*nm + t]
    double *du;             // [(l*nm) + t] or NULL
    double *bcs_save;       // FS_FIELD only: [(c*nm + t)], c = 0..3 = Re/Im at the bottom, Re/Im at the top (BC data)
    // LU factors of the modes, SoA [(k*n + j)*nm + t], k = 0..4 = a, b (forward), 1/c, -d, -e (backward), exactly as the elimination below
    // produces them: fac_out != NULL stores them (plan creation of the low-mode sub-plan), fac != NULL reads them instead of eliminating
    // (its per-call solves: the chain of dependent divisions is what a handful of marching threads spends its time on)
    double *fac_out;
    const double *fac;
    int fpart;              // SPLIT launches of k_int1 (one line per thread): which component of the complex FS_FIELD source this thread takes
    // 3- / 7-diagonal integral systems (int1_generic.cpp): everything factorized on the host, per mode -- g_fac [ndi][n][nm] (rows 2..n-1: the factors
    // of TRIDFS / HEPTADFS; rows 1, n: the reduced boundary rows), g_rb / g_rt [40][nm] (rhs_b(1:5, 0:7), rhs_t(0:4, 1:8)), g_R [n][nri].  g_fac != NULL
    // sends launch_int1 to k_int1g.
    const double *g_fac, *g_rb, *g_rt, *g_R;
    int g_ndi, g_nri;
};

// Non-fused arithmetic for everything that builds or factorizes the per-mode matrices: the reference's CPU build rounds every product and
// every sum, and the solution of these boundary-value problems is sensitive to the last bit of the matrix and of its LU factors (a table
// of the form L0 + lambda L1 with the row normalisation folded in, evaluated and eliminated with fused multiply-adds, sits 7-10x above the
// floor that one ulp of forcing noise sets: 4e-12 in p and 2e-11 in dp/dy on the 512-point lines of a projection step, measured).
__device__ __forceinline__ double nf_madd(double a, double b, double c) {   // a + b * c, two roundings
#pragma clang fp contract(off)
    const double t = b * c;
    return a + t;
}
__device__ __forceinline__ double nf_msub(double a, double b, double c) {   // a - b * c, two roundings
#pragma clang fp contract(off)
    const double t = b * c;
    return a - t;
}

// row j of lhs = (B + lambda A) * normalisation, in the operation order of FDM_Int1_CreateSystem (fdm_integral.f90:150-201); the
// normalisation of row j is stored behind the [n][5] block of L0
__device__ __forceinline__ void lhs_row(const Int1Dev &T, int j, double lam, double (&r)[5]) {
    const double sj = T.L0[5 * T.n + j];
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = nf_madd(T.L0[j * 5 + k], lam, T.L1[j * 5 + k]) * sj;
}

template <class TT>
__device__ __forceinline__ void lhs_row_t(const TT &T, int j, double lam, double (&r)[5]) {
    const double sj = T.L0[(unsigned)(5 * T.n + j)];
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = nf_madd(T.L0[(unsigned)(j * 5 + k)], lam, T.L1[(unsigned)(j * 5 + k)]) * sj;
}

template <int NL, int FS>
__device__ __forceinline__ void load_f(const Int1Args &a, int j, long long t, long long fidx0, double (&f)[NL]) {
    if (FS == FS_FIELD) {
        if (NL == 1 && a.fpart) {       // (SPLIT launch: the imaginary part alone)
            f[0] = reinterpret_cast<const double *>(a.fsrc)[2 * (fidx0 + (long long)j * a.nxh) + 1] * a.fscale;
            return;
        }
        const double2 v = reinterpret_cast<const double2 *>(a.fsrc)[fidx0 + (long long)j * a.nxh];
        f[0] = v.x * a.fscale;
        if (NL > 1) f[1] = v.y * a.fscale;
    } else if (FS == FS_LINEAR) {
#pragma unroll
        for (int l = 0; l < NL; ++l) f[l] = (l < a.nlf) ? a.fsrc[((long long)l * a.T.n + j) * a.nm + t] : 0.0;
    } else {
#pragma unroll
        for (int l = 0; l < NL; ++l) f[l] = (l == 0 && j == a.unit_row) ? 1.0 : 0.0;
    }
}

// One FDM_Int1_Solve per thread (mode).  BC = 1: value given at the bottom (BCS_MIN), BC = 2: at the top (BCS_MAX).
// SPLIT (with NL = 1): the two lines of a mode (real and imaginary part) on two threads, thread gid -> (mode gid % nm, line gid / nm).  The few
// modes of the low-mode sub-plan are a latency chain of n dependent rows bound by the instructions per row: half of them per thread.
// LDSV (with SPLIT): the few lines of the low-mode sub-plan are a chain of 2 n dependent rows whose every block of U rows waited for a round trip to
// memory -- 0.7 + 0.3 ms per substep at 512 rows beside a k_ode_nn that keeps the memory system busy, and on z-slabs / kx-pencils the critical path of
// the ranks that own the low kx.  Here a workgroup stages what its LV1 = 4 (2 from 1024 rows on) lines read in a sweep (source and forward factors, then
// the backward factors; the two right-hand-side coefficients) in LDS with all its threads, four lanes run the same recurrences on LDS operands (same
// expressions, same order: the results are the marching kernel's to the bit), and the intermediate of the forward sweep stays in LDS.
template <int BC, int NL, int FS, int U, bool STORED, bool SPLIT = false, bool LDSV = false, int LV1 = 4>
__global__ void __launch_bounds__(256) k_int1(Int1Args a) {
#pragma clang fp contract(off)
    static_assert(!SPLIT || (NL == 1 && STORED), "SPLIT: one line per thread, stored factors");
    static_assert(!LDSV || (SPLIT && FS != FS_UNIT), "LDSV: the low-mode form");
    extern __shared__ double s_i1[];
    const int n = a.T.n;
    const long long nm = a.nm;
    // LDS per line: four rows of n doubles -- forward sweep: source, a, b (forward factors), intermediate out; backward sweep: 1/c, -d, -e, intermediate
    // (16 KiB per line at 512 rows: a workgroup of four lines fits beside ONE workgroup of k_ode_nn on a CU, so it is scheduled while that kernel runs)
    double *s_b = s_i1, *s_R = s_b + LV1 * 4 * n;      // [LV1][4][n], [n][2]
    auto stage = [&](bool forward) {
        for (int idx = threadIdx.x; idx < LV1 * n; idx += blockDim.x) {
            const int k = idx / n, j = idx - k * n;
            const long long g = (long long)blockIdx.x * LV1 + k;
            if (g >= 2 * nm) continue;
            const long long tk = g % nm;
            const int pk = (int)(g / nm);
            if (forward) {
                double fv;
                if (FS == FS_FIELD) {
                    const long long f0 = (tk % a.nxh) + (long long)a.nxh * a.ny * (tk / a.nxh);
                    fv = a.fsrc[2 * (f0 + (long long)j * a.nxh) + pk];
                } else {
                    fv = (pk < a.nlf) ? a.fsrc[((long long)pk * n + j) * nm + tk] : 0.0;      // (line pk of the stored lines; lines >= nlf are zero)
                }
                s_b[(k * 4 + 0) * n + j] = fv;
                s_b[(k * 4 + 1) * n + j] = a.fac[((long long)0 * n + j) * nm + tk];
                s_b[(k * 4 + 2) * n + j] = a.fac[((long long)1 * n + j) * nm + tk];
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) s_b[(k * 4 + q) * n + j] = a.fac[((long long)(2 + q) * n + j) * nm + tk];
            }
        }
    };
    bool active = true;      // LDSV: lanes beyond the workgroup's lines (and beyond the last line) repeat the work of its first line and store nothing: every
                             // thread reaches the barriers between the sweeps
    if constexpr (LDSV) {
        stage(true);
        for (int idx = threadIdx.x; idx < n; idx += blockDim.x) { s_R[idx * 2] = a.T.R[idx * 3]; s_R[idx * 2 + 1] = a.T.R[idx * 3 + 1]; }
        __syncthreads();
        active = threadIdx.x < LV1 && (long long)blockIdx.x * LV1 + threadIdx.x < 2 * nm;
    }
    const int myk = (LDSV && active) ? (int)threadIdx.x : 0;
    const long long gid = LDSV ? (long long)blockIdx.x * LV1 + myk : (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!LDSV && gid >= (SPLIT ? 2 : 1) * a.nm) return;
    const long long t = SPLIT ? gid % a.nm : gid;
    const int part = SPLIT ? (int)(gid / a.nm) : 0;
    if (SPLIT && part) {      // line 1 of every array becomes this thread's line 0
        a.scratch += (long long)n * nm;
        a.dst += (long long)n * nm;
        if (a.du) a.du += nm;
        if (a.bv_ptr) a.bv_ptr += nm;
        a.bv[0] = a.bv[1];
        if (FS == FS_LINEAR) { a.fsrc += (long long)n * nm; a.nlf -= 1; }
        a.fpart = 1;
    }
    const double lam = a.lam_sign * a.lam[t];
    const long long fidx0 = (FS == FS_FIELD) ? (t % a.nxh) + (long long)a.nxh * a.ny * (t / a.nxh) : 0;
    auto ldf = [&](int j, double (&f)[NL]) {      // row j of the source: from LDS (LDSV: the staged raw value, scaled / masked as load_f does) or from memory
        if constexpr (LDSV) {
            if (FS == FS_FIELD) f[0] = s_b[(myk * 4 + 0) * n + j] * a.fscale;
            else f[0] = (0 < a.nlf) ? s_b[(myk * 4 + 0) * n + j] : 0.0;
        } else {
            load_f<NL, FS>(a, j, t, fidx0, f);
        }
    };

    // ---- boundary rows of the system of this mode (fdm_integral.f90:203-211 -> FDM_Bcs_Reduce at the opposite end) ----
    double l0[5], l1[5], l2[5], lN[5], lN1[5], lN2[5], rb[3][4], rt[3][4];
    lhs_row(a.T, 0, lam, l0); lhs_row(a.T, 1, lam, l1); lhs_row(a.T, 2, lam, l2);
    lhs_row(a.T, n - 1, lam, lN); lhs_row(a.T, n - 2, lam, lN1); lhs_row(a.T, n - 3, lam, lN2);
    if (BC == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) rb[j][c] = a.T.rb[j][c];
        const double d = 1.0 / lN[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) lN[k] = -lN[k] * d;
        lN[2] = 1.0;
        lN1[0] = nf_madd(lN1[0], lN1[3], lN[4]); lN1[1] = nf_madd(lN1[1], lN1[3], lN[0]); lN1[2] = nf_madd(lN1[2], lN1[3], lN[1]);
        lN2[1] = nf_madd(lN2[1], lN2[4], lN[4]); lN2[2] = nf_madd(lN2[2], lN2[4], lN[0]); lN2[3] = nf_madd(lN2[3], lN2[4], lN[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rt[2][c] = a.T.R[(n - 1) * 3 + c] * d;
            rt[1][c] = a.T.R[(n - 2) * 3 + c];
            rt[0][c] = a.T.R[(n - 3) * 3 + c];
        }
        rt[0][3] = rt[1][3] = rt[2][3] = 0.0;
        rt[1][0] = nf_msub(rt[1][0], lN1[3], rt[2][2]); rt[1][1] = nf_msub(rt[1][1], lN1[3], rt[2][0]); rt[1][2] = nf_msub(rt[1][2], lN1[3], rt[2][1]);
        rt[0][1] = nf_msub(rt[0][1], lN2[4], rt[2][2]); rt[0][2] = nf_msub(rt[0][2], lN2[4], rt[2][0]); rt[0][3] = nf_msub(rt[0][3], lN2[4], rt[2][1]);
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) rt[j][c] = a.T.rt[j][c];
        const double d = 1.0 / l0[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) l0[k] = -l0[k] * d;
        l0[2] = 1.0;
        l1[2] = nf_madd(l1[2], l1[1], l0[3]); l1[3] = nf_madd(l1[3], l1[1], l0[4]); l1[4] = nf_madd(l1[4], l1[1], l0[0]);
        l2[1] = nf_madd(l2[1], l2[0], l0[3]); l2[2] = nf_madd(l2[2], l2[0], l0[4]); l2[3] = nf_madd(l2[3], l2[0], l0[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rb[0][c + 1] = a.T.R[0 * 3 + c] * d;
            rb[1][c + 1] = a.T.R[1 * 3 + c];
            rb[2][c + 1] = a.T.R[2 * 3 + c];
        }
        rb[0][0] = rb[1][0] = rb[2][0] = 0.0;
        rb[1][1] = nf_msub(rb[1][1], l1[1], rb[0][2]); rb[1][2] = nf_msub(rb[1][2], l1[1], rb[0][3]); rb[1][3] = nf_msub(rb[1][3], l1[1], rb[0][1]);
        rb[2][0] = nf_msub(rb[2][0], l2[0], rb[0][2]); rb[2][1] = nf_msub(rb[2][1], l2[0], rb[0][3]); rb[2][2] = nf_msub(rb[2][2], l2[0], rb[0][1]);
    }

    // ---- boundary values: res0 (row 0) and resN (row n-1) as MatMul_3d sees them (fdm_integral.f90:240-245) ----
    double fb0[NL], fbN[NL], res0[NL], resN[NL];
    ldf(0, fb0);
    ldf(n - 1, fbN);
    if (FS == FS_FIELD && a.bcs_save != nullptr) {  // Neumann data travel in the forcing planes (opr_elliptic.f90:285-286,310-311)
        if (SPLIT) {
            if (active) {
                a.bcs_save[(long long)part * nm + t] = fb0[0];
                a.bcs_save[(long long)(2 + part) * nm + t] = fbN[0];
            }
        } else {
            a.bcs_save[0 * nm + t] = fb0[0]; a.bcs_save[1 * nm + t] = fb0[NL > 1 ? 1 : 0];
            a.bcs_save[2 * nm + t] = fbN[0]; a.bcs_save[3 * nm + t] = fbN[NL > 1 ? 1 : 0];
        }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double given = a.bv_ptr ? a.bv_ptr[(long long)l * nm + t] : a.bv[l];
        if (BC == 1) { res0[l] = given; resN[l] = a.zero_bsave ? 0.0 : fbN[l]; }
        else { resN[l] = given; res0[l] = a.zero_bsave ? 0.0 : fb0[l]; }
    }

    // ---- forward: right-hand side (MatMul_3d, BCS_BOTH), LU on the fly (PENTADFS), forward substitution (PENTADSS) ----
    double fm[NL], fc[NL], fp[NL];           // f[j-1], f[j], f[j+1]
    ldf(1, fc);
    ldf(2, fp);
    double f1[NL], fn2[NL];                   // f[1] and f[n-2] are needed again for du
    double bcs_b[NL], bcs_t[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        f1[l] = fc[l];
        bcs_b[l] = res0[l] * rb[0][2] + fc[l] * rb[0][3] + fp[l] * rb[0][1];
        fm[l] = 0.0;
    }
    double c1 = 0.0, c2 = 0.0, d1 = 0.0, d2 = 0.0, e1 = 0.0, e2 = 0.0;  // pivots of rows m-1, m-2
    double y1[NL], y2[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) y1[l] = y2[l] = 0.0;
    const int nmax = n - 2;
    // U = rows per block: the loads of a block are issued together so that only one memory latency is exposed per U rows.  Large U
    // pays on small slabs (few modes -> few waves -> latency-bound), small U keeps the registers down when the grid fills the chip.
    constexpr bool stored = STORED;          // a.fac != nullptr (launch_int1): the factors of every row are read instead of regenerated
    for (int jb = 1; jb <= nmax; jb += U) {
        double fqb[U][NL], fab[U][2];         // f[jb+2 .. jb+U+1]; stored forward factors of rows jb .. jb+U-1
        double Rb[U][2];                      // right-hand-side coefficients of the rows of the block: requested with the rest, BEFORE the first store
        //                                       of the block (the output arrays may alias the tables as far as the compiler knows: left inside the row loop,
        //                                       every row waited for its own scalar load -- 7 us per block of 8 rows with few modes in flight)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jc = (jb + u <= nmax) ? jb + u : nmax;
            if constexpr (LDSV) { Rb[u][0] = s_R[jc * 2 + 0]; Rb[u][1] = s_R[jc * 2 + 1]; }
            else { Rb[u][0] = a.T.R[jc * 3 + 0]; Rb[u][1] = a.T.R[jc * 3 + 1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jr = jb + u + 2;
            if (jr <= n - 1) ldf(jr, fqb[u]);
            else {
#pragma unroll
                for (int l = 0; l < NL; ++l) fqb[u][l] = 0.0;
            }
            fab[u][0] = fab[u][1] = 0.0;
            if (stored) {
                const int jf = (jb + u <= nmax) ? jb + u : nmax;
                if constexpr (LDSV) { fab[u][0] = s_b[(myk * 4 + 1) * n + jf]; fab[u][1] = s_b[(myk * 4 + 2) * n + jf]; }
                else {
                    fab[u][0] = a.fac[((long long)0 * n + jf) * nm + t];
                    fab[u][1] = a.fac[((long long)1 * n + jf) * nm + t];
                }
            }
        }
        // A block without one of the boundary rows 1, 2, n-3, n-2 (all but the first and the last one or two) takes the plain form of every
        // expression: with few modes in flight (the low-mode sub-plan: 2 waves) the kernel is bound by the instructions per row, and the
        // row-number selects of the general form are most of them.
        const bool edge_blk = jb < 3 || jb + U - 1 > n - 4;
        auto fwd_row = [&](auto edge_c, int u) {
            constexpr bool EDGE = decltype(edge_c)::value;
            const int j = jb + u;
            if (EDGE && j > nmax) return;
            double r[5] = {0.0, 0.0, 1.0, 0.0, 0.0};
            if (!stored) {
                if (EDGE && j == 1) { for (int k = 0; k < 5; ++k) r[k] = l1[k]; }
                else if (EDGE && j == 2) { for (int k = 0; k < 5; ++k) r[k] = l2[k]; }
                else if (EDGE && j == n - 3) { for (int k = 0; k < 5; ++k) r[k] = lN2[k]; }
                else if (EDGE && j == n - 2) { for (int k = 0; k < 5; ++k) r[k] = lN1[k]; }
                else lhs_row(a.T, j, lam, r);
            }
            // right-hand side of row j
            double rhs[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (EDGE && j == 1) rhs[l] = res0[l] * rb[1][1] + fc[l] * rb[1][2] + fp[l] * rb[1][3];
                else if (EDGE && j == 2) rhs[l] = res0[l] * rb[2][0] + fm[l] * rb[2][1] + fc[l] * rb[2][2] + fp[l] * rb[2][3];
                else if (EDGE && j == n - 3) rhs[l] = fm[l] * rt[0][0] + fc[l] * rt[0][1] + fp[l] * rt[0][2] + resN[l] * rt[0][3];
                else if (EDGE && j == n - 2) rhs[l] = fm[l] * rt[1][0] + fc[l] * rt[1][1] + resN[l] * rt[1][2];
                else rhs[l] = fm[l] * Rb[u][0] + fc[l] * Rb[u][1] + fp[l];
            }
            if (EDGE && j == n - 2) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    fn2[l] = fc[l];
                    bcs_t[l] = fm[l] * rt[2][2] + fc[l] * rt[2][0] + resN[l] * rt[2][1];
                }
            }
            // PENTADFS row m = j
            double am = 0.0, bm = 0.0, cm = r[2], dm = r[3], em = r[4], cinv = 1.0;
            if (stored) {
                am = fab[u][0]; bm = fab[u][1];
            } else {
                if (EDGE && j == 2) {
                    bm = r[1] / c1;
                    cm = nf_msub(r[2], bm, d1);
                    dm = nf_msub(r[3], bm, e1);
                } else if (!EDGE || j >= 3) {
                    am = r[0] / c2;
                    bm = nf_msub(r[1], am, d2) / c1;
                    cm = nf_msub(nf_msub(r[2], bm, d1), am, e2);
                    dm = nf_msub(r[3], bm, e1);
                }
                cinv = 1.0 / cm;
            }
            // PENTADSS forward: f(n) = f(n) + f(n-1)*b(n) + f(n-2)*a(n) with a, b negated
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double y = rhs[l] - y1[l] * bm - y2[l] * am;
                if constexpr (LDSV) s_b[(myk * 4 + 3) * n + j] = y;      // (the lanes that repeat line 0 write the same value)
                else a.scratch[((long long)l * n + j) * nm + t] = y;
                y2[l] = y1[l];
                y1[l] = y;
            }
            if (!stored) {
                a.scratch[((long long)(NL + 0) * n + j) * nm + t] = cinv;
                a.scratch[((long long)(NL + 1) * n + j) * nm + t] = -dm;
                a.scratch[((long long)(NL + 2) * n + j) * nm + t] = -em;
                if (a.fac_out) {
                    a.fac_out[((long long)0 * n + j) * nm + t] = am; a.fac_out[((long long)1 * n + j) * nm + t] = bm;
                    a.fac_out[((long long)2 * n + j) * nm + t] = cinv; a.fac_out[((long long)3 * n + j) * nm + t] = -dm;
                    a.fac_out[((long long)4 * n + j) * nm + t] = -em;
                }
            }
            c2 = c1; d2 = d1; e2 = e1;
            c1 = cm; d1 = dm; e1 = em;
#pragma unroll
            for (int l = 0; l < NL; ++l) { fm[l] = fc[l]; fc[l] = fp[l]; fp[l] = fqb[u][l]; }
        };
        if (edge_blk) {
#pragma unroll
            for (int u = 0; u < U; ++u) fwd_row(std::true_type{}, u);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) fwd_row(std::false_type{}, u);
        }
    }

    if constexpr (LDSV) {      // the backward factors take the place of the source and the forward factors
        __syncthreads();
        stage(false);
        __syncthreads();
    }
    // ---- backward substitution ----
    double x1[NL], x2[NL];                    // x[j+1], x[j+2]
    double xs1[NL], xs2[NL], xs3[NL];         // x[1], x[2], x[3]
    double xe2[NL], xe3[NL], xe4[NL];         // x[n-2], x[n-3], x[n-4]
#pragma unroll
    for (int l = 0; l < NL; ++l) x1[l] = x2[l] = xs1[l] = xs2[l] = xs3[l] = xe2[l] = xe3[l] = xe4[l] = 0.0;
    const double *fsrc = stored ? a.fac + (long long)2 * n * nm : a.scratch + (long long)NL * n * nm;      // 1/c, -d, -e of every row
    for (int jb = nmax; jb >= 1; jb -= U) {
        double yb[U][NL], cb[U], db[U], eb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb - u;
            const int jr = j >= 1 ? j : 1;
            if constexpr (LDSV) {
                cb[u] = s_b[(myk * 4 + 0) * n + jr]; db[u] = s_b[(myk * 4 + 1) * n + jr]; eb[u] = s_b[(myk * 4 + 2) * n + jr];
                yb[u][0] = s_b[(myk * 4 + 3) * n + jr];
            } else {
                cb[u] = fsrc[((long long)0 * n + jr) * nm + t];
                db[u] = fsrc[((long long)1 * n + jr) * nm + t];
                eb[u] = fsrc[((long long)2 * n + jr) * nm + t];
#pragma unroll
                for (int l = 0; l < NL; ++l) yb[u][l] = a.scratch[((long long)l * n + jr) * nm + t];
            }
        }
        const bool edge_blk = jb > n - 5 || jb - U + 1 < 4;       // holds one of the rows 1, 2, 3, n-4, n-3, n-2 (kept for the boundary formulas), or runs past row 1
        auto bwd_row = [&](auto edge_c, int u) {
            constexpr bool EDGE = decltype(edge_c)::value;
            const int j = jb - u;
            if (EDGE && j < 1) return;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double x = (yb[u][l] + x1[l] * db[u] + x2[l] * eb[u]) * cb[u];
                if (active) a.dst[((long long)l * n + j) * nm + t] = x;
                x2[l] = x1[l];
                x1[l] = x;
                if (EDGE) {
                    if (j == 1) xs1[l] = x;
                    if (j == 2) xs2[l] = x;
                    if (j == 3) xs3[l] = x;
                    if (j == n - 2) xe2[l] = x;
                    if (j == n - 3) xe3[l] = x;
                    if (j == n - 4) xe4[l] = x;
                }
            }
        };
        if (edge_blk) {
#pragma unroll
            for (int u = 0; u < U; ++u) bwd_row(std::true_type{}, u);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) bwd_row(std::false_type{}, u);
        }
    }

    // ---- boundary value at the free end and derivative at the given end (fdm_integral.f90:265-311) ----
    if (!active) return;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (BC == 2) {
            const double r0 = bcs_b[l] + l0[3] * xs1[l] + l0[4] * xs2[l] + l0[0] * xs3[l];
            a.dst[((long long)l * n + 0) * nm + t] = r0;
            a.dst[((long long)l * n + (n - 1)) * nm + t] = resN[l];
            if (a.du) a.du[(long long)l * nm + t] = lN[2] * resN[l] + lN[1] * xe2[l] + lN[0] * xe3[l] + lN[4] * xe4[l] + a.T.R[(n - 1) * 3 + 0] * fn2[l];
        } else {
            const double rN = bcs_t[l] + lN[1] * xe2[l] + lN[0] * xe3[l] + lN[4] * xe4[l];
            a.dst[((long long)l * n + (n - 1)) * nm + t] = rN;
            a.dst[((long long)l * n + 0) * nm + t] = res0[l];
            if (a.du) a.du[(long long)l * nm + t] = l0[2] * res0[l] + l0[3] * xs1[l] + l0[4] * xs2[l] + l0[0] * xs3[l] + a.T.R[0 * 3 + 2] * f1[l];
        }
    }
}

// ================================================================================================
// k_int1g : FDM_Int1_Solve (fdm/fdm_integral.f90:219-314) for the 3- and 7-diagonal integral systems of SpaceOrder1 = CompactJacobian4 /
// CompactDirect4 / CompactJacobian6Penta, factorized on the host (int1_generic.cpp).  One thread per mode, the reference's operations in the
// reference's order, no fused multiply-adds: right-hand side (MatMul_3d / MatMul_5d with BCS_BOTH, fdm_matmul.f90:70-121 / :267-320), substitution
// (TRIDSS utils/linear3.f90:56-150 / HEPTADSS utils/linear7.f90:98-142), value at the free end and derivative at the given one (:265-311).
// Nobody selects these schemes with the factorized solver: correctness first, every operand re-read where it is used.
// Compiled at -O3 like everything else since round 4.  Rounds 2-3 carried __attribute__((optnone)) here because "-O3 gave O(1) errors that vanished when a
// printf was added".  Root cause (tools/repro/int1g_O3.hip, a stand-alone reduction: the same function body on host and device): hipcc 7.2's loop
// unroller mis-transforms the two substitution loops below -- loops whose first three / last three iterations take other branches and whose
// iterations communicate through memory -- at -O2 and -O3; -O1, -O0 and -fno-unroll-loops give the host's bits, the host replay is clean under
// AddressSanitizer and UBSan (no undefined behaviour in the source), fences and volatile accesses change nothing.  `#pragma clang loop
// unroll(disable)` on those two loops is the whole work-around; every instantiation in use is bitwise equal to the oracle at -O3
// (tests/test_gpu_poisson.py, tlab_debug_int1_solve variants 0-2).
template <int BC, int NL, int FS, int NDI>
__global__ void __launch_bounds__(256) k_int1g(Int1Args a) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nm) return;
    constexpr int ndi = NDI, nri = NDI == 3 ? 3 : 5, idl = ndi / 2 + 1, idr = nri / 2 + 1;      // (3, 3): CompactJacobian4 / Direct4; (7, 5): CompactJacobian6Penta
    const int n = a.T.n;
    const long long nm = a.nm;
    const long long fidx0 = (FS == FS_FIELD) ? (t % a.nxh) + (long long)a.nxh * a.ny * (t / a.nxh) : 0;
    auto F = [&](int k, int j) { return a.g_fac[((long long)k * n + j) * nm + t]; };                 // diagonal k (0-based) of row j (0-based)
    auto RB = [&](int j1, int c) { return a.g_rb[(long long)((j1 - 1) + 5 * c) * nm + t]; };          // rhs_b(j1, c)
    auto RT = [&](int r, int c1) { return a.g_rt[(long long)(r + 5 * (c1 - 1)) * nm + t]; };          // rhs_t(r, c1)
    auto Rr = [&](int j, int k1) { return a.g_R[j * nri + (k1 - 1)]; };                               // rhs(j+1, k1)
    auto fv = [&](int j, int l) -> double {                                                            // f(l, j+1): one value, no private array
        if (FS == FS_FIELD) return reinterpret_cast<const double *>(a.fsrc)[2 * (fidx0 + (long long)j * a.nxh) + l] * a.fscale;
        if (FS == FS_LINEAR) return (l < a.nlf) ? a.fsrc[((long long)l * n + j) * nm + t] : 0.0;
        return (l == 0 && j == a.unit_row) ? 1.0 : 0.0;
    };
    double res0[NL], resN[NL];
    {
        double fb0[NL], fbN[NL];
        load_f<NL, FS>(a, 0, t, fidx0, fb0);
        load_f<NL, FS>(a, n - 1, t, fidx0, fbN);
        if (FS == FS_FIELD && a.bcs_save != nullptr) {
            a.bcs_save[0 * nm + t] = fb0[0]; a.bcs_save[1 * nm + t] = fb0[NL > 1 ? 1 : 0];
            a.bcs_save[2 * nm + t] = fbN[0]; a.bcs_save[3 * nm + t] = fbN[NL > 1 ? 1 : 0];
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double given = a.bv_ptr ? a.bv_ptr[(long long)l * nm + t] : a.bv[l];
            if (BC == 1) { res0[l] = given; resN[l] = a.zero_bsave ? 0.0 : fbN[l]; }
            else { resN[l] = given; res0[l] = a.zero_bsave ? 0.0 : fb0[l]; }
        }
    }
    const int nmax = n - 2;                       // the systems are those of rows 2 .. n-1; sub-row m <-> row j = m + 1 (0-based)
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        // ---- right-hand side of row j (0-based) ----
        auto rhs_row = [&](int j) -> double {
            if (nri == 3) {
                if (j == 1) return res0[l] * RB(2, 1) + fv(1, l) * RB(2, 2) + fv(2, l) * RB(2, 3);
                if (j == 2) return res0[l] * RB(3, 0) + fv(1, l) * RB(3, 1) + fv(2, l) * RB(3, 2) + fv(3, l) * RB(3, 3);
                if (j == n - 3) return fv(n - 4, l) * RT(0, 1) + fv(n - 3, l) * RT(0, 2) + fv(n - 2, l) * RT(0, 3) + resN[l] * RT(0, 4);
                if (j == n - 2) return fv(n - 3, l) * RT(1, 1) + fv(n - 2, l) * RT(1, 2) + resN[l] * RT(1, 3);
                return fv(j - 1, l) * Rr(j, 1) + fv(j, l) * Rr(j, 2) + fv(j + 1, l);
            }
            if (j == 1) return res0[l] * RB(2, 2) + fv(1, l) * RB(2, 3) + fv(2, l) * RB(2, 4) + fv(3, l) * RB(2, 5);
            if (j == 2) return res0[l] * RB(3, 1) + fv(1, l) * RB(3, 2) + fv(2, l) * RB(3, 3) + fv(3, l) * RB(3, 4) + fv(4, l) * RB(3, 5);
            if (j == 3) return res0[l] * RB(4, 0) + fv(1, l) * RB(4, 1) + fv(2, l) * RB(4, 2) + fv(3, l) * RB(4, 3) + fv(4, l) * RB(4, 4) + fv(5, l) * RB(4, 5);
            if (j == n - 4)
                return fv(n - 6, l) * RT(0, 1) + fv(n - 5, l) * RT(0, 2) + fv(n - 4, l) * RT(0, 3) + fv(n - 3, l) * RT(0, 4) + fv(n - 2, l) * RT(0, 5) +
                       resN[l] * RT(0, 6);
            if (j == n - 3) return fv(n - 5, l) * RT(1, 1) + fv(n - 4, l) * RT(1, 2) + fv(n - 3, l) * RT(1, 3) + fv(n - 2, l) * RT(1, 4) + resN[l] * RT(1, 5);
            if (j == n - 2) return fv(n - 4, l) * RT(2, 1) + fv(n - 3, l) * RT(2, 2) + fv(n - 2, l) * RT(2, 3) + resN[l] * RT(2, 4);
            return fv(j - 2, l) * Rr(j, 1) + fv(j - 1, l) * Rr(j, 2) + fv(j, l) * Rr(j, 3) + fv(j + 1, l) + fv(j + 2, l) * Rr(j, 5);
        };
        double bcs_b, bcs_t;
        if (nri == 3) {
            bcs_b = res0[l] * RB(1, 2) + fv(1, l) * RB(1, 3) + fv(2, l) * RB(1, 1);
            bcs_t = fv(n - 3, l) * RT(2, 3) + fv(n - 2, l) * RT(2, 1) + resN[l] * RT(2, 2);
        } else {
            bcs_b = res0[l] * RB(1, 3) + fv(1, l) * RB(1, 4) + fv(2, l) * RB(1, 5) + fv(3, l) * RB(1, 1);
            bcs_t = fv(n - 4, l) * RT(3, 5) + fv(n - 3, l) * RT(3, 1) + fv(n - 2, l) * RT(3, 2) + resN[l] * RT(3, 3);
        }
        double *y = a.scratch + (long long)l * n * nm + t;                   // y(j) at y[j * nm]
        double *x = a.dst + (long long)l * n * nm + t;
        // ---- forward substitution (the rows before come back from memory: same thread, program order) ----
        auto Y = [&](int j) { return y[(long long)j * nm]; };
#pragma clang loop unroll(disable)      // hipcc 7.2 miscompiles these two loops when its loop unroller peels them: tools/repro/int1g_O3.hip
        for (int m = 0; m < nmax; ++m) {
            const int j = m + 1;
            const double r = rhs_row(j);
            double v;
            if constexpr (NDI == 3) {
                v = m == 0 ? r : r + F(0, j) * Y(j - 1);                      // f(n) = f(n) + a(n) f(n-1)
            } else {
                if (m == 0) v = r * F(2, j);                                  // normalise the first equation (c(1) = 1 / d(1), HEPTADFS)
                else if (m == 1) v = r - Y(j - 1) * F(2, j);
                else if (m == 2) v = r - Y(j - 1) * F(2, j) - Y(j - 2) * F(1, j);
                else v = r - Y(j - 1) * F(2, j) - Y(j - 2) * F(1, j) - Y(j - 3) * F(0, j);
            }
            y[(long long)j * nm] = v;
        }
        // ---- backward substitution ----
        auto XX = [&](int j) { return x[(long long)j * nm]; };
#pragma clang loop unroll(disable)
        for (int m = nmax - 1; m >= 0; --m) {
            const int j = m + 1;
            const double yv = Y(j);
            double v;
            if constexpr (NDI == 3) {
                v = m == nmax - 1 ? yv * F(1, j) : (yv + F(2, j) * XX(j + 1)) * F(1, j);
            } else {
                if (m == nmax - 1) v = yv / F(3, j);
                else if (m == nmax - 2) v = (yv - XX(j + 1) * F(4, j)) / F(3, j);
                else if (m == nmax - 3) v = (yv - XX(j + 1) * F(4, j) - XX(j + 2) * F(5, j)) / F(3, j);
                else v = (yv - XX(j + 1) * F(4, j) - XX(j + 2) * F(5, j) - XX(j + 3) * F(6, j)) / F(3, j);
            }
            x[(long long)j * nm] = v;
        }
        // ---- value at the free end, derivative at the given end (fdm_integral.f90:265-311); idl: centre of the integral system ----
        auto X = [&](int j) { return x[(long long)j * nm]; };
        if (BC == 2) {
            double r0 = bcs_b;
            for (int ic = 1; ic <= idl - 1; ++ic) r0 = r0 + F(idl + ic - 1, 0) * X(ic);
            r0 = r0 + F(0, 0) * X(idl);
            x[0] = r0;
            x[(long long)(n - 1) * nm] = resN[l];
            if (a.du) {
                double du = F(idl - 1, n - 1) * resN[l];
                for (int ic = 1; ic <= idl - 1; ++ic) du = du + F(idl - ic - 1, n - 1) * X(n - 1 - ic);
                du = du + F(ndi - 1, n - 1) * X(n - 1 - idl);
                for (int ic = 1; ic <= idr - 1; ++ic) du = du + Rr(n - 1, idr - ic) * fv(n - 1 - ic, l);
                a.du[(long long)l * nm + t] = du;
            }
        } else {
            double rN = bcs_t;
            for (int ic = 1; ic <= idl - 1; ++ic) rN = rN + F(idl - ic - 1, n - 1) * X(n - 1 - ic);
            rN = rN + F(ndi - 1, n - 1) * X(n - 1 - idl);
            x[(long long)(n - 1) * nm] = rN;
            x[0] = res0[l];
            if (a.du) {
                double du = F(idl - 1, 0) * res0[l];
                for (int ic = 1; ic <= idl - 1; ++ic) du = du + F(idl + ic - 1, 0) * X(ic);
                du = du + F(0, 0) * X(idl);
                for (int ic = 1; ic <= idr - 1; ++ic) du = du + Rr(0, idr + ic) * fv(ic, l);
                a.du[(long long)l * nm + t] = du;
            }
        }
    }
}

// ================================================================================================
// k_int2 : FDM_Int2_Solve of the DIRECT elliptic solver (EllipticOrder = CompactDirect6; OPR_Poisson_FourierXZ_Direct,
// opr_elliptic.f90:368-455): ONE pentadiagonal solve per Fourier mode, (B - lambda2 A) p^ = A f^ with both boundary data in the wall
// planes of f^.  Same marching scheme as k_int1 (thread = mode, LU of the mode regenerated on the fly, forward-substituted line and U
// factors through a scratch array), and the solution goes straight into the spectral field p^ (no superposition stage).  The Neumann
// problem amplifies last-bit differences of the matrix entries and of the elimination to 1e-12 .. 5e-12 in p (measured with an affine
// table and fused multiply-adds), so this kernel repeats the reference's operations in the reference's order with FP contraction OFF:
// given the same f^ it returns the same bits as FDM_Int2_Initialize + FDM_Int2_Solve on the CPU.
// ================================================================================================
struct Int2Dev {
    const double *Bt, *A5, *s, *R;   // row-major [n][5], [n][5], [n], [n][3] (poisson_host.hpp)
    double rb[3][4], rt[3][4];
    double c1[3], e1, nb[2], cn[3], en, nt[2];
    int n;
};

struct Int2Args {
    Int2Dev T;
    const double *lam;      // [nm] lambda2 = mwn2_x + mwn2_z of each mode
    double alpha;           // Helmholtz: the system constant is lambda2 - alpha (opr_elliptic.f90:604); 0 for Poisson
    long long nm;           // modes of the spectral box
    long long first, count; // threads cover modes [first, first + count)
    long long skip;         // mode left out (the singular one, solved by its own launch with the BCS_DN tables), or -1
    const double *fsrc;     // complex field (nxh, ny, nz)
    double *dst;            // complex field (nxh, ny, nz); may alias fsrc (a thread reads its whole column before it writes it)
    double fscale;          // 1/(nx*nz) (opr_elliptic.f90:402)
    int nxh, ny;
    int zero_bottom;        // compatibility constraint of the singular mode: p = 0 at the bottom (opr_elliptic.f90:420-421)
    int neumann_b, neumann_t;
    double *scratch;        // SoA [(k*n + j)*nm + t], k < 5
};

template <int U>
__global__ void __launch_bounds__(256) k_int2(Int2Args a) {
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.count) return;
    const long long t = a.first + q;
    if (t == a.skip) return;
    const int n = a.T.n;
    const long long nm = a.nm;
    const double lam = a.lam[t] - a.alpha;
    const long long fidx0 = (t % a.nxh) + (long long)a.nxh * a.ny * (t / a.nxh);
    const double2 *__restrict__ fs = reinterpret_cast<const double2 *>(a.fsrc);
    auto loadf = [&](int j, double (&f)[2]) {
        const double2 v = fs[fidx0 + (long long)j * a.nxh];
        f[0] = v.x * a.fscale; f[1] = v.y * a.fscale;
    };
    // rows 1 and n of the Neumann system (fdm_integral.f90:446-452, 485-491)
    double l1[3] = {a.T.c1[0], a.T.c1[1], a.T.c1[2]}, lN[3] = {a.T.cn[0], a.T.cn[1], a.T.cn[2]};
    l1[0] = l1[0] + lam * a.T.e1;
    lN[2] = lN[2] + lam * a.T.en;
    double res0[2], resN[2], fm[2] = {0.0, 0.0}, fc[2], fp[2];
    loadf(0, res0); loadf(n - 1, resN);                       // u(1:2) = f(1:2), u(2ny-1:2ny) = f(...) (opr_elliptic.f90:416-417)
    if (a.zero_bottom) res0[0] = res0[1] = 0.0;
    loadf(1, fc); loadf(2, fp);
    double bcs_b[2], bcs_t[2] = {0.0, 0.0};
#pragma unroll
    for (int l = 0; l < 2; ++l) bcs_b[l] = res0[l] * a.T.rb[0][2] + fc[l] * a.T.rb[0][3] + fp[l] * a.T.rb[0][1];   // MatMul_3d, BCS_BOTH

    // ---- forward: right-hand side (MatMul_3d), LU on the fly (PENTADFS), forward substitution (PENTADSS) ----
    double c1 = 0.0, c2 = 0.0, d1 = 0.0, d2 = 0.0, e1 = 0.0, e2 = 0.0;
    double y1[2] = {0.0, 0.0}, y2[2] = {0.0, 0.0};
    const int nmax = n - 2;
    for (int jb = 1; jb <= nmax; jb += U) {
        double fqb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jr = jb + u + 2;
            if (jr <= n - 1) loadf(jr, fqb[u]);
            else fqb[u][0] = fqb[u][1] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u;
            if (j > nmax) break;
            double r[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) r[k] = a.T.Bt[j * 5 + k] - lam * a.T.A5[j * 5 + k];      // :412-432
            if (a.neumann_b && (j == 1 || j == 2)) {       // rows 2, 3: lhs(1+ir, idr-ir+1 : idr-ir+3) -= rhs_b(1+ir, idl-ir) * lhs(1, 1:3)  (:462-464)
                const int k0 = 3 - j;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) r[k0 + qq] = r[k0 + qq] - a.T.nb[j - 1] * l1[qq];
            }
            if (a.neumann_t && (j == n - 2 || j == n - 3)) {   // rows n-1, n-2: lhs(nx-ir, ir : ir+2) -= rhs_t(idl-ir, idl+ir) * lhs(nx, ndr-2:ndr)  (:501-503)
                const int ir = n - 1 - j;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) r[ir - 1 + qq] = r[ir - 1 + qq] - a.T.nt[ir - 1] * lN[qq];
            }
            {
                const double sj = a.T.s[j];                  // :518-540
#pragma unroll
                for (int k = 0; k < 5; ++k) r[k] = r[k] * sj;
            }
            double rhs[2];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                if (j == 1) rhs[l] = res0[l] * a.T.rb[1][1] + fc[l] * a.T.rb[1][2] + fp[l] * a.T.rb[1][3];
                else if (j == 2) rhs[l] = res0[l] * a.T.rb[2][0] + fm[l] * a.T.rb[2][1] + fc[l] * a.T.rb[2][2] + fp[l] * a.T.rb[2][3];
                else if (j == n - 3) rhs[l] = fm[l] * a.T.rt[0][0] + fc[l] * a.T.rt[0][1] + fp[l] * a.T.rt[0][2] + resN[l] * a.T.rt[0][3];
                else if (j == n - 2) rhs[l] = fm[l] * a.T.rt[1][0] + fc[l] * a.T.rt[1][1] + resN[l] * a.T.rt[1][2];
                else rhs[l] = fm[l] * a.T.R[j * 3 + 0] + fc[l] * a.T.R[j * 3 + 1] + fp[l];
            }
            if (j == n - 2) {
#pragma unroll
                for (int l = 0; l < 2; ++l) bcs_t[l] = fm[l] * a.T.rt[2][2] + fc[l] * a.T.rt[2][0] + resN[l] * a.T.rt[2][1];
            }
            double am = 0.0, bm = 0.0, cm = r[2], dm = r[3], em = r[4];
            if (j == 2) {
                bm = r[1] / c1;
                cm = r[2] - bm * d1;
                dm = r[3] - bm * e1;
            } else if (j >= 3) {
                am = r[0] / c2;
                bm = (r[1] - am * d2) / c1;
                cm = r[2] - bm * d1 - am * e2;
                dm = r[3] - bm * e1;
            }
            const double cinv = 1.0 / cm;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const double y = rhs[l] - y1[l] * bm - y2[l] * am;
                a.scratch[((long long)l * n + j) * nm + t] = y;
                y2[l] = y1[l];
                y1[l] = y;
            }
            a.scratch[((long long)2 * n + j) * nm + t] = cinv;
            a.scratch[((long long)3 * n + j) * nm + t] = -dm;
            a.scratch[((long long)4 * n + j) * nm + t] = -em;
            c2 = c1; d2 = d1; e2 = e1;
            c1 = cm; d1 = dm; e1 = em;
#pragma unroll
            for (int l = 0; l < 2; ++l) { fm[l] = fc[l]; fc[l] = fp[l]; fp[l] = fqb[u][l]; }
        }
    }

    // ---- backward substitution, straight into the spectral field ----
    double2 *__restrict__ ds = reinterpret_cast<double2 *>(a.dst);
    double x1[2] = {0.0, 0.0}, x2[2] = {0.0, 0.0};
    double xs[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};     // x[1..3]
    double xe[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};     // x[n-2], x[n-3], x[n-4]
    for (int jb = nmax; jb >= 1; jb -= U) {
        double yb[U][2], cb[U], db[U], eb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb - u;
            const int jr = j >= 1 ? j : 1;
            cb[u] = a.scratch[((long long)2 * n + jr) * nm + t];
            db[u] = a.scratch[((long long)3 * n + jr) * nm + t];
            eb[u] = a.scratch[((long long)4 * n + jr) * nm + t];
            yb[u][0] = a.scratch[((long long)0 * n + jr) * nm + t];
            yb[u][1] = a.scratch[((long long)1 * n + jr) * nm + t];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb - u;
            if (j < 1) break;
            double x[2];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                x[l] = (yb[u][l] + x1[l] * db[u] + x2[l] * eb[u]) * cb[u];
                x2[l] = x1[l];
                x1[l] = x[l];
                if (j <= 3) xs[j - 1][l] = x[l];
                if (j >= n - 4) xe[n - 2 - j][l] = x[l];
            }
            ds[fidx0 + (long long)j * a.nxh] = make_double2(x[0], x[1]);
        }
    }

    // ---- end values: given (Dirichlet) or from the biased first-derivative formula (Neumann), fdm_integral.f90:659-668 ----
    double r0[2], rN[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) { r0[l] = res0[l]; rN[l] = resN[l]; }
    if (a.neumann_b) {
#pragma unroll
        for (int l = 0; l < 2; ++l) r0[l] = bcs_b[l] + l1[0] * xs[0][l] + l1[1] * xs[1][l] + l1[2] * xs[2][l];
    }
    if (a.neumann_t) {
#pragma unroll
        for (int l = 0; l < 2; ++l) rN[l] = bcs_t[l] + lN[2] * xe[0][l] + lN[1] * xe[1][l] + lN[0] * xe[2][l];
    }
    ds[fidx0] = make_double2(r0[0], r0[1]);
    ds[fidx0 + (long long)(n - 1) * a.nxh] = make_double2(rN[0], rN[1]);
}


// ================================================================================================
// k_ode_nn : OPR_ODE2_Factorize_NN for a group of modes with the y-line cut into chunks of 8 rows that live in registers.
//
// k_int1 marches one thread per mode along the whole line: 512 dependent steps, twice, with every intermediate of the
// pentadiagonal solve (5 doubles per row) written to and read back from HBM -- latency-bound on small slabs, traffic-bound
// (22 GB per Poisson solve at 512^3) on large ones.  Here a workgroup owns NM modes x all rows, thread (m, c) owns rows
// [8c, 8c+8) of mode m:
//   * the LU factors of its rows are regenerated from a CHECKPOINT of the PENTADFS recurrence (pivots of the two rows before
//     the chunk, 6 doubles per chunk and mode, written once at plan creation by k_ode_checkpoint): the same numbers the serial
//     elimination produces (the system B + lambda A is not diagonally dominant -- partitioned eliminations with their own
//     local pivots lose up to 6 digits for small lambda, measured -- so the serial pivot sequence is kept);
//   * forward and backward substitution are two-term linear recurrences: every chunk computes its particular end values and its
//     2x2 transfer matrix, a parallel scan over the chunks (lane shuffles inside a wave, wave totals through LDS) gives every
//     chunk its inflow, and the chunk repeats its 8 rows with it;
//   * v0, u0 stay in registers until the three constants of the 3x3 constraint system are known, and the superposition with the
//     homogeneous solutions of the mode (5 arrays computed at plan creation, opr_odes.f90:350-367) is the epilogue of the same kernel.
//     (Running the pair of solves a second time with the final boundary values instead of reading them was measured: the kernel is
//     bound by dependent fp64 latency at 8 waves per CU, not by HBM, and the second pass doubled its time.)
// HBM traffic per mode and row: f^ 16 B, p^ + dp^/dy 32 B, homogeneous solutions 40 B, checkpoints 12 B; no scratch.
// ================================================================================================
// Knock-out builds (-DODE_KO=<bits>: 1 no workgroup barriers, 2 no scan steps, 4 divisions as multiplications, 8 no homogeneous-solution loads, 16 no
// table loads): wrong results, right timings -- what DESIGN.md section 4 quotes for the parts of k_ode_nn (profiles/r04/ode_knockout.txt).  0: the product.
#ifndef ODE_KO
#define ODE_KO 0
#endif
#define ODE_SYNC() do { if (!(ODE_KO & 1)) __syncthreads(); } while (0)
#define ODE_DIV(x, y) ((ODE_KO & 4) ? (x) * (y) : (x) / (y))
constexpr int OM = 8;   // rows per thread

struct OdeSys {                  // Int1Dev without the by-value boundary constants (they would sit in ~100 SGPRs)
    const double *L0, *L1, *R;   // row-major [n][5], [n][5], [n][3]
    const double *pk;            // the same numbers packed per row, [n][16] = L0[5], L1[5], row scale, R[3], 0, 0: one 128-B line and seven 16-B loads
                                 // per row where the separate arrays take 13 loads from four lines (k_ode_nn's per-row loads were a fifth of its time)
    const double *bt;            // [3][4]: rhs_b of the BCS_MIN system / rhs_t of the BCS_MAX system
    int n;
};

struct OdeArgs {
    OdeSys T1, T2;               // BCS_MIN (+lambda) and BCS_MAX (-lambda) tables
    const double *lam;           // [nm]
    const unsigned char *skip;   // [nm] singular modes: computed elsewhere
    const double *chk1, *chk2;   // [C][6][nm] PENTADFS state before the first row of each chunk
    const double *cst;           // [9][nm] LU of the constraint matrix (k_nn_constants)
    const double *hom;           // [5][n][nm] homogeneous solutions v1, em, u1, sp, ep (build_homogeneous)
    const int *band;             // [2][nm] (may be NULL): rows (jb, jt) exclusive where all five homogeneous solutions of the mode are negligible
    int pair_xcd;                // see k_ode_nn
    const double *f_hat;
    double *p_hat, *dp_hat;
    double fscale;
    int n, nxh, ny, C;
    long long nm;
};

// boundary rows of the system of one mode (the prologue of k_int1)
struct OdeRows {
    double l0[5], l1[5], l2[5], lN[5], lN1[5], lN2[5], rb[3][4], rt[3][4];
};

// lhs_row_t and R(j, 1:3) of row j from the packed table (OdeSys::pk): the same numbers by the same operations
__device__ __forceinline__ void ode_row_pk(const OdeSys &T, int j, double lam, double (&r)[5], double (&R)[3]) {
    const double2 *pk = reinterpret_cast<const double2 *>(T.pk) + (unsigned)(j * 8);
    const double2 q0 = pk[0], q1 = pk[1], q2 = pk[2], q3 = pk[3], q4 = pk[4], q5 = pk[5], q6 = pk[6];
    const double sj = q5.x;
    r[0] = nf_madd(q0.x, lam, q2.y) * sj; r[1] = nf_madd(q0.y, lam, q3.x) * sj; r[2] = nf_madd(q1.x, lam, q3.y) * sj;
    r[3] = nf_madd(q1.y, lam, q4.x) * sj; r[4] = nf_madd(q2.x, lam, q4.y) * sj;
    R[0] = q5.y; R[1] = q6.x; R[2] = q6.y;
}

// (the two ends are independent of each other: a caller that stores one end only -- ode_solve -- pays for that end only)
template <int BC>
__device__ __forceinline__ void ode_boundary_rows(const OdeSys &T, double lam, OdeRows &k) {
    const int n = T.n;
    double R0[3], R1[3], R2[3], RN[3], RN1[3], RN2[3];
    ode_row_pk(T, 0, lam, k.l0, R0); ode_row_pk(T, 1, lam, k.l1, R1); ode_row_pk(T, 2, lam, k.l2, R2);
    ode_row_pk(T, n - 1, lam, k.lN, RN); ode_row_pk(T, n - 2, lam, k.lN1, RN1); ode_row_pk(T, n - 3, lam, k.lN2, RN2);
    if (BC == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) k.rb[j][c] = T.bt[j * 4 + c];
        const double d = 1.0 / k.lN[2];
#pragma unroll
        for (int q = 0; q < 5; ++q) k.lN[q] = -k.lN[q] * d;
        k.lN[2] = 1.0;
        k.lN1[0] = nf_madd(k.lN1[0], k.lN1[3], k.lN[4]); k.lN1[1] = nf_madd(k.lN1[1], k.lN1[3], k.lN[0]); k.lN1[2] = nf_madd(k.lN1[2], k.lN1[3], k.lN[1]);
        k.lN2[1] = nf_madd(k.lN2[1], k.lN2[4], k.lN[4]); k.lN2[2] = nf_madd(k.lN2[2], k.lN2[4], k.lN[0]); k.lN2[3] = nf_madd(k.lN2[3], k.lN2[4], k.lN[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            k.rt[2][c] = RN[c] * d;
            k.rt[1][c] = RN1[c];
            k.rt[0][c] = RN2[c];
        }
        k.rt[0][3] = k.rt[1][3] = k.rt[2][3] = 0.0;
        k.rt[1][0] = nf_msub(k.rt[1][0], k.lN1[3], k.rt[2][2]); k.rt[1][1] = nf_msub(k.rt[1][1], k.lN1[3], k.rt[2][0]); k.rt[1][2] = nf_msub(k.rt[1][2], k.lN1[3], k.rt[2][1]);
        k.rt[0][1] = nf_msub(k.rt[0][1], k.lN2[4], k.rt[2][2]); k.rt[0][2] = nf_msub(k.rt[0][2], k.lN2[4], k.rt[2][0]); k.rt[0][3] = nf_msub(k.rt[0][3], k.lN2[4], k.rt[2][1]);
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) k.rt[j][c] = T.bt[j * 4 + c];
        const double d = 1.0 / k.l0[2];
#pragma unroll
        for (int q = 0; q < 5; ++q) k.l0[q] = -k.l0[q] * d;
        k.l0[2] = 1.0;
        k.l1[2] = nf_madd(k.l1[2], k.l1[1], k.l0[3]); k.l1[3] = nf_madd(k.l1[3], k.l1[1], k.l0[4]); k.l1[4] = nf_madd(k.l1[4], k.l1[1], k.l0[0]);
        k.l2[1] = nf_madd(k.l2[1], k.l2[0], k.l0[3]); k.l2[2] = nf_madd(k.l2[2], k.l2[0], k.l0[4]); k.l2[3] = nf_madd(k.l2[3], k.l2[0], k.l0[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            k.rb[0][c + 1] = R0[c] * d;
            k.rb[1][c + 1] = R1[c];
            k.rb[2][c + 1] = R2[c];
        }
        k.rb[0][0] = k.rb[1][0] = k.rb[2][0] = 0.0;
        k.rb[1][1] = nf_msub(k.rb[1][1], k.l1[1], k.rb[0][2]); k.rb[1][2] = nf_msub(k.rb[1][2], k.l1[1], k.rb[0][3]); k.rb[1][3] = nf_msub(k.rb[1][3], k.l1[1], k.rb[0][1]);
        k.rb[2][0] = nf_msub(k.rb[2][0], k.l2[0], k.rb[0][2]); k.rb[2][1] = nf_msub(k.rb[2][1], k.l2[0], k.rb[0][3]); k.rb[2][2] = nf_msub(k.rb[2][2], k.l2[0], k.rb[0][1]);
    }
}

// matrix row j of the reduced system of one mode
__device__ __forceinline__ void ode_row(const OdeSys &T, const OdeRows &k, int j, double lam, double (&r)[5]) {
    const int n = T.n;
    if (j == 1) { for (int q = 0; q < 5; ++q) r[q] = k.l1[q]; }
    else if (j == 2) { for (int q = 0; q < 5; ++q) r[q] = k.l2[q]; }
    else if (j == n - 3) { for (int q = 0; q < 5; ++q) r[q] = k.lN2[q]; }
    else if (j == n - 2) { for (int q = 0; q < 5; ++q) r[q] = k.lN1[q]; }
    else lhs_row_t(T, j, lam, r);
}

// The boundary rows depend on the mode only: one thread per mode computes them into LDS (54 doubles per mode), the two chunks that
// touch a boundary read what they need from there, and no thread keeps them in registers.  Layout: [field][NM], fields:
//   0-4 l0, 5-9 l1, 10-14 l2, 15-19 lN, 20-24 lN1, 25-29 lN2, 30-41 rb[3][4], 42-53 rt[3][4]
constexpr int OK_L0 = 0, OK_L1 = 5, OK_L2 = 10, OK_LN = 15, OK_LN1 = 20, OK_LN2 = 25, OK_RB = 30, OK_RT = 42, OK_FS = 54, OK_CST = 58, OK_BAND = 67, OK_SIZE = 70;      // OK_FS: one f row per line, parked by the chunk that needs it after the solve; OK_CST, OK_BAND: the mode's constants and band (k_ode_nn: fetched at the start)
template <int NM, int END = 0>      // END = 1 / 2: the rows of the bottom / the top only
__device__ __forceinline__ void ode_rows_to_lds(const OdeRows &k, double *s_k, int m) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (END != 2) { s_k[(OK_L0 + q) * NM + m] = k.l0[q]; s_k[(OK_L1 + q) * NM + m] = k.l1[q]; s_k[(OK_L2 + q) * NM + m] = k.l2[q]; }
        if (END != 1) { s_k[(OK_LN + q) * NM + m] = k.lN[q]; s_k[(OK_LN1 + q) * NM + m] = k.lN1[q]; s_k[(OK_LN2 + q) * NM + m] = k.lN2[q]; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (END != 2) s_k[(OK_RB + j * 4 + c) * NM + m] = k.rb[j][c];
            if (END != 1) s_k[(OK_RT + j * 4 + c) * NM + m] = k.rt[j][c];
        }
}
template <int NM>
__device__ __forceinline__ void ode_row_lds(const OdeSys &T, const double *s_k, int m, int j, double lam, double (&r)[5]) {
    const int n = T.n;
    const int off = (j == 1) ? OK_L1 : (j == 2) ? OK_L2 : (j == n - 3) ? OK_LN2 : (j == n - 2) ? OK_LN1 : -1;
    if (off >= 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = s_k[(off + q) * NM + m];
    } else {
        lhs_row_t(T, j, lam, r);
    }
}

// one PENTADFS step (linear5.f90:30-71) for row j; st = (c1, d1, e1, c2, d2, e2) of rows j-1, j-2
__device__ __forceinline__ void ode_factor_step(int j, const double (&r)[5], double (&st)[6], double &am, double &bm, double &cinv, double &nd,
                                                double &ne) {
    double cm = r[2], dm = r[3];
    const double em = r[4];
    am = 0.0; bm = 0.0;
    if (j == 2) {
        bm = r[1] / st[0];
        cm = nf_msub(r[2], bm, st[1]);
        dm = nf_msub(r[3], bm, st[2]);
    } else if (j >= 3) {
        am = r[0] / st[3];
        bm = nf_msub(r[1], am, st[4]) / st[0];
        cm = nf_msub(nf_msub(r[2], bm, st[1]), am, st[5]);
        dm = nf_msub(r[3], bm, st[2]);
    }
    cinv = 1.0 / cm; nd = -dm; ne = -em;
    st[3] = st[0]; st[4] = st[1]; st[5] = st[2];
    st[0] = cm; st[1] = dm; st[2] = em;
}

// checkpoints of the factor recurrence: state before rows 8, 16, ... (chunk 0 starts from zeros)
template <int BC>
// Layout of everything k_ode_nn reads per mode: blocked by the NM modes of a workgroup, [block][...][NM], so that a workgroup's reads are
// one contiguous stream (mode-minor [..][nm] rows would be 64-B pieces of 128-B lines at NM = 8: measured 2x over-fetch).
__global__ void __launch_bounds__(256) k_ode_checkpoint(OdeSys T, const double *__restrict__ lamv, double lam_sign, double *__restrict__ chk,
                                                        long long nm, int NM, int C, int om) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nm) return;
    const int n = T.n;
    const double lam = lam_sign * lamv[t];
    OdeRows k;
    ode_boundary_rows<BC>(T, lam, k);
    double st[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 1; j <= n - 2; ++j) {
        if ((j % om) == 0) {
            const int c = j / om;
#pragma unroll
            for (int q = 0; q < 6; ++q) chk[(((t / NM) * C + c) * 6 + q) * NM + (t % NM)] = st[q];
        }
        double r[5], am, bm, cinv, nd, ne;
        ode_row(T, k, j, lam, r);
        ode_factor_step(j, r, st, am, bm, cinv, nd, ne);
    }
}

// Inflow of every chunk from the chunks before it (DIR = +1: c-1, c-2, ... ; DIR = -1: c+1, c+2, ...), i.e. the exclusive prefix of the
// affine maps in -> Phi in + e of the chunks, composed in the direction of the sweep.  Lanes hold (mode m, chunk c) with m fastest, so a
// wave owns 64/NM consecutive chunks of NM modes: Hillis-Steele with lane shuffles inside the wave, the wave totals through LDS.
//   phi = {p00, p01, p10, p11}, e[l] = {e1, e2} per line; returns in[l] = {in1, in2}.     s_w: [nwaves][4 + 2 NL][NM] doubles
template <int NM, int DIR, int NL = 2>
__device__ __forceinline__ void ode_chain(double (&phi)[4], double (&e)[NL][2], int c, int C, int m, double *s_w, double (&in)[NL][2]) {
    constexpr int SW = 4 + 2 * NL;                     // doubles per wave total: phi, then (e1, e2) of every line
    constexpr int CPW = 64 / NM;                       // chunks per wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cw = lane / NM;                          // chunk index inside the wave
    // position along the sweep inside the wave: DIR = +1 -> cw, DIR = -1 -> reversed
#pragma unroll
    for (int d = 1; d < ((ODE_KO & 2) ? 1 : CPW); d <<= 1) {
        double q[4], f[NL][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = (DIR > 0) ? __shfl_up(phi[k], d * NM) : __shfl_down(phi[k], d * NM);
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int k = 0; k < 2; ++k) f[l][k] = (DIR > 0) ? __shfl_up(e[l][k], d * NM) : __shfl_down(e[l][k], d * NM);
        const bool has = (DIR > 0) ? (cw >= d) : (cw + d < CPW && c + d < C);
        if (has) {      // (phi, e) <- (phi * q, phi * f + e): the partner's chunks come first in the sweep
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double n1 = phi[0] * f[l][0] + phi[1] * f[l][1] + e[l][0];
                const double n2 = phi[2] * f[l][0] + phi[3] * f[l][1] + e[l][1];
                e[l][0] = n1; e[l][1] = n2;
            }
            const double r00 = phi[0] * q[0] + phi[1] * q[2], r01 = phi[0] * q[1] + phi[1] * q[3];
            const double r10 = phi[2] * q[0] + phi[3] * q[2], r11 = phi[2] * q[1] + phi[3] * q[3];
            phi[0] = r00; phi[1] = r01; phi[2] = r10; phi[3] = r11;
        }
    }
    // wave totals = the inclusive value of the last chunk of the wave along the sweep
    const bool last_in_wave = (DIR > 0) ? (cw == CPW - 1 || c == C - 1) : (cw == 0);
    if (last_in_wave) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_w[(w * SW + k) * NM + m] = phi[k];
#pragma unroll
        for (int l = 0; l < NL; ++l) { s_w[(w * SW + 4 + 2 * l) * NM + m] = e[l][0]; s_w[(w * SW + 5 + 2 * l) * NM + m] = e[l][1]; }
    }
    ODE_SYNC();
    // what enters my wave: the waves before it along the sweep, composed in order
    const int nw = (blockDim.x + 63) >> 6;
    double pe[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) pe[l][0] = pe[l][1] = 0.0;
    if (DIR > 0) {
        for (int v = 0; v < w; ++v) {
            const double a0 = s_w[(v * SW + 0) * NM + m], a1 = s_w[(v * SW + 1) * NM + m], a2 = s_w[(v * SW + 2) * NM + m], a3 = s_w[(v * SW + 3) * NM + m];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double n1 = a0 * pe[l][0] + a1 * pe[l][1] + s_w[(v * SW + 4 + 2 * l) * NM + m];
                const double n2 = a2 * pe[l][0] + a3 * pe[l][1] + s_w[(v * SW + 5 + 2 * l) * NM + m];
                pe[l][0] = n1; pe[l][1] = n2;
            }
        }
    } else {
        for (int v = nw - 1; v > w; --v) {
            const double a0 = s_w[(v * SW + 0) * NM + m], a1 = s_w[(v * SW + 1) * NM + m], a2 = s_w[(v * SW + 2) * NM + m], a3 = s_w[(v * SW + 3) * NM + m];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double n1 = a0 * pe[l][0] + a1 * pe[l][1] + s_w[(v * SW + 4 + 2 * l) * NM + m];
                const double n2 = a2 * pe[l][0] + a3 * pe[l][1] + s_w[(v * SW + 5 + 2 * l) * NM + m];
                pe[l][0] = n1; pe[l][1] = n2;
            }
        }
    }
    // inclusive value of my chunk over the whole line, then the previous chunk's along the sweep = my inflow
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double f1 = phi[0] * pe[l][0] + phi[1] * pe[l][1] + e[l][0];
        const double f2 = phi[2] * pe[l][0] + phi[3] * pe[l][1] + e[l][1];
        const double g1 = (DIR > 0) ? __shfl_up(f1, NM) : __shfl_down(f1, NM);
        const double g2 = (DIR > 0) ? __shfl_up(f2, NM) : __shfl_down(f2, NM);
        const bool first_in_wave = (DIR > 0) ? (cw == 0) : (cw == CPW - 1 || c == C - 1);
        in[l][0] = first_in_wave ? pe[l][0] : g1;
        in[l][1] = first_in_wave ? pe[l][1] : g2;
    }
    ODE_SYNC();      // s_w is reused by the next scan
}

// Per mode: the rows between which all five homogeneous solutions are below 1e-40 of their own maximum, found from the middle of the line
// outwards (band[t] = last significant row of the lower half, band[nm + t] = first one of the upper half).  hom: [5][n][nm].
__global__ void __launch_bounds__(256) k_ode_hom_band(const double *__restrict__ hom, int n, long long nm, int *__restrict__ band, double rel) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nm) return;
    double thr[5];
    for (int a = 0; a < 5; ++a) {
        double mx = 0.0;
        for (int j = 0; j < n; ++j) mx = fmax(mx, fabs(hom[((size_t)a * n + j) * nm + t]));
        thr[a] = mx * rel;
    }
    const int mid = n / 2;
    int jb = -1, jt = n;
    for (int j = 0; j < n; ++j) {
        bool sig = false;
        for (int a = 0; a < 5; ++a) sig = sig || !(fabs(hom[((size_t)a * n + j) * nm + t]) <= thr[a]);      // NaN counts as significant
        if (sig && j < mid) jb = j;
        if (sig && j >= mid && j < jt) jt = j;
    }
    band[t] = jb;
    band[nm + t] = jt;
}

// src[a][j][nm] -> dst[blk][a][j][NM]
// (the five solutions of a mode and row side by side, [blk][j][NM][6] with three 16-B loads per row in k_ode_nn, was measured: 3 % slower)
__global__ void __launch_bounds__(256) k_ode_block_layout(const double *__restrict__ src, double *__restrict__ dst, int A, int n, long long nm, int NM) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)A * n * nm) return;
    const long long t = i % nm, aj = i / nm;          // aj = a * n + j
    dst[((t / NM) * A * n + aj) * NM + (t % NM)] = src[i];
}

// One FDM_Int1_Solve of BOTH lines (Re, Im) for the rows of this thread.
//   fl[p][l], p = 0..9: f rows j0-1 .. j0+8 (the halo rows are only read where they exist)
//   res0 / resN: the boundary values as MatMul_3d sees them (fdm_integral.f90:240-245)
//   x[p][l]: solution rows j0..j0+7 (boundary rows included after the reconstruction)
//   ext[l]: derivative at the given end: BC == 1 at the bottom (valid in chunk 0), BC == 2 at the top (valid in the last chunk)
// LDS: s_w [nwaves][4 + 2 NL][NM] (scan), s_k [OK_SIZE][NM] (boundary rows), s_fac [threads][3 OM + 1] (backward factors)
// OM rows per thread (8, or 4 with the line cut into twice as many chunks), NL lines sharing the factors (2 = Re, Im of one mode; 4 = of two
// modes with the same lambda)
template <int BC, int NM, int OM = 8, int NL = 2>
__device__ __forceinline__ void ode_solve(const OdeSys &T, double lam, const double *__restrict__ chk, int nm, int t, int c, int C, int m,
                                          const double (&fl)[OM + 2][NL], const double (&res0)[NL], const double (&resN)[NL],
                                          double (&x)[OM][NL], double (&ext)[NL], double *s_w, double *s_k, double *s_fac) {
    static_assert(OM == 4 || OM == 8, "rows per thread");
    const int n = T.n, j0 = c * OM;
    // the boundary rows of the mode: the bottom ones by the thread of chunk 1, the top ones by that of chunk 2 (every other thread of the workgroup waits
    // for them at the barrier below: two threads side by side halve that wait; what a thread does not store is not computed)
    if (C >= 3) {
        if (c == 1) {
            OdeRows k;
            ode_boundary_rows<BC>(T, lam, k);
            ode_rows_to_lds<NM, 1>(k, s_k, m);
        }
        if (c == 2) {
            OdeRows k;
            ode_boundary_rows<BC>(T, lam, k);
            ode_rows_to_lds<NM, 2>(k, s_k, m);
        }
    } else if (c == 1) {       // C >= 2; chunk 1 never touches a boundary row itself
        OdeRows k;
        ode_boundary_rows<BC>(T, lam, k);
        ode_rows_to_lds<NM>(k, s_k, m);
    }
    ODE_SYNC();
#define KK(field, q) s_k[((field) + (q)) * NM + m]
#define KRB(j, cc) s_k[(OK_RB + (j) * 4 + (cc)) * NM + m]
#define KRT(j, cc) s_k[(OK_RT + (j) * 4 + (cc)) * NM + m]
    // ---- factors of my rows, from the checkpoint ----
    double st[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) st[q] = (c == 0) ? 0.0 : chk[(unsigned)((((t / NM) * C + c) * 6 + q) * NM + m)];      // 32-bit indices: checked on the host
    double am[OM], bm[OM];             // forward multipliers in registers; the backward factors (1/c, -d, -e) wait in LDS
    double *my_fac = s_fac + threadIdx.x * (3 * OM + 1);      // thread-major with an odd stride: constant offsets, no bank conflicts
#define FAC(p, q) my_fac[(p) * 3 + (q)]
    double (&rhs)[OM][NL] = x;          // right-hand side -> y -> x in place
    double bcs_b[NL], bcs_t[NL];
    static_assert(NL <= OK_CST - OK_FS, "parking rows (the mode constants start at OK_CST)");
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        // the only use of fl after the right-hand side: f(n-2) of the last chunk (BCS_MAX) / f(1) of the first one (BCS_MIN), for du -- parked in
        // LDS by the thread that reads it back (8 VGPRs less through both sweeps)
        if (BC == 2 && c == C - 1) s_k[(OK_FS + l) * NM + m] = fl[OM - 1][l];
        if (BC == 1 && c == 0) s_k[(OK_FS + l) * NM + m] = fl[2][l];
        bcs_b[l] = bcs_t[l] = 0.0;
    }
    // The special rows sit at fixed positions of the first and the last chunk (requires n = 8 C): row 0 / n-1 are not part of the
    // system, rows 1, 2 / n-3, n-2 carry the reduced boundary closures.  Conditions are written on the unrolled p so that they fold away
    // everywhere else, and the special cases are selections of coefficients, not branches.
    const bool lo = (c == 0), hi = (c == C - 1);
#pragma unroll
    for (int p = 0; p < OM; ++p) {
        const int j = j0 + p;
        const bool off = (p == 0 && lo) || (p == OM - 1 && hi);         // boundary rows
        double r[5], c0, c1, c2 = 1.0, cb = 0.0, ct = 0.0;     // rhs = c0 f(j-1) + c1 f(j) + c2 f(j+1) + cb res0 + ct resN
        if (ODE_KO & 16) { r[0] = 0.01 * lam; r[1] = 0.3; r[2] = 1.0 + lam; r[3] = 0.3; r[4] = 0.01; c0 = 0.5; c1 = 0.25; } else
        {   // lhs_row_t and R(j, 1:2) from the packed row (same numbers, same operations)
            const double2 *pk = reinterpret_cast<const double2 *>(T.pk) + (unsigned)(j * 8);
            const double2 q0 = pk[0], q1 = pk[1], q2 = pk[2], q3 = pk[3], q4 = pk[4], q5 = pk[5], q6 = pk[6];
            const double sj = q5.x;
            r[0] = nf_madd(q0.x, lam, q2.y) * sj; r[1] = nf_madd(q0.y, lam, q3.x) * sj; r[2] = nf_madd(q1.x, lam, q3.y) * sj;
            r[3] = nf_madd(q1.y, lam, q4.x) * sj; r[4] = nf_madd(q2.x, lam, q4.y) * sj;
            c0 = q5.y; c1 = q6.x;
        }
        if (p == 1 && lo) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = KK(OK_L1, q);
            c0 = 0.0; c1 = KRB(1, 2); c2 = KRB(1, 3); cb = KRB(1, 1);
        }
        if (p == 2 && lo) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = KK(OK_L2, q);
            c0 = KRB(2, 1); c1 = KRB(2, 2); c2 = KRB(2, 3); cb = KRB(2, 0);
        }
        if (p == OM - 3 && hi) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = KK(OK_LN2, q);
            c0 = KRT(0, 0); c1 = KRT(0, 1); c2 = KRT(0, 2); ct = KRT(0, 3);
        }
        if (p == OM - 2 && hi) {
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = KK(OK_LN1, q);
            c0 = KRT(1, 0); c1 = KRT(1, 1); c2 = 0.0; ct = KRT(1, 2);
        }
        // PENTADFS step (linear5.f90:30-71): row 1 starts the elimination, row 2 has one sub-diagonal, the rest two
        double a_m = 0.0, b_m = 0.0, cm = r[2], dm = r[3];
        const double em = r[4];
        if (p >= 3 || !lo) {
            a_m = ODE_DIV(r[0], st[3]);
            b_m = ODE_DIV(nf_msub(r[1], a_m, st[4]), st[0]);
            cm = nf_msub(nf_msub(r[2], b_m, st[1]), a_m, st[5]);
            dm = nf_msub(r[3], b_m, st[2]);
        } else if (p == 2) {
            b_m = ODE_DIV(r[1], st[0]);
            cm = nf_msub(r[2], b_m, st[1]);
            dm = nf_msub(r[3], b_m, st[2]);
        }
        if (off) { a_m = 0.0; b_m = 0.0; }
        am[p] = a_m; bm[p] = b_m;
        FAC(p, 0) = off ? 1.0 : ODE_DIV(1.0, cm); FAC(p, 1) = off ? 0.0 : -dm; FAC(p, 2) = off ? 0.0 : -em;
        if (!off) {
            st[3] = st[0]; st[4] = st[1]; st[5] = st[2];
            st[0] = cm; st[1] = dm; st[2] = em;
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double fm = fl[p][l], fc = fl[p + 1][l], fp = fl[p + 2][l];
            double v = fm * c0 + fc * c1 + fp * c2;
            if ((p == 1 || p == 2) && lo) v = res0[l] * cb + v;          // (order of the reference: boundary term first, fdm_matmul.f90:93-94)
            if ((p == OM - 3 || p == OM - 2) && hi) v = v + resN[l] * ct;
            rhs[p][l] = off ? 0.0 : v;
            if (p == 1 && lo) bcs_b[l] = res0[l] * KRB(0, 2) + fc * KRB(0, 3) + fp * KRB(0, 1);
            if (p == OM - 2 && hi) bcs_t[l] = fm * KRT(2, 2) + fc * KRT(2, 0) + resN[l] * KRT(2, 1);
        }
        if (p & 1) __builtin_amdgcn_sched_barrier(0);      // table loads of two rows in flight, not of all eight (96 doubles)
    }
    // ---- forward substitution: particular end values + transfer matrix, scan, repeat with the inflow ----
    double inflow[NL][2];
    {
        double y1[NL], y2[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) y1[l] = y2[l] = 0.0;
        double h1a = 1.0, h2a = 0.0, h1b = 0.0, h2b = 1.0;     // responses to unit inflows (y[j0-1], y[j0-2]) = (1,0), (0,1)
#pragma unroll
        for (int p = 0; p < OM; ++p) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double y = rhs[p][l] - y1[l] * bm[p] - y2[l] * am[p];
                y2[l] = y1[l]; y1[l] = y;
            }
            const double ha = -h1a * bm[p] - h2a * am[p]; h2a = h1a; h1a = ha;
            const double hb = -h1b * bm[p] - h2b * am[p]; h2b = h1b; h1b = hb;
        }
        // out = (y[j0+7], y[j0+6]) = Phi (in1, in2) + end
        double phi[4] = {h1a, h1b, h2a, h2b}, ee[NL][2];
#pragma unroll
        for (int l = 0; l < NL; ++l) { ee[l][0] = y1[l]; ee[l][1] = y2[l]; }
        ode_chain<NM, +1, NL>(phi, ee, c, C, m, s_w, inflow);
    }
    double (&y)[OM][NL] = x;
    {
        double y1[NL], y2[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) { y1[l] = inflow[l][0]; y2[l] = inflow[l][1]; }
#pragma unroll
        for (int p = 0; p < OM; ++p)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double v = rhs[p][l] - y1[l] * bm[p] - y2[l] * am[p];
                y[p][l] = v; y2[l] = y1[l]; y1[l] = v;
            }
    }
    ODE_SYNC();
    // ---- backward substitution, same scheme downwards: in = (x[j0+8], x[j0+9]), out = (x[j0], x[j0+1]) ----
    {
        double x1[NL], x2[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) x1[l] = x2[l] = 0.0;
        double h1a = 1.0, h2a = 0.0, h1b = 0.0, h2b = 1.0;
#pragma unroll
        for (int p = OM - 1; p >= 0; --p) {
            const double cinv_p = FAC(p, 0), nd_p = FAC(p, 1), ne_p = FAC(p, 2);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double v = (y[p][l] + x1[l] * nd_p + x2[l] * ne_p) * cinv_p;
                x2[l] = x1[l]; x1[l] = v;
            }
            const double ha = (h1a * nd_p + h2a * ne_p) * cinv_p; h2a = h1a; h1a = ha;
            const double hb = (h1b * nd_p + h2b * ne_p) * cinv_p; h2b = h1b; h1b = hb;
        }
        double phi[4] = {h1a, h1b, h2a, h2b}, ee[NL][2];
#pragma unroll
        for (int l = 0; l < NL; ++l) { ee[l][0] = x1[l]; ee[l][1] = x2[l]; }
        ode_chain<NM, -1, NL>(phi, ee, c, C, m, s_w, inflow);
    }
    {
        double x1[NL], x2[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) { x1[l] = inflow[l][0]; x2[l] = inflow[l][1]; }
#pragma unroll
        for (int p = OM - 1; p >= 0; --p) {
            const double cinv_p = FAC(p, 0), nd_p = FAC(p, 1), ne_p = FAC(p, 2);
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const double v = (y[p][l] + x1[l] * nd_p + x2[l] * ne_p) * cinv_p;
                x[p][l] = v; x2[l] = x1[l]; x1[l] = v;
            }
        }
    }
    ODE_SYNC();
    // ---- boundary value at the free end, derivative at the given end (fdm_integral.f90:265-311) ----
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        ext[l] = 0.0;
        if (BC == 2) {
            if (c == 0) x[0][l] = bcs_b[l] + KK(OK_L0, 3) * x[1][l] + KK(OK_L0, 4) * x[2][l] + KK(OK_L0, 0) * x[3][l];
            if (c == C - 1) {
                x[OM - 1][l] = resN[l];
                // rows n-2, n-3, n-4 = p 6, 5, 4 ; f[n-2] = fl[7]
                ext[l] = KK(OK_LN, 2) * resN[l] + KK(OK_LN, 1) * x[OM - 2][l] + KK(OK_LN, 0) * x[OM - 3][l] + KK(OK_LN, 4) * x[OM - 4][l] +
                         T.R[(n - 1) * 3 + 0] * s_k[(OK_FS + l) * NM + m];
            }
        } else {
            if (c == C - 1) x[OM - 1][l] = bcs_t[l] + KK(OK_LN, 1) * x[OM - 2][l] + KK(OK_LN, 0) * x[OM - 3][l] + KK(OK_LN, 4) * x[OM - 4][l];
            if (c == 0) {
                x[0][l] = res0[l];
                ext[l] = KK(OK_L0, 2) * res0[l] + KK(OK_L0, 3) * x[1][l] + KK(OK_L0, 4) * x[2][l] + KK(OK_L0, 0) * x[3][l] + T.R[0 * 3 + 2] * s_k[(OK_FS + l) * NM + m];
            }
        }
    }
    ODE_SYNC();       // s_k is rewritten by the next solve
#undef KK
#undef KRB
#undef KRT
#undef FAC
}

// DD: OPR_ODE2_Factorize_DD (opr_odes.f90:391-478) instead of _NN: the same two solves with the top value of u GIVEN (bcs(:,2)), two constants
// instead of three (a.cst = [5][nm]: aa, bb, 1 / (aa sp(1) - bb u1(1)), sp(1), u1(1) from k_dd_constants), no e^(+) term in the superposition.
// NL = 4: MIRROR PAIRS.  lambda(kx, kz) = lambda(kx, nz - kz) to the bit (the modified wavenumbers of +-omega, fdm_derivative.f90:198-204), so the
// pivots, the checkpoints, the constants and the homogeneous solutions of the two modes are the same numbers: one thread carries the four lines
// (Re, Im of both modes) through one regeneration of the factors and one read of the tables.  Workgroup = NM values of kx x one kz <= nz/2 (and its
// mirror plane); kz = 0 and nz/2 are their own partners (the second store is dropped).  The plan checks the symmetry of lambda and of the skip
// flags on the host before it picks this form.  OMR = 4 rows per thread there: the same 16 values per thread as 8 rows x 2 lines.
template <int NM, bool DD = false, int OMR = 8, int NL = 2>
__global__ void __launch_bounds__(512) k_ode_nn(OdeArgs a) {
    constexpr int NQ = NL / 2;                                                                    // modes per thread
    extern __shared__ double lds[];
    const int C = a.C, n = a.n;
    const int m = threadIdx.x % NM, c = threadIdx.x / NM;
    double *s_w = lds, *s_x = lds + 8 * (4 + 2 * NL) * NM;                                       // s_w: [8 waves][4 + 2 NL][NM]; s_x: [C][NL lines][2][NM]
    double *s_sc = s_x + (size_t)C * 2 * NL * NM;                                                // [5][NL][NM]
    double *s_k = s_sc + 5 * NL * NM;                                                            // [OK_SIZE][NM]
    double *s_fac = s_k + OK_SIZE * NM;                                                          // [threads][3 OMR + 1]
#define SC(q, l) s_sc[((q) * NL + (l)) * NM + m]
    // 32-bit index arithmetic throughout (the host checks that every array has < 2^31 elements): 64-bit address pairs for the ~50
    // distinct rows this thread touches would otherwise be precomputed and kept in registers
    const int nm = (int)a.nm;
    // NM = 4: a workgroup's row of f^ / p^ / dp^ is 64 B, half a 128-B line.  Workgroups go round-robin over the 8 XCDs (each with its own L2), so
    // the neighbour that owns the other half would sit on another XCD and the line would cross the fabric twice (PMC: 9.3 GB per launch, 8.3 with the pairing, against
    // 6.7 algorithmic).  Pair them: of every 16 consecutive workgroups, XCD x gets the adjacent blocks 2x and 2x + 1.
    unsigned blk = blockIdx.x;
    // pair_xcd = G (16, 32, 64, ...): of every G consecutive workgroups XCD x gets the G/8 ADJACENT blocks x G/8 .. -- the rows of f^ / p^ / dp^ are
    // nxh = nx/2 + 1 complex numbers long, an ODD number of 16-B elements, so the 128-B lines are not aligned with any fixed window of kx: a line is
    // shared by neighbouring blocks in every row but one of eight, and only neighbours on the same XCD (same L2) share it for free.  Counters at 512^3
    // (profiles/r06/poisson_requests.txt): G = 16 read 4.06 GB per launch in 128-B requests where f^ + checkpoints + homogeneous solutions are 1.75.
    if (a.pair_xcd >= 16) {
        const unsigned G = (unsigned)a.pair_xcd;
        if ((blk | (G - 1u)) < gridDim.x) blk = (blk & ~(G - 1u)) + (blk & 7u) * (G >> 3) + ((blk & (G - 1u)) >> 3);
    }
    int t = (int)blk * NM + m;                        // the mode whose tables are read (NL = 4: kz <= nz/2, so the pair index is the mode index)
    const int nlive = (NL == 4) ? a.nxh * (nm / a.nxh / 2 + 1) : nm;
    const bool live = t < nlive;
    if (!live) t = nlive - 1;
    const double lam = a.lam[t];
    unsigned fidx0[NQ];
    bool store[NQ];                                   // modes solved elsewhere (singular, low) are left alone, each of a pair on its own
    fidx0[0] = (unsigned)((t % a.nxh) + a.nxh * a.ny * (t / a.nxh));
    store[0] = live && !a.skip[t];
    if (NL == 4) {
        const int nz = nm / a.nxh, kz = t / a.nxh, kz2 = (nz - kz) % nz;
        fidx0[NQ - 1] = (unsigned)((t % a.nxh) + a.nxh * a.ny * kz2);
        store[NQ - 1] = live && kz2 != kz && !a.skip[(t % a.nxh) + a.nxh * kz2];      // kz = 0, nz/2: their own mirror, stored once
    }
    const int j0 = c * OMR;
    const double2 *F = reinterpret_cast<const double2 *>(a.f_hat);
    double2 *P = reinterpret_cast<double2 *>(a.p_hat), *D = reinterpret_cast<double2 *>(a.dp_hat);
    // the constants of the mode and its band of negligible homogeneous solutions are needed after the two solves, by every chunk: one thread per mode
    // fetches them now (their trip to HBM was exposed in front of the epilogue, and 64 chunks issued the same nine loads)
    if (c == (C > 3 ? 3 : 0)) {
#pragma unroll
        for (int k = 0; k < (DD ? 5 : 9); ++k) s_k[(OK_CST + k) * NM + m] = a.cst[(unsigned)(k * nm + t)];
        s_k[(OK_BAND + 0) * NM + m] = a.band != nullptr ? (double)a.band[t] : (double)n;
        s_k[(OK_BAND + 1) * NM + m] = a.band != nullptr ? (double)a.band[nm + t] : 0.0;
    }

    double u[OMR][NL], ext[NL];
    double v_1[NL], u_n[NL], fn[NL];      // (the Neumann data bb = SC(0, l), bt = SC(1, l) stay in LDS until the constants are formed)
#pragma unroll
    for (int l = 0; l < NL; ++l) v_1[l] = u_n[l] = fn[l] = 0.0;
    double vh[OMR + 2][NL];      // rows j0-1 .. j0+OMR of the u-solve's right-hand side v; vh[1..OMR] is where the v-solve puts v
    {
        // ---- f rows j0-1 .. j0+OMR (normalised).  f(n) itself is never read by the solves: the callers' f(n) = 0 enters as resN (opr_odes.f90:303)
        double fl[OMR + 2][NL];
#pragma unroll
        for (int p = 0; p < OMR + 2; ++p) {
            const int j = j0 - 1 + p;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double2 w = make_double2(0.0, 0.0);
                if (j >= 0 && j <= n - 1) w = F[fidx0[q] + (unsigned)(j * a.nxh)];
                fl[p][2 * q] = w.x * a.fscale; fl[p][2 * q + 1] = w.y * a.fscale;
            }
        }
        // Neumann data travel in the boundary rows of the forcing (opr_elliptic.f90:310-311)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (c == 0) SC(0, l) = fl[1][l];
            if (c == C - 1) SC(1, l) = fl[OMR][l];
        }
        // ---- v0' + lambda v0 = f, v0(1) = 0 ; f(n) = 0 ----
        ode_solve<1, NM, OMR, NL>(a.T1, lam, a.chk1, nm, t, c, C, m, fl, v_1, fn, reinterpret_cast<double (&)[OMR][NL]>(vh[1]), ext, s_w, s_k, s_fac);
    }
    // halo rows of v0 for the right-hand side of the u-solve
#pragma unroll
    for (int l = 0; l < NL; ++l) { s_x[((c * NL + l) * 2 + 0) * NM + m] = vh[1][l]; s_x[((c * NL + l) * 2 + 1) * NM + m] = vh[OMR][l]; }
    ODE_SYNC();
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        vh[0][l] = (c > 0) ? s_x[(((c - 1) * NL + l) * 2 + 1) * NM + m] : 0.0;
        vh[OMR + 1][l] = (c < C - 1) ? s_x[(((c + 1) * NL + l) * 2 + 0) * NM + m] : 0.0;
    }
    if (c == C - 1) {     // v0(n)
#pragma unroll
        for (int l = 0; l < NL; ++l) SC(3, l) = vh[OMR][l];
    }
    // ---- u0' - lambda u0 = v0, u0(n) = 0 ; the "opposite boundary value" is v0(1) = 0 (res(1) = f(1), fdm_integral.f90:243) ----
    if (DD) {      // u(:, nx) = bcs(:, 2)  (:440)
#pragma unroll
        for (int l = 0; l < NL; ++l) u_n[l] = SC(1, l);
    }
    ode_solve<2, NM, OMR, NL>(a.T2, -lam, a.chk2, nm, t, c, C, m, vh, v_1, u_n, u, ext, s_w, s_k, s_fac);
    // ---- u0(1), v0(n), du0(n) -> the three constants (opr_odes.f90:350-356 with the LU of k_nn_constants) ----
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        if (c == 0) SC(2, l) = u[0][l];
        if (c == C - 1) SC(4, l) = ext[l];
    }
    ODE_SYNC();
    if (DD) {      // :452-456
        const double aa = s_k[(OK_CST + 0) * NM + m], bc = s_k[(OK_CST + 1) * NM + m], dummy = s_k[(OK_CST + 2) * NM + m];
        const double sp1 = s_k[(OK_CST + 3) * NM + m], u11 = s_k[(OK_CST + 4) * NM + m];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double u0_1 = SC(2, l), v0_n = SC(3, l), du0n = SC(4, l), bbl = SC(0, l), btl = SC(1, l);
            const double w = lam * btl - du0n + v0_n;
            v_1[l] = (aa * (bbl - u0_1) - u11 * w) * dummy;
            fn[l] = (sp1 * w - bc * (bbl - u0_1)) * dummy;
        }
    } else {
        const double a11 = s_k[(OK_CST + 0) * NM + m], a21 = s_k[(OK_CST + 1) * NM + m], a31 = s_k[(OK_CST + 2) * NM + m];
        const double a12 = s_k[(OK_CST + 3) * NM + m], a22 = s_k[(OK_CST + 4) * NM + m], a32 = s_k[(OK_CST + 5) * NM + m];
        const double a13 = s_k[(OK_CST + 6) * NM + m], a23 = s_k[(OK_CST + 7) * NM + m], a33 = s_k[(OK_CST + 8) * NM + m];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double u0_1 = SC(2, l), v0_n = SC(3, l), du0n = SC(4, l), bbl = SC(0, l), btl = SC(1, l);
            v_1[l] = (bbl - lam * u0_1) / a11;
            u_n[l] = (btl - v0_n - a21 * v_1[l]) / a22;
            fn[l] = (btl - du0n - a31 * v_1[l] - a32 * u_n[l]) / a33;
            u_n[l] = u_n[l] - a23 * fn[l];
            v_1[l] = v_1[l] - a12 * u_n[l] - a13 * fn[l];
        }
    }
    // ---- superposition with the stored homogeneous solutions (opr_odes.f90:358-367); p^ = u, dp^/dy = v ----
    if (!store[0] && !store[NQ - 1]) return;
    // The homogeneous solutions decay like exp(-sqrt(lambda) distance from their wall): for all but the lowest modes they are below 1e-40 of
    // their maximum a few tens of rows away from the walls, where adding them changes no bit of the sum.  The plan records that band per
    // mode (k_ode_hom_band); chunks inside it skip the five loads (40 of the 100 B per mode and row this kernel would otherwise move).
    bool need = true;
    if (ODE_KO & 8) need = false; else need = (j0 <= (int)s_k[(OK_BAND + 0) * NM + m]) || (j0 + OMR - 1 >= (int)s_k[(OK_BAND + 1) * NM + m]);
#pragma unroll
    for (int p = 0; p < OMR; ++p) {
        const int j = j0 + p;
        const unsigned h = (unsigned)(((t / NM) * 5 * n + j) * NM + m), hs = (unsigned)(n * NM);       // hom_blocked[blk][5][n][NM]
        double hv1 = 0.0, hem = 0.0, hu1 = 0.0, hsp = 0.0, hep = 0.0;
        if (need) { hv1 = a.hom[h]; hem = a.hom[h + hs]; hu1 = a.hom[h + 2 * hs]; hsp = a.hom[h + 3 * hs]; if (!DD) hep = a.hom[h + 4 * hs]; }
        double uu[NL], vv[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const double u0 = u[p][l], v0 = vh[p + 1][l];
            if (DD) {           // :459-465: rows nx .. 2 by the general formula (u0(nx) = bcs(:,2), u1(nx) = sp(nx) = 0), row 1 = the bottom value
                if (j == 0) {
                    uu[l] = SC(0, l);
                    vv[l] = v_1[l] + lam * uu[l];
                } else {
                    uu[l] = u0 + fn[l] * hu1 + v_1[l] * hsp;
                    vv[l] = v0 + fn[l] * hv1 + v_1[l] * hem + lam * uu[l];
                }
            } else if (j == n - 1) {
                uu[l] = u_n[l];
                vv[l] = v0 + fn[l] * hv1 + v_1[l] * hem + lam * uu[l];
            } else if (j == 0) {
                uu[l] = u0 + fn[l] * hu1 + v_1[l] * hsp + u_n[l] * hep;
                vv[l] = v_1[l] + lam * uu[l];
            } else {
                uu[l] = u0 + fn[l] * hu1 + v_1[l] * hsp + u_n[l] * hep;
                vv[l] = v0 + fn[l] * hv1 + v_1[l] * hem + lam * uu[l];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!store[q]) continue;
            const unsigned idx = fidx0[q] + (unsigned)(j * a.nxh);
            P[idx] = make_double2(uu[2 * q], uu[2 * q + 1]);
            D[idx] = make_double2(vv[2 * q], vv[2 * q + 1]);
        }
    }
#undef SC
}

// ================================================================================================
// k_int2c : FDM_Int2_Solve (the DIRECT elliptic solver, k_int2 above) on the chunked scheme of k_ode_nn: ONE pentadiagonal system per mode,
// thread (mode m, chunk c) owns 8 rows, the PENTADFS pivots of its rows regenerated from a checkpoint of the serial recurrence, forward and
// backward substitution as particular end values + 2 x 2 transfer matrix, parallel scan over the chunks (ode_chain), repeat with the inflow.
// Rows and right-hand sides are built exactly as k_int2 builds them (same operations, no contraction); what differs from the marching kernel is
// the association of the substitution sums.  No scratch: f^ 16 B in, p^ 16 B out, checkpoints 6 B per mode and row.
// ================================================================================================
struct Int2cArgs {
    Int2Dev T;
    const double *lam;
    double alpha;
    long long nm, skip;
    const double *chk;          // [blk][C][6][NM] PENTADFS state before the first row of each chunk
    const double *fsrc;
    double *dst;
    double fscale;
    int nxh, ny, C, neumann_b, neumann_t;
};

__device__ __forceinline__ void int2_row(const Int2Dev &T, int j, double lam, const double (&l1)[3], const double (&lN)[3], int nb_on, int nt_on,
                                         double (&r)[5]) {
#pragma clang fp contract(off)
    const int n = T.n;
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = T.Bt[(unsigned)(j * 5 + k)] - lam * T.A5[(unsigned)(j * 5 + k)];      // fdm_integral.f90:412-432
    if (nb_on && (j == 1 || j == 2)) {       // :462-464
        const int k0 = 3 - j;
#pragma unroll
        for (int q = 0; q < 3; ++q) r[k0 + q] = r[k0 + q] - T.nb[j - 1] * l1[q];
    }
    if (nt_on && (j == n - 2 || j == n - 3)) {   // :501-503
        const int ir = n - 1 - j;
#pragma unroll
        for (int q = 0; q < 3; ++q) r[ir - 1 + q] = r[ir - 1 + q] - T.nt[ir - 1] * lN[q];
    }
    const double sj = T.s[j];                    // :518-540
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = r[k] * sj;
}

// checkpoints of the factor recurrence of every mode: state before rows 8, 16, ...
__global__ void __launch_bounds__(256) k_int2_checkpoint(Int2Dev T, const double *__restrict__ lamv, double alpha, int nb_on, int nt_on,
                                                         double *__restrict__ chk, long long nm, int NM, int C) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nm) return;
    const int n = T.n;
    const double lam = lamv[t] - alpha;
    double l1[3] = {T.c1[0], T.c1[1], T.c1[2]}, lN[3] = {T.cn[0], T.cn[1], T.cn[2]};
    l1[0] = l1[0] + lam * T.e1;
    lN[2] = lN[2] + lam * T.en;
    double st[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 1; j <= n - 2; ++j) {
        if ((j % OM) == 0) {
            const int c = j / OM;
#pragma unroll
            for (int q = 0; q < 6; ++q) chk[(((t / NM) * C + c) * 6 + q) * NM + (t % NM)] = st[q];
        }
        double r[5], am, bm, cinv, nd, ne;
        int2_row(T, j, lam, l1, lN, nb_on, nt_on, r);
        ode_factor_step(j, r, st, am, bm, cinv, nd, ne);
    }
}

template <int NM>
__global__ void __launch_bounds__(512) k_int2c(Int2cArgs a) {
    extern __shared__ double lds[];
    const int C = a.C, n = a.T.n;
    const int m = threadIdx.x % NM, c = threadIdx.x / NM;
    double *s_w = lds, *s_fac = lds + 8 * 8 * NM;
    const int nm = (int)a.nm;
    int t = (int)blockIdx.x * NM + m;
    const bool live = t < nm;
    if (!live) t = nm - 1;
    const bool store = live && ((long long)t != a.skip);
    const double lam = a.lam[t] - a.alpha;
    const unsigned fidx0 = (unsigned)((t % a.nxh) + a.nxh * a.ny * (t / a.nxh));
    const int j0 = c * OM;
    const bool lo = (c == 0), hi = (c == C - 1);
    const double2 *F = reinterpret_cast<const double2 *>(a.fsrc);
    double fl[OM + 2][2];        // f rows j0-1 .. j0+8 (normalised)
#pragma unroll
    for (int p = 0; p < OM + 2; ++p) {
        const int j = j0 - 1 + p;
        double2 w = make_double2(0.0, 0.0);
        if (j >= 0 && j <= n - 1) w = F[fidx0 + (unsigned)(j * a.nxh)];
        fl[p][0] = w.x * a.fscale; fl[p][1] = w.y * a.fscale;
    }
    const double res0[2] = {fl[1][0], fl[1][1]}, resN[2] = {fl[OM][0], fl[OM][1]};      // rows 0 / n-1: meaningful in the first / last chunk only
    double l1[3] = {a.T.c1[0], a.T.c1[1], a.T.c1[2]}, lN[3] = {a.T.cn[0], a.T.cn[1], a.T.cn[2]};
    l1[0] = nf_madd(l1[0], lam, a.T.e1);
    lN[2] = nf_madd(lN[2], lam, a.T.en);
    double bcs_b[2] = {0, 0}, bcs_t[2] = {0, 0};
    // ---- factors of my rows from the checkpoint, right-hand side ----
    double st[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) st[q] = (c == 0) ? 0.0 : a.chk[(unsigned)((((t / NM) * C + c) * 6 + q) * NM + m)];
    double am[OM], bm[OM], x[OM][2];
    double *my_fac = s_fac + threadIdx.x * (3 * OM + 1);
#define FAC(p, q) my_fac[(p) * 3 + (q)]
#pragma unroll
    for (int p = 0; p < OM; ++p) {
        const int j = j0 + p;
        const bool off = (p == 0 && lo) || (p == OM - 1 && hi);
        double r[5];
        int2_row(a.T, j, lam, l1, lN, a.neumann_b, a.neumann_t, r);
        double a_m = 0.0, b_m = 0.0, cm = r[2], dm = r[3];
        const double em = r[4];
        if (p >= 3 || !lo) {
            a_m = r[0] / st[3];
            b_m = nf_msub(r[1], a_m, st[4]) / st[0];
            cm = nf_msub(nf_msub(r[2], b_m, st[1]), a_m, st[5]);
            dm = nf_msub(r[3], b_m, st[2]);
        } else if (p == 2) {
            b_m = r[1] / st[0];
            cm = nf_msub(r[2], b_m, st[1]);
            dm = nf_msub(r[3], b_m, st[2]);
        }
        if (off) { a_m = 0.0; b_m = 0.0; }
        am[p] = a_m; bm[p] = b_m;
        FAC(p, 0) = off ? 1.0 : 1.0 / cm; FAC(p, 1) = off ? 0.0 : -dm; FAC(p, 2) = off ? 0.0 : -em;
        if (!off) {
            st[3] = st[0]; st[4] = st[1]; st[5] = st[2];
            st[0] = cm; st[1] = dm; st[2] = em;
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
#pragma clang fp contract(off)
            const double fm = fl[p][l], fc = fl[p + 1][l], fp = fl[p + 2][l];
            double v = fm * a.T.R[(unsigned)(j * 3 + 0)] + fc * a.T.R[(unsigned)(j * 3 + 1)] + fp;      // MatMul_3d interior row
            if (p == 1 && lo) v = res0[l] * a.T.rb[1][1] + fc * a.T.rb[1][2] + fp * a.T.rb[1][3];
            if (p == 2 && lo) v = res0[l] * a.T.rb[2][0] + fm * a.T.rb[2][1] + fc * a.T.rb[2][2] + fp * a.T.rb[2][3];
            if (p == OM - 3 && hi) v = fm * a.T.rt[0][0] + fc * a.T.rt[0][1] + fp * a.T.rt[0][2] + resN[l] * a.T.rt[0][3];
            if (p == OM - 2 && hi) v = fm * a.T.rt[1][0] + fc * a.T.rt[1][1] + resN[l] * a.T.rt[1][2];
            x[p][l] = off ? 0.0 : v;
            if (p == 1 && lo) bcs_b[l] = res0[l] * a.T.rb[0][2] + fc * a.T.rb[0][3] + fp * a.T.rb[0][1];
            if (p == OM - 2 && hi) bcs_t[l] = fm * a.T.rt[2][2] + fc * a.T.rt[2][0] + resN[l] * a.T.rt[2][1];
        }
        if (p & 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- forward substitution ----
    double inflow[2][2];
    {
        double y1[2] = {0, 0}, y2[2] = {0, 0};
        double h1a = 1.0, h2a = 0.0, h1b = 0.0, h2b = 1.0;
#pragma unroll
        for (int p = 0; p < OM; ++p) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const double y = x[p][l] - y1[l] * bm[p] - y2[l] * am[p];
                y2[l] = y1[l]; y1[l] = y;
            }
            const double ha = -h1a * bm[p] - h2a * am[p]; h2a = h1a; h1a = ha;
            const double hb = -h1b * bm[p] - h2b * am[p]; h2b = h1b; h1b = hb;
        }
        double phi[4] = {h1a, h1b, h2a, h2b}, ee[2][2] = {{y1[0], y2[0]}, {y1[1], y2[1]}};
        ode_chain<NM, +1>(phi, ee, c, C, m, s_w, inflow);
    }
    {
        double y1[2] = {inflow[0][0], inflow[1][0]}, y2[2] = {inflow[0][1], inflow[1][1]};
#pragma unroll
        for (int p = 0; p < OM; ++p)
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const double v = x[p][l] - y1[l] * bm[p] - y2[l] * am[p];
                x[p][l] = v; y2[l] = y1[l]; y1[l] = v;
            }
    }
    __syncthreads();
    // ---- backward substitution ----
    {
        double x1[2] = {0, 0}, x2[2] = {0, 0};
        double h1a = 1.0, h2a = 0.0, h1b = 0.0, h2b = 1.0;
#pragma unroll
        for (int p = OM - 1; p >= 0; --p) {
            const double cinv_p = FAC(p, 0), nd_p = FAC(p, 1), ne_p = FAC(p, 2);
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const double v = (x[p][l] + x1[l] * nd_p + x2[l] * ne_p) * cinv_p;
                x2[l] = x1[l]; x1[l] = v;
            }
            const double ha = (h1a * nd_p + h2a * ne_p) * cinv_p; h2a = h1a; h1a = ha;
            const double hb = (h1b * nd_p + h2b * ne_p) * cinv_p; h2b = h1b; h1b = hb;
        }
        double phi[4] = {h1a, h1b, h2a, h2b}, ee[2][2] = {{x1[0], x2[0]}, {x1[1], x2[1]}};
        ode_chain<NM, -1>(phi, ee, c, C, m, s_w, inflow);
    }
    {
        double x1[2] = {inflow[0][0], inflow[1][0]}, x2[2] = {inflow[0][1], inflow[1][1]};
#pragma unroll
        for (int p = OM - 1; p >= 0; --p) {
            const double cinv_p = FAC(p, 0), nd_p = FAC(p, 1), ne_p = FAC(p, 2);
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const double v = (x[p][l] + x1[l] * nd_p + x2[l] * ne_p) * cinv_p;
                x[p][l] = v; x2[l] = x1[l]; x1[l] = v;
            }
        }
    }
#undef FAC
    // ---- end values: given (Dirichlet) or from the biased first-derivative formula (Neumann), fdm_integral.f90:659-668 ----
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma clang fp contract(off)
        if (lo) x[0][l] = a.neumann_b ? bcs_b[l] + l1[0] * x[1][l] + l1[1] * x[2][l] + l1[2] * x[3][l] : res0[l];
        if (hi) x[OM - 1][l] = a.neumann_t ? bcs_t[l] + lN[2] * x[OM - 2][l] + lN[1] * x[OM - 3][l] + lN[0] * x[OM - 4][l] : resN[l];
    }
    if (!store) return;
    double2 *D = reinterpret_cast<double2 *>(a.dst);
#pragma unroll
    for (int p = 0; p < OM; ++p) D[fidx0 + (unsigned)((j0 + p) * a.nxh)] = make_double2(x[p][0], x[p][1]);
}

// The <= 4 singular modes (lambda = 0): OPR_ODE2_Factorize_NN_Sing -> _DN_Sing (opr_odes.f90:165-183, 37-96) with the same chunked
// solves, one workgroup: v0' = f (f(1) = 0), v0(n) = bcs_t ; u0' = v0, u0(1) = 0 ; u = u0 + c u1, v = v0 + c v1 with
// c = (v0(1) - du0(1)) / (du1(1) - v1(1)); u1, v1, du1 depend on the mode only (plan creation).
struct OdeSingArgs {
    OdeSys T1, T2;
    const double *chk1, *chk2;          // checkpoints of the ns singular modes, blocked [0][C][6][NM]
    const int *modes;                   // [ns] flat mode indices
    const double *v1, *u1, *du1;        // [n][ns] (line 0 of the stored pairs), [ns]
    const double *f_hat;
    double *p_hat, *dp_hat;
    double fscale;
    int n, nxh, ny, C, ns;
};

template <int NM>
__global__ void __launch_bounds__(512) k_ode_sing(OdeSingArgs a) {
    extern __shared__ double lds[];
    const int C = a.C, n = a.n;
    const int m = threadIdx.x % NM, c = threadIdx.x / NM;
    double *s_w = lds, *s_x = lds + 8 * 8 * NM;
    double *s_sc = s_x + (size_t)C * 4 * NM;
    double *s_k = s_sc + 10 * NM;
    double *s_fac = s_k + OK_SIZE * NM;
    const bool live = m < a.ns;
    const int t = a.modes[live ? m : 0];
    const unsigned fidx0 = (unsigned)((t % a.nxh) + a.nxh * a.ny * (t / a.nxh));
    const int j0 = c * OM;
    const double2 *F = reinterpret_cast<const double2 *>(a.f_hat);
    double vh[OM + 2][2], u[OM][2], ext[2];
    double zero[2] = {0, 0}, bct[2];
    {
        double fl[OM + 2][2];
#pragma unroll
        for (int p = 0; p < OM + 2; ++p) {
            const int j = j0 - 1 + p;
            double2 w = make_double2(0.0, 0.0);
            if (j >= 0 && j <= n - 1) w = F[fidx0 + (unsigned)(j * a.nxh)];
            fl[p][0] = w.x * a.fscale; fl[p][1] = w.y * a.fscale;
        }
        if (c == C - 1) { s_sc[2 * NM + m] = fl[OM][0]; s_sc[3 * NM + m] = fl[OM][1]; }      // Neumann datum at the top (opr_elliptic.f90:310-311)
        if (c == 0) { fl[1][0] = 0.0; fl[1][1] = 0.0; }                                     // f(1) = 0 (opr_odes.f90:59 via :179)
        __syncthreads();
        bct[0] = s_sc[2 * NM + m]; bct[1] = s_sc[3 * NM + m];
        ode_solve<2, NM>(a.T2, 0.0, a.chk2, 0, m, c, C, m, fl, zero, bct, reinterpret_cast<double (&)[OM][2]>(vh[1]), ext, s_w, s_k, s_fac);
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) { s_x[((c * 2 + l) * 2 + 0) * NM + m] = vh[1][l]; s_x[((c * 2 + l) * 2 + 1) * NM + m] = vh[OM][l]; }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        vh[0][l] = (c > 0) ? s_x[(((c - 1) * 2 + l) * 2 + 1) * NM + m] : 0.0;
        vh[OM + 1][l] = (c < C - 1) ? s_x[(((c + 1) * 2 + l) * 2 + 0) * NM + m] : 0.0;
    }
    if (c == 0) { s_sc[4 * NM + m] = vh[1][0]; s_sc[5 * NM + m] = vh[1][1]; }                 // v0(1)
    ode_solve<1, NM>(a.T1, 0.0, a.chk1, 0, m, c, C, m, vh, zero, bct, u, ext, s_w, s_k, s_fac);
    if (c == 0) { s_sc[6 * NM + m] = ext[0]; s_sc[7 * NM + m] = ext[1]; }                   // du0 at the bottom
    __syncthreads();
    if (!live) return;
    const int ns = a.ns;
    const double f1 = 1.0 / (a.du1[m] - a.v1[(0 * n + 0) * ns + m]);
    double cc[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) cc[l] = (s_sc[(4 + l) * NM + m] - s_sc[(6 + l) * NM + m]) * f1;
    double2 *P = reinterpret_cast<double2 *>(a.p_hat), *D = reinterpret_cast<double2 *>(a.dp_hat);
#pragma unroll
    for (int p = 0; p < OM; ++p) {
        const int j = j0 + p;
        const double hu = a.u1[j * ns + m], hv = a.v1[j * ns + m];
        const unsigned idx = fidx0 + (unsigned)(j * a.nxh);
        P[idx] = make_double2(u[p][0] + cc[0] * hu, u[p][1] + cc[1] * hu);
        D[idx] = make_double2(vh[p + 1][0] + cc[0] * hv, vh[p + 1][1] + cc[1] * hv);
    }
}

// ------------------------------------------------------------------------------------------------
// per-mode constants of OPR_ODE2_Factorize_NN: LU of the 3x3 constraint matrix (opr_odes.f90:329-348)
// hom: SoA [(c*n + j)*nm + t], c = 0 v1, 1 em, 2 u1, 3 sp, 4 ep ; der: [(c*nm + t)], c = 0 du1_n, 1 dsp_n, 2 dep_n
// cst: [(c*nm + t)], c = 0..8 = a11 a21 a31 a12 a22 a32 a13 a23 a33 (after the LU)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nn_constants(const double *__restrict__ hom, const double *__restrict__ der,
                                                      const double *__restrict__ lamv, double *__restrict__ cst, int n, long long nm) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nm) return;
    const double lam = lamv[t];
    auto H = [&](int c, int j) { return hom[((long long)c * n + j) * nm + t]; };
    double a11 = 1.0 + lam * H(3, 0), a21 = H(1, n - 1), a31 = der[1 * nm + t];
    double a12 = lam * H(4, 0), a22 = lam, a32 = der[2 * nm + t];
    double a13 = lam * H(2, 0), a23 = H(0, n - 1), a33 = der[0 * nm + t];
    a12 = a12 / a11;
    a22 = a22 - a21 * a12;
    a32 = a32 - a31 * a12;
    a13 = a13 / a11;
    a23 = (a23 - a21 * a13) / a22;
    a33 = a33 - a31 * a13 - a32 * a23;
    cst[0 * nm + t] = a11; cst[1 * nm + t] = a21; cst[2 * nm + t] = a31;
    cst[3 * nm + t] = a12; cst[4 * nm + t] = a22; cst[5 * nm + t] = a32;
    cst[6 * nm + t] = a13; cst[7 * nm + t] = a23; cst[8 * nm + t] = a33;
}

// ------------------------------------------------------------------------------------------------
// superposition (opr_odes.f90:350-367): writes p^ and dp^/dy in the spectral field layout.
// u0, v0: SoA [(l*n + j)*nm + t] (l = Re, Im); du0: [(l*nm + t)]; bcs: [(c*nm + t)] c = ReB, ImB, ReT, ImT
// ------------------------------------------------------------------------------------------------
struct CombineArgs {
    const double *u0, *v0, *du0, *bcs, *hom, *cst, *lam;
    const unsigned char *skip;   // [nm]: 1 for the singular modes (handled separately)
    double *p_hat, *dp_hat;      // complex fields (nxh, ny, nz)
    int n, nxh, ny;
    long long nm;
};

__global__ void __launch_bounds__(256) k_nn_combine(CombineArgs a) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nm) return;
    if (a.skip[t]) return;
    const int n = a.n;
    const long long nm = a.nm;
    const double lam = a.lam[t];
    const long long fidx0 = (t % a.nxh) + (long long)a.nxh * a.ny * (t / a.nxh);
    const double a11 = a.cst[0 * nm + t], a21 = a.cst[1 * nm + t], a31 = a.cst[2 * nm + t];
    const double a12 = a.cst[3 * nm + t], a22 = a.cst[4 * nm + t], a32 = a.cst[5 * nm + t];
    const double a13 = a.cst[6 * nm + t], a23 = a.cst[7 * nm + t], a33 = a.cst[8 * nm + t];
    double v_1[2], u_n[2], fn[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const double bb = a.bcs[(long long)l * nm + t], bt = a.bcs[(long long)(2 + l) * nm + t];
        const double u0_1 = a.u0[((long long)l * n + 0) * nm + t];
        const double v0_n = a.v0[((long long)l * n + (n - 1)) * nm + t];
        const double du0n = a.du0[(long long)l * nm + t];
        v_1[l] = (bb - lam * u0_1) / a11;
        u_n[l] = (bt - v0_n - a21 * v_1[l]) / a22;
        fn[l] = (bt - du0n - a31 * v_1[l] - a32 * u_n[l]) / a33;
        u_n[l] = u_n[l] - a23 * fn[l];
        v_1[l] = v_1[l] - a12 * u_n[l] - a13 * fn[l];
    }
    double2 *P = reinterpret_cast<double2 *>(a.p_hat), *D = reinterpret_cast<double2 *>(a.dp_hat);
    // rows [jlo, jhi) of this thread: gridDim.y row blocks (few modes: the rows are the parallelism; many modes: one block does them all)
    const int rows = (n + (int)gridDim.y - 1) / (int)gridDim.y;
    const int jlo = (int)blockIdx.y * rows, jhi = (jlo + rows < n) ? jlo + rows : n;
    for (int j = jlo; j < jhi; ++j) {
        const double hv1 = a.hom[((long long)0 * n + j) * nm + t], hem = a.hom[((long long)1 * n + j) * nm + t];
        const double hu1 = a.hom[((long long)2 * n + j) * nm + t], hsp = a.hom[((long long)3 * n + j) * nm + t];
        const double hep = a.hom[((long long)4 * n + j) * nm + t];
        double u[2], v[2];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const double u0 = a.u0[((long long)l * n + j) * nm + t], v0 = a.v0[((long long)l * n + j) * nm + t];
            if (j == n - 1) {
                u[l] = u_n[l];
                v[l] = v0 + fn[l] * hv1 + v_1[l] * hem + lam * u[l];
            } else if (j == 0) {
                u[l] = u0 + fn[l] * hu1 + v_1[l] * hsp + u_n[l] * hep;
                v[l] = v_1[l] + lam * u[l];
            } else {
                u[l] = u0 + fn[l] * hu1 + v_1[l] * hsp + u_n[l] * hep;
                v[l] = v0 + fn[l] * hv1 + v_1[l] * hem + lam * u[l];
            }
        }
        const long long idx = fidx0 + (long long)j * a.nxh;
        P[idx] = make_double2(u[0], u[1]);
        D[idx] = make_double2(v[0], v[1]);
    }
}

// ------------------------------------------------------------------------------------------------
// singular modes (lambda = 0: OPR_ODE2_Factorize_NN_Sing -> _DN_Sing, opr_odes.f90:165-183, 37-96): <= 4 modes
// ------------------------------------------------------------------------------------------------
// gather f^ * norm of the singular modes into SoA [(l*n + j)*ns + s], with row 0 zeroed (f(:,1) = 0), and the top BC
__global__ void k_sing_gather(const double *__restrict__ f_hat, const int *__restrict__ modes, int ns, int n, int nxh, int ny,
                              double scale, double *__restrict__ fs, double *__restrict__ bct) {
    const int s = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    const long long idx = (t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh;
    const double2 v = reinterpret_cast<const double2 *>(f_hat)[idx];
    fs[((long long)0 * n + j) * ns + s] = (j == 0) ? 0.0 : v.x * scale;
    fs[((long long)1 * n + j) * ns + s] = (j == 0) ? 0.0 : v.y * scale;
    if (j == n - 1) { bct[0 * ns + s] = v.x * scale; bct[1 * ns + s] = v.y * scale; }
}

// columns of a list of modes between the spectral field layout (nxh, ny, nz) and a compact (ns, ny, 1) field (low-mode sub-plan)
__global__ void k_modes_gather(const double2 *__restrict__ f_hat, const int *__restrict__ modes, int ns, int n, int nxh, int ny,
                               double2 *__restrict__ out) {
    const int s = threadIdx.x + blockIdx.x * blockDim.x, j = blockIdx.y;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    out[(long long)j * ns + s] = f_hat[(t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh];
}
__global__ void k_modes_scatter(const double2 *__restrict__ p_low, const double2 *__restrict__ dp_low, const int *__restrict__ modes, int ns,
                                int n, int nxh, int ny, double2 *__restrict__ p_hat, double2 *__restrict__ dp_hat) {
    const int s = threadIdx.x + blockIdx.x * blockDim.x, j = blockIdx.y;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    const long long idx = (t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh;
    p_hat[idx] = p_low[(long long)j * ns + s];
    dp_hat[idx] = dp_low[(long long)j * ns + s];
}

// u = u0 + c u1, v = v0 + c v1, c = (v0(1) - du0_n) / (du1_n - v1(1)); scatter into the spectral fields
__global__ void k_sing_combine(const double *__restrict__ u0, const double *__restrict__ v0, const double *__restrict__ u1,
                               const double *__restrict__ v1, const double *__restrict__ du0, const double *__restrict__ du1,
                               const int *__restrict__ modes, int ns, int n, int nxh, int ny, double *__restrict__ p_hat,
                               double *__restrict__ dp_hat) {
    const int s = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    const long long idx = (t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh;
    const double f1 = 1.0 / (du1[0 * ns + s] - v1[((long long)0 * n + 0) * ns + s]);
    double u[2], v[2];
    for (int l = 0; l < 2; ++l) {
        const double c = (v0[((long long)l * n + 0) * ns + s] - du0[l * ns + s]) * f1;
        u[l] = u0[((long long)l * n + j) * ns + s] + c * u1[((long long)0 * n + j) * ns + s];
        v[l] = v0[((long long)l * n + j) * ns + s] + c * v1[((long long)0 * n + j) * ns + s];
    }
    reinterpret_cast<double2 *>(p_hat)[idx] = make_double2(u[0], u[1]);
    reinterpret_cast<double2 *>(dp_hat)[idx] = make_double2(v[0], v[1]);
}

// ------------------------------------------------------------------------------------------------
// ibc = BCS_DD of the factorized solver: OPR_ODE2_Factorize_DD (opr_odes.f90:391-478) and _DD_Sing (:188-260).  Marching kernels only
// (the chunked kernel is the BCS_NN solver of the RHS); same two integral solves as BCS_NN with the top value of u given, other constants.
// ------------------------------------------------------------------------------------------------
struct DDCombineArgs {
    const double *u0, *v0, *du0, *bcs, *hom, *der, *lam;
    const int *sing;             // singular modes (handled separately)
    int ns;
    int hom_nm_block;            // 0: hom is SoA [(c*n + j)*nm + t]; NM > 0: blocked [blk][5][n][NM] (k_ode_block_layout)
    double *p_hat, *dp_hat;
    int n, nxh, ny;
    long long nm;
};

// constants of OPR_ODE2_Factorize_DD that depend on the mode only (opr_odes.f90:452-454), for the chunked kernel: cst[5][nm]
__global__ void __launch_bounds__(256) k_dd_constants(const double *__restrict__ hom, int hom_nm_block, const double *__restrict__ der,
                                                      double *__restrict__ cst, int n, long long nm) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nm) return;
    auto H = [&](int c, int j) {
        if (hom_nm_block > 0) {
            const int NM = hom_nm_block;
            return hom[(((t / NM) * 5 + c) * n + j) * NM + (t % NM)];
        }
        return hom[((long long)c * n + j) * nm + t];
    };
    const double aa = der[0 * nm + t] - H(0, n - 1);
    const double bb = der[1 * nm + t] - H(1, n - 1);
    cst[0 * nm + t] = aa;
    cst[1 * nm + t] = bb;
    cst[2 * nm + t] = 1.0 / (aa * H(3, 0) - bb * H(2, 0));
    cst[3 * nm + t] = H(3, 0);
    cst[4 * nm + t] = H(2, 0);
}

__global__ void __launch_bounds__(256) k_dd_combine(DDCombineArgs a) {
#pragma clang fp contract(off)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.nm) return;
    for (int s = 0; s < a.ns; ++s)
        if (a.sing[s] == t) return;
    const int n = a.n;
    const long long nm = a.nm;
    const double lam = a.lam[t];
    const long long fidx0 = (t % a.nxh) + (long long)a.nxh * a.ny * (t / a.nxh);
    auto H = [&](int c, int j) {
        if (a.hom_nm_block > 0) {
            const int NM = a.hom_nm_block;
            return a.hom[(((t / NM) * 5 + c) * n + j) * NM + (t % NM)];
        }
        return a.hom[((long long)c * n + j) * nm + t];
    };
    // c = 0 v1, 1 em, 2 u1, 3 sp ; der: 0 du1_n, 1 dsp_n
    const double aa = a.der[0 * nm + t] - H(0, n - 1);
    const double bb = a.der[1 * nm + t] - H(1, n - 1);
    const double dummy = 1.0 / (aa * H(3, 0) - bb * H(2, 0));
    double q1[2], fn[2], bb_[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const double bcb = a.bcs[(long long)l * nm + t], bct = a.bcs[(long long)(2 + l) * nm + t];
        const double u0_1 = a.u0[((long long)l * n + 0) * nm + t];
        const double v0_n = a.v0[((long long)l * n + (n - 1)) * nm + t];
        const double w = lam * bct - a.du0[(long long)l * nm + t] + v0_n;
        q1[l] = (aa * (bcb - u0_1) - H(2, 0) * w) * dummy;
        fn[l] = (H(3, 0) * w - bb * (bcb - u0_1)) * dummy;
        bb_[l] = bcb;
    }
    double2 *P = reinterpret_cast<double2 *>(a.p_hat), *D = reinterpret_cast<double2 *>(a.dp_hat);
    const int rows = (n + (int)gridDim.y - 1) / (int)gridDim.y;      // row blocks as in k_nn_combine
    const int jlo = (int)blockIdx.y * rows, jhi = (jlo + rows < n) ? jlo + rows : n;
    for (int j = jhi - 1; j >= (jlo > 1 ? jlo : 1); --j) {
        const double hv1 = H(0, j), hem = H(1, j), hu1 = H(2, j), hsp = H(3, j);
        double u[2], v[2];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            u[l] = a.u0[((long long)l * n + j) * nm + t] + fn[l] * hu1 + q1[l] * hsp;
            v[l] = a.v0[((long long)l * n + j) * nm + t] + fn[l] * hv1 + q1[l] * hem + lam * u[l];
        }
        P[fidx0 + (long long)j * a.nxh] = make_double2(u[0], u[1]);
        D[fidx0 + (long long)j * a.nxh] = make_double2(v[0], v[1]);
    }
    if (jlo == 0) {
        P[fidx0] = make_double2(bb_[0], bb_[1]);
        D[fidx0] = make_double2(q1[0] + lam * bb_[0], q1[1] + lam * bb_[1]);
    }
}

// singular modes: f^ * norm into SoA [(l*n + j)*ns + s] (all rows), bottom / top values into bcb / bct [l*ns + s]
__global__ void k_sing_gather_dd(const double *__restrict__ f_hat, const int *__restrict__ modes, int ns, int n, int nxh, int ny, double scale,
                                 double *__restrict__ fs, double *__restrict__ bcb, double *__restrict__ bct) {
    const int s = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    const double2 v = reinterpret_cast<const double2 *>(f_hat)[(t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh];
    fs[((long long)0 * n + j) * ns + s] = v.x * scale;
    fs[((long long)1 * n + j) * ns + s] = v.y * scale;
    if (j == 0) { bcb[0 * ns + s] = v.x * scale; bcb[1 * ns + s] = v.y * scale; }
    if (j == n - 1) { bct[0 * ns + s] = v.x * scale; bct[1 * ns + s] = v.y * scale; }
}
__global__ void k_fill_ones(double *__restrict__ a, long long m) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) a[i] = 1.0;
}
// opr_odes.f90:238-251
__global__ void k_sing_combine_dd(const double *__restrict__ u0, const double *__restrict__ v0, const double *__restrict__ u1,
                                  const double *__restrict__ v1, const double *__restrict__ sp, const double *__restrict__ du0,
                                  const double *__restrict__ du1, const double *__restrict__ bcb, const int *__restrict__ modes, int ns, int n,
                                  int nxh, int ny, double *__restrict__ p_hat, double *__restrict__ dp_hat) {
#pragma clang fp contract(off)
    const int s = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
    if (s >= ns || j >= n) return;
    const long long t = modes[s];
    const long long idx = (t % nxh) + (long long)nxh * ny * (t / nxh) + (long long)j * nxh;
    const double fn = 1.0 / (du1[0 * ns + s] - v1[((long long)0 * n + (n - 1)) * ns + s]);
    const double dummy = 1.0 / sp[((long long)0 * n + 0) * ns + s];
    double u[2], v[2];
    for (int l = 0; l < 2; ++l) {
        const double c = (v0[((long long)l * n + (n - 1)) * ns + s] - du0[l * ns + s]) * fn;
        const double q = (bcb[l * ns + s] - (u0[((long long)l * n + 0) * ns + s] + c * u1[((long long)0 * n + 0) * ns + s])) * dummy;
        if (j == 0) {
            u[l] = bcb[l * ns + s];
            v[l] = q;
        } else {
            u[l] = u0[((long long)l * n + j) * ns + s] + c * u1[((long long)0 * n + j) * ns + s] + q * sp[((long long)0 * n + j) * ns + s];
            v[l] = v0[((long long)l * n + j) * ns + s] + c * v1[((long long)0 * n + j) * ns + s] + q;
        }
    }
    reinterpret_cast<double2 *>(p_hat)[idx] = make_double2(u[0], u[1]);
    reinterpret_cast<double2 *>(dp_hat)[idx] = make_double2(v[0], v[1]);
}

// p(:,1,:) = bcs_hb, p(:,ny,:) = bcs_ht  (opr_elliptic.f90:285-286)
__global__ void __launch_bounds__(256) k_set_wall_planes(double *__restrict__ p, const double *__restrict__ hb,
                                                          const double *__restrict__ ht, int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)(i % nx);
    const long long k = i / nx;
    p[ix + (long long)nx * (0 + (long long)ny * k)] = hb[i];
    p[ix + (long long)nx * ((ny - 1) + (long long)ny * k)] = ht[i];
}

}  // namespace tlab

// ================================================================================================
// host side: plan, rocFFT, orchestration
// ================================================================================================
using namespace tlab;

namespace {

void hipc(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP ") + what + ": " + hipGetErrorString(e));
}
void fftc(rocfft_status s, const char *what) {
    if (s != rocfft_status_success) throw std::runtime_error(std::string("rocFFT ") + what + " failed (status " + std::to_string((int)s) + ")");
}

struct DBuf {
    double *p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = count;
        if (count) hipc(hipMalloc((void **)&p, count * sizeof(double)), "hipMalloc");
    }
    void upload(const std::vector<double> &h) {
        alloc(h.size());
        if (n) hipc(hipMemcpy(p, h.data(), n * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy");
    }
    ~DBuf() { if (p) (void)hipFree(p); }
};

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    ~FftPlan() {
        if (info) rocfft_execution_info_destroy(info);
        if (plan) rocfft_plan_destroy(plan);
        if (work) (void)hipFree(work);
    }
    void finish() {
        fftc(rocfft_plan_get_work_buffer_size(plan, &work_bytes), "work size");
        fftc(rocfft_execution_info_create(&info), "info");
        if (work_bytes) {
            hipc(hipMalloc(&work, work_bytes), "hipMalloc(fft work)");
            fftc(rocfft_execution_info_set_work_buffer(info, work, work_bytes), "set work");
        }
    }
    void exec(void *in, void *out, hipStream_t st, double bytes = 0.0) {
        ProfScope ps("rocfft", st, bytes);
        fftc(rocfft_execution_info_set_stream(info, st), "set stream");
        void *ib[1] = {in}, *ob[1] = {out};
        fftc(rocfft_execute(plan, ib, ob, info), "execute");
    }
};

bool g_rocfft_up = false;
bool g_poisson_exact = [] { const char *e = getenv("TLAB_POISSON_EXACT"); return e && atoi(e) != 0; }();

}  // namespace

struct tlab_poisson_plan {
    int nx = 0, ny = 0, nz = 0, nxh = 0;   // nz = local number of z planes (kmax)
    int nzt = 0, koff = 0, nproc = 1;      // global nz, first global plane of this slab, number of z slabs
    int ioff = 0;                          // first global kx of the local spectral box (kx-pencil plans; nxh is then the local count)
    int fx_nxh = 0, fx_nz = 0;             // x-transform geometry: complex row length nx/2+1 and number of planes it is batched over
    long long nm = 0;                 // local modes = nxh * nz
    double norm = 1.0;
    Int1Tables tmin, tmax;            // host copies
    // SpaceOrder1 with (3, 3) or (5, 7) diagonals: integral systems factorized on the host (int1_generic.cpp), one table set per system and
    // lambda array in use (all modes; the singular modes' zeros), built the first time base_args meets it
    bool generic = false;
    DerTables gder;
    struct GenSet { int which; const double *lam; long long nm; DBuf fac, rb, rt, R; int ndi = 0, nri = 0; };
    mutable std::vector<std::unique_ptr<GenSet>> gen;
    const GenSet &gen_set(int which, const double *lam_dev, long long nm_) const {
        for (const auto &e : gen)
            if (e->which == which && e->lam == lam_dev && e->nm == nm_) return *e;
        std::vector<double> hl((size_t)nm_);
        if (hipMemcpy(hl.data(), lam_dev, (size_t)nm_ * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("hipMemcpy (lambda)");
        Int1Gen G;
        int1_generic_build(gder, which == 0 ? 1 : 2, hl.data(), nm_, which == 0 ? 1.0 : -1.0, G);
        auto e = std::make_unique<GenSet>();
        e->which = which; e->lam = lam_dev; e->nm = nm_; e->ndi = G.ndi; e->nri = G.nri;
        e->fac.upload(G.fac); e->rb.upload(G.rb); e->rt.upload(G.rt); e->R.upload(G.R);
        gen.push_back(std::move(e));
        return *gen.back();
    }
    DBuf d_L0[2], d_L1[2], d_R[2];    // [0] BCS_MIN tables, [1] BCS_MAX tables
    DBuf d_pk[2];                     // OdeSys::pk
    DBuf lam;                         // [nm]  sqrt(kx'^2 + kz'^2)
    DBuf hom, der, cst;               // homogeneous solutions [5][ny][nm], their boundary derivatives [3][nm], 3x3 LU [9][nm]
    DBuf scratch, v0, u0, du0, bcs;   // per-call work: [5][ny][nm], [2][ny][nm] x2, [2][nm], [4][nm]
    DBuf cwork;                       // complex work field (nxh*ny*nz complex)
    DBuf d_bt[2], chk[2], chk_s[2], homb;       // chunked ODE kernel: boundary constants [3][4], PENTADFS checkpoints [blk][C][6][NM] of both systems,
                                      // homogeneous solutions re-laid out as [blk][5][ny][NM]
    bool use_chunked = false;
    int ode_nm_per_wg = 0;
    int ode_om = OM;                  // rows per thread of k_ode_nn
    bool ode_pair = false;            // k_ode_nn on mirror pairs (kx, kz), (kx, nz - kz): lambda symmetric to the bit, checked at creation
    // The lowest-lambda modes of a chunked plan go through a marching sub-plan on the side stream (see build_low_modes)
    std::unique_ptr<tlab_poisson_plan> low;
    DBuf fac[2];                      // sub-plan only: stored LU factors of its two systems (Int1Args::fac)
    int *d_low_modes = nullptr;
    int *d_hom_band = nullptr;                  // [2][nm] rows between which the homogeneous solutions of a mode are negligible (k_ode_hom_band)
    int n_low = 0;
    DBuf low_f, low_p, low_dp;
    std::vector<int> sing_modes;      // flat mode indices t = kx + nxh*kz of the singular modes
    int *d_sing = nullptr;
    unsigned char *d_skip = nullptr;
    DBuf s_lam, s_f, s_unit, s_bct, s_v0, s_v1, s_u0, s_u1, s_du0, s_du1, s_scr;
    DBuf dd_v1, dd_u1, dd_du1, dd_sp, dd_ones, dd_bcb;      // BCS_DD: homogeneous solutions of the singular modes (built on first use)
    bool dd_ready = false;
    DBuf cst_dd;                                // [5][nm] constants of the chunked BCS_DD solver (k_dd_constants), built on first use
    FftPlan fx_r2c, fx_c2r, fz_f, fz_b;
    FftPlan f2_fwd, f2_bwd;           // optional fused 2-D (x,z) transforms, batch over y
    std::unique_ptr<FftzPlan> fz_own;  // own strided z-transform (fftz.hip) where its lengths apply; rocFFT's fz_f / fz_b otherwise
    std::unique_ptr<FftxPlan> fx_own;  // own one-pass real-to-complex x-transform (fftz.hip: k_fftx_r2c); rocFFT's two-kernel fx_r2c otherwise
    // one-shot request of the RHS driver (tlab_internal_poisson_arm_v_final): the inverse x-transform of dp^/dy finishes the v equation
    // (FftxPlan::exec_inverse_final) instead of writing dp/dy
    struct VFinal { double *q = nullptr, *h = nullptr; double dte = 0.0, kco = 0.0; int scale = 0; bool armed = false; } vfinal;
    void x_backward_dpdy(void *in, double *dpdy, hipStream_t st, const VFinal &f) {
        if (f.armed && fx_own) fx_own->exec_inverse_final(static_cast<const double *>(in), f.q, f.h, f.dte, f.kco, f.scale, ny, st);
        else fx_c2r.exec(in, dpdy, st);
    }
    // inverse x-transform of p^: rocFFT's c2r runs at the copy rate at 512 points but at 2.1 TB/s from 1024 on (4.06 ms per call on one rank's share of
    // BASELINE configs[4], where the own kernel moves the same bytes at 5.9 TB/s); TLAB_FFTX_C2R_OWN = 0 / 1 forces the choice
    void x_backward_p(void *in, double *p, hipStream_t st) {
        static const int own = [] { const char *e = getenv("TLAB_FFTX_C2R_OWN"); return e ? atoi(e) : -1; }();
        if (fx_own && (own == 1 || (own < 0 && nx >= 1024))) fx_own->exec_inverse(static_cast<const double *>(in), p, st);
        else fx_c2r.exec(in, p, st);
    }
    void x_forward(void *in, void *out, hipStream_t st) {
        if (fx_own) fx_own->exec(static_cast<const double *>(in), static_cast<double *>(out), st);
        else fx_r2c.exec(in, out, st);
    }
    // pack-layout maps of the own x-transforms (tlab_poisson_fft_x_packed), one per distinct block map; a slab driver uses one or two
    struct KxMap { std::vector<long long> key; long long *off = nullptr; int *w = nullptr; };
    std::vector<KxMap> kxmaps;
    const KxMap &kx_map(int nblocks, const int *start, const long long *base) {
        std::vector<long long> key;
        key.reserve((size_t)2 * nblocks);
        for (int b = 0; b < nblocks; ++b) { key.push_back(start[b]); key.push_back(base[b]); }
        for (const KxMap &m : kxmaps) if (m.key == key) return m;
        std::vector<long long> off((size_t)fx_nxh);
        std::vector<int> w((size_t)fx_nxh);
        if (tlab_debug_pack_map(fx_nxh, nblocks, start, base, off.data(), w.data()) != TLAB_OK) throw std::invalid_argument("pack map: bad block map");
        KxMap m;
        m.key = key;
        hipc(hipMalloc((void **)&m.off, off.size() * sizeof(long long)), "hipMalloc");
        hipc(hipMalloc((void **)&m.w, w.size() * sizeof(int)), "hipMalloc");
        hipc(hipMemcpy(m.off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice), "hipMemcpy");
        hipc(hipMemcpy(m.w, w.data(), w.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
        kxmaps.push_back(m);
        return kxmaps.back();
    }
    bool use_2d = false;
    bool fz_inplace = false;          // z-transform plans built in place (kx-pencil plans: rocFFT then picks its column kernel, ~3x faster)
    hipStream_t side = nullptr;       // the <= 4 singular modes are solved beside the regular ones
    // DIRECT elliptic solver (EllipticOrder = CompactDirect6): one second-order integral operator per boundary type, built on first use
    bool direct = false;
    bool exact_mode = false;                  // tlab_poisson_set_exact(1) at creation: marching kernels only (k_int2 instead of k_int2c)
    tlab_fdm_plan_t gy_der = nullptr;         // y plan of the derivatives (dp/dy = OPR_Partial_Y(p), opr_elliptic.f90:447-449); not owned
    // factorized Helmholtz (opr_elliptic.f90:466-557): the per-mode tables depend on alpha, so every alpha in use is a sub-plan of its own
    // (tables only; transforms and work field are the parent's).  The implicit RK cycles through a few alphas: the last 4 are kept.
    tlab_fdm_plan_t g3[3] = {nullptr, nullptr, nullptr};      // x, y, z plans of a single-device factorized plan; not owned
    bool helmholtz = false;
    std::vector<std::pair<double, std::unique_ptr<tlab_poisson_plan>>> helm;
    DerTables ell_der2;                       // second derivative of the elliptic y plan (fdm_loc%der2)
    std::vector<double> ell_nodes;
    struct Int2Set { Int2Tables host; DBuf Bt, A5, s, R, chk; double chk_alpha = 0.0; bool chk_ok = false; };      // chk: checkpoints of k_int2c for one alpha
    std::unique_ptr<Int2Set> int2[4];
    long long sing_direct = -1;               // local index of the mode (1,1), or -1 when another rank owns it
    Int2Dev dev2(int ibc) {
        if (!int2[ibc]) {
            auto e = std::make_unique<Int2Set>();
            int2_build_tables(ell_der2, ell_nodes, ibc, e->host);
            e->Bt.upload(e->host.Bt); e->A5.upload(e->host.A5); e->s.upload(e->host.s); e->R.upload(e->host.R);
            int2[ibc] = std::move(e);
        }
        Int2Set &E = *int2[ibc];
        Int2Dev d;
        d.Bt = E.Bt.p; d.A5 = E.A5.p; d.s = E.s.p; d.R = E.R.p; d.n = ny;
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 4; ++c) { d.rb[j][c] = E.host.rb[j][c]; d.rt[j][c] = E.host.rt[j][c]; }
        for (int q = 0; q < 3; ++q) { d.c1[q] = E.host.c1[q]; d.cn[q] = E.host.cn[q]; }
        d.e1 = E.host.e1; d.en = E.host.en;
        for (int q = 0; q < 2; ++q) { d.nb[q] = E.host.nb[q]; d.nt[q] = E.host.nt[q]; }
        return d;
    }
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t side_low = nullptr;   // the low-mode sub-plan runs beside the singular modes, not behind them
    hipEvent_t ev_join_low = nullptr;
    ~tlab_poisson_plan() {
        if (d_sing) (void)hipFree(d_sing);
        if (d_skip) (void)hipFree(d_skip);
        if (d_low_modes) (void)hipFree(d_low_modes);
        if (d_hom_band) (void)hipFree(d_hom_band);
        if (side) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side_low) (void)hipStreamDestroy(side_low);
        if (ev_join_low) (void)hipEventDestroy(ev_join_low);
        for (KxMap &m : kxmaps) { (void)hipFree(m.off); (void)hipFree(m.w); }
    }
    OdeSys sys(int which) const {
        OdeSys d;
        d.L0 = d_L0[which].p; d.L1 = d_L1[which].p; d.R = d_R[which].p; d.bt = d_bt[which].p; d.n = ny;
        d.pk = d_pk[which].p;
        return d;
    }
    Int1Dev dev(int which) const {
        const Int1Tables &T = which == 0 ? tmin : tmax;
        Int1Dev d;
        if (generic) {      // k_int1g reads its own tables (Int1Args::g_*)
            d = Int1Dev{};
            d.n = ny;
            return d;
        }
        d.L0 = d_L0[which].p; d.L1 = d_L1[which].p; d.R = d_R[which].p; d.n = ny;
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 4; ++c) { d.rb[j][c] = T.rb[j][c]; d.rt[j][c] = T.rt[j][c]; }
        return d;
    }
};

namespace {

template <int BC, int NL, int FS>
void launch_int1(const Int1Args &a, hipStream_t st) {
    const int grid = (int)((a.nm + 255) / 256);
    // operand traffic of one integral solve: read NL lines, write NL lines (scratch traffic is overhead, not algorithmic)
    const bool few = a.nm <= 8;   // the <= 4 singular modes, solved beside the regular ones on the side stream
    ProfScope ps(few ? "k_int1<singular modes>" : (FS == FS_FIELD ? "k_int1<field>" : (FS == FS_LINEAR ? "k_int1<linear>" : "k_int1<unit>")), st,
                 (double)a.nm * a.T.n * 16.0 * NL);
    if (a.g_fac) {
        if (a.g_ndi == 3) hipLaunchKernelGGL((k_int1g<BC, NL, FS, 3>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_int1g<BC, NL, FS, 7>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.fac && NL == 2 && a.nm <= 2048) {      // the low-mode sub-plan: one line per thread
        static const bool ldsv = [] { const char *e = getenv("TLAB_INT1_LDS"); return !(e && atoi(e) == 0); }();
        // four rows per line (source / factors of the sweep, intermediate) + the rhs coefficients; at most what is left of a CU beside one workgroup of k_ode_nn
        auto lds_of = [&](int lv) { return ((size_t)lv * 4 * a.T.n + (size_t)2 * a.T.n) * sizeof(double); };
        if constexpr (FS != FS_UNIT) {
            auto go = [&](auto lv_c) {
                constexpr int LV = decltype(lv_c)::value;
                static bool attr = false;
                if (!attr) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_int1<BC, 1, FS, 8, true, true, true, LV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
                    (void)hipGetLastError();
                    attr = true;
                }
                hipLaunchKernelGGL((k_int1<BC, 1, FS, 8, true, true, true, LV>), dim3((unsigned)((2 * a.nm + LV - 1) / LV)), dim3(256), lds_of(LV), st, a);
                hipc(hipGetLastError(), "k_int1 (LDS)");
            };
            if (ldsv && lds_of(4) <= (size_t)84 * 1024) { go(std::integral_constant<int, 4>{}); return; }
            if (ldsv && lds_of(2) <= (size_t)84 * 1024) { go(std::integral_constant<int, 2>{}); return; }
        }
        hipLaunchKernelGGL((k_int1<BC, 1, FS, 8, true, true>), dim3((unsigned)((2 * a.nm + 127) / 128)), dim3(128), 0, st, a);
    } else if (a.fac) {
        if (a.nm < 65536) hipLaunchKernelGGL((k_int1<BC, NL, FS, 8, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_int1<BC, NL, FS, 2, true>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (a.nm < 65536) hipLaunchKernelGGL((k_int1<BC, NL, FS, 8, false>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_int1<BC, NL, FS, 2, false>), dim3(grid), dim3(256), 0, st, a);
    }
    hipc(hipGetLastError(), "k_int1");
}

Int1Args base_args(const tlab_poisson_plan &P, int which, const double *lam, long long nm, double *scratch) {
    Int1Args a{};
    a.T = P.dev(which);
    a.lam = lam;
    a.lam_sign = which == 0 ? 1.0 : -1.0;
    a.nm = nm;
    a.fscale = 1.0;
    a.nxh = P.nxh;
    a.ny = P.ny;
    a.scratch = scratch;
    if (P.generic) {
        const tlab_poisson_plan::GenSet &G = P.gen_set(which, lam, nm);
        a.g_fac = G.fac.p; a.g_rb = G.rb.p; a.g_rt = G.rt.p; a.g_R = G.R.p; a.g_ndi = G.ndi; a.g_nri = G.nri;
    }
    return a;
}

// ---- chunked ODE kernel: geometry, checkpoints, launch ----
int ode_modes_per_wg(int C) {
    // 256 threads per workgroup where the line allows it: two workgroups then share a CU (the kernel needs ~230 VGPRs, i.e. 8 waves per CU
    // either way) and one runs while the other waits at one of its ~40 barriers: 2.85 -> 2.55 ms at 512^3 against one 512-thread workgroup
    int nmw = 64;
    while (nmw > 4 && nmw * C > 256) nmw >>= 1;
    if (const char *e = getenv("TLAB_ODE_NM")) {      // experiments
        const int v = atoi(e);
        if ((v == 4 || v == 8 || v == 16 || v == 32 || v == 64) && v * C <= 512) nmw = v;
    }
    return (nmw * C <= 512) ? nmw : 0;
}
size_t ode_lds_bytes(int C, int NM, int om = OM, int NL = 2) {
    return ((size_t)(8 * (4 + 2 * NL) + 2 * NL * C + 5 * NL + OK_SIZE) * NM + (size_t)(3 * om + 1) * NM * C) * sizeof(double);
}

template <int NM, bool DD, int OMR = OM, int NL = 2>
void launch_ode_nm(const OdeArgs &a, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ode_nn<NM, DD, OMR, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        (void)hipGetLastError();
        attr_done = true;
    }
    const long long nlive = NL == 4 ? (long long)a.nxh * (a.nm / a.nxh / 2 + 1) : a.nm;
    const unsigned grid = (unsigned)((nlive + NM - 1) / NM);
    hipLaunchKernelGGL((k_ode_nn<NM, DD, OMR, NL>), dim3(grid), dim3(NM * a.C), lds, st, a);
}
template <bool DD>
void launch_ode_pair(const OdeArgs &a, int NM, int om, size_t lds, hipStream_t st) {
    // 8 rows x 4 lines per thread (256 VGPRs, ~20 of them spilled) beats 4 rows x 4 lines in twice as many chunks (217 VGPRs, but 512-thread
    // workgroups that do not share a CU and a scan twice as long): 1.86 against 2.83 ms at 512^3, one mode per thread 2.30 (profiles/r03)
    if (om != OM) throw std::logic_error("k_ode_nn: no pair form for this geometry");
    switch (NM) {
    case 4: launch_ode_nm<4, DD, OM, 4>(a, lds, st); break;
    case 8: launch_ode_nm<8, DD, OM, 4>(a, lds, st); break;
    case 16: launch_ode_nm<16, DD, OM, 4>(a, lds, st); break;
    case 32: launch_ode_nm<32, DD, OM, 4>(a, lds, st); break;
    default: launch_ode_nm<64, DD, OM, 4>(a, lds, st); break;
    }
}
bool ode_pair_geometry(int NM, int om) { return om == OM; }

void launch_ode(tlab_poisson_plan &P, double *f_hat, double *p_hat, double *dp_hat, hipStream_t st, bool dd = false) {
    OdeArgs a{};
    a.T1 = P.sys(0); a.T2 = P.sys(1);
    a.lam = P.lam.p; a.skip = P.d_skip; a.chk1 = P.chk[0].p; a.chk2 = P.chk[1].p; a.cst = dd ? P.cst_dd.p : P.cst.p; a.hom = P.homb.p; a.band = P.d_hom_band;
    a.f_hat = f_hat; a.p_hat = p_hat; a.dp_hat = dp_hat; a.fscale = P.norm;
    a.n = P.ny; a.nxh = P.nxh; a.ny = P.ny; a.C = P.ny / P.ode_om; a.nm = P.nm;
    const int NM = P.ode_nm_per_wg;
    {
        static int pair = -1;
        if (pair < 0) {      // TLAB_ODE_PAIR_XCD = 0 (blocks in dispatch order) or the group size G, a power of two >= 16 (1 = 16, the round-5 pairing)
            const char *e = getenv("TLAB_ODE_PAIR_XCD");
            pair = e ? atoi(e) : 128;      // 128: XCD x takes 16 adjacent blocks of every 128 (HBM reads of the launch 4.07 -> 3.15 GB, requests 3.18e7 -> 2.46e7 at 512^3)
            if (pair == 1) pair = 16;
            if (pair != 0 && (pair < 16 || (pair & (pair - 1)) != 0)) pair = 16;
        }
        a.pair_xcd = pair;
    }
    const size_t lds = ode_lds_bytes(a.C, NM, P.ode_om, P.ode_pair ? 4 : 2);
    ProfScope ps(dd ? "k_ode_nn<DD>" : "k_ode_nn", st, (double)P.nm * P.ny * 48.0);      // algorithmic bytes: f^ in, p^ and dp^/dy out (its own tables -- checkpoints 12 B, the band of the homogeneous solutions -- come on top)
    if (P.ode_pair) {
        if (dd) launch_ode_pair<true>(a, NM, P.ode_om, lds, st);
        else launch_ode_pair<false>(a, NM, P.ode_om, lds, st);
    } else if (dd) {
        switch (NM) {
        case 4: launch_ode_nm<4, true>(a, lds, st); break;
        case 8: launch_ode_nm<8, true>(a, lds, st); break;
        case 16: launch_ode_nm<16, true>(a, lds, st); break;
        case 32: launch_ode_nm<32, true>(a, lds, st); break;
        default: launch_ode_nm<64, true>(a, lds, st); break;
        }
    } else {
        switch (NM) {
        case 4: launch_ode_nm<4, false>(a, lds, st); break;
        case 8: launch_ode_nm<8, false>(a, lds, st); break;
        case 16: launch_ode_nm<16, false>(a, lds, st); break;
        case 32: launch_ode_nm<32, false>(a, lds, st); break;
        default: launch_ode_nm<64, false>(a, lds, st); break;
        }
    }
    hipc(hipGetLastError(), "k_ode_nn");
}

template <int NM>
static void launch_int2c(const Int2cArgs &k, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_int2c<NM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        (void)hipGetLastError();
        attr_done = true;
    }
    hipLaunchKernelGGL((k_int2c<NM>), dim3((unsigned)((k.nm + NM - 1) / NM)), dim3(NM * k.C), lds, st, k);
}

void build_checkpoints(tlab_poisson_plan &P, hipStream_t st) {
    const int C = P.ny / P.ode_om, NM = P.ode_nm_per_wg;
    const long long nblk = (P.nm + NM - 1) / NM;
    for (int w = 0; w < 2; ++w) {
        const Int1Tables &T = w == 0 ? P.tmin : P.tmax;
        std::vector<double> bt(12);
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 4; ++c) bt[j * 4 + c] = (w == 0) ? T.rb[j][c] : T.rt[j][c];
        P.d_bt[w].upload(bt);
        P.chk[w].alloc((size_t)C * 6 * nblk * NM);
    }
    const int grid = (int)((P.nm + 255) / 256);
    hipLaunchKernelGGL((k_ode_checkpoint<1>), dim3(grid), dim3(256), 0, st, P.sys(0), P.lam.p, 1.0, P.chk[0].p, P.nm, NM, C, P.ode_om);
    hipLaunchKernelGGL((k_ode_checkpoint<2>), dim3(grid), dim3(256), 0, st, P.sys(1), P.lam.p, -1.0, P.chk[1].p, P.nm, NM, C, P.ode_om);
    hipc(hipGetLastError(), "k_ode_checkpoint");
    P.homb.alloc((size_t)5 * P.ny * nblk * NM);
    const long long tot = (long long)5 * P.ny * P.nm;
    hipLaunchKernelGGL(k_ode_block_layout, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, P.hom.p, P.homb.p, 5, P.ny, P.nm, NM);
    hipc(hipGetLastError(), "k_ode_block_layout");
    static const bool band_on = [] { const char *e = getenv("TLAB_ODE_HOM_BAND"); return !(e && atoi(e) == 0); }();
    if (band_on) {
        hipc(hipMalloc((void **)&P.d_hom_band, (size_t)2 * P.nm * sizeof(int)), "hipMalloc");
        static const double rel = [] { const char *e = getenv("TLAB_ODE_HOM_THR"); return e ? atof(e) : 1.0e-40; }();
        hipLaunchKernelGGL(k_ode_hom_band, dim3(grid), dim3(256), 0, st, P.hom.p, P.ny, P.nm, P.d_hom_band, rel);
        hipc(hipGetLastError(), "k_ode_hom_band");
    }
}

// v1, u1, du1 of the singular modes depend on the mode only (opr_odes.f90:64-73): once per plan
void build_singular_homogeneous(tlab_poisson_plan &P, hipStream_t st) {
    const int ns = (int)P.sing_modes.size();
    if (ns == 0) return;
    Int1Args s2 = base_args(P, 1, P.s_lam.p, ns, P.s_scr.p);   // v1' = delta_1, v1(n) = 0
    s2.unit_row = 0; s2.zero_bsave = 0; s2.dst = P.s_v1.p;
    launch_int1<2, 2, FS_UNIT>(s2, st);
    Int1Args s4 = base_args(P, 0, P.s_lam.p, ns, P.s_scr.p);   // u1' = v1, u1(1) = 0
    s4.fsrc = P.s_v1.p; s4.nlf = 2; s4.zero_bsave = 0; s4.dst = P.s_u1.p; s4.du = P.s_du1.p;
    launch_int1<1, 2, FS_LINEAR>(s4, st);
}

// lanes per chunk of the singular-mode kernel: 8 (<= 4 modes in use), 4 when the line has more than 64 chunks (512 threads at most)
// 256 threads from 512 rows on: the one workgroup then fits the slot any retiring workgroup of k_ode_nn (256 threads, ~250 VGPRs: two per CU)
// leaves; with 512 threads it needs a whole CU and waited for the tail of k_ode_nn (measured: 1.84 ms in the queue beside the pair form)
inline int ode_sing_nm(int C) { return C >= 64 ? 4 : 8; }
// the streams of the singular / low modes.  NOT high-priority ones: the presence of a high-priority stream in the process slowed every kernel of
// the normal streams on this stack (measured A/B on one box: substep 17.7 -> 21.9 ms, k_fftz 0.53 -> 0.66 ms; 8 loopback slabs 30.6 -> 62.9 ms);
// TLAB_SIDE_PRIORITY=1 brings it back for experiments
static void create_side_stream(hipStream_t *s) {
    int lo = 0, hi = 0;
    static const bool prio = [] { const char *e = getenv("TLAB_SIDE_PRIORITY"); return e && atoi(e) != 0; }();
    if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) hipc(hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi), "stream");
    else { (void)hipGetLastError(); hipc(hipStreamCreateWithFlags(s, hipStreamNonBlocking), "stream"); }
}

void build_singular_checkpoints(tlab_poisson_plan &P, hipStream_t st) {
    const int ns = (int)P.sing_modes.size();
    if (ns == 0) return;
    const int C = P.ny / OM, NM = ode_sing_nm(C);
    for (int w = 0; w < 2; ++w) P.chk_s[w].alloc((size_t)C * 6 * NM);
    hipLaunchKernelGGL((k_ode_checkpoint<1>), dim3(1), dim3(256), 0, st, P.sys(0), P.s_lam.p, 1.0, P.chk_s[0].p, (long long)ns, NM, C, OM);
    hipLaunchKernelGGL((k_ode_checkpoint<2>), dim3(1), dim3(256), 0, st, P.sys(1), P.s_lam.p, -1.0, P.chk_s[1].p, (long long)ns, NM, C, OM);
    hipc(hipGetLastError(), "k_ode_checkpoint (singular)");
}

template <int NM>
void launch_ode_sing_nm(const OdeSingArgs &a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ode_sing<NM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        (void)hipGetLastError();
        attr_done = true;
    }
    hipLaunchKernelGGL((k_ode_sing<NM>), dim3(1), dim3(NM * a.C), ode_lds_bytes(a.C, NM), st, a);
}

void launch_ode_sing(tlab_poisson_plan &P, double *f_hat, double *p_hat, double *dp_hat, hipStream_t st) {
    OdeSingArgs a{};
    a.T1 = P.sys(0); a.T2 = P.sys(1);
    a.chk1 = P.chk_s[0].p; a.chk2 = P.chk_s[1].p; a.modes = P.d_sing;
    a.v1 = P.s_v1.p; a.u1 = P.s_u1.p; a.du1 = P.s_du1.p;
    a.f_hat = f_hat; a.p_hat = p_hat; a.dp_hat = dp_hat; a.fscale = P.norm;
    a.n = P.ny; a.nxh = P.nxh; a.ny = P.ny; a.C = P.ny / OM; a.ns = (int)P.sing_modes.size();
    ProfScope ps("k_ode_sing", st, (double)a.ns * P.ny * 48.0);
    if (ode_sing_nm(a.C) == 8) launch_ode_sing_nm<8>(a, st);
    else launch_ode_sing_nm<4>(a, st);
    hipc(hipGetLastError(), "k_ode_sing");
}

void build_fft(tlab_poisson_plan &P) {
    if (!g_rocfft_up) {
        fftc(rocfft_setup(), "setup");
        g_rocfft_up = true;
    }
    const size_t nx = P.nx, ny = P.ny, nxh = P.fx_nxh;
    size_t nz = P.fx_nz;
    {   // x: real -> complex, batch ny*nz (dfftw_plan_many_dft_r2c, opr_fourier.f90:163-166)
        rocfft_plan_description d = nullptr;
        fftc(rocfft_plan_description_create(&d), "desc");
        size_t is[1] = {1}, os[1] = {1};
        fftc(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
                                                     1, is, nx, 1, os, nxh), "layout r2c");
        size_t len[1] = {nx};
        fftc(rocfft_plan_create(&P.fx_r2c.plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_double,
                                1, len, ny * nz, d), "plan r2c");
        rocfft_plan_description_destroy(d);
        P.fx_r2c.finish();
        const char *e = getenv("TLAB_FFTX");              // TLAB_FFTX=0 keeps rocFFT for the forward x-transform
        if (!(e && atoi(e) == 0) && FftxPlan::supported((int)nx) && nxh == nx / 2 + 1) P.fx_own = std::make_unique<FftxPlan>((int)nx, (long long)(ny * nz));
    }
    {   // x: complex -> real (dfftw_plan_many_dft_c2r, :167-170)
        rocfft_plan_description d = nullptr;
        fftc(rocfft_plan_description_create(&d), "desc");
        size_t is[1] = {1}, os[1] = {1};
        fftc(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr,
                                                     1, is, nxh, 1, os, nx), "layout c2r");
        size_t len[1] = {nx};
        fftc(rocfft_plan_create(&P.fx_c2r.plan, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_double,
                                1, len, ny * nz, d), "plan c2r");
        rocfft_plan_description_destroy(d);
        P.fx_c2r.finish();
    }
    if (P.nzt > 1) {  // z: complex <-> complex, stride = batch = nlines with distance 1 (dfftw_plan_many_dft, :111-119);
        // nlines = (imax/2+1)*jmax, or tmpi_plan_fftz%nlines = that / npro_k after the K-transposition (opr_fourier.f90:85-98)
        nz = P.nz;
        const size_t nlines = (size_t)P.nxh * ny * nz / (size_t)P.nzt;
        {
            const char *e = getenv("TLAB_FFTZ");          // TLAB_FFTZ=0 keeps rocFFT for the z-transform
            if (!(e && atoi(e) == 0) && FftzPlan::supported(P.nzt)) P.fz_own = std::make_unique<FftzPlan>(P.nzt, (long long)nlines);
        }
        for (int dir = 0; dir < 2; ++dir) {
            rocfft_plan_description d = nullptr;
            fftc(rocfft_plan_description_create(&d), "desc");
            size_t st[1] = {nlines};
            fftc(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                                         nullptr, nullptr, 1, st, 1, 1, st, 1), "layout c2c");
            size_t len[1] = {(size_t)P.nzt};
            FftPlan &F = dir == 0 ? P.fz_f : P.fz_b;
            fftc(rocfft_plan_create(&F.plan, P.fz_inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                    dir == 0 ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                                    rocfft_precision_double, 1, len, nlines, d), "plan c2c");
            rocfft_plan_description_destroy(d);
            F.finish();
        }
    }
}

// fused 2-D transforms over (x, z), one per y plane: same arithmetic as r2c(x) followed by c2c(z)
void build_fft_2d(tlab_poisson_plan &P) {
    const size_t nx = P.nx, ny = P.ny, nz = P.nz, nxh = P.nxh;
    for (int dir = 0; dir < 2; ++dir) {
        rocfft_plan_description d = nullptr;
        fftc(rocfft_plan_description_create(&d), "desc");
        size_t rs[2] = {1, nx * ny}, cs[2] = {1, nxh * ny};
        if (dir == 0)
            fftc(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
                                                         2, rs, nx, 2, cs, nxh), "layout 2d fwd");
        else
            fftc(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr, nullptr,
                                                         2, cs, nxh, 2, rs, nx), "layout 2d bwd");
        size_t len[2] = {nx, nz};
        FftPlan &F = dir == 0 ? P.f2_fwd : P.f2_bwd;
        fftc(rocfft_plan_create(&F.plan, rocfft_placement_notinplace,
                                dir == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                rocfft_precision_double, 2, len, ny, d), "plan 2d");
        rocfft_plan_description_destroy(d);
        F.finish();
    }
}

// homogeneous solutions and constraint LU of every mode (opr_odes.f90:308-348), once per plan
void build_homogeneous(tlab_poisson_plan &P, hipStream_t st) {
    const long long nm = P.nm;
    const int n = P.ny;
    // v^(1), e^(-): v' + l v = (delta_n, 0), v(1) = (0, 1)   [third line of the reference is identically zero]
    Int1Args a = base_args(P, 0, P.lam.p, nm, P.scratch.p);
    a.unit_row = n - 1;
    a.zero_bsave = 0;
    a.bv[0] = 0.0; a.bv[1] = 1.0; a.bv[2] = 0.0;
    a.dst = P.hom.p;                                   // lines 0,1 -> v1, em
    launch_int1<1, 2, FS_UNIT>(a, st);
    // u^(1), s^(+), e^(+): u' - l u = (v1, em, 0), u(n) = (0, 0, 1)
    Int1Args b = base_args(P, 1, P.lam.p, nm, P.scratch.p);
    b.fsrc = P.hom.p;
    b.nlf = 2;
    b.zero_bsave = 0;
    b.bv[0] = 0.0; b.bv[1] = 0.0; b.bv[2] = 1.0;
    b.dst = P.hom.p + (size_t)2 * n * nm;              // lines 2,3,4 -> u1, sp, ep
    b.du = P.der.p;
    launch_int1<2, 3, FS_LINEAR>(b, st);
    const int grid = (int)((nm + 255) / 256);
    hipLaunchKernelGGL(k_nn_constants, dim3(grid), dim3(256), 0, st, P.hom.p, P.der.p, P.lam.p, P.cst.p, n, nm);
    hipc(hipGetLastError(), "k_nn_constants");
}

// The difference between k_ode_nn and the reference's serial sweeps lives in the few modes with lambda h^2 << 1: the
// substitution recurrences of B +- lambda A are neutral there and the rounding of the chunk transfers adds up.  Those modes -- sqrt(lambda) *
// mean(h) <= 0.06, at most 128 per box, i.e. the energy-carrying ones of a smooth field -- are taken out of k_ode_nn (skip flag) and solved by a
// marching sub-plan (the reference's operations one by one) on the side stream, beside the regular modes: 4.5e-12 -> 7.6e-13 in p on the
// projection forcing of tests/test_gpu_poisson.py, i.e. the FFT-noise floor.  TLAB_POISSON_LOW_MODES=0 disables it.
void build_low_modes(tlab_poisson_plan &P, const std::vector<double> &nodes, const std::vector<double> &lam, std::vector<unsigned char> &skip,
                     hipStream_t st) {
    // Cost: the sub-plan's chain of latency-bound launches (2.6 ms beside k_ode_nn's 2.4 ms at 512^3) sticks out by ~0.3 ms on a single device
    // (1.4 % of the substep).  Decomposed plans (z-slabs, kx-pencils) take it too: parity with the single domain at <= 1e-12 comes first, and
    // with the staged pencil exchange the low modes sit in the first kx half of rank 0, whose solve runs under the transfer of the second half.
    // Every plan applies the same threshold to its own modes, so the union over the ranks is the single-domain set (below the cap of 128).
    bool on = true;
    if (const char *e = getenv("TLAB_POISSON_LOW_MODES")) on = atoi(e) != 0;
    if (!on) return;
    const int ny = P.ny;
    if ((int)nodes.size() != ny || ny < 2) return;               // host-built plans without nodes: feature off
    const double hbar = (nodes[ny - 1] - nodes[0]) / (ny - 1.0);
    std::vector<int> cand;
    for (long long t = 0; t < P.nm; ++t)
        if (!skip[t] && lam[t] * hbar <= 0.06) cand.push_back((int)t);
    if (cand.empty()) return;
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return lam[a] < lam[b] || (lam[a] == lam[b] && a < b); });
    if (cand.size() > 128) cand.resize(128);
    const int ns = (int)cand.size();
    auto L = std::make_unique<tlab_poisson_plan>();
    L->nx = 2; L->ny = ny; L->nz = 1; L->nxh = ns; L->nzt = 1; L->nproc = 1; L->fx_nxh = ns; L->fx_nz = 1;
    L->nm = ns; L->norm = P.norm;
    L->tmin = P.tmin; L->tmax = P.tmax;
    L->d_L0[0].upload(L->tmin.L0); L->d_L1[0].upload(L->tmin.L1); L->d_R[0].upload(L->tmin.R);
    L->d_L0[1].upload(L->tmax.L0); L->d_L1[1].upload(L->tmax.L1); L->d_R[1].upload(L->tmax.R);
    std::vector<double> sub(ns);
    for (int s = 0; s < ns; ++s) { sub[s] = lam[cand[s]]; skip[cand[s]] = 1; }
    L->lam.upload(sub);
    hipc(hipMalloc((void **)&L->d_skip, (size_t)ns), "hipMalloc");
    hipc(hipMemset(L->d_skip, 0, (size_t)ns), "hipMemset");
    hipc(hipMalloc((void **)&L->d_sing, sizeof(int)), "hipMalloc");
    const size_t n = ny;
    L->hom.alloc(5 * n * ns); L->der.alloc(3 * (size_t)ns); L->cst.alloc(9 * (size_t)ns);
    L->scratch.alloc(6 * n * ns);
    L->v0.alloc(2 * n * ns); L->u0.alloc(2 * n * ns); L->du0.alloc(2 * (size_t)ns); L->bcs.alloc(4 * (size_t)ns);
    create_side_stream(&L->side);
    hipc(hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming), "event");
    hipc(hipEventCreateWithFlags(&L->ev_join, hipEventDisableTiming), "event");
    build_homogeneous(*L, st);
    {   // LU factors of both systems, stored once: the per-call solves of these few threads then only substitute
        L->fac[0].alloc(5 * n * ns); L->fac[1].alloc(5 * n * ns);
        Int1Args a = base_args(*L, 0, L->lam.p, ns, L->scratch.p);
        a.unit_row = 1; a.dst = L->v0.p; a.fac_out = L->fac[0].p;
        launch_int1<1, 2, FS_UNIT>(a, st);
        Int1Args b = base_args(*L, 1, L->lam.p, ns, L->scratch.p);
        b.unit_row = 1; b.dst = L->u0.p; b.fac_out = L->fac[1].p;
        launch_int1<2, 2, FS_UNIT>(b, st);
    }
    hipc(hipMalloc((void **)&P.d_low_modes, ns * sizeof(int)), "hipMalloc");
    hipc(hipMemcpy(P.d_low_modes, cand.data(), ns * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
    hipc(hipMemcpy(P.d_skip, skip.data(), (size_t)P.nm, hipMemcpyHostToDevice), "hipMemcpy");      // k_ode_nn leaves these columns alone
    P.n_low = ns;
    P.low_f.alloc(2 * n * ns); P.low_p.alloc(2 * n * ns); P.low_dp.alloc(2 * n * ns);
    P.low = std::move(L);
    create_side_stream(&P.side_low);
    hipc(hipEventCreateWithFlags(&P.ev_join_low, hipEventDisableTiming), "event");
}

}  // namespace

extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();

bool tlab_internal_poisson_has_own_x(tlab_poisson_plan_t P) { return P && P->fx_own; }
bool tlab_internal_poisson_can_v_final(tlab_poisson_plan_t P);

extern "C" {

// nz: planes of the local spectral box; [ioff, ioff+nxl) its kx range (nxl = 0: all nx/2+1); fx_nz: planes of the local PHYSICAL box
static int poisson_plan_create_impl(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                                    int nz, int nzt, int koff, int nproc, int ioff = 0, int nxl = 0, int fx_nz = 0,
                                    tlab_fdm_plan_t gy_ell = nullptr, bool helmholtz = false, double alpha = 0.0) {
    try {
        if (!out || !gx || !gy || !gz) throw std::invalid_argument("tlab_poisson_plan_create: null argument");
        if (!tlab_device_ready()) throw std::runtime_error("tlab_init has not been called (no CPU fallback exists)");
        if (gx->t.n != nx || gy->t.n != ny || gz->t.n != nzt) throw std::invalid_argument("plan sizes do not match nx, ny, nz");
        if (!gx->t.periodic || (nzt > 1 && !gz->t.periodic) || gy->t.periodic)
            throw std::invalid_argument("OPR_Poisson_FourierXZ needs periodic x, z and non-periodic y");
        if (nx % 2 != 0) throw std::invalid_argument("Imax must be a multiple of 2 for the FFT operations (opr_fourier.f90:72-75)");
        {   // host-built plans (tlab_fdm_plan_create_from_arrays) carry the modified wavenumbers only after tlab_fdm_plan_set_aux
            auto no_mwn = [](const tlab::DerTables &d) {
                for (double v : d.mwn) if (v != 0.0) return false;
                return true;
            };
            if (!gy_ell && (no_mwn(gx->t.der1) || (nzt > 1 && no_mwn(gz->t.der1))))
                throw std::invalid_argument("the x / z plans carry no modified wavenumbers (der1%mwn): call tlab_fdm_plan_set_aux");
            if (gy_ell && (no_mwn(gx->t.der2) || (nzt > 1 && no_mwn(gz->t.der2))))
                throw std::invalid_argument("the x / z plans carry no second-derivative modified wavenumbers (der2%mwn): call tlab_fdm_plan_set_aux");
            if (gy_ell && (gy_ell->t.n != ny || gy_ell->t.periodic || !gy_ell->t.der2.direct || gy_ell->t.der2.ndl != 3 || gy_ell->t.der2.ndr != 5 ||
                           (int)gy_ell->t.nodes.size() != ny))
                throw std::invalid_argument("direct elliptic solver: the elliptic y plan must hold a CompactDirect6 second derivative (3/5 diagonals) and its nodes");
        }
        if (nproc < 1 || nz * nproc != nzt || koff < 0 || koff + nz > nzt) throw std::invalid_argument("bad z-slab decomposition");
        if (((long long)(nx / 2 + 1) * ny) % nproc != 0) throw std::invalid_argument("(imax/2+1)*jmax must be divisible by the number of z slabs (tlab_mpi_transpose.f90:292)");
        auto P = std::make_unique<tlab_poisson_plan>();
        if (nxl < 0 || ioff < 0 || ioff + nxl > nx / 2 + 1) throw std::invalid_argument("bad kx range");
        P->nx = nx; P->ny = ny; P->nz = nz; P->nxh = nxl > 0 ? nxl : nx / 2 + 1;
        P->ioff = nxl > 0 ? ioff : 0;
        P->fx_nxh = nx / 2 + 1; P->fx_nz = fx_nz > 0 ? fx_nz : nz;
        {
            const char *e = getenv("TLAB_FFTZ_INPLACE");
            P->fz_inplace = e ? atoi(e) != 0 : false;
        }
        P->nzt = nzt; P->koff = koff; P->nproc = nproc;
        P->nm = (long long)P->nxh * nz;
        P->norm = 1.0 / ((double)nx * (double)nzt);                     // opr_elliptic.f90:130
        if (gy_ell) {   // TYPE_DIRECT (opr_elliptic.f90:152-163, 228-245)
            P->direct = true;
            P->exact_mode = g_poisson_exact;
            P->gy_der = gy;
            P->ell_der2 = gy_ell->t.der2;
            P->ell_nodes = gy_ell->t.nodes;
            const long long nm = P->nm;
            std::vector<double> lam((size_t)nm);
            for (int k = 0; k < nz; ++k)
                for (int i = 0; i < P->nxh; ++i) {
                    double l2 = gx->t.der2.mwn[P->ioff + i];                       // lambda = mwn2_x + mwn2_z (:230-234)
                    if (nzt > 1) l2 += gz->t.der2.mwn[koff + k];
                    lam[(size_t)i + (size_t)P->nxh * k] = l2;
                }
            if (P->ioff == 0 && koff == 0) P->sing_direct = 0;                      // i_sing = k_sing = [1, 1] (:160-161)
            P->lam.upload(lam);
            (void)P->dev2(TLAB_BCS_NN);
            if (P->sing_direct >= 0) (void)P->dev2(TLAB_BCS_DN);
            P->scratch.alloc((size_t)5 * ny * nm);
            P->cwork.alloc((size_t)2 * P->nxh * ny * nz);
            build_fft(*P);
            *out = P.release();
            return TLAB_OK;
        }
        if (int1_generic_applies(gy->t.der1)) {      // (3, 3) / (5, 7) diagonals: TRIDFS / HEPTADFS systems, factorized on the host (int1_generic.cpp)
            if (gy->t.periodic) throw std::invalid_argument("Poisson: the wall-normal direction must not be periodic");
            P->generic = true;
            P->gder = gy->t.der1;
        } else {
            int1_build_tables(gy->t.der1, 1, P->tmin);
            int1_build_tables(gy->t.der1, 2, P->tmax);
            P->d_L0[0].upload(P->tmin.L0); P->d_L1[0].upload(P->tmin.L1); P->d_R[0].upload(P->tmin.R);
            P->d_L0[1].upload(P->tmax.L0); P->d_L1[1].upload(P->tmax.L1); P->d_R[1].upload(P->tmax.R);
            for (int w = 0; w < 2; ++w) {
                const Int1Tables &T = w == 0 ? P->tmin : P->tmax;
                if (T.L0.size() < (size_t)6 * ny || T.L1.size() < (size_t)5 * ny || T.R.size() < (size_t)3 * ny) continue;      // (generic tables: k_int1g)
                std::vector<double> pk((size_t)16 * ny, 0.0);
                for (int j = 0; j < ny; ++j) {
                    for (int k = 0; k < 5; ++k) { pk[(size_t)16 * j + k] = T.L0[(size_t)5 * j + k]; pk[(size_t)16 * j + 5 + k] = T.L1[(size_t)5 * j + k]; }
                    pk[(size_t)16 * j + 10] = T.L0[(size_t)5 * ny + j];
                    pk[(size_t)16 * j + 11] = T.R[(size_t)3 * j + 0];
                    pk[(size_t)16 * j + 12] = T.R[(size_t)3 * j + 1];
                    pk[(size_t)16 * j + 13] = T.R[(size_t)3 * j + 2];
                }
                P->d_pk[w].upload(pk);
            }
        }
        // lambda(k,i) = mwn_x(i)^2 + mwn_z(k)^2 (opr_elliptic.f90:199-203), stored as sqrt (:205-209)
        const long long nm = P->nm;
        std::vector<double> lam((size_t)nm);
        std::vector<unsigned char> skip((size_t)nm, 0);
        for (int k = 0; k < nz; ++k)
            for (int i = 0; i < P->nxh; ++i) {
                double l2 = std::pow(gx->t.der1.mwn[P->ioff + i], 2.0);
                if (nzt > 1) l2 += std::pow(gz->t.der1.mwn[koff + k], 2.0);   // kglobal = k + ims_offset_k (:191)
                if (helmholtz) {                                             // sqrt(lambda(k,i) - alpha) (:518-522)
                    if (!(l2 - alpha > 0.0)) throw std::invalid_argument("OPR_Helmholtz (factorized): lambda - alpha must be positive for every mode");
                    l2 = l2 - alpha;
                }
                lam[(size_t)i + (size_t)P->nxh * k] = std::sqrt(l2);
            }
        // i_sing, k_sing (:148-149), 0-based, global; with the staggered pressure grid only (1, 1) is singular: the interpolatory modified
        // wavenumbers do not vanish at the Nyquist modes (:144-146)
        const bool stag = gx->t.stagger || (nzt > 1 && gz->t.stagger);
        const int isg[2] = {0, stag ? 0 : nx / 2}, ksg[2] = {0, (nzt > 1 && !stag) ? nzt / 2 : 0};
        for (int a = 0; a < 2 && !helmholtz; ++a)                           // Helmholtz: every mode is a regular one (:512-531)
            for (int b = 0; b < 2; ++b) {
                const int kl = ksg[b] - koff;                               // task-local index (:177-178)
                const int il = isg[a] - P->ioff;
                if (kl < 0 || kl >= nz || il < 0 || il >= P->nxh) continue;
                const int t = il + P->nxh * kl;
                if (!skip[t]) { skip[t] = 1; P->sing_modes.push_back(t); }
            }
        P->lam.upload(lam);
        hipc(hipMalloc((void **)&P->d_skip, (size_t)nm), "hipMalloc");
        hipc(hipMemcpy(P->d_skip, skip.data(), (size_t)nm, hipMemcpyHostToDevice), "hipMemcpy");
        const int ns = (int)P->sing_modes.size();
        hipc(hipMalloc((void **)&P->d_sing, (ns + 1) * sizeof(int)), "hipMalloc");
        if (ns) hipc(hipMemcpy(P->d_sing, P->sing_modes.data(), ns * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy");
        std::vector<double> slam(ns);
        for (int s = 0; s < ns; ++s) slam[s] = lam[P->sing_modes[s]];
        P->s_lam.upload(slam);
        const size_t n = ny;
        P->hom.alloc(5 * n * nm); P->der.alloc(3 * nm); P->cst.alloc(9 * nm);
        P->scratch.alloc(6 * n * nm);      // NL + 3 components, NL <= 3
        P->v0.alloc(2 * n * nm); P->u0.alloc(2 * n * nm); P->du0.alloc(2 * nm); P->bcs.alloc(4 * nm);
        P->helmholtz = helmholtz;
        if (!helmholtz) P->cwork.alloc((size_t)2 * P->nxh * ny * nz);
        if (nproc == 1 && nxl == 0) { P->g3[0] = gx; P->g3[1] = gy; P->g3[2] = gz; }
        P->s_f.alloc(2 * n * ns); P->s_bct.alloc(2 * ns); P->s_v0.alloc(2 * n * ns); P->s_v1.alloc(2 * n * ns);
        P->s_u0.alloc(2 * n * ns); P->s_u1.alloc(2 * n * ns); P->s_du0.alloc(2 * ns); P->s_du1.alloc(2 * ns); P->s_scr.alloc(5 * n * ns);
        if (!helmholtz) build_fft(*P);
        {   // fused 2-D (x,z) transforms are ~2x faster than r2c(x) + strided c2c(z) at 512^3, but rocFFT does not build them
            // for every layout: fall back to the two 1-D plans when plan creation fails (TLAB_FFT2D=0 forces the 1-D path)
            const char *e = getenv("TLAB_FFT2D");
            // ... and slower than r2c(x) + the own strided z-transform (fftz.hip: 0.49 + 0.53 ms against 1.14 ms at 512^3), so they are only
            // built where that kernel does not apply (TLAB_FFTZ=0 or a length that is not 8^a * {1,2,4})
            if (!helmholtz && nzt > 1 && nproc == 1 && nxl == 0 && !P->fz_own && !(e && atoi(e) == 0)) {
                try {
                    build_fft_2d(*P);
                    P->use_2d = true;
                } catch (const std::exception &) {
                    P->use_2d = false;
                }
            }
        }
        create_side_stream(&P->side);
        hipc(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming), "event");
        hipc(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming), "event");
        hipStream_t st = tlab_current_stream();
        build_homogeneous(*P, st);
        build_singular_homogeneous(*P, st);
        {   // chunked ODE kernel (k_ode_nn) when the line splits into 8-row chunks and 32-bit indices suffice; tlab_poisson_set_exact(1)
            // (or TLAB_ODE_CHUNKED=0) keeps the marching kernels, which repeat the reference's operations one by one
            const char *e = g_poisson_exact ? "0" : getenv("TLAB_ODE_CHUNKED");
            const int C = ny / OM;
            const long long big = std::max<long long>((long long)5 * ny * (nm + 64), std::max<long long>(9 * nm, (long long)P->nxh * ny * nz));
            if (!P->generic && !(e && atoi(e) == 0) && ny % OM == 0 && C >= 2 && ode_modes_per_wg(C) > 0 && big < (1LL << 31) &&
                ode_lds_bytes(C, ode_modes_per_wg(C)) <= (size_t)160 * 1024) {
                P->ode_nm_per_wg = ode_modes_per_wg(C);
                {   // mirror pairs (see k_ode_nn): TLAB_ODE_PAIR=0 keeps one mode per thread
                    const char *pe = getenv("TLAB_ODE_PAIR");
                    const int om = OM;
                    const int nzm = (int)(nm / P->nxh);
                    bool sym = !(pe && atoi(pe) == 0) && nzm >= 4 && (om == 4 || om == 8) && ny % om == 0;
                    for (int kz = 1; sym && kz < nzm; ++kz)
                        for (long long i = 0; i < P->nxh; ++i) sym = sym && lam[i + P->nxh * kz] == lam[i + P->nxh * (nzm - kz)];
                    const int Cp = ny / om, NMp = sym ? ode_modes_per_wg(Cp) : 0;
                    if (sym && NMp > 0 && ode_pair_geometry(NMp, om) && ode_lds_bytes(Cp, NMp, om, 4) <= (size_t)160 * 1024) {
                        P->ode_pair = true; P->ode_om = om; P->ode_nm_per_wg = NMp;
                    }
                }
                build_checkpoints(*P, st);
                P->use_chunked = true;
                if (ode_sing_nm(C) * C <= 512 && (int)P->sing_modes.size() <= ode_sing_nm(C) && ode_lds_bytes(C, ode_sing_nm(C)) <= (size_t)160 * 1024)
                    build_singular_checkpoints(*P, st);
                else P->use_chunked = false;
            }
        }
        if (P->use_chunked) build_low_modes(*P, gy->t.nodes, lam, skip, st);
        hipc(hipStreamSynchronize(st), "sync");
        if (P->use_chunked) {   // the scratch of the marching kernels is not needed any more
            P->scratch.alloc(0); P->v0.alloc(0); P->u0.alloc(0); P->hom.alloc(0);      // hom lives on in its blocked copy
        }
        *out = P.release();
        return TLAB_OK;
    } catch (const std::invalid_argument &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EHIP;
    }
}

int tlab_poisson_plan_create(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                             int nz) {
    return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, nz, nz, 0, 1);
}

int tlab_poisson_set_exact(int on) {
    g_poisson_exact = on != 0;
    return TLAB_OK;
}

int tlab_poisson_plan_create_direct(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                                    int nz, tlab_fdm_plan_t gy_elliptic) {
    if (!gy_elliptic) {
        tlab_set_error("tlab_poisson_plan_create_direct: null elliptic plan");
        return TLAB_EINVAL;
    }
    return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, nz, nz, 0, 1, 0, 0, 0, gy_elliptic);
}

int tlab_poisson_plan_create_slab(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                                  int kmax, int nz_total, int koffset, int nproc_k) {
    return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, kmax, nz_total, koffset, nproc_k);
}

int tlab_poisson_plan_create_pencil(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx, int ny,
                                    int kmax, int nz_total, int ioffset, int nxl) {
    if (nxl <= 0 || kmax <= 0 || nz_total % kmax != 0) {
        tlab_set_error("tlab_poisson_plan_create_pencil: bad decomposition");
        return TLAB_EINVAL;
    }
    return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, nz_total, nz_total, 0, 1, ioffset, nxl, kmax);
}

// decomposed variants of a direct plan (EllipticOrder = CompactDirect6): mode = 0 z-slab (K-transposes), 1 kx-pencil; a, b as in the
// factorized creators: (koffset, nproc_k) or (ioffset, nxl)
int tlab_poisson_plan_create_direct_decomposed(tlab_poisson_plan_t *out, tlab_fdm_plan_t gx, tlab_fdm_plan_t gy, tlab_fdm_plan_t gz, int nx,
                                               int ny, int kmax, int nz_total, int mode, int a, int b, tlab_fdm_plan_t gy_elliptic) {
    if (!gy_elliptic || kmax <= 0 || nz_total % kmax != 0 || (mode != 0 && mode != 1) || (mode == 1 && b <= 0)) {
        tlab_set_error("tlab_poisson_plan_create_direct_decomposed: bad arguments");
        return TLAB_EINVAL;
    }
    if (mode == 0) return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, kmax, nz_total, a, b, 0, 0, 0, gy_elliptic);
    return poisson_plan_create_impl(out, gx, gy, gz, nx, ny, nz_total, nz_total, 0, 1, a, b, kmax, gy_elliptic);
}

int tlab_poisson_plan_destroy(tlab_poisson_plan_t p) {
    delete p;
    return TLAB_OK;
}

// ODE stage on the local modes: f_hat (complex (nxh, ny, kmax), unnormalised FFT output) -> p_hat, dp_hat.
// p_hat may alias f_hat (the reference also overwrites); dp_hat must be a different array.
// FDM_Int2_Solve of every local mode (opr_elliptic.f90:413-434)
// helmholtz: OPR_Helmholtz_FourierXZ_Direct (:562-628): system constant lambda2 - alpha for every mode, no singular-mode treatment
static void poisson_direct_stage(tlab_poisson_plan_t P, int ibc, double *f_hat, double *p_hat, hipStream_t st, bool helmholtz = false, double alpha = 0.0) {
    Int2Args a{};
    a.T = P->dev2(ibc);
    a.lam = P->lam.p; a.nm = P->nm; a.first = 0; a.count = P->nm;
    a.alpha = helmholtz ? alpha : 0.0;
    a.skip = (ibc == TLAB_BCS_NN && !helmholtz) ? P->sing_direct : -1;   // singular mode: BCS_DN system with p = 0 at the bottom (:236-240, :420-424)
    a.fsrc = f_hat; a.dst = p_hat; a.fscale = P->norm; a.nxh = P->nxh; a.ny = P->ny;
    a.zero_bottom = 0;
    a.neumann_b = (ibc == TLAB_BCS_ND || ibc == TLAB_BCS_NN) ? 1 : 0;
    a.neumann_t = (ibc == TLAB_BCS_DN || ibc == TLAB_BCS_NN) ? 1 : 0;
    a.scratch = P->scratch.p;
    // chunked kernel (k_int2c) where the line splits into 8-row chunks and 32-bit indices suffice; TLAB_INT2_CHUNKED=0 (read per call) or
    // tlab_poisson_set_exact(1) keep the marching kernel, which repeats the reference's operation order
    const int C = P->ny / OM, NM = (P->ny % OM == 0 && P->ny >= 2 * OM) ? ode_modes_per_wg(C) : 0;
    const char *ce = getenv("TLAB_INT2_CHUNKED");
    const bool chunked = NM > 0 && !(ce && atoi(ce) == 0) && !P->exact_mode && (double)P->nm * P->ny * 2.0 < 2.0e9;
    if (chunked) {
        auto &E = *P->int2[ibc];
        const long long nblk = (P->nm + NM - 1) / NM;
        if (!E.chk_ok || E.chk_alpha != a.alpha) {
            if (E.chk.n != (size_t)C * 6 * nblk * NM) E.chk.alloc((size_t)C * 6 * nblk * NM);
            hipLaunchKernelGGL(k_int2_checkpoint, dim3((unsigned)((P->nm + 255) / 256)), dim3(256), 0, st, a.T, P->lam.p, a.alpha, a.neumann_b, a.neumann_t,
                               E.chk.p, P->nm, NM, C);
            hipc(hipGetLastError(), "k_int2_checkpoint");
            E.chk_alpha = a.alpha; E.chk_ok = true;
        }
        Int2cArgs k{};
        k.T = a.T; k.lam = a.lam; k.alpha = a.alpha; k.nm = P->nm; k.skip = a.skip; k.chk = E.chk.p; k.fsrc = f_hat; k.dst = p_hat; k.fscale = a.fscale;
        k.nxh = a.nxh; k.ny = a.ny; k.C = C; k.neumann_b = a.neumann_b; k.neumann_t = a.neumann_t;
        const size_t lds = ((size_t)64 * NM + (size_t)(3 * OM + 1) * NM * C) * sizeof(double);
        ProfScope ps("k_int2c", st, (double)P->nm * P->ny * 32.0);
        switch (NM) {
        case 4: launch_int2c<4>(k, lds, st); break;
        case 8: launch_int2c<8>(k, lds, st); break;
        case 16: launch_int2c<16>(k, lds, st); break;
        case 32: launch_int2c<32>(k, lds, st); break;
        default: launch_int2c<64>(k, lds, st); break;
        }
    } else {
        ProfScope ps("k_int2", st, (double)P->nm * P->ny * 32.0);
        hipLaunchKernelGGL((k_int2<4>), dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, st, a);
    }
    if (a.skip >= 0) {
        Int2Args b = a;
        b.T = P->dev2(TLAB_BCS_DN);
        b.first = a.skip; b.count = 1; b.skip = -1; b.zero_bottom = 1; b.neumann_b = 0; b.neumann_t = 1;
        hipLaunchKernelGGL((k_int2<4>), dim3(1), dim3(64), 0, st, b);
    }
    hipc(hipGetLastError(), "k_int2");
}

// ibc = BCS_DD on a factorized plan (opr_elliptic.f90:322-329).  Chunked plans: the regular modes in k_ode_nn<DD> (the BCS_NN kernel with the
// top value given and the two constants of OPR_ODE2_Factorize_DD), the <= 4 singular modes (_DD_Sing) and the lowest-lambda modes (marching
// sub-plan) beside it on the side stream, as for BCS_NN; other plans: marching kernels for every mode.  TLAB_ODE_DD_CHUNKED=0 keeps the marching route.
static void poisson_dd_stage(tlab_poisson_plan_t P, double *f_hat, double *p_hat, double *dp_hat, hipStream_t st) {
    const long long nm = P->nm;
    const int n = P->ny, nxh = P->nxh, ny = P->ny;
    const int ns = (int)P->sing_modes.size();
    const char *dd_env = getenv("TLAB_ODE_DD_CHUNKED");      // read per call: the tests switch between the two routes of one plan
    const bool chunked = P->use_chunked && !(dd_env && atoi(dd_env) == 0);
    hipStream_t main_st = st;
    if (chunked) {
        if (P->cst_dd.n == 0) {
            P->cst_dd.alloc((size_t)5 * nm);
            hipLaunchKernelGGL(k_dd_constants, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, P->homb.p, P->ode_nm_per_wg, P->der.p, P->cst_dd.p, n, nm);
            hipc(hipGetLastError(), "k_dd_constants");
        }
        hipc(hipEventRecord(P->ev_fork, st), "event record");
        hipc(hipStreamWaitEvent(P->side, P->ev_fork, 0), "stream wait");
        st = P->side;            // the singular and the low modes run beside k_ode_nn<DD>, which leaves their columns alone
    } else if (P->scratch.n == 0) {     // a chunked plan released the work arrays of the marching kernels: they come back
        P->scratch.alloc((size_t)6 * n * nm); P->v0.alloc((size_t)2 * n * nm); P->u0.alloc((size_t)2 * n * nm);
    }
    // ---- singular modes first (they read f^ before the regular combine may overwrite it when p_hat aliases f_hat) ----
    if (ns > 0) {
        if (!P->dd_ready) {      // v^(1): v' = delta_n, v(1) = 0 ; u^(1): u' = v1, u(n) = 0 ; s^(+): u' = 1, u(n) = 0   (opr_odes.f90:216-236)
            P->dd_v1.alloc((size_t)2 * n * ns); P->dd_u1.alloc((size_t)2 * n * ns); P->dd_du1.alloc((size_t)2 * ns);
            P->dd_sp.alloc((size_t)2 * n * ns); P->dd_ones.alloc((size_t)n * ns); P->dd_bcb.alloc((size_t)2 * ns);
            hipLaunchKernelGGL(k_fill_ones, dim3((unsigned)(((long long)n * ns + 255) / 256)), dim3(256), 0, st, P->dd_ones.p, (long long)n * ns);
            Int1Args h1 = base_args(*P, 0, P->s_lam.p, ns, P->s_scr.p);
            h1.unit_row = n - 1; h1.zero_bsave = 0; h1.dst = P->dd_v1.p;
            launch_int1<1, 2, FS_UNIT>(h1, st);
            Int1Args h2 = base_args(*P, 1, P->s_lam.p, ns, P->s_scr.p);
            h2.fsrc = P->dd_v1.p; h2.nlf = 1; h2.zero_bsave = 0; h2.dst = P->dd_u1.p; h2.du = P->dd_du1.p;
            launch_int1<2, 2, FS_LINEAR>(h2, st);
            Int1Args h3 = base_args(*P, 1, P->s_lam.p, ns, P->s_scr.p);
            h3.fsrc = P->dd_ones.p; h3.nlf = 1; h3.zero_bsave = 0; h3.dst = P->dd_sp.p;
            launch_int1<2, 2, FS_LINEAR>(h3, st);
            P->dd_ready = true;
        }
        dim3 g(ns, (n + 63) / 64), blk(64);
        hipLaunchKernelGGL(k_sing_gather_dd, g, blk, 0, st, f_hat, P->d_sing, ns, n, nxh, ny, P->norm, P->s_f.p, P->dd_bcb.p, P->s_bct.p);
        Int1Args s1 = base_args(*P, 0, P->s_lam.p, ns, P->s_scr.p);        // v' = f (f(n) = 0), v(1) = 0
        s1.fsrc = P->s_f.p; s1.nlf = 2; s1.zero_bsave = 1; s1.dst = P->s_v0.p;
        launch_int1<1, 2, FS_LINEAR>(s1, st);
        Int1Args s2 = base_args(*P, 1, P->s_lam.p, ns, P->s_scr.p);        // u' = v, u(n) = bcs_t
        s2.fsrc = P->s_v0.p; s2.nlf = 2; s2.zero_bsave = 0; s2.bv_ptr = P->s_bct.p; s2.dst = P->s_u0.p; s2.du = P->s_du0.p;
        launch_int1<2, 2, FS_LINEAR>(s2, st);
    }
    if (chunked) {
        if (ns > 0) {
            dim3 g(ns, (n + 63) / 64), blk(64);
            hipLaunchKernelGGL(k_sing_combine_dd, g, blk, 0, st, P->s_u0.p, P->s_v0.p, P->dd_u1.p, P->dd_v1.p, P->dd_sp.p, P->s_du0.p, P->dd_du1.p,
                               P->dd_bcb.p, P->d_sing, ns, n, nxh, ny, p_hat, dp_hat);
        }
        if (P->low) {            // the lowest-lambda modes through the marching sub-plan (poisson_ode_stage does the same for BCS_NN)
            const int nl = P->n_low;
            const dim3 g((nl + 63) / 64, n), blk(64);
            hipLaunchKernelGGL(k_modes_gather, g, blk, 0, st, reinterpret_cast<const double2 *>(f_hat), P->d_low_modes, nl, n, nxh, ny,
                               reinterpret_cast<double2 *>(P->low_f.p));
            poisson_dd_stage(P->low.get(), P->low_f.p, P->low_p.p, P->low_dp.p, st);
            hipLaunchKernelGGL(k_modes_scatter, g, blk, 0, st, reinterpret_cast<const double2 *>(P->low_p.p),
                               reinterpret_cast<const double2 *>(P->low_dp.p), P->d_low_modes, nl, n, nxh, ny, reinterpret_cast<double2 *>(p_hat),
                               reinterpret_cast<double2 *>(dp_hat));
        }
        launch_ode(*P, f_hat, p_hat, dp_hat, main_st, true);
        hipc(hipEventRecord(P->ev_join, st), "event record");
        hipc(hipStreamWaitEvent(main_st, P->ev_join, 0), "stream wait");
        hipc(hipGetLastError(), "BCS_DD kernels");
        return;
    }
    // ---- regular modes ----
    Int1Args a = base_args(*P, 0, P->lam.p, nm, P->scratch.p);             // v' + l v = f, v(1) = 0
    a.fsrc = f_hat; a.fscale = P->norm; a.zero_bsave = 1; a.bcs_save = P->bcs.p; a.dst = P->v0.p;
    launch_int1<1, 2, FS_FIELD>(a, st);
    Int1Args b = base_args(*P, 1, P->lam.p, nm, P->scratch.p);             // u' - l u = v, u(n) = bcs_t
    b.fsrc = P->v0.p; b.nlf = 2; b.zero_bsave = 0; b.bv_ptr = P->bcs.p + (size_t)2 * nm; b.dst = P->u0.p; b.du = P->du0.p;
    launch_int1<2, 2, FS_LINEAR>(b, st);
    DDCombineArgs c{};
    c.u0 = P->u0.p; c.v0 = P->v0.p; c.du0 = P->du0.p; c.bcs = P->bcs.p; c.der = P->der.p; c.lam = P->lam.p;
    c.hom = P->use_chunked ? P->homb.p : P->hom.p;
    c.hom_nm_block = P->use_chunked ? P->ode_nm_per_wg : 0;
    c.sing = P->d_sing; c.ns = ns;
    c.p_hat = p_hat; c.dp_hat = dp_hat; c.n = n; c.nxh = nxh; c.ny = ny; c.nm = nm;
    hipLaunchKernelGGL(k_dd_combine, dim3((unsigned)((nm + 255) / 256), nm <= 4096 ? (unsigned)((n + 15) / 16) : 1u), dim3(256), 0, st, c);
    if (ns > 0) {
        dim3 g(ns, (n + 63) / 64), blk(64);
        hipLaunchKernelGGL(k_sing_combine_dd, g, blk, 0, st, P->s_u0.p, P->s_v0.p, P->dd_u1.p, P->dd_v1.p, P->dd_sp.p, P->s_du0.p, P->dd_du1.p,
                           P->dd_bcb.p, P->d_sing, ns, n, nxh, ny, p_hat, dp_hat);
    }
    hipc(hipGetLastError(), "BCS_DD kernels");
}

static void poisson_ode_stage(tlab_poisson_plan_t P, double *f_hat, double *p_hat, double *dp_hat, hipStream_t st) {
    if (P->direct) throw std::invalid_argument("direct elliptic plan: use tlab_poisson_direct_ode (there is no dp^/dy; dp/dy is OPR_Partial_Y of p)");
    const long long nm = P->nm;
    const int n = P->ny, nxh = P->nxh, ny = P->ny;
    hipc(hipEventRecord(P->ev_fork, st), "event record");
    hipc(hipStreamWaitEvent(P->side, P->ev_fork, 0), "stream wait");
    // ---- regular modes: OPR_ODE2_Factorize_NN (opr_odes.f90:302-318) ----
    if (!P->use_chunked) {
        Int1Args a = base_args(*P, 0, P->lam.p, nm, P->scratch.p);   // v' + l v = f, v(1) = 0
        a.fsrc = f_hat; a.fscale = P->norm; a.zero_bsave = 1; a.bcs_save = P->bcs.p; a.dst = P->v0.p;
        if (P->fac[0].n) a.fac = P->fac[0].p;
        launch_int1<1, 2, FS_FIELD>(a, st);
        Int1Args b = base_args(*P, 1, P->lam.p, nm, P->scratch.p);   // u' - l u = v, u(n) = 0
        b.fsrc = P->v0.p; b.nlf = 2; b.zero_bsave = 0; b.dst = P->u0.p; b.du = P->du0.p;
        if (P->fac[1].n) b.fac = P->fac[1].p;
        launch_int1<2, 2, FS_LINEAR>(b, st);
    }
    // ---- singular modes: OPR_ODE2_Factorize_NN_Sing -> _DN_Sing (opr_odes.f90:165-183, 37-96) ----
    const int ns = (int)P->sing_modes.size();
    hipStream_t ss = P->side;   // independent of the regular modes until the scatter below
    if (ns > 0 && P->use_chunked) {
        launch_ode_sing(*P, f_hat, p_hat, dp_hat, ss);      // one workgroup beside the regular modes; writes only the singular entries
    }
    if (P->use_chunked && P->low) {                         // the lowest-lambda modes: marching sub-plan, also beside the regular ones
        const int nl = P->n_low;
        const dim3 g((nl + 63) / 64, n), blk(64);
        hipStream_t ls = P->side_low;
        hipc(hipStreamWaitEvent(ls, P->ev_fork, 0), "stream wait");
        hipLaunchKernelGGL(k_modes_gather, g, blk, 0, ls, reinterpret_cast<const double2 *>(f_hat), P->d_low_modes, nl, n, nxh, ny,
                           reinterpret_cast<double2 *>(P->low_f.p));
        poisson_ode_stage(P->low.get(), P->low_f.p, P->low_p.p, P->low_dp.p, ls);
        hipLaunchKernelGGL(k_modes_scatter, g, blk, 0, ls, reinterpret_cast<const double2 *>(P->low_p.p),
                           reinterpret_cast<const double2 *>(P->low_dp.p), P->d_low_modes, nl, n, nxh, ny, reinterpret_cast<double2 *>(p_hat),
                           reinterpret_cast<double2 *>(dp_hat));
        hipc(hipEventRecord(P->ev_join_low, ls), "event record");
    }
    if (ns > 0 && P->use_chunked) {
    } else if (ns > 0) {
        dim3 g(ns, (n + 63) / 64), blk(64);
        hipLaunchKernelGGL(k_sing_gather, g, blk, 0, ss, f_hat, P->d_sing, ns, n, nxh, ny, P->norm, P->s_f.p, P->s_bct.p);
        Int1Args s1 = base_args(*P, 1, P->s_lam.p, ns, P->s_scr.p);   // v' = f (f(1)=0), v(n) = bcs_t
        s1.fsrc = P->s_f.p; s1.nlf = 2; s1.zero_bsave = 0; s1.bv_ptr = P->s_bct.p; s1.dst = P->s_v0.p;
        launch_int1<2, 2, FS_LINEAR>(s1, ss);
        Int1Args s3 = base_args(*P, 0, P->s_lam.p, ns, P->s_scr.p);   // u' = v, u(1) = bcs_b = 0
        s3.fsrc = P->s_v0.p; s3.nlf = 2; s3.zero_bsave = 0; s3.dst = P->s_u0.p; s3.du = P->s_du0.p;
        launch_int1<1, 2, FS_LINEAR>(s3, ss);
    }
    // ---- superposition ----
    if (!P->use_chunked) {
        CombineArgs c{};
        c.u0 = P->u0.p; c.v0 = P->v0.p; c.du0 = P->du0.p; c.bcs = P->bcs.p; c.hom = P->hom.p; c.cst = P->cst.p; c.lam = P->lam.p;
        c.skip = P->d_skip; c.p_hat = p_hat; c.dp_hat = dp_hat; c.n = n; c.nxh = nxh; c.ny = ny; c.nm = nm;
        ProfScope ps("k_nn_combine", st, (double)nm * n * (9 + 4) * 8.0);
        hipLaunchKernelGGL(k_nn_combine, dim3((unsigned)((nm + 255) / 256), nm <= 4096 ? (unsigned)((n + 15) / 16) : 1u), dim3(256), 0, st, c);
    } else {
        launch_ode(*P, f_hat, p_hat, dp_hat, st);      // both solves, the constants and the final pass in one kernel
    }
    // the singular modes write other entries than k_nn_combine (which skips them), but f^ must have been consumed by the regular
    // v-solve first when p_hat aliases it: order the scatter after it through the main stream
    hipc(hipEventRecord(P->ev_join, ss), "event record");
    hipc(hipStreamWaitEvent(st, P->ev_join, 0), "stream wait");
    if (P->use_chunked && P->low) hipc(hipStreamWaitEvent(st, P->ev_join_low, 0), "stream wait");
    if (ns > 0 && !P->use_chunked) {
        dim3 g(ns, (n + 63) / 64), blk(64);
        hipLaunchKernelGGL(k_sing_combine, g, blk, 0, st, P->s_u0.p, P->s_v0.p, P->s_u1.p, P->s_v1.p, P->s_du0.p, P->s_du1.p,
                           P->d_sing, ns, n, nxh, ny, p_hat, dp_hat);
    }
    hipc(hipGetLastError(), "poisson kernels");
}

#define POISSON_GUARD_BEGIN try {
#define POISSON_GUARD_END                           \
    return TLAB_OK;                                 \
    }                                               \
    catch (const std::invalid_argument &e) {        \
        tlab_set_error(e.what());                   \
        return TLAB_EINVAL;                         \
    }                                               \
    catch (const std::exception &e) {               \
        tlab_set_error(e.what());                   \
        return TLAB_EHIP;                           \
    }

int tlab_opr_poisson(tlab_poisson_plan_t P, int nx, int ny, int nz, int ibc, double *p, double *tmp1, double *tmp2,
                     const double *bcs_hb, const double *bcs_ht, double *dpdy) {
    POISSON_GUARD_BEGIN
    if (!P || !p || !tmp1 || !tmp2 || !bcs_hb || !bcs_ht) throw std::invalid_argument("tlab_opr_poisson: null argument");
    const tlab_poisson_plan::VFinal vf = P->vfinal;      // the request holds for THIS call only, whatever its outcome
    P->vfinal.armed = false;
    if (vf.armed && (!dpdy || !tlab_internal_poisson_can_v_final(P))) throw std::invalid_argument("tlab_opr_poisson: internal: the fused v update was requested from a plan that cannot do it");
    if (nx != P->nx || ny != P->ny || nz != P->nz) throw std::invalid_argument("tlab_opr_poisson: sizes do not match the plan");
    if (P->nproc != 1 || P->nxh != P->fx_nxh || P->fx_nz != P->nz)
        throw std::invalid_argument("tlab_opr_poisson: plan is a z-slab / kx-pencil plan; drive its stages with the transposes in between");
    if (ibc != TLAB_BCS_NN && ibc != TLAB_BCS_DD && !P->direct) {
        tlab_set_error("OPR_Poisson (factorized): BCS_NN and BCS_DD only, like the reference (opr_elliptic.f90:312-331)");
        return TLAB_EUNSUPPORTED;
    }
    if (ibc < TLAB_BCS_DD || ibc > TLAB_BCS_NN) throw std::invalid_argument("tlab_opr_poisson: bad ibc");
    if (p == tmp1 || p == tmp2 || tmp1 == tmp2 || dpdy == p || dpdy == tmp1 || dpdy == tmp2) throw std::invalid_argument("arrays must be distinct");
    if (P->direct) {    // OPR_Poisson_FourierXZ_Direct (opr_elliptic.f90:368-455)
        hipStream_t st = tlab_current_stream();
        hipLaunchKernelGGL(k_set_wall_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, p, bcs_hb, bcs_ht, nx, ny, nz);
        if (nz > 1) {
            P->x_forward(p, tmp2, st);
            if (P->fz_own) P->fz_own->exec(1, tmp2, tmp1, st);
            else P->fz_f.exec(tmp2, tmp1, st);
        } else {
            P->x_forward(p, tmp1, st);
        }
        poisson_direct_stage(P, ibc, tmp1, tmp1, st);
        if (nz > 1) {
            if (P->fz_own) P->fz_own->exec(-1, tmp1, P->cwork.p, st);
            else P->fz_b.exec(tmp1, P->cwork.p, st);
            P->x_backward_p(P->cwork.p, p, st);
        } else {
            P->x_backward_p(tmp1, p, st);
        }
        if (dpdy) {     // :447-449, with the y plan of the derivatives
            const int rc = tlab_opr_partial(2, P->gy_der, TLAB_OPR_P1, nx, ny, nz, 0, p, dpdy, tmp1);
            if (rc != TLAB_OK) return rc;
        }
        return TLAB_OK;
    }
    hipStream_t st = tlab_current_stream();
    // BC planes into the forcing (opr_elliptic.f90:285-286)
    hipLaunchKernelGGL(k_set_wall_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, p, bcs_hb, bcs_ht, nx, ny, nz);
    // forward transforms: p -> tmp2 -> tmp1 (:288-293); the scaling by norm (:295) is folded into the loads of the ODE stage
    if (P->use_2d) {
        P->f2_fwd.exec(p, tmp1, st);
    } else if (nz > 1) {
        P->x_forward(p, tmp2, st);
        if (P->fz_own) P->fz_own->exec(1, tmp2, tmp1, st);
        else P->fz_f.exec(tmp2, tmp1, st);
    } else {
        P->x_forward(p, tmp1, st);
    }
    if (ibc == TLAB_BCS_DD) poisson_dd_stage(P, tmp1, tmp1, tmp2, st);
    else poisson_ode_stage(P, tmp1, tmp1, tmp2, st);      // p^ -> tmp1 (over f^), dp^/dy -> tmp2
    // backward transforms (:341-356)
    if (P->use_2d) {
        P->f2_bwd.exec(tmp1, p, st);
        if (dpdy) P->f2_bwd.exec(tmp2, dpdy, st);
    } else if (nz > 1) {
        if (P->fz_own) P->fz_own->exec(-1, tmp1, P->cwork.p, st);
        else P->fz_b.exec(tmp1, P->cwork.p, st);
        P->x_backward_p(P->cwork.p, p, st);
        if (dpdy) {
            if (P->fz_own) P->fz_own->exec(-1, tmp2, P->cwork.p, st);
            else P->fz_b.exec(tmp2, P->cwork.p, st);
            P->x_backward_dpdy(P->cwork.p, dpdy, st, vf);
        }
    } else {
        P->x_backward_p(tmp1, p, st);
        if (dpdy) P->x_backward_dpdy(tmp2, dpdy, st, vf);
    }
    POISSON_GUARD_END
}

// ---- stages, for the z-slab (multi-GPU) driver: the K-transposes of OPR_Fourier_Z_* (opr_fourier.f90:343-376) happen between them ----
int tlab_poisson_set_wall_planes(tlab_poisson_plan_t P, double *p, const double *bcs_hb, const double *bcs_ht) {
    POISSON_GUARD_BEGIN
    if (!P || !p || !bcs_hb || !bcs_ht) throw std::invalid_argument("null argument");
    hipLaunchKernelGGL(k_set_wall_planes, dim3((unsigned)(((long long)P->nx * P->fx_nz + 255) / 256)), dim3(256), 0, tlab_current_stream(), p,
                       bcs_hb, bcs_ht, P->nx, P->ny, P->fx_nz);
    POISSON_GUARD_END
}
// dir = +1: real (nx,ny,kmax) -> complex (nx/2+1,ny,kmax)  [OPR_Fourier_X_Forward]; dir = -1: the inverse [OPR_Fourier_X_Backward]
// dir = -2: the inverse by the library's own kernel (k_fftx_c2r; the solver itself uses rocFFT's, which already runs at the copy rate) -- tests
int tlab_poisson_fft_x(tlab_poisson_plan_t P, int dir, double *in, double *out) {
    POISSON_GUARD_BEGIN
    if (!P || !in || !out || in == out) throw std::invalid_argument("tlab_poisson_fft_x: bad arguments");
    if (dir > 0) P->x_forward(in, out, tlab_current_stream());
    else if (dir == -2) {
        if (!P->fx_own) throw std::invalid_argument("tlab_poisson_fft_x: no own transform for this length");
        P->fx_own->exec_inverse(in, out, tlab_current_stream());
    } else P->fx_c2r.exec(in, out, tlab_current_stream());
    POISSON_GUARD_END
}
// The x-transforms with the complex side in the pack layout of tlab_pencil_repack_blocks (the repack pass folded into the transform):
// dir = +1: real in (nx,ny,kmax) -> pack buffer out; dir = -1: pack buffer in -> real out.  Own kernels only.
static void packed_check(tlab_poisson_plan_t P, const void *a, const void *b, int nblocks, const int *start, const long long *base, const char *who) {
    if (!P || !a || !b || !start || !base || nblocks < 1 || nblocks > 16 || start[0] != 0) throw std::invalid_argument(std::string(who) + ": bad arguments");
    for (int p = 0; p + 1 < nblocks; ++p)
        if (start[p + 1] < start[p] || start[p + 1] > P->fx_nxh) throw std::invalid_argument(std::string(who) + ": block starts must increase within [0, nx/2+1]");
}
#define PACKED_NEEDS_OWN(P, who) if ((P) && !(P)->fx_own) { tlab_set_error(who ": no own transform for this length"); return TLAB_EUNSUPPORTED; }
int tlab_poisson_fft_x_packed(tlab_poisson_plan_t P, int dir, double *in, double *out, int nblocks, const int *start, const long long *base) {
    PACKED_NEEDS_OWN(P, "tlab_poisson_fft_x_packed")
    POISSON_GUARD_BEGIN
    packed_check(P, in, out, nblocks, start, base, "tlab_poisson_fft_x_packed");
    if (in == out) throw std::invalid_argument("tlab_poisson_fft_x_packed: out of place only");
    const auto &m = P->kx_map(nblocks, start, base);
    if (dir > 0) P->fx_own->exec(in, out, tlab_current_stream(), m.off, m.w);
    else P->fx_own->exec_inverse(in, out, tlab_current_stream(), m.off, m.w);
    POISSON_GUARD_END
}
// ... and the inverse of dp^/dy finishing the v equation (h = h - dp/dy, wall planes zeroed, q += dte h, h *= kco when scale) in its epilogue
int tlab_poisson_fft_x_packed_final(tlab_poisson_plan_t P, double *in, double *q, double *h, double dte, double kco, int scale, int nblocks,
                                    const int *start, const long long *base) {
    PACKED_NEEDS_OWN(P, "tlab_poisson_fft_x_packed_final")
    POISSON_GUARD_BEGIN
    packed_check(P, in, q, nblocks, start, base, "tlab_poisson_fft_x_packed_final");
    if (!h) throw std::invalid_argument("tlab_poisson_fft_x_packed_final: bad arguments");
    const auto &m = P->kx_map(nblocks, start, base);
    P->fx_own->exec_inverse_final(in, q, h, dte, kco, scale, (int)P->ny, tlab_current_stream(), m.off, m.w);
    POISSON_GUARD_END
}
// complex (nlines, nz_total) lines-fastest (the K-transposed layout; nlines = (nx/2+1)*ny/nproc_k), out of place
int tlab_poisson_fft_z(tlab_poisson_plan_t P, int dir, double *in, double *out) {
    POISSON_GUARD_BEGIN
    if (!P || !in || !out || (in == out && !P->fz_inplace && !P->fz_own)) throw std::invalid_argument("tlab_poisson_fft_z: bad arguments");
    if (P->nzt <= 1) throw std::invalid_argument("tlab_poisson_fft_z: no z direction");
    if (P->fz_own) {
        P->fz_own->exec(dir, in, out, tlab_current_stream());
    } else if (P->fz_inplace) {      // in place on out (in == out allowed and cheapest)
        if (in != out)
            hipc(hipMemcpyAsync(out, in, (size_t)2 * P->nxh * P->ny * P->nz * sizeof(double), hipMemcpyDeviceToDevice, tlab_current_stream()), "copy");
        if (dir > 0) P->fz_f.exec(out, out, tlab_current_stream());
        else P->fz_b.exec(out, out, tlab_current_stream());
    } else if (dir > 0) {
        P->fz_f.exec(in, out, tlab_current_stream());
    } else {
        P->fz_b.exec(in, out, tlab_current_stream());
    }
    POISSON_GUARD_END
}
// per-mode ODE solves on the local (kx, kz) modes: f_hat -> p_hat (may alias f_hat), dp_hat
int tlab_poisson_ode(tlab_poisson_plan_t P, double *f_hat, double *p_hat, double *dp_hat) {
    POISSON_GUARD_BEGIN
    if (!P || !f_hat || !p_hat || !dp_hat || dp_hat == f_hat || dp_hat == p_hat) throw std::invalid_argument("tlab_poisson_ode: bad arguments");
    poisson_ode_stage(P, f_hat, p_hat, dp_hat, tlab_current_stream());
    POISSON_GUARD_END
}

// OPR_Helmholtz(nx, ny, nz, ibc, alpha, a, tmp1, tmp2, bcs_hb, bcs_ht): OPR_Helmholtz_FourierXZ_Direct (operators/opr_elliptic.f90:562-628) on a
// direct plan, OPR_Helmholtz_FourierXZ_Factorize (:466-557) on a factorized one
int tlab_opr_helmholtz(tlab_poisson_plan_t P, int nx, int ny, int nz, int ibc, double alpha, double *a, double *tmp1, double *tmp2,
                       const double *bcs_hb, const double *bcs_ht) {
    POISSON_GUARD_BEGIN
    if (!P || !a || !tmp1 || !tmp2 || !bcs_hb || !bcs_ht) throw std::invalid_argument("tlab_opr_helmholtz: null argument");
    if (nx != P->nx || ny != P->ny || nz != P->nz) throw std::invalid_argument("tlab_opr_helmholtz: sizes do not match the plan");
    if (ibc < TLAB_BCS_DD || ibc > TLAB_BCS_NN) throw std::invalid_argument("tlab_opr_helmholtz: bad ibc");
    if (a == tmp1 || a == tmp2 || tmp1 == tmp2) throw std::invalid_argument("arrays must be distinct");
    tlab_poisson_plan *H = nullptr;
    if (!P->direct) {
        if (ibc != TLAB_BCS_NN && ibc != TLAB_BCS_DD) {
            tlab_set_error("OPR_Helmholtz (factorized): BCS_NN and BCS_DD only, like the reference (opr_elliptic.f90:524-532)");
            return TLAB_EUNSUPPORTED;
        }
        if (!P->g3[0] || P->helmholtz) throw std::invalid_argument("tlab_opr_helmholtz: needs a single-device plan of tlab_poisson_plan_create");
        for (size_t i = 0; i < P->helm.size(); ++i)
            if (P->helm[i].first == alpha) {      // most recently used last
                std::rotate(P->helm.begin() + i, P->helm.begin() + i + 1, P->helm.end());
                H = P->helm.back().second.get();
                break;
            }
        if (!H) {
            tlab_poisson_plan_t h = nullptr;
            const int rc = poisson_plan_create_impl(&h, P->g3[0], P->g3[1], P->g3[2], nx, ny, nz, nz, 0, 1, 0, 0, 0, nullptr, true, alpha);
            if (rc != TLAB_OK) return rc;
            if (P->helm.size() >= 4) P->helm.erase(P->helm.begin());
            P->helm.emplace_back(alpha, std::unique_ptr<tlab_poisson_plan>(h));
            H = h;
        }
    }
    hipStream_t st = tlab_current_stream();
    hipLaunchKernelGGL(k_set_wall_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, a, bcs_hb, bcs_ht, nx, ny, nz);
    if (P->use_2d) {
        P->f2_fwd.exec(a, tmp1, st);
    } else if (nz > 1) {
        P->x_forward(a, tmp2, st);
        if (P->fz_own) P->fz_own->exec(1, tmp2, tmp1, st);
        else P->fz_f.exec(tmp2, tmp1, st);
    } else {
        P->x_forward(a, tmp1, st);
    }
    if (P->direct) poisson_direct_stage(P, ibc, tmp1, tmp1, st, true, alpha);
    else if (ibc == TLAB_BCS_DD) poisson_dd_stage(H, tmp1, tmp1, tmp2, st);      // u over f^; v = u' + sqrt(lambda - alpha) u (not returned) in tmp2
    else poisson_ode_stage(H, tmp1, tmp1, tmp2, st);
    if (P->use_2d) {
        P->f2_bwd.exec(tmp1, a, st);
    } else if (nz > 1) {
        if (P->fz_own) P->fz_own->exec(-1, tmp1, P->cwork.p, st);
        else P->fz_b.exec(tmp1, P->cwork.p, st);
        P->x_backward_p(P->cwork.p, a, st);
    } else {
        P->x_backward_p(tmp1, a, st);
    }
    POISSON_GUARD_END
}

// direct plans: FDM_Int2_Solve of the local modes, f_hat -> p_hat (may alias)
int tlab_poisson_direct_ode(tlab_poisson_plan_t P, int ibc, double *f_hat, double *p_hat) {
    POISSON_GUARD_BEGIN
    if (!P || !f_hat || !p_hat) throw std::invalid_argument("tlab_poisson_direct_ode: bad arguments");
    if (!P->direct) throw std::invalid_argument("tlab_poisson_direct_ode: not a direct plan");
    if (ibc < TLAB_BCS_DD || ibc > TLAB_BCS_NN) throw std::invalid_argument("tlab_poisson_direct_ode: bad ibc");
    poisson_direct_stage(P, ibc, f_hat, p_hat, tlab_current_stream());
    POISSON_GUARD_END
}


// include/tlab_amd.h: debug aid -- one FDM_Int1_Solve of the 3- / 7-diagonal path on the device (k_int1g), two lines per mode, for tests
int tlab_debug_int1_solve(tlab_fdm_plan_t gy, int ibc, int variant, int nm, const double *lam, const double *f, const double *bv, double *res, double *du) {
    try {
        if (!gy || !lam || !f || !bv || !res || !du || nm < 1 || (ibc != 1 && ibc != 2)) throw std::invalid_argument("tlab_debug_int1_solve: bad arguments");
        if (!tlab_device_ready()) throw std::runtime_error("tlab_init has not been called");
        const int n = gy->t.n;
        Int1Gen G;
        int1_generic_build(gy->t.der1, ibc, lam, nm, 1.0, G);
        DBuf fac, rb, rt, R, df, dbv, dres, ddu, scr;
        fac.upload(G.fac); rb.upload(G.rb); rt.upload(G.rt); R.upload(G.R);
        df.upload(std::vector<double>(f, f + (size_t)2 * n * nm));
        dbv.upload(std::vector<double>(bv, bv + (size_t)2 * nm));
        dres.alloc((size_t)2 * n * nm); ddu.alloc((size_t)2 * nm); scr.alloc((size_t)5 * n * nm);
        Int1Args a{};
        a.T.n = n; a.nm = nm; a.fscale = 1.0; a.fsrc = df.p; a.nlf = 2; a.bv_ptr = dbv.p; a.dst = dres.p; a.du = ddu.p; a.scratch = scr.p;
        a.g_fac = fac.p; a.g_rb = rb.p; a.g_rt = rt.p; a.g_R = R.p; a.g_ndi = G.ndi; a.g_nri = G.nri;
        hipStream_t st = tlab_current_stream();
        if (variant == 1) {                 // FS_UNIT variants of build_homogeneous / build_singular_homogeneous
            a.bv_ptr = nullptr; a.bv[0] = 0.0; a.bv[1] = 1.0; a.bv[2] = 0.0;
            if (ibc == 1) { a.unit_row = n - 1; launch_int1<1, 2, FS_UNIT>(a, st); }
            else { a.unit_row = 0; launch_int1<2, 2, FS_UNIT>(a, st); }
        } else if (variant == 2) {          // three lines, two stored (build_homogeneous, u-solve); ibc = 2
            DBuf d3, u3;
            d3.alloc((size_t)3 * n * nm); u3.alloc((size_t)3 * nm);
            a.bv_ptr = nullptr; a.bv[0] = 0.0; a.bv[1] = 0.0; a.bv[2] = 1.0;
            a.dst = d3.p; a.du = u3.p;
            launch_int1<2, 3, FS_LINEAR>(a, st);
            hipc(hipStreamSynchronize(st), "sync");
            hipc(hipMemcpy(dres.p, d3.p + (size_t)n * nm, (size_t)2 * n * nm * sizeof(double), hipMemcpyDeviceToDevice), "copy");      // lines 1, 2
            hipc(hipMemcpy(ddu.p, u3.p + (size_t)nm, (size_t)2 * nm * sizeof(double), hipMemcpyDeviceToDevice), "copy");
        } else if (ibc == 1) launch_int1<1, 2, FS_LINEAR>(a, st);
        else launch_int1<2, 2, FS_LINEAR>(a, st);
        hipc(hipStreamSynchronize(st), "sync");
        hipc(hipMemcpy(res, dres.p, (size_t)2 * n * nm * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        hipc(hipMemcpy(du, ddu.p, (size_t)2 * nm * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        return TLAB_OK;
    } catch (const std::invalid_argument &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EHIP;
    }
}

}  // extern "C"

// ---- hooks of the RHS driver (rhs.cpp) ----
// The v equation needs dp/dy only as the operand of its final update: when the plan has the own x-transform and takes the 1-D transform route, the
// driver arms the NEXT tlab_opr_poisson call with (q, h, dte, kco, scale) and that call's last inverse transform finishes v instead of storing dp/dy.
bool tlab_internal_poisson_can_v_final(tlab_poisson_plan_t P) {
    static const bool on = [] { const char *e = getenv("TLAB_V_FINAL"); return !(e && atoi(e) == 0); }();
    return on && P && P->fx_own && !P->use_2d && !P->direct && !P->helmholtz && P->nproc == 1;
}
void tlab_internal_poisson_arm_v_final(tlab_poisson_plan_t P, double *q, double *h, double dte, double kco, int scale) {
    P->vfinal.q = q; P->vfinal.h = h; P->vfinal.dte = dte; P->vfinal.kco = kco; P->vfinal.scale = scale; P->vfinal.armed = true;
}
