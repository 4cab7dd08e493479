// k_pentatile : SpaceOrder1 = CompactJacobian6Penta (fdm_com1_jacobian.f90:136-192) along y / z (and along x behind the transposes of OPR_Partial_X)
// on register tiles.
//
// The reference solves A u' = B u with the seven-diagonal antisymmetric right-hand side (MatMul_7d_antisym, fdm_matmul.f90) and the pentadiagonal
// system by its stored LU factors: PENTADSS2 (linear5.f90:207-244) = a descending two-term recurrence, then an ascending one with a division per row;
// periodic lines add the rank-two correction of PENTADPSS (linear5.f90:352-411).  k_penta1 repeats that one line per thread (three passes through
// memory, ~1 TB/s).  Here a workgroup owns 32 memory-contiguous lines x all rows, two chunks of 32 rows per wave (lanes 0-31 / 32-63: the layout of
// k_htile), the line in registers:
//   * both recurrences are linear with coefficients that belong to the plan, so a chunk runs them with zero inflow and adds the response of its rows
//     to the two values that enter it (two response vectors per sweep, computed on the host in long double);
//   * what enters a chunk is a 2-vector combination of the chunk-end values of the chunks before it along the sweep, with 2 x 2 blocks that are
//     products of the chunks' transfer matrices (host, long double; they fall off like the recurrence, 1e-10 per 32 rows): one pass through LDS;
//   * one read and one write of the field.
// Same additions and multiplications per row as the reference (the division by the pivot included); the superposition replaces the serial carry, so
// results agree to rounding, not to the bit (asserted against the oracle at 1e-12 like every operator).
#include <hip/hip_runtime.h>

#include <cmath>
#include <stdexcept>
#include <vector>

#include "kernels.hpp"
#include "profile.hpp"

namespace tlab {

// tables of one (plan, Neumann variant, chunk count): everything per row is [9][n] = D, E (descending), A, B, C (ascending), the two response vectors of
// each sweep; the block matrices are [2][C][C][4]; periodic lines: the two vectors of the rank-two correction [2][n] and its 8 constants
void pentatile_build(int n, int C, bool periodic, int ibc, const double *lu, std::vector<double> &rows, std::vector<double> &blocks, std::vector<double> &smw) {
    const int M = n / C;
    if (C < 2 || C > 16 || M * C != n || M < 8) throw std::invalid_argument("pentatile_build: bad chunking");
    const int ip = periodic ? 0 : ibc * 5;
    const bool nb = !periodic && (ibc == 1 || ibc == 3), nt = !periodic && (ibc == 2 || ibc == 3);
    const int jmin = nb ? 1 : 0, jmax = nt ? n - 2 : n - 1;
    rows.assign((size_t)9 * n, 0.0);
    double *D = rows.data(), *E = D + n, *A = E + n, *B = A + n, *Cc = B + n, *PD1 = Cc + n, *PD2 = PD1 + n, *PA1 = PD2 + n, *PA2 = PA1 + n;
    for (int j = 0; j < n; ++j) {
        A[j] = lu[(size_t)n * (ip + 0) + j]; B[j] = lu[(size_t)n * (ip + 1) + j]; Cc[j] = lu[(size_t)n * (ip + 2) + j];
        D[j] = lu[(size_t)n * (ip + 3) + j]; E[j] = lu[(size_t)n * (ip + 4) + j];
    }
    // rows outside the system (Neumann walls) stay zero; the first rows of the ascending sweep and the last ones of the descending sweep have no
    // predecessors (PENTADSS2: G(m-1) = G(m-1) - G(m) D, G(1) = G(1) / C, G(2) = (G(2) - G(1) B) / C)
    for (int j = 0; j < n; ++j)
        if (j < jmin || j > jmax) { A[j] = B[j] = D[j] = E[j] = 0.0; Cc[j] = 1.0; }
    D[jmax] = 0.0; E[jmax] = 0.0;
    if (jmax >= 1) E[jmax - 1] = 0.0;
    A[jmin] = 0.0; B[jmin] = 0.0;
    if (jmin + 1 < n) A[jmin + 1] = 0.0;
    // response vectors (long double) and transfer matrices
    std::vector<long double> Td((size_t)C * 4), Ta((size_t)C * 4);
    for (int c = 0; c < C; ++c) {
        const int r0 = c * M;
        for (int v = 0; v < 2; ++v) {      // descending: unit inflow (g[r0+M], g[r0+M+1]) = (1,0), (0,1)
            long double g1 = v == 0 ? 1.0L : 0.0L, g2 = v == 0 ? 0.0L : 1.0L;
            for (int p = M - 1; p >= 0; --p) {
                const long double r = -g1 * (long double)D[r0 + p] - g2 * (long double)E[r0 + p];
                (v == 0 ? PD1 : PD2)[r0 + p] = (double)r;
                if (p < 2) Td[(size_t)c * 4 + p * 2 + v] = r;      // T = [[PD1[0], PD2[0]], [PD1[1], PD2[1]]]
                g2 = g1; g1 = r;
            }
        }
        for (int v = 0; v < 2; ++v) {      // ascending: unit inflow (g[r0-1], g[r0-2])
            long double g1 = v == 0 ? 1.0L : 0.0L, g2 = v == 0 ? 0.0L : 1.0L;
            for (int p = 0; p < M; ++p) {
                const long double r = (-g1 * (long double)B[r0 + p] - g2 * (long double)A[r0 + p]) / (long double)Cc[r0 + p];
                (v == 0 ? PA1 : PA2)[r0 + p] = (double)r;
                if (p >= M - 2) Ta[(size_t)c * 4 + (M - 1 - p) * 2 + v] = r;      // U = [[PA1[M-1], PA2[M-1]], [PA1[M-2], PA2[M-2]]]
                g2 = g1; g1 = r;
            }
        }
    }
    // inflow of chunk c, descending: H_{c+1} = sum_{k >= c+1} (T_{c+1} ... T_{k-1}) h_k ; ascending: Tl_{c-1} = sum_{k <= c-1} (U_{c-1} ... U_{k+1}) t_k
    blocks.assign((size_t)2 * C * C * 4, 0.0);
    auto mul = [](const long double *a, const long double *b, long double *o) {
        o[0] = a[0] * b[0] + a[1] * b[2]; o[1] = a[0] * b[1] + a[1] * b[3];
        o[2] = a[2] * b[0] + a[3] * b[2]; o[3] = a[2] * b[1] + a[3] * b[3];
    };
    for (int c = 0; c < C; ++c) {
        long double P[4] = {1, 0, 0, 1};
        for (int k = c + 1; k < C; ++k) {           // descending: block (c, k) multiplies h_k in the inflow of chunk c
            for (int q = 0; q < 4; ++q) blocks[((size_t)(0 * C + c) * C + k) * 4 + q] = (double)P[q];
            long double N[4];
            mul(P, &Td[(size_t)k * 4], N);
            for (int q = 0; q < 4; ++q) P[q] = N[q];
        }
        long double Q[4] = {1, 0, 0, 1};
        for (int k = c - 1; k >= 0; --k) {          // ascending: block (c, k) multiplies t_k in the inflow of chunk c
            for (int q = 0; q < 4; ++q) blocks[((size_t)(1 * C + c) * C + k) * 4 + q] = (double)Q[q];
            long double N[4];
            mul(Q, &Ta[(size_t)k * 4], N);
            for (int q = 0; q < 4; ++q) Q[q] = N[q];
        }
    }
    smw.assign((size_t)2 * n + 8, 0.0);
    if (periodic) {      // PENTADPSS (linear5.f90:352-411) with the stored vectors f, g: the same expressions as k_penta1, evaluated once
        const double *a = lu, *b = lu + n, *d = lu + 3 * (size_t)n, *e = lu + 4 * (size_t)n, *Fv = lu + 5 * (size_t)n, *Gv = lu + 6 * (size_t)n;
        for (int j = 0; j < n; ++j) { smw[j] = Fv[j]; smw[(size_t)n + j] = Gv[j]; }
        const double m1 = e[n - 1] * Fv[0] + a[0] * Fv[n - 2] + b[0] * Fv[n - 1] + 1.0;
        const double m2 = e[n - 1] * Gv[0] + a[0] * Gv[n - 2] + b[0] * Gv[n - 1];
        const double m3 = d[n - 1] * Fv[0] + e[n - 1] * Fv[1] + a[0] * Fv[n - 1];
        const double m4 = d[n - 1] * Gv[0] + e[n - 1] * Gv[1] + a[0] * Gv[n - 1] + 1.0;
        const double di = 1 / (m1 * m4 - m2 * m3);
        double *k = smw.data() + 2 * (size_t)n;
        k[0] = di * (m4 * e[n - 1] - m2 * d[n - 1]); k[1] = di * (m4 * b[0] - m2 * a[0]); k[2] = di * m4 * a[0]; k[3] = di * m2 * e[n - 1];
        k[4] = di * (m1 * d[n - 1] - m3 * e[n - 1]); k[5] = di * (m1 * a[0] - m3 * b[0]); k[6] = di * m3 * a[0]; k[7] = di * m1 * e[n - 1];
    }
}

namespace {

constexpr int PM = 32, PL = 32;      // rows per chunk, lines per tile

// LDS: rows [9][n], blocks [2][C][C][4], periodic vectors [2][n], chunk-end values [C][2][PL] of the current sweep, four corner values [4][PL]
// XD (lines along x, contiguous in memory): PL = 16 lines per workgroup are brought into an LDS tile [PL][n + 1] by coalesced row loads, the threads
// pick their rows out of it (a thread owns a 32-row chunk of ONE line, so its rows are contiguous in the tile and the lines a wave holds are n + 1
// doubles apart), and the result leaves through the same tile -- one read and one write of the field where OPR_Partial_X went through two
// transposes around the y-direction kernel (the reference: TLab_Transpose + solve + TLab_Transpose, opr_partial.f90:185-195)
template <int PL, bool XD>
__global__ void __launch_bounds__(512, 1) k_pentatile(PentaTileArgs a) {
    extern __shared__ double s_pt[];
    const int n = a.g.n, C = n / PM;
    // XD: the tables stay in global memory (read-only, the same for every workgroup: L2 / L1 hits, and the lanes of a wave ask for four addresses only) so
    // that the LDS holds the tile and little else: two to four workgroups per CU instead of one
    const double *s_rows = XD ? a.rows : s_pt;
    const double *s_blk = XD ? a.blocks : s_pt + 9 * n;
    const double *s_fg = XD ? a.smw : s_pt + 9 * n + 2 * C * C * 4;
    double *s_end = XD ? s_pt : s_pt + 9 * n + 2 * C * C * 4 + (a.periodic ? 2 * n : 0), *s_cor = s_end + C * 2 * PL;
    double *s_tile = s_cor + 4 * PL;      // XD: [PL][n + 1]
    const int l32 = threadIdx.x & (PL - 1), c = threadIdx.x / PL;
    const long long rs = XD ? 1 : a.g.row_stride;
    if constexpr (!XD) {
        double *w = s_pt;
        for (int i = threadIdx.x; i < 9 * n; i += blockDim.x) w[i] = a.rows[i];
        for (int i = threadIdx.x; i < 2 * C * C * 4; i += blockDim.x) w[9 * n + i] = a.blocks[i];
        if (a.periodic)
            for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) w[9 * n + 2 * C * C * 4 + i] = a.smw[i];
    }
    const int tiles_inner = (a.g.lines_inner + PL - 1) / PL;
    const long long outer = blockIdx.x / tiles_inner;
    const int l0 = (int)(blockIdx.x % tiles_inner) * PL;
    const bool valid = (l0 + l32) < a.g.lines_inner;
    long long base = outer * a.g.outer_stride + l0 + (valid ? l32 : 0);
    const int row0 = c * PM;
    const bool per = a.periodic != 0;
    const double *src = a.in0;
    if constexpr (XD) {      // the tile's lines are lines [blockIdx.x PL, + PL) of nlines; line L starts at L n
        const long long line0 = (long long)blockIdx.x * PL;
        const int nl = (int)((a.g.nlines - line0 < PL) ? a.g.nlines - line0 : PL);
        const double2 *g2 = reinterpret_cast<const double2 *>(a.in0 + line0 * n);
        const int tot = nl * (n / 2), bd = (int)blockDim.x;      // 16 B per lane along the lines, four requests in flight per lane
        for (int i0 = threadIdx.x; i0 < tot; i0 += 4 * bd) {
            double2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + q * bd; v[q] = i < tot ? g2[i] : make_double2(0.0, 0.0); }      // (the lines of a tile are contiguous: g2[ln n/2 + r2] = g2[i])
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * bd;
                if (i < tot) { const int ln = i / (n / 2), r2 = i - ln * (n / 2); s_tile[ln * (n + 1) + 2 * r2] = v[q].x; s_tile[ln * (n + 1) + 2 * r2 + 1] = v[q].y; }
            }
        }
        __syncthreads();
        src = s_tile;
        base = (long long)(l32 < nl ? l32 : 0) * (n + 1);
    }
    const bool valid_x = XD ? ((long long)blockIdx.x * PL + l32 < a.g.nlines) : valid;
    // ---- operand rows + three rows on either side ----
    double e[PM + 6];
#pragma unroll
    for (int p = 0; p < PM; ++p) e[p + 3] = src[base + (long long)(row0 + p) * rs];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int rl = row0 - 3 + k, rr = row0 + PM + k;
        const bool okl = per || rl >= 0, okr = per || rr < n;
        if (rl < 0) rl += n;
        if (rr >= n) rr -= n;
        e[k] = okl ? src[base + (long long)rl * rs] : 0.0;
        e[PM + 3 + k] = okr ? src[base + (long long)rr * rs] : 0.0;
    }
    // ---- MatMul_7d_antisym ----
    const double r6 = a.r6, r7 = a.r7;
    double f[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) f[p] = e[p + 4] - e[p + 2] + r6 * (e[p + 5] - e[p + 1]) + r7 * (e[p + 6] - e[p]);
    if (!per) {
        const int ibc = a.ibc;
        const bool nb = (ibc == 1 || ibc == 3), nt = (ibc == 2 || ibc == 3);
#define U_(i) e[(i) + 2]                       /* U(i), i = 1 .. 7, in the first chunk */
#define V_(k) e[PM + 2 - (k)]                  /* U(n - k), k = 0 .. 6, in the last chunk */
#define RI(i, k) a.rhs[((i)-1) + (size_t)n * ((k)-1)]
#define RB(j, cc) a.rb[((j)-1) + 4 * (cc)]
#define RT(r, cc) a.rt[(r) + 5 * ((cc)-1)]
        if (c == 0) {
            if (nb) {
                const double f1 = 0.0;
                f[0] = 0.0;
                f[1] = f1 * RB(2, 3) + U_(2) * RB(2, 4) + U_(3) * RB(2, 5) + U_(4) * RB(2, 6) + U_(5) * RB(2, 7);
                f[2] = f1 * RB(3, 2) + U_(2) * RB(3, 3) + U_(3) * RB(3, 4) + U_(4) * RB(3, 5) + U_(5) * RB(3, 6) + U_(6) * RB(3, 7);
                f[3] = f1 * RB(4, 1) + U_(2) * RB(4, 2) + U_(3) * RB(4, 3) + U_(4) * RB(4, 4) + U_(5) * RB(4, 5) + U_(6) * RB(4, 6) + U_(7) * RB(4, 7);
            } else {
                f[0] = U_(1) * RI(1, 4) + U_(2) * RI(1, 5) + U_(3) * RI(1, 6) + U_(4) * RI(1, 7) + U_(5) * RI(1, 1);
                f[1] = U_(1) * RI(2, 3) + U_(2) * RI(2, 4) + U_(3) * RI(2, 5) + U_(4) * RI(2, 6) + U_(5) * RI(2, 7);
                f[2] = U_(1) * RI(3, 2) + U_(2) * RI(3, 3) + U_(3) * RI(3, 4) + U_(4) * RI(3, 5) + U_(5) * RI(3, 6) + U_(6) * RI(3, 7);
                f[3] = U_(1) * RI(4, 1) + U_(2) * RI(4, 2) + U_(3) * RI(4, 3) + U_(4) * RI(4, 4) + U_(5) * RI(4, 5) + U_(6) * RI(4, 6) + U_(7) * RI(4, 7);
            }
        }
        if (c == C - 1) {
            if (nt) {
                const double fn = 0.0;
                f[PM - 4] = V_(6) * RT(1, 1) + V_(5) * RT(1, 2) + V_(4) * RT(1, 3) + V_(3) * RT(1, 4) + V_(2) * RT(1, 5) + V_(1) * RT(1, 6) + fn * RT(1, 7);
                f[PM - 3] = V_(5) * RT(2, 1) + V_(4) * RT(2, 2) + V_(3) * RT(2, 3) + V_(2) * RT(2, 4) + V_(1) * RT(2, 5) + fn * RT(2, 6);
                f[PM - 2] = V_(4) * RT(3, 1) + V_(3) * RT(3, 2) + V_(2) * RT(3, 3) + V_(1) * RT(3, 4) + fn * RT(3, 5);
                f[PM - 1] = 0.0;
            } else {
                f[PM - 4] = V_(6) * RI(n - 3, 1) + V_(5) * RI(n - 3, 2) + V_(4) * RI(n - 3, 3) + V_(3) * RI(n - 3, 4) + V_(2) * RI(n - 3, 5) + V_(1) * RI(n - 3, 6) + V_(0) * RI(n - 3, 7);
                f[PM - 3] = V_(5) * RI(n - 2, 1) + V_(4) * RI(n - 2, 2) + V_(3) * RI(n - 2, 3) + V_(2) * RI(n - 2, 4) + V_(1) * RI(n - 2, 5) + V_(0) * RI(n - 2, 6);
                f[PM - 2] = V_(4) * RI(n - 1, 1) + V_(3) * RI(n - 1, 2) + V_(2) * RI(n - 1, 3) + V_(1) * RI(n - 1, 4) + V_(0) * RI(n - 1, 5);
                f[PM - 1] = V_(4) * RI(n, 7) + V_(3) * RI(n, 1) + V_(2) * RI(n, 2) + V_(1) * RI(n, 3) + V_(0) * RI(n, 4);
            }
        }
#undef U_
#undef V_
#undef RI
#undef RB
#undef RT
    }
    __syncthreads();      // tables are in LDS
    const double *D = s_rows + row0, *E = D + n, *A = E + n, *B = A + n, *Cp = B + n, *PD1 = Cp + n, *PD2 = PD1 + n, *PA1 = PD2 + n, *PA2 = PA1 + n;
    // ---- descending sweep with zero inflow, chunk heads to LDS, inflow from the chunks above, correction ----
    {
        double g1 = 0.0, g2 = 0.0;
#pragma unroll
        for (int p = PM - 1; p >= 0; --p) {
            const double v = f[p] - g1 * D[p] - g2 * E[p];
            f[p] = v;
            g2 = g1; g1 = v;
        }
        s_end[(c * 2 + 0) * PL + l32] = f[0];
        s_end[(c * 2 + 1) * PL + l32] = f[1];
        __syncthreads();
        double x1 = 0.0, x2 = 0.0;
        for (int k = c + 1; k < C; ++k) {
            const double *m = s_blk + ((size_t)(0 * C + c) * C + k) * 4;
            const double h0 = s_end[(k * 2 + 0) * PL + l32], h1 = s_end[(k * 2 + 1) * PL + l32];
            x1 += m[0] * h0 + m[1] * h1;
            x2 += m[2] * h0 + m[3] * h1;
        }
#pragma unroll
        for (int p = 0; p < PM; ++p) f[p] = f[p] + PD1[p] * x1 + PD2[p] * x2;
        __syncthreads();      // s_end is reused
    }
    // ---- ascending sweep ----
    {
        double g1 = 0.0, g2 = 0.0;
#pragma unroll
        for (int p = 0; p < PM; ++p) {
            const double v = (f[p] - g1 * B[p] - g2 * A[p]) / Cp[p];
            f[p] = v;
            g2 = g1; g1 = v;
        }
        s_end[(c * 2 + 0) * PL + l32] = f[PM - 1];
        s_end[(c * 2 + 1) * PL + l32] = f[PM - 2];
        __syncthreads();
        double x1 = 0.0, x2 = 0.0;
        for (int k = c - 1; k >= 0; --k) {
            const double *m = s_blk + ((size_t)(1 * C + c) * C + k) * 4;
            const double t0 = s_end[(k * 2 + 0) * PL + l32], t1 = s_end[(k * 2 + 1) * PL + l32];
            x1 += m[0] * t0 + m[1] * t1;
            x2 += m[2] * t0 + m[3] * t1;
        }
#pragma unroll
        for (int p = 0; p < PM; ++p) f[p] = f[p] + PA1[p] * x1 + PA2[p] * x2;
    }
    // ---- periodic lines: the rank-two correction of PENTADPSS with F(1), F(2), F(n-1), F(n) of the open solve ----
    if (per) {
        if (c == 0) { s_cor[0 * PL + l32] = f[0]; s_cor[1 * PL + l32] = f[1]; }
        if (c == C - 1) { s_cor[2 * PL + l32] = f[PM - 2]; s_cor[3 * PL + l32] = f[PM - 1]; }
        __syncthreads();
        const double F1 = s_cor[0 * PL + l32], F2 = s_cor[1 * PL + l32], Fn1 = s_cor[2 * PL + l32], Fn = s_cor[3 * PL + l32];
        const double *k = a.smw + 2 * (size_t)n;
        const double dummy1 = k[0] * F1 + k[1] * Fn + k[2] * Fn1 - k[3] * F2;
        const double dummy2 = k[4] * F1 + k[5] * Fn - k[6] * Fn1 + k[7] * F2;
        const double *Fv = s_fg + row0, *Gv = s_fg + n + row0;
#pragma unroll
        for (int p = 0; p < PM; ++p) f[p] = f[p] - dummy1 * Fv[p] - dummy2 * Gv[p];
    }
    if constexpr (XD) {
        // (every thread read its operand rows from the tile before the first barrier after the right-hand side: the tile is free)
        if (valid_x) {
#pragma unroll
            for (int p = 0; p < PM; ++p) s_tile[base + row0 + p] = f[p];
        }
        __syncthreads();
        const long long line0 = (long long)blockIdx.x * PL;
        const int nl = (int)((a.g.nlines - line0 < PL) ? a.g.nlines - line0 : PL);
        double2 *o2 = reinterpret_cast<double2 *>(a.out0 + line0 * n);
        for (int i = threadIdx.x; i < nl * (n / 2); i += blockDim.x) {
            const int ln = i / (n / 2), r2 = i - ln * (n / 2);
            o2[i] = make_double2(s_tile[ln * (n + 1) + 2 * r2], s_tile[ln * (n + 1) + 2 * r2 + 1]);
        }
    } else if (valid) {
#pragma unroll
        for (int p = 0; p < PM; ++p) a.out0[base + (long long)(row0 + p) * rs] = f[p];
    }
}

}  // namespace

bool pentatile_ok(const LineGeom &g) {
    static const bool off = [] { const char *e = getenv("TLAB_PENTA_TILE"); return e && atoi(e) == 0; }();
    const int C = g.n / PM;
    return !off && g.n % PM == 0 && C >= 2 && C <= 16 && g.row_stride > 1;
}

hipError_t launch_pentatile(const PentaTileArgs &a, hipStream_t st) {
    const int n = a.g.n, C = n / PM;
    const long long tiles_inner = (a.g.lines_inner + PL - 1) / PL;
    const long long tiles = tiles_inner * (a.g.nlines / a.g.lines_inner);
    const size_t lds = ((size_t)9 * n + (size_t)2 * C * C * 4 + (a.periodic ? (size_t)2 * n : 0) + (size_t)C * 2 * PL + 4 * PL) * sizeof(double);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pentatile<PL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr = true;
    }
    ProfScope ps("k_pentatile", st, (double)a.g.nlines * n * 16.0);
    hipLaunchKernelGGL((k_pentatile<PL, false>), dim3((unsigned)tiles), dim3(PL * C), lds, st, a);
    return hipGetLastError();
}

// lines along x (row_stride == 1): 16 lines per workgroup through an LDS tile
constexpr int PLX = 16;
bool pentatile_x_ok(const LineGeom &g) {
    static const bool off = [] { const char *e = getenv("TLAB_PENTA_TILE_X"); return e && atoi(e) == 0; }();
    const int C = g.n / PM;
    const size_t lds = ((size_t)C * 2 * PLX + 4 * PLX + (size_t)PLX * (g.n + 1)) * sizeof(double);
    return !off && g.n % PM == 0 && g.n % 2 == 0 && C >= 2 && C <= 16 && g.row_stride == 1 && lds <= (size_t)160 * 1024;
}
template <int L>
static hipError_t launch_pentatile_x_l(const PentaTileArgs &a, hipStream_t st) {
    const int n = a.g.n, C = n / PM;
    const long long tiles = (a.g.nlines + L - 1) / L;
    const size_t lds = ((size_t)C * 2 * L + 4 * L + (size_t)L * (n + 1)) * sizeof(double);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pentatile<L, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr = true;
    }
    ProfScope ps("k_pentatile<x>", st, (double)a.g.nlines * n * 16.0);
    hipLaunchKernelGGL((k_pentatile<L, true>), dim3((unsigned)tiles), dim3(L * C), lds, st, a);
    return hipGetLastError();
}
hipError_t launch_pentatile_x(const PentaTileArgs &a, hipStream_t st) {
    static const int plx = [] { const char *e = getenv("TLAB_PENTA_PLX"); return e ? atoi(e) : PLX; }();      // lines per workgroup (experiments: 8, 16, 32)
    if (plx == 8) return launch_pentatile_x_l<8>(a, st);
    if (plx == 32 && a.g.n <= 256) return launch_pentatile_x_l<32>(a, st);
    return launch_pentatile_x_l<PLX>(a, st);
}

}  // namespace tlab
