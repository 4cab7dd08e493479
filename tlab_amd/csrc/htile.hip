// k_htile : y/z-direction derivative kernel on HALF-wave tiles (32 memory-contiguous lines x n rows per workgroup).
//
// Same chunked Thomas as k_rtile (kernels.hip), but a wave holds TWO chunks: lanes 0-31 own rows [2w*M, (2w+1)*M) of the 32
// lines, lanes 32-63 rows [(2w+1)*M, (2w+2)*M).  A tile is then 32 x n x 8 B = 128 KiB at n = 512 instead of 256 KiB, which
// leaves room in the register file for TWO line-sets at once: the first and the second derivative of OPR_Burgers /
// OPR_P2_P1 are computed from one load of the operand (24 B/point instead of the 48 B/point of the two-launch k_rtile path),
// including the Jacobian correction of stretched grids (the first derivative never leaves the registers; only the two edge
// values of each chunk go through LDS).  It also reaches n = 1024 (32 chunks of 32 rows, or 16 of 64).
// Price: 256-B instead of 512-B contiguous rows, and coefficient rows that differ between the two halves of a wave (vector
// loads that hit L1 instead of scalar loads).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_tables.hpp"
#include "kernels.hpp"
#include "profile.hpp"

namespace tlab {

typedef __attribute__((address_space(1))) char gchar;        // global address space kept through the integer round trip below (else: flat_load)
typedef __attribute__((address_space(1))) double gdouble;
// a wave-uniform pointer pinned into a scalar register pair and made opaque to the optimiser
__device__ __forceinline__ gchar *sgpr_ptr(const void *p) {
    unsigned long long v = (unsigned long long)p;
    asm("" : "+s"(v));
    return (gchar *)v;
}

__device__ __forceinline__ gchar *sgpr_ptr(gchar *p) {
    unsigned long long v = (unsigned long long)p;
    asm("" : "+s"(v));
    return (gchar *)v;
}

__device__ __forceinline__ unsigned vgpr_u32(unsigned v) {
    asm("" : "+v"(v));
    return v;
}

template <bool SYM>
__device__ __forceinline__ double h_stencil(const StencilDev &s, double um3, double um2, double um1, double u0, double up1, double up2,
                                            double up3) {
    if (SYM) return s.c0 * u0 + up1 + um1 + s.c2 * (up2 + um2) + s.c3 * (up3 + um3);
    return up1 - um1 + s.c2 * (up2 - um2);
}
__device__ __forceinline__ double h_dense6(const double (&c)[6], double a0, double a1, double a2, double a3, double a4, double a5) {
    return a0 * c[0] + a1 * c[1] + a2 * c[2] + a3 * c[3] + a4 * c[4] + a5 * c[5];
}

// local Thomas solve of one chunk + separator system through LDS.  f: right-hand side in, solution out.
// BAND: ginv holds the five central cyclic diagonals [C][5] of the inverse (SystemDev::band; C a power of two), the rest being negligible
template <int M, int L, bool BAND = false>
__device__ __forceinline__ void h_solve(double (&f)[M], const double *rowtab, const double *ginv, int n, int row0, int c, int C, int l32,
                                        double *s_yl, double *s_r) {
    const double *Lm = rowtab + row0, *Di = rowtab + n + row0, *Cm = rowtab + 2 * n + row0;
    const double *Vt = rowtab + 3 * n + row0, *Wt = rowtab + 4 * n + row0;
    double g = 0.0;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        g = f[p] + Lm[p] * g;
        f[p] = g;
    }
    double yn = 0.0;
#pragma unroll
    for (int p = M - 1; p >= 1; --p) {
        yn = f[p] * Di[p] + Cm[p] * yn;
        f[p] = yn;
    }
    s_yl[c * L + l32] = f[M - 1];
    __syncthreads();
    const int cm = (c + C - 1) % C, cp = (c + 1) % C;
    const double yLprev = s_yl[cm * L + l32];
    s_r[c * L + l32] = f[0] - Lm[0] * yLprev - Cm[0] * f[1];
    __syncthreads();
    double X = 0.0, Xr = 0.0;
    if constexpr (BAND) {
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            X += ginv[c * 5 + d] * s_r[((c + d - 2) & (C - 1)) * L + l32];
            Xr += ginv[cp * 5 + d] * s_r[((cp + d - 2) & (C - 1)) * L + l32];
        }
    } else {
        const double *g0 = ginv + c * C, *g1 = ginv + cp * C;
        for (int q = 0; q < C; ++q) {
            const double rq = s_r[q * L + l32];
            X += g0[q] * rq;
            Xr += g1[q] * rq;
        }
    }
    f[0] = X;
#pragma unroll
    for (int p = 1; p < M; ++p) f[p] = f[p] + Vt[p] * X + Wt[p] * Xr;
}

// Both systems of OPR_Burgers / OPR_P2_P1 at once (uniform grids: the second right-hand side does not need the first solution): two independent
// recurrences per lane hide the latency of the dependent fp64 multiply-adds (two waves per SIMD do not), and the two separator systems share their
// two barriers.  Same operations per system, in the same order, as h_solve.
template <int M, int L, bool BAND = false>
__device__ __forceinline__ void h_solve2(double (&f)[M], double (&h)[M], const double *tf, const double *th, const double *gif, const double *gih, int n,
                                         int row0, int c, int C, int l32, double *s_ylf, double *s_rf, double *s_ylh, double *s_rh) {
    // (n = distance between the five arrays of a table, row0 = first row of the chunk in them: (n, c M) for the full tables, (M, 0) for a chunk's own copy)
    const double *Lf = tf + row0, *Df = tf + n + row0, *Cf = tf + 2 * n + row0, *Vf = tf + 3 * n + row0, *Wf = tf + 4 * n + row0;
    const double *Lh = th + row0, *Dh = th + n + row0, *Ch = th + 2 * n + row0, *Vh = th + 3 * n + row0, *Wh = th + 4 * n + row0;
    double gf = 0.0, gh = 0.0;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        gf = f[p] + Lf[p] * gf;
        gh = h[p] + Lh[p] * gh;
        f[p] = gf;
        h[p] = gh;
    }
    double yf = 0.0, yh = 0.0;
#pragma unroll
    for (int p = M - 1; p >= 1; --p) {
        yf = f[p] * Df[p] + Cf[p] * yf;
        yh = h[p] * Dh[p] + Ch[p] * yh;
        f[p] = yf;
        h[p] = yh;
    }
    s_ylf[c * L + l32] = f[M - 1];
    s_ylh[c * L + l32] = h[M - 1];
    __syncthreads();
    const int cm = (c + C - 1) % C, cp = (c + 1) % C;
    s_rf[c * L + l32] = f[0] - Lf[0] * s_ylf[cm * L + l32] - Cf[0] * f[1];
    s_rh[c * L + l32] = h[0] - Lh[0] * s_ylh[cm * L + l32] - Ch[0] * h[1];
    __syncthreads();
    double Xf = 0.0, Xrf = 0.0, Xh = 0.0, Xrh = 0.0;
    if constexpr (BAND) {
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int q0 = ((c + d - 2) & (C - 1)) * L + l32, q1 = ((cp + d - 2) & (C - 1)) * L + l32;
            Xf += gif[c * 5 + d] * s_rf[q0];
            Xrf += gif[cp * 5 + d] * s_rf[q1];
            Xh += gih[c * 5 + d] * s_rh[q0];
            Xrh += gih[cp * 5 + d] * s_rh[q1];
        }
    } else {
        const double *g0f = gif + c * C, *g1f = gif + cp * C, *g0h = gih + c * C, *g1h = gih + cp * C;
        for (int q = 0; q < C; ++q) {
            const double rf = s_rf[q * L + l32], rh = s_rh[q * L + l32];
            Xf += g0f[q] * rf;
            Xrf += g1f[q] * rf;
            Xh += g0h[q] * rh;
            Xrh += g1h[q] * rh;
        }
    }
    f[0] = Xf;
    h[0] = Xh;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        f[p] = f[p] + Vf[p] * Xf + Wf[p] * Xrf;
        h[p] = h[p] + Vh[p] * Xh + Wh[p] * Xrh;
    }
}

// L = lines per tile: 32 (a wave holds two chunks) or 16 (four chunks; half the tile, so that TWO workgroups fit a CU and one can
// load or store while the other solves)
// DIV (MODE_BURGERS, one field = the advecting velocity, at most 16 chunks): RTileArgs::fdiv, the forcing term of this direction from the finished
// tendency while the lines are still in registers.
// ANEL (MODE_BURGERS): 1 / 2 = anelastic diffusion weight along y / z lines (RTileArgs::ari); a template parameter because the register allocation
// of the incompressible kernel sits at 256 and must not see the extra path
// UNI (two systems): uniform grid and Jacobian schemes only -- no Jacobian correction, no per-row right-hand-side coefficients -- so that both systems are
// solved in one pass (h_solve2); its own instantiation because the allocation of the general kernel sits at 256 registers
template <int M, int MODE, int MAXT, int L, bool DIV = false, int ANEL = 0, bool UNI = false>
__global__ void __launch_bounds__(MAXT, ((L == 16 && MAXT == 256) ? 2 : 1)) k_htile(RTileArgs a) {
    __shared__ double s_yl[32 * L];
    __shared__ double s_r[32 * L];
    __shared__ double s_e[2 * 32 * L];   // first-derivative edge values of every chunk (Jacobian correction)
    __shared__ double s_h[DIV ? 6 * MAXT : 1];        // DIV: first and last three rows of every chunk's forcing operand (C * L = MAXT threads at most)
    extern __shared__ double s_tab[];     // coefficient rows of both systems, separator inverses, Jacobian-correction diagonals
    constexpr bool NEED1 = (MODE == MODE_P1 || MODE == MODE_P2_P1 || MODE == MODE_BURGERS);
    constexpr bool NEED2 = (MODE != MODE_P1);
    // A wave holds 64 / L chunks: its first chunk cw is wave-uniform (readfirstlane makes that visible to the compiler), the lane adds csub.  Addresses are
    // then   field + [tile + cw M rs + p rs]  (64-bit, SCALAR registers and scalar adds)  +  [l32 + csub M rs] (one 32-bit byte offset per lane, the same
    // for every row p and every field): the loads and stores take the SGPR-base + VGPR-offset form and no 64-bit vector multiply-add per access is left
    // (they were a third of the kernel's vector instructions).  launch_htile refuses row strides for which the lane part leaves 32 bits.
    const int l32 = threadIdx.x & (L - 1);
    const int cw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (64 / L);
    const int csub = (threadIdx.x & 63) / L;
    const int c = cw + csub;
    const int C = blockDim.x / L;
    const int n = a.g.n;
    const long long rs = a.g.row_stride;
    const bool per = a.s1.periodic != 0;
    const bool corr = !UNI && a.jc.j != nullptr;        // non-uniform grid: second derivative needs the first one

    // MODE_BURGERS: one launch serves every transported field that shares the advecting velocity.  Workgroup ids are dealt round-robin
    // to the 8 XCDs (each with its own L2), so the nf workgroups of a tile get ids with the same residue mod 8 and follow each other
    // closely in the dispatch order: bid = x + 8 (f + nf y), tile = x + 8 y.  The velocity tile is then fetched from HBM once and the
    // other fields' reads of it hit (or merge in) that XCD's L2.
    const int tiles_inner = (a.g.lines_inner + L - 1) / L;
    long long tile = blockIdx.x;
    int fi = 0;
    if (MODE == MODE_BURGERS) {
        const long long q = blockIdx.x >> 3;
        fi = (int)(q % a.nf);
        tile = (blockIdx.x & 7) + 8 * (q / a.nf);
        if (tile >= (long long)tiles_inner * (a.g.nlines / a.g.lines_inner)) return;     // whole workgroup, before any barrier
    }
    const long long outer = tile / tiles_inner;
    const int l0 = (int)(tile % tiles_inner) * L;
    const bool valid = (l0 + l32) < a.g.lines_inner;
    const int lv = valid ? l32 : 0;       // lanes beyond the last line of a partial tile LOAD line l0 of the tile (finite values, never stored): no branch per load
    const long long base = outer * a.g.outer_stride + l0 + lv;
    const int row0 = c * M;
    const long long ub = outer * a.g.outer_stride + l0 + (long long)cw * M * rs;      // wave-uniform element offset of row cw M, line l0
    const unsigned vb = (unsigned)((lv + (long long)csub * M * rs) * 8);              // the lane's byte offset from there
    const long long rs8 = rs * 8;
    // (the row pointers go through an empty asm with an SGPR constraint: left visible, the compiler re-associates field + ub + vb first and adds p rs
    // to that 64-bit VECTOR address again; they are advanced by scalar adds, row after row)
#define H_ROW0(f) sgpr_ptr((f) + ub)
#define H_NEXT(r) r = sgpr_ptr(r + rs8)
#define H_AT(r) (reinterpret_cast<gdouble *>(r + vgpr_u32(vb)))
    const double *__restrict__ in0 = (MODE == MODE_BURGERS) ? a.fs[fi] : a.in0;
    double *__restrict__ out0 = (MODE == MODE_BURGERS) ? a.fo[fi] : a.out0;
    double nu = (MODE == MODE_BURGERS) ? a.fnu[fi] : a.nu;
    if (ANEL == 2) nu = nu * a.ari[(l0 / a.ari_nx) % a.ari_ny];      // z lines: ribackground of the tile's y row

    // ---- operand rows + 3-row halos: requested BEFORE the tables are staged, so that the two latencies overlap ----
    double e[M + 6];
    {
        gchar *r = H_ROW0(in0);
#pragma unroll
        for (int p = 0; p < M; ++p) { e[p + 3] = *H_AT(r); H_NEXT(r); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int rl = row0 - 3 + k, rr = row0 + M + k;
        const bool okl = per || rl >= 0, okr = per || rr < n;
        if (rl < 0) rl += n;
        if (rr >= n) rr -= n;
        e[k] = okl ? in0[base + (long long)rl * rs] : 0.0;
        e[M + 3 + k] = okr ? in0[base + (long long)rr * rs] : 0.0;
    }
    if constexpr (MODE == MODE_P1) {
        if (a.in0b != nullptr) {   // operand = in0 + s * in0b (same rows, same halos), as in k_rtile
            gchar *r = H_ROW0(a.in0b);
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {      // 8 rows in flight: the operand rows are all live here (128 VGPRs on 1024 threads)
                double t[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) { t[p] = *H_AT(r); H_NEXT(r); }
#pragma unroll
                for (int p = 0; p < 8; ++p) e[p0 + p + 3] = e[p0 + p + 3] + t[p] * a.in0b_scale;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int rl = row0 - 3 + k, rr = row0 + M + k;
                const bool okl = per || rl >= 0, okr = per || rr < n;
                if (rl < 0) rl += n;
                if (rr >= n) rr -= n;
                if (okl) e[k] = e[k] + a.in0b[base + (long long)rl * rs] * a.in0b_scale;
                if (okr) e[M + 3 + k] = e[M + 3 + k] + a.in0b[base + (long long)rr * rs] * a.in0b_scale;
            }
        }
    }


    // stage the tables once per workgroup: the two halves of a wave work on different rows, so the coefficient rows are not
    // wave-uniform here; LDS reads with two distinct addresses per wave are broadcast, vector loads from L1 cost ~120 cycles each
    double *t1 = s_tab, *t2 = s_tab + 5 * n, *gi1 = s_tab + 10 * n, *gi2 = gi1 + C * C, *tj = gi2 + C * C;
    {
        const bool use1 = NEED1 || corr;
        for (int i = threadIdx.x; i < 5 * n; i += blockDim.x) {
            if (use1) t1[i] = a.y1.rowtab[i];
            if (NEED2) t2[i] = a.y2.rowtab[i];
        }
        for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
            if (use1) gi1[i] = a.y1.red[i];
            if (NEED2) gi2[i] = a.y2.red[i];
        }
        if (NEED2 && corr)
            for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) tj[i] = a.jc.j[i];
        if (!UNI && NEED2 && a.s2.rowc != nullptr)      // direct second derivative: per-row RHS coefficients (never together with the correction)
            for (int i = threadIdx.x; i < 5 * n; i += blockDim.x) tj[i] = a.s2.rowc[i];
    }
    // (the first __syncthreads inside h_solve would be too late for the coefficient reads of the local sweeps)
    __syncthreads();

    // ---- right-hand sides of both systems from the one operand ----
    double x1[M], x2[NEED2 ? M : 1];
    if (NEED1 || (NEED2 && corr)) {
#pragma unroll
        for (int p = 0; p < M; ++p) x1[p] = h_stencil<false>(a.s1, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
        if (!UNI && a.s1.rowc != nullptr) {   // direct first derivative (MatMul_3d / MatMul_5d): per-row coefficients, read where they lie (rare scheme)
            const double *rc = a.s1.rowc + row0 * 5;
#pragma unroll
            for (int p = 0; p < M; ++p)
                x1[p] = e[p + 1] * rc[p * 5 + 0] + e[p + 2] * rc[p * 5 + 1] + e[p + 3] * rc[p * 5 + 2] + e[p + 4] * rc[p * 5 + 3] + e[p + 5] * rc[p * 5 + 4];
        }
        if (!per) {
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) x1[r] = h_dense6(a.s1.bb[r], e[3], e[4], e[5], e[6], e[7], e[8]);
            }
            if (c == C - 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) x1[M - 3 + r] = h_dense6(a.s1.bt[r], e[M - 3], e[M - 2], e[M - 1], e[M], e[M + 1], e[M + 2]);
            }
        }
    }
    if constexpr (NEED2) {
#pragma unroll
        for (int p = 0; p < M; ++p) x2[p] = h_stencil<true>(a.s2, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
        if (!UNI && a.s2.rowc != nullptr) {   // MatMul_5d interior (fdm_matmul.f90:303-305) with the per-row coefficients staged in LDS
            const double *rc = tj + row0 * 5;
#pragma unroll
            for (int p = 0; p < M; ++p)
                x2[p] = e[p + 1] * rc[p * 5 + 0] + e[p + 2] * rc[p * 5 + 1] + e[p + 3] * rc[p * 5 + 2] + e[p + 4] * rc[p * 5 + 3] + e[p + 5] * rc[p * 5 + 4];
        }
        if (!per) {
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) x2[r] = h_dense6(a.s2.bb[r], e[3], e[4], e[5], e[6], e[7], e[8]);
            }
            if (c == C - 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) x2[M - 3 + r] = h_dense6(a.s2.bt[r], e[M - 3], e[M - 2], e[M - 1], e[M], e[M + 1], e[M + 2]);
            }
        }
    }

    // advecting velocity: issued now (the operand registers are dead) so that its latency hides behind the two solves
    double vl[MODE == MODE_BURGERS ? M : 1];
    if constexpr (MODE == MODE_BURGERS) {
        gchar *r = H_ROW0(a.in2);
#pragma unroll
        for (int p = 0; p < M; ++p) { vl[p] = *H_AT(r); H_NEXT(r); }
    }

    if constexpr (UNI && NEED1 && NEED2) {      // both systems in one pass: the second right-hand side is complete (no Jacobian correction)
        h_solve2<M, L>(x1, x2, t1, t2, gi1, gi2, n, row0, c, C, l32, s_yl, s_r, s_e, s_e + 32 * L);
    } else {
    // ---- first derivative ----
    if (NEED1 || (NEED2 && corr)) h_solve<M, L>(x1, t1, gi1, n, row0, c, C, l32, s_yl, s_r);

    // ---- second derivative (+ Jacobian correction  f += A2 dx2 du, MatMul_3d_add fdm_matmul.f90:126-153) ----
    if constexpr (NEED2) {
        if (corr) {
            s_e[(2 * c + 0) * L + l32] = x1[0];
            s_e[(2 * c + 1) * L + l32] = x1[M - 1];
            __syncthreads();
            const double dl = (c > 0) ? s_e[(2 * (c - 1) + 1) * L + l32] : 0.0;   // du at row0 - 1
            const double dr = (c < C - 1) ? s_e[(2 * (c + 1) + 0) * L + l32] : 0.0;  // du at row0 + M
            const double *j1 = tj + row0, *j2 = tj + n + row0, *j3 = tj + 2 * n + row0;
#pragma unroll
            for (int p = 0; p < M; ++p) {
                const double dm = (p == 0) ? dl : x1[p > 0 ? p - 1 : 0];
                const double dp = (p == M - 1) ? dr : x1[p < M - 1 ? p + 1 : M - 1];
                double add = dm * j1[p] + x1[p] * j2[p] + dp * j3[p];
                if (p == 0 && c == 0) add = x1[0] * j2[0] + x1[1] * j3[0] + x1[2] * j1[0];                          // r1(1) extended stencil
                if (p == M - 1 && c == C - 1) add = x1[M - 3] * j3[M - 1] + x1[M - 2] * j1[M - 1] + x1[M - 1] * j2[M - 1];  // r3(n)
                x2[p] = x2[p] + add;
            }
        }
        h_solve<M, L>(x2, t2, gi2, n, row0, c, C, l32, s_yl, s_r);
    }
    }

    // ---- epilogue ----
    if (valid) {
    gchar *ro = H_ROW0(out0);
    if constexpr (MODE == MODE_P1) {
        if (a.fq != nullptr) {
            // final-update epilogue of k_rtile (z-direction lines: lane = (ix, j) inside the plane, rows = k): h -= dp/dz, wall planes, q += dte h, h *= kco;
            // 8 rows at a time (the 1024-thread launch has 128 VGPRs)
            const int li = l0 + l32;
            const int j = (li / a.fnx) % a.fny;
            const bool wall = (j == 0) || (j == a.fny - 1);
            const double *wp = wall ? (j == 0 ? a.fpb : a.fpt) : nullptr;      // given wall tendencies (Neumann walls), or zero
            if (wp != nullptr) wp += li % a.fnx;
            gchar *rq = H_ROW0(a.fq);
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {
                double h[8], qv[8];
                gchar *rh2 = ro, *rq2 = rq;
#pragma unroll
                for (int p = 0; p < 8; ++p) { h[p] = *H_AT(rh2); qv[p] = *H_AT(rq2); H_NEXT(rh2); H_NEXT(rq2); }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const double hv = wall ? (wp ? wp[(long long)(row0 + p0 + p) * a.fnx] : 0.0) : h[p] - x1[p0 + p];
                    qv[p] = qv[p] + a.fdte * hv;
                    h[p] = a.fscale ? a.fkco * hv : hv;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) { *H_AT(rq) = qv[p]; *H_AT(ro) = h[p]; H_NEXT(rq); H_NEXT(ro); }
            }
        } else if (a.acc) {      // result = old +- derivative: loads first, 8 rows at a time
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {
                double o[8];
                gchar *r2 = ro;
#pragma unroll
                for (int p = 0; p < 8; ++p) { o[p] = *H_AT(r2); H_NEXT(r2); }
#pragma unroll
                for (int p = 0; p < 8; ++p) { *H_AT(ro) = (a.acc == 2) ? o[p] - x1[p0 + p] : o[p] + x1[p0 + p]; H_NEXT(ro); }
            }
        } else {
#pragma unroll
            for (int p = 0; p < M; ++p) { *H_AT(ro) = x1[p]; H_NEXT(ro); }
        }
    } else if constexpr (MODE == MODE_P2) {
#pragma unroll
        for (int p = 0; p < M; ++p) { *H_AT(ro) = x2[p]; H_NEXT(ro); }
    } else if constexpr (MODE == MODE_P2_P1) {
        gchar *r1 = H_ROW0(a.out1);
#pragma unroll
        for (int p = 0; p < M; ++p) {
            *H_AT(ro) = x2[p];
            *H_AT(r1) = x1[p];
            H_NEXT(ro); H_NEXT(r1);
        }
    } else {  // MODE_BURGERS: result = nu d2 - vel d1 (opr_burgers.f90:513)
        if (ANEL == 1) {      // y lines: ribackground(j) of every row on the diffusion term (opr_burgers.f90:128-183)
#pragma unroll
            for (int p = 0; p < M; ++p) x2[p] = (nu * a.ari[row0 + p]) * x2[p] - vl[p] * x1[p];
        } else {
#pragma unroll
            for (int p = 0; p < M; ++p) x2[p] = nu * x2[p] - vl[p] * x1[p];
        }
        // the old tendency is read once and the new one is not read again before 4 GB of other traffic have passed: non-temporal accesses keep
        // them out of the way of the operand rows in L2 (3 % of the launch, measured A/B in one binary)
        if (a.acc && !((a.fresh_mask >> fi) & 1u)) {   // accumulate into the tendency: all loads first (the compiler cannot move them across the stores itself)
#pragma unroll
            for (int p = 0; p < M; ++p) { x1[p] = __builtin_nontemporal_load(H_AT(ro)); H_NEXT(ro); }
            ro = H_ROW0(out0);
#pragma unroll
            for (int p = 0; p < M; ++p) x2[p] = x1[p] + x2[p];
        }
#pragma unroll
        for (int p = 0; p < M; ++p) { __builtin_nontemporal_store(x2[p], H_AT(ro)); H_NEXT(ro); }
    }
    }   // valid
    if constexpr (DIV && MODE == MODE_BURGERS) {
        // x2 = the finished tendency h of the velocity component of this direction, vl = that component: forcing term d/dy (h + fidte v), the same
        // operand, stencil and system as the separate fused launch (k_rtile MODE_P1 with in0b), added to fdiv
        double tt[M + 6];
#pragma unroll
        for (int p = 0; p < M; ++p) tt[p + 3] = valid ? x2[p] + vl[p] * a.fidte : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s_h[(c * 6 + k) * L + l32] = tt[3 + k];
            s_h[(c * 6 + 3 + k) * L + l32] = tt[M + k];
        }
        __syncthreads();
        {
            const int cl = (c + C - 1) % C, cr = (c + 1) % C;
            const bool okl = per || c > 0, okr = per || c < C - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                tt[k] = okl ? s_h[(cl * 6 + 3 + k) * L + l32] : 0.0;
                tt[M + 3 + k] = okr ? s_h[(cr * 6 + k) * L + l32] : 0.0;
            }
        }
        double gg[M];
#pragma unroll
        for (int p = 0; p < M; ++p) gg[p] = h_stencil<false>(a.s1, tt[p], tt[p + 1], tt[p + 2], tt[p + 3], tt[p + 4], tt[p + 5], tt[p + 6]);
        if (!per) {
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) gg[r] = h_dense6(a.s1.bb[r], tt[3], tt[4], tt[5], tt[6], tt[7], tt[8]);
            }
            if (c == C - 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) gg[M - 3 + r] = h_dense6(a.s1.bt[r], tt[M - 3], tt[M - 2], tt[M - 1], tt[M], tt[M + 1], tt[M + 2]);
            }
        }
        h_solve<M, L>(gg, t1, gi1, n, row0, c, C, l32, s_yl, s_r);
        if (valid) {
            double fo[M];
            gchar *r = H_ROW0(a.fdiv);
#pragma unroll
            for (int p = 0; p < M; ++p) { fo[p] = *H_AT(r); H_NEXT(r); }
            r = H_ROW0(a.fdiv);
#pragma unroll
            for (int p = 0; p < M; ++p) { *H_AT(r) = fo[p] + gg[p]; H_NEXT(r); }
        }
    }
#undef H_AT
#undef H_NEXT
#undef H_ROW0
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// k_ptile: OPR_Burgers along y / z on uniform grids with PERSISTENT workgroups whose operand tiles arrive by LDS-DMA.
//
// k_htile is one workgroup per CU (256 registers x 512 threads) whose waves load, solve and store in lockstep: the knock-out timings of round 4
// (profiles/r04/htile_knockout.txt) say the memory phases of a launch take 1.8 ms and the two solves plus the staging of the tables another 0.5 ms
// that nothing hides.  Here a workgroup walks over its work items (tile, field): while it computes item k, `global_load_lds_dwordx4` instructions
// (no destination registers) bring the operand tile of item k + 1 into LDS -- 32 lines x n rows x 8 B = 128 KiB at n = 512 -- so that the operand
// trip from HBM runs beside the stencils, the velocity load and the solves.  The halo rows of a chunk are rows of the same LDS tile (no second read
// of the operand through L2), and the tables are staged once per workgroup, not once per tile.  To fit beside the tile the tables are kept in their
// chunk-invariant form (first / interior / last chunk: 3 x 10 x M doubles): uniform grids only, checked on the host (launch_htile).
// LDS: tile 128 KiB + separator values of both systems 16 KiB + tables 7.5 KiB + separator inverses 4 KiB = 155.5 of 160 KiB.
// The DMA is inline assembly (the compiler drains its own `__builtin_amdgcn_global_load_lds` with vmcnt(0) at the next ordinary load or barrier): loads
// return in order, so the wait for the velocity / old tendency of item k, which are issued AFTER the DMA of item k + 1, retires that DMA as well; an
// explicit vmcnt(0) before the stores states it.
// DIV: the one-field launch of the velocity component of this direction with the forcing term of the pressure equation as a third solve (k_htile's DIV);
// the operand IS the advecting velocity then, which is taken from the LDS tile instead of a second load.
// <32, 16, 32, ., BAND>: lines of 1024 points in 16-line tiles (the same 128 KiB); the separator inverses (2 x 32 x 32 doubles) no longer fit beside the
// tile and are kept as their five central diagonals (SystemDev::band: what lies further out is below 1e-30 of the diagonal, checked per plan).
template <int M, int L, int C, bool DIV = false, bool BAND = false>
__global__ void __launch_bounds__(L * C, 1) k_ptile(RTileArgs a, long long nitems) {
    constexpr int N = M * C;                 // points per line
    constexpr int GI = BAND ? 5 * C : C * C; // separator inverse of one system
    __shared__ double s_sep[4 * C * L];      // s_yl, s_r of both systems
    __shared__ double s_t[2 * 3 * 5 * M];    // [system][first / interior / last chunk][Lm, Dinv, Cm, V, W][M]
    __shared__ double s_gi[2 * GI];
    extern __shared__ double s_op[];         // operand tile [N][L]
    const int l32 = threadIdx.x & (L - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cw = wv * (64 / L);
    const int csub = (threadIdx.x & 63) / L;
    const int c = cw + csub;
    const long long rs = a.g.row_stride;
    const long long rs8 = rs * 8;
    const bool per = a.s1.periodic != 0;
    const int tiles_inner = a.g.lines_inner / L;
    const long long ntiles = (long long)tiles_inner * (a.g.nlines / a.g.lines_inner);

    // tables: once per workgroup
    for (int i = threadIdx.x; i < 2 * 3 * 5 * M; i += blockDim.x) {
        const int sys = i / (15 * M), v = (i / (5 * M)) % 3, arr = (i / M) % 5, p = i % M;
        const int chunk = v == 0 ? 0 : (v == 1 ? 1 : C - 1);
        s_t[i] = (sys == 0 ? a.y1.rowtab : a.y2.rowtab)[arr * N + chunk * M + p];
    }
    for (int i = threadIdx.x; i < GI; i += blockDim.x) {
        s_gi[i] = BAND ? a.y1.band[i] : a.y1.red[i];
        s_gi[GI + i] = BAND ? a.y2.band[i] : a.y2.red[i];
    }
    const int var = (c == 0) ? 0 : (c == C - 1 ? 2 : 1);
    const double *t1 = s_t + var * 5 * M, *t2 = s_t + (3 + var) * 5 * M;

    // item -> (tile, field): as k_htile (the fields of a tile on one XCD at the same time: their velocity reads share its L2)
    auto decode = [&](long long it, long long &tile, int &fi) {
        const long long q = it >> 3;
        fi = (int)(q % a.nf);
        tile = (it & 7) + 8 * (q / a.nf);
    };
    // LDS-DMA of the operand tile of an item: wave w brings rows [RW w, RW w + RW), 1 KiB (RI rows of L doubles) per instruction, lane = (row of the
    // instruction, 16-B piece of the row)
    constexpr int PR = L / 2, RI = 64 / PR, RW = N / (L * C / 64), NI = RW / RI;      // pieces per row, rows per instruction, rows per wave, instructions
    const unsigned lane = threadIdx.x & 63;
    const unsigned dma_voff = (unsigned)(((long long)(lane / PR) * rs + (lane % PR) * 2) * 8);
    const unsigned lds_base = (unsigned)(unsigned long long)s_op;
    auto prefetch = [&](long long it) {
        long long tile;
        int fi;
        decode(it, tile, fi);
        if (tile >= ntiles) return;
        const long long outer = tile / tiles_inner;
        const int l0 = (int)(tile % tiles_inner) * L;
        gchar *src = sgpr_ptr(a.fs[fi] + outer * a.g.outer_stride + l0 + (long long)(RW * wv) * rs);
        unsigned dst = lds_base + (unsigned)(RW * wv) * (L * 8);
        const long long step = RI * rs8;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(dma_voff), "s"(src), "s"(dst)
                         : "memory");
            src = sgpr_ptr(src + step);
            dst += RI * L * 8;
        }
    };

    long long item = blockIdx.x;
    prefetch(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // tables and the first tile are in LDS
    for (; item < nitems; item += gridDim.x) {
        long long tile;
        int fi;
        decode(item, tile, fi);
        if (tile >= ntiles) {                 // padding of the item range: nothing to compute (its own prefetch was skipped as well), but the operand
            // tile of the workgroup's NEXT item must still be brought in -- that item is padding too only when gridDim.x is a multiple of 8
            if (item + gridDim.x < nitems) prefetch(item + gridDim.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            continue;
        }
        const long long outer = tile / tiles_inner;
        const int l0 = (int)(tile % tiles_inner) * L;
        const int row0 = c * M;
        const long long ub = outer * a.g.outer_stride + l0 + (long long)cw * M * rs;
        const unsigned vb = (unsigned)((l32 + (long long)csub * M * rs) * 8);
        double *__restrict__ out0 = a.fo[fi];
        const double nu = a.fnu[fi];

        // ---- operand rows + halos from the LDS tile ----
        double e[M + 6];
        {
            const double *col = s_op + l32 + row0 * L;
#pragma unroll
            for (int p = 0; p < M; ++p) e[p + 3] = col[p * L];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int rl = row0 - 3 + k, rr = row0 + M + k;
                const bool okl = per || rl >= 0, okr = per || rr < N;
                if (rl < 0) rl += N;
                if (rr >= N) rr -= N;
                e[k] = okl ? s_op[l32 + rl * L] : 0.0;
                e[M + 3 + k] = okr ? s_op[l32 + rr * L] : 0.0;
            }
        }
        __syncthreads();      // every wave has read its rows: the tile may be overwritten
        if (item + gridDim.x < nitems) prefetch(item + gridDim.x);

        // ---- right-hand sides of both systems ----
        double x1[M], x2[M];
#pragma unroll
        for (int p = 0; p < M; ++p) x1[p] = h_stencil<false>(a.s1, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
#pragma unroll
        for (int p = 0; p < M; ++p) x2[p] = h_stencil<true>(a.s2, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
        if (!per) {
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    x1[r] = h_dense6(a.s1.bb[r], e[3], e[4], e[5], e[6], e[7], e[8]);
                    x2[r] = h_dense6(a.s2.bb[r], e[3], e[4], e[5], e[6], e[7], e[8]);
                }
            }
            if (c == C - 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    x1[M - 3 + r] = h_dense6(a.s1.bt[r], e[M - 3], e[M - 2], e[M - 1], e[M], e[M + 1], e[M + 2]);
                    x2[M - 3 + r] = h_dense6(a.s2.bt[r], e[M - 3], e[M - 2], e[M - 1], e[M], e[M + 1], e[M + 2]);
                }
            }
        }
#define H_ROW0(f) sgpr_ptr((f) + ub)
#define H_NEXT(r) r = sgpr_ptr(r + rs8)
#define H_AT(r) (reinterpret_cast<gdouble *>(r + vgpr_u32(vb)))
        double vl[M];
        if constexpr (DIV) {
#pragma unroll
            for (int p = 0; p < M; ++p) vl[p] = e[p + 3];
        } else {
            gchar *r = H_ROW0(a.in2);
#pragma unroll
            for (int p = 0; p < M; ++p) { vl[p] = *H_AT(r); H_NEXT(r); }
        }
        h_solve2<M, L, BAND>(x1, x2, t1, t2, s_gi, s_gi + GI, M, 0, c, C, l32, s_sep, s_sep + C * L, s_sep + 2 * C * L, s_sep + 3 * C * L);

        // ---- epilogue: result = nu d2 - vel d1 (opr_burgers.f90:513), accumulated into the tendency ----
#pragma unroll
        for (int p = 0; p < M; ++p) x2[p] = nu * x2[p] - vl[p] * x1[p];
        gchar *ro = H_ROW0(out0);
        if (a.acc && !((a.fresh_mask >> fi) & 1u)) {
#pragma unroll
            for (int p = 0; p < M; ++p) { x1[p] = __builtin_nontemporal_load(H_AT(ro)); H_NEXT(ro); }
            ro = H_ROW0(out0);
#pragma unroll
            for (int p = 0; p < M; ++p) x2[p] = x1[p] + x2[p];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every load of this wave has landed, the DMA of the next tile included
#pragma unroll
        for (int p = 0; p < M; ++p) { __builtin_nontemporal_store(x2[p], H_AT(ro)); H_NEXT(ro); }
        if constexpr (DIV) {
            // x2 = the finished tendency h of this velocity component, vl = the component: forcing term d/dz (h + fidte w) (rhs_global_incompressible_1.f90:
            // 197-230), same stencil and system as the first derivative above, added to fdiv.  The three rows a chunk needs from each neighbour travel
            // through the separator buffer in two rounds (first rows to the left neighbour, last rows to the right one).
            double tt[M + 6];
#pragma unroll
            for (int p = 0; p < M; ++p) tt[p + 3] = x2[p] + vl[p] * a.fidte;
            const int cl = (c + C - 1) % C, cr = (c + 1) % C;
            const bool okl = per || c > 0, okr = per || c < C - 1;
            __syncthreads();      // the separator values of the dual solve have been read by everyone
#pragma unroll
            for (int k = 0; k < 3; ++k) s_sep[(c * 3 + k) * L + l32] = tt[3 + k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) tt[M + 3 + k] = okr ? s_sep[(cr * 3 + k) * L + l32] : 0.0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) s_sep[(c * 3 + k) * L + l32] = tt[M + k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) tt[k] = okl ? s_sep[(cl * 3 + k) * L + l32] : 0.0;
            __syncthreads();      // ... before h_solve writes its separator values there
            double gg[M];
#pragma unroll
            for (int p = 0; p < M; ++p) gg[p] = h_stencil<false>(a.s1, tt[p], tt[p + 1], tt[p + 2], tt[p + 3], tt[p + 4], tt[p + 5], tt[p + 6]);
            if (!per) {
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) gg[r] = h_dense6(a.s1.bb[r], tt[3], tt[4], tt[5], tt[6], tt[7], tt[8]);
                }
                if (c == C - 1) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) gg[M - 3 + r] = h_dense6(a.s1.bt[r], tt[M - 3], tt[M - 2], tt[M - 1], tt[M], tt[M + 1], tt[M + 2]);
                }
            }
            h_solve<M, L, BAND>(gg, t1, s_gi, M, 0, c, C, l32, s_sep, s_sep + C * L);
            double fo[M];
            gchar *r = H_ROW0(a.fdiv);
#pragma unroll
            for (int p = 0; p < M; ++p) { fo[p] = *H_AT(r); H_NEXT(r); }
            r = H_ROW0(a.fdiv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < M; ++p) { *H_AT(r) = fo[p] + gg[p]; H_NEXT(r); }
        }
#undef H_AT
#undef H_NEXT
#undef H_ROW0
        __syncthreads();      // the next tile is in LDS for every wave (and the separator values of this item are no longer read)
    }
}

static int g_htile_persist = [] { const char *e = getenv("TLAB_HTILE_PERSIST"); return (e && atoi(e) == 0) ? 0 : 1; }();      // 0: k_htile<UNI> instead of k_ptile (A/B)
// 512-point lines: the banded separator inverses as well -- 2.07 -> 2.00 ms for the three-field launch, 1.05 -> 1.10 ms for the one with the forcing term
// (A/B on one box): 1 = the first only (default), 2 = both, 0 = neither
static int g_ptile_band = [] { const char *e = getenv("TLAB_PTILE_BAND"); return e ? atoi(e) : 1; }();
static int g_ncu = 0;
static int g_ptile_grid = 0;       // > 0: forced number of persistent workgroups (tests: a count that is not a multiple of 8)
void ptile_set_grid(int n) { g_ptile_grid = n > 0 ? n : 0; }
static long long ptile_cus() {
    if (g_ptile_grid) return g_ptile_grid;
    if (!g_ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) g_ncu = pr.multiProcessorCount;
        if (g_ncu <= 0) g_ncu = 256;
        // items of one tile-octet sit on the 8 XCDs (item & 7): a grid that is a multiple of 8 keeps (item & 7) fixed per workgroup, so that a workgroup whose
        // first item is padding only ever sees padding (partitioned parts report 38 / 228 CUs; the kernel is correct without this, only slower)
        g_ncu = g_ncu >= 8 ? (g_ncu / 8) * 8 : 8;
    }
    return g_ncu;
}
// the persistent kernel takes 512-point lines in 32-line tiles whose tables are the same for every interior chunk (periodic z of a uniform grid)
// ... or 1024-point lines in 16-line tiles whose separator inverses are banded (five diagonals of them in LDS)
static bool ptile_ok(const RTileArgs &a, int L, int C) {
    if (!g_htile_persist || a.g.lines_inner % L != 0 || !a.y1.chunk_invariant || !a.y2.chunk_invariant) return false;
    if (L == 32 && C == 16 && a.g.n == 512) return true;
    static const bool long_ok = [] { const char *e = getenv("TLAB_PTILE_1024"); return !(e && atoi(e) == 0); }();
    return long_ok && L == 16 && C == 32 && a.g.n == 1024 && a.y1.band != nullptr && a.y2.band != nullptr;
}
static int g_htile_uni = [] { const char *e = getenv("TLAB_HTILE_UNI"); return (e && atoi(e) == 0) ? 0 : 1; }();      // 0: general kernel on uniform grids too (A/B)
static int g_htile_lines = [] { const char *e = getenv("TLAB_HTILE_LINES"); return (e && atoi(e) == 16) ? 16 : 32; }();
void htile_set_lines(int lines) { g_htile_lines = lines; }
bool htile_narrow() { return g_htile_lines == 16; }

// chunk length of the half-wave-tile kernel for a line length n and a mode (0 = unsupported)
int htile_chunk(int n, int mode) {
    const bool two = (mode == MODE_P2_P1 || mode == MODE_BURGERS);   // two line-sets live in registers
    auto ok = [&](int m, int cmax) { return n % m == 0 && (n / m) % 2 == 0 && n / m <= cmax; };   // two chunks per wave -> even count
    if (two) {
        if (ok(32, 16)) return 32;      // <= 512 threads -> 256-VGPR budget
        if (ok(16, 16)) return 16;
        if (n == 1024) return 32;       // 32 chunks of 32 rows on 16-LINE tiles: still 512 threads (launch_htile)
        return 0;
    }
    if (ok(32, 32)) return 32;          // up to n = 1024 with 1024 threads (P1/P2: < 128 VGPRs)
    if (ok(64, 16)) return 64;
    if (ok(16, 32)) return 16;
    return 0;
}

template <int M, int MAXT, int L = 32>
static hipError_t launch_htile_m(int mode, int C, long long tiles, const RTileArgs &a, hipStream_t st) {
    const long long nwg = (mode == MODE_BURGERS) ? 8LL * a.nf * ((tiles + 7) / 8) : tiles;      // see the blockIdx mapping in the kernel
    const dim3 grid((unsigned)nwg), block(L * C);
    const size_t lds = ((size_t)(a.s2.rowc ? 15 : 13) * a.g.n + (size_t)2 * C * C) * sizeof(double);     // 10n tables + max(3n correction, 5n per-row RHS)
    const double pts = (double)a.g.nlines * a.g.n;
    const char *name = mode == MODE_P1 ? "k_htile<P1>" : mode == MODE_P2 ? "k_htile<P2>" : mode == MODE_P2_P1 ? "k_htile<P2_P1>" :
                       a.fdiv ? "k_htile<BURGERS+div>" : "k_htile<BURGERS>";
    const double bpp = (mode == MODE_P1 || mode == MODE_P2) ? 16 : 24;
    double bytes = pts * (bpp + (a.acc ? 8 : 0));
    if (mode == MODE_P1) bytes += pts * ((a.in0b ? 8 : 0) + (a.fq ? 24 : 0));
    if (mode == MODE_BURGERS) {   // velocity once (re-reads are L2 hits by construction) + per field: operand unless it is the velocity, result, old result
        bytes = pts * 8;
        for (int f = 0; f < a.nf; ++f) bytes += pts * ((a.fs[f] == a.in2 ? 0 : 8) + 8 + ((a.acc && !((a.fresh_mask >> f) & 1u)) ? 8 : 0));
        if (a.fdiv) bytes += pts * 16;      // forcing term: read + write
    }
    const bool uni = g_htile_uni && a.jc.j == nullptr && a.s1.rowc == nullptr && a.s2.rowc == nullptr;      // uniform grid, Jacobian schemes
    if constexpr (M == 32 && MAXT == 512)
        if (mode == MODE_BURGERS && uni && !a.ari && ptile_ok(a, L, C) && (!a.fdiv || (a.nf == 1 && a.fs[0] == a.in2)))
            name = a.fdiv ? "k_ptile<BURGERS+div>" : "k_ptile<BURGERS>";
    ProfScope ps(name, st, bytes);
    switch (mode) {
    case MODE_P1: hipLaunchKernelGGL((k_htile<M, MODE_P1, MAXT, L>), grid, block, lds, st, a); break;
    case MODE_P2: hipLaunchKernelGGL((k_htile<M, MODE_P2, MAXT, L>), grid, block, lds, st, a); break;
    case MODE_P2_P1: hipLaunchKernelGGL((k_htile<M, MODE_P2_P1, MAXT, L>), grid, block, lds, st, a); break;
    case MODE_BURGERS:
        if (a.fdiv) {
            if constexpr (M == 32 && MAXT == 512) {      // L = 32 with up to 16 chunks, or the 16-line tiles of 1024-point lines (32 chunks)
                if (a.nf != 1 || a.fs[0] != a.in2 || C * L > MAXT || a.s1.rowc != nullptr || a.ari) return hipErrorInvalidValue;
                if (uni && ptile_ok(a, L, C)) {
                    const unsigned pg = (unsigned)(nwg < ptile_cus() ? nwg : ptile_cus());
                    if constexpr (L == 32) {
                        if (g_ptile_band == 2 && a.y1.band && a.y2.band) hipLaunchKernelGGL((k_ptile<32, 32, 16, true, true>), dim3(pg), dim3(512), (size_t)a.g.n * 32 * sizeof(double), st, a, nwg);
                        else hipLaunchKernelGGL((k_ptile<32, 32, 16, true>), dim3(pg), dim3(512), (size_t)a.g.n * 32 * sizeof(double), st, a, nwg);
                    }
                    else hipLaunchKernelGGL((k_ptile<32, 16, 32, true, true>), dim3(pg), dim3(512), (size_t)a.g.n * 16 * sizeof(double), st, a, nwg);
                } else if (uni) hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L, true, 0, true>), grid, block, lds, st, a);
                else hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L, true>), grid, block, lds, st, a);
            } else {
                return hipErrorInvalidValue;
            }
        } else if (a.ari) {
            if constexpr (M == 32 && MAXT == 512 && L == 32) {
                if (a.ari_mode == 1) hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L, false, 1>), grid, block, lds, st, a);
                else if (a.ari_mode == 2 && a.ari_nx % L == 0) hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L, false, 2>), grid, block, lds, st, a);
                else return hipErrorInvalidValue;
            } else {
                return hipErrorInvalidValue;
            }
        } else {
            if constexpr (M == 32 && MAXT == 512) {
                if (uni && ptile_ok(a, L, C)) {
                    const unsigned pg = (unsigned)(nwg < ptile_cus() ? nwg : ptile_cus());      // items = the (padded) workgroup ids of k_htile
                    if constexpr (L == 32) {
                        if (g_ptile_band && a.y1.band && a.y2.band) hipLaunchKernelGGL((k_ptile<32, 32, 16, false, true>), dim3(pg), dim3(512), (size_t)a.g.n * 32 * sizeof(double), st, a, nwg);
                        else hipLaunchKernelGGL((k_ptile<32, 32, 16>), dim3(pg), dim3(512), (size_t)a.g.n * 32 * sizeof(double), st, a, nwg);
                    }
                    else hipLaunchKernelGGL((k_ptile<32, 16, 32, false, true>), dim3(pg), dim3(512), (size_t)a.g.n * 16 * sizeof(double), st, a, nwg);
                } else if (uni) hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L, false, 0, true>), grid, block, lds, st, a);
                else hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L>), grid, block, lds, st, a);
            } else {
                hipLaunchKernelGGL((k_htile<M, MODE_BURGERS, MAXT, L>), grid, block, lds, st, a);
            }
        }
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_htile(int mode, const RTileArgs &a, hipStream_t st) {
    const int n = a.g.n;
    const int M = htile_chunk(n, mode);
    if (M == 0) return hipErrorInvalidValue;
    const int C = n / M;
    if (C & 1) return hipErrorInvalidValue;      // two chunks per wave
    // the lane part of an address, (line in tile + chunk-in-wave x M rows) x 8 B, is a 32-bit byte offset in the kernel
    if ((double)(64 / 16 - 1) * M * (double)a.g.row_stride * 8.0 + 512.0 >= 4294967296.0) return hipErrorInvalidValue;
    const bool two = (mode == MODE_P2_P1 || mode == MODE_BURGERS);
    const bool narrow = (g_htile_lines == 16) && mode == MODE_BURGERS && M == 32 && C <= 16;
    const bool long16 = two && M == 32 && C == 32;     // lines of 1024 points with two line-sets in registers: 16-line tiles, 4 chunks per wave
    const int L = (narrow || long16) ? 16 : 32;
    const long long tiles_inner = (a.g.lines_inner + L - 1) / L;
    const long long tiles = tiles_inner * (a.g.nlines / a.g.lines_inner);
    if (narrow) return launch_htile_m<32, 256, 16>(mode, C, tiles, a, st);
    if (long16) return launch_htile_m<32, 512, 16>(mode, C, tiles, a, st);
    if (M == 64) return launch_htile_m<64, 512>(mode, C, tiles, a, st);
    if (M == 32) return (C <= 16) ? launch_htile_m<32, 512>(mode, C, tiles, a, st) : launch_htile_m<32, 1024>(mode, C, tiles, a, st);
    return (C <= 16) ? launch_htile_m<16, 512>(mode, C, tiles, a, st) : launch_htile_m<16, 1024>(mode, C, tiles, a, st);
}

}  // namespace tlab
