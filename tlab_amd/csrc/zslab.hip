// z-derivative operators on a z-SLAB without transposing the field (multi-GPU, one slab of kmax planes per GPU).
//
// The reference computes d/dz on a decomposed domain by transposing every operand to complete z-lines with an all-to-all
// (TLabMPI_Trp_ExecK_Forward/Backward around OPR_Partial_Z / OPR_Burgers_Z, opr_partial.f90:154-262, opr_burgers.f90:331-440).
// On point-to-point xGMI that moves the whole field twice per operator.  Here the implicit (compact) system is split at the
// slab boundaries instead -- the same partitioned Thomas algorithm the single-GPU kernels use between the waves of a workgroup
// (chunked.hpp), one level up:
//
//   slab of rank r = rows [k0, k0+kmax) of the periodic line; row k0 is the slab's separator S_r, the rest its interior.
//   interior:   x = y + X_r V + X_{r+1} W,   y = T_loc^-1 [0, f_1 .. f_{kmax-1}]   (T_loc: slab rows, identity first row)
//   separator:  alpha X_{r-1} + beta X_r + gamma X_{r+1} = f_S - a_S y^{r-1}_last - c_S y^r_1
//
// alpha, gamma are the far ends of the slab-level spikes, ~0.38^kmax for the sixth-order compact schemes: below 1e-19 beta for
// kmax >~ 50 (64 with the 16- or 32-row sub-chunks), i.e. exactly zero in double precision (checked at plan creation; thinner slabs are refused and the caller keeps
// the transpose path).  The interface system is then diagonal and each slab needs only
//   tail  = y_last            from its left  neighbour    (8 B per line and system)
//   head  = f_S - c_S y_1     from its right neighbour
// plus 3 halo planes of the operand for the explicit stencils: a neighbour exchange of a few planes instead of two
// all-to-alls of the field.  Phase A computes head/tail, the caller exchanges them, phase B recomputes y (cheaper than storing
// it), adds the spikes and runs the usual epilogue (first derivative / Burgers, optional accumulation into the tendency).
//
// Kernel layout as k_rtile: 64 memory-contiguous lines per workgroup, wave w owns rows [w*M, (w+1)*M) of the slab in registers,
// coefficient rows are wave-uniform scalar loads, sub-chunks are coupled through LDS with the dense inverse of their separator
// system.  Fields must carry 3 valid planes before and after the slab (the caller's halo exchange fills them).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tlab_amd.h"
#include "chunked.hpp"
#include "device_tables.hpp"
#include "plan.hpp"
#include "profile.hpp"

extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();

namespace tlab {

struct ZSysDev {
    const double *rowtab;   // [5][kmax] Lm, Dinv, Cm, V, W of the sub-chunked local system T_loc
    const double *ginv;     // [C][C]
    const double *vw;       // [2][kmax] slab-level spikes V, W
    double cS, aS, binv;    // separator row of this slab: head = f_S - cS y_1 ; X = (head - aS tail_left) * binv
    double aSn, binvn;      // separator row of the right neighbour: X_right = (head_right - aSn tail_mine) * binvn
};

struct ZSlabArgs {
    const double *in0;      // operand (first plane of the slab)
    const double *in0b;     // optional: operand = in0 + scale * in0b
    // the neighbours' 3 planes before (lo) and after (hi) the slab of every operand: in place (lo = in - 3 planes, hi = in + kmax planes: the
    // caller's arrays carry the room) or in buffers of their own (the native slab driver: a Fortran host's module arrays have no room)
    const double *lo0, *hi0, *lo0b, *hi0b;
    const double *flo[4], *fhi[4];
    double scale;
    const double *vel;      // advecting velocity (Burgers), no halo needed
    double *out0;
    int acc;
    long long nlines;       // nx*ny = row stride
    int kmax;
    double c2_1;                    // first-derivative stencil  f = (u+1 - u-1) + c2 (u+2 - u-2)
    double c0_2, c2_2, c3_2;        // second-derivative stencil f = c0 u + (u+1 + u-1) + c2 (u+2 + u-2) + c3 (u+3 + u-3)
    ZSysDev y1, y2;
    double nu;
    double *head, *tail;                    // phase A out: [nsys][nlines]   (Burgers with nf fields: [nf][2][nlines])
    const double *tail_left, *head_right;   // phase B in:  [nsys][nlines]
    // Burgers: nf transported fields share the advecting velocity (one launch; the workgroups of a tile sit on one XCD, see k_htile)
    int nf;
    const double *fs[4];
    double *fo[4];
    double fnu[4];
    // Burgers phase B: ffin[f] != 0: the tendency of field f is complete with this term -> wall planes, Runge-Kutta update of the operand in
    // place, scaling (k_final_update's arithmetic with Dirichlet walls; the scalars of the slab driver, whose z term comes last)
    int ffin[4];
    // MODE_P1 phase B "final update" epilogue (fq != NULL), as in k_xline / k_rtile: out0 = tendency h, hv = h - d/dz, walls, q += dte hv
    double *fq;
    double fdte, fkco;
    int fscale, fnx, fny;
    int dual;               // Burgers: both systems in one pass (z_solve2); 0 (TLAB_ZSLAB_DUAL=0): one after the other
    int early;              // ... and the old tendencies requested ahead of the solves
};

// local solve of the slab system (sub-chunks through LDS), then phase handling. f: RHS in, y (phase A) / x (phase B) out
template <int M, int PHASE>
__device__ __forceinline__ void z_solve(double (&f)[M], const ZSysDev &sy, int kmax, int w, int C, int lane, bool valid, long long line,
                                        long long nlines, int isys, const ZSlabArgs &a, double *s_yl, double *s_r, double *s_x) {
    const int row0 = w * M;
    double fS = 0.0;
    if (w == 0) { fS = f[0]; f[0] = 0.0; }
    const double *Lm = sy.rowtab + row0, *Di = sy.rowtab + kmax + row0, *Cm = sy.rowtab + 2 * kmax + row0;
    const double *Vt = sy.rowtab + 3 * kmax + row0, *Wt = sy.rowtab + 4 * kmax + row0;
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
    s_yl[w * 64 + lane] = f[M - 1];
    __syncthreads();
    const int wm = (w + C - 1) % C, wp = (w + 1) % C;
    const double yLprev = s_yl[wm * 64 + lane];
    s_r[w * 64 + lane] = f[0] - Lm[0] * yLprev - Cm[0] * f[1];
    __syncthreads();
    double X = 0.0, Xr = 0.0;
    for (int q = 0; q < C; ++q) {
        const double rq = s_r[q * 64 + lane];
        X += sy.ginv[w * C + q] * rq;
        Xr += sy.ginv[wp * C + q] * rq;
    }
    f[0] = X;
#pragma unroll
    for (int p = 1; p < M; ++p) f[p] = f[p] + Vt[p] * X + Wt[p] * Xr;
    // f = y now.  Slab-level interface
    if (PHASE == 1) {
        if (valid) {
            if (w == 0) a.head[(long long)isys * nlines + line] = fS - sy.cS * f[1];
            if (w == C - 1) a.tail[(long long)isys * nlines + line] = f[M - 1];
        }
        __syncthreads();   // s_yl / s_r are reused by the next system
    } else {
        if (w == 0) s_x[lane] = fS - sy.cS * f[1];
        if (w == C - 1) s_x[64 + lane] = f[M - 1];
        __syncthreads();
        const double tl = valid ? a.tail_left[(long long)isys * nlines + line] : 0.0;
        const double hr = valid ? a.head_right[(long long)isys * nlines + line] : 0.0;
        const double XS = (s_x[lane] - sy.aS * tl) * sy.binv;
        const double XR = (hr - sy.aSn * s_x[64 + lane]) * sy.binvn;
        const double *Vr = sy.vw + row0, *Wr = sy.vw + kmax + row0;
#pragma unroll
        for (int p = 0; p < M; ++p) f[p] = f[p] + Vr[p] * XS + Wr[p] * XR;
        __syncthreads();
    }
}

// Both systems of the Burgers operator in one pass (same arithmetic per system as z_solve, to the bit): the two recurrences of a chunk are
// independent, so their dependent chains interleave and every barrier serves both -- three barriers per phase instead of six.
template <int M, int PHASE>
__device__ __forceinline__ void z_solve2(double (&f)[M], double (&g2)[M], const ZSysDev &sa, const ZSysDev &sb, int kmax, int w, int C, int lane, bool valid,
                                         long long line, long long nlines, int isys, const ZSlabArgs &a, double *s_yl, double *s_r, double *s_x) {
    const int row0 = w * M;
    double fSa = 0.0, fSb = 0.0;
    if (w == 0) { fSa = f[0]; f[0] = 0.0; fSb = g2[0]; g2[0] = 0.0; }
    const double *La = sa.rowtab + row0, *Da = sa.rowtab + kmax + row0, *Ca = sa.rowtab + 2 * kmax + row0;
    const double *Va = sa.rowtab + 3 * kmax + row0, *Wa = sa.rowtab + 4 * kmax + row0;
    const double *Lb = sb.rowtab + row0, *Db = sb.rowtab + kmax + row0, *Cb = sb.rowtab + 2 * kmax + row0;
    const double *Vb = sb.rowtab + 3 * kmax + row0, *Wb = sb.rowtab + 4 * kmax + row0;
    double ga = 0.0, gb = 0.0;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        ga = f[p] + La[p] * ga;
        gb = g2[p] + Lb[p] * gb;
        f[p] = ga;
        g2[p] = gb;
    }
    double ya = 0.0, yb = 0.0;
#pragma unroll
    for (int p = M - 1; p >= 1; --p) {
        ya = f[p] * Da[p] + Ca[p] * ya;
        yb = g2[p] * Db[p] + Cb[p] * yb;
        f[p] = ya;
        g2[p] = yb;
    }
    double *s_yl2 = s_yl + 8 * 64, *s_r2 = s_r + 8 * 64, *s_x2 = s_x + 2 * 64;
    s_yl[w * 64 + lane] = f[M - 1];
    s_yl2[w * 64 + lane] = g2[M - 1];
    __syncthreads();
    const int wm = (w + C - 1) % C, wp = (w + 1) % C;
    s_r[w * 64 + lane] = f[0] - La[0] * s_yl[wm * 64 + lane] - Ca[0] * f[1];
    s_r2[w * 64 + lane] = g2[0] - Lb[0] * s_yl2[wm * 64 + lane] - Cb[0] * g2[1];
    __syncthreads();
    double Xa = 0.0, Xra = 0.0, Xb = 0.0, Xrb = 0.0;
    for (int q = 0; q < C; ++q) {
        const double ra = s_r[q * 64 + lane], rb = s_r2[q * 64 + lane];
        Xa += sa.ginv[w * C + q] * ra;
        Xra += sa.ginv[wp * C + q] * ra;
        Xb += sb.ginv[w * C + q] * rb;
        Xrb += sb.ginv[wp * C + q] * rb;
    }
    f[0] = Xa;
    g2[0] = Xb;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        f[p] = f[p] + Va[p] * Xa + Wa[p] * Xra;
        g2[p] = g2[p] + Vb[p] * Xb + Wb[p] * Xrb;
    }
    if (PHASE == 1) {
        if (valid) {
            if (w == 0) {
                a.head[(long long)isys * nlines + line] = fSa - sa.cS * f[1];
                a.head[(long long)(isys + 1) * nlines + line] = fSb - sb.cS * g2[1];
            }
            if (w == C - 1) {
                a.tail[(long long)isys * nlines + line] = f[M - 1];
                a.tail[(long long)(isys + 1) * nlines + line] = g2[M - 1];
            }
        }
    } else {
        if (w == 0) { s_x[lane] = fSa - sa.cS * f[1]; s_x2[lane] = fSb - sb.cS * g2[1]; }
        if (w == C - 1) { s_x[64 + lane] = f[M - 1]; s_x2[64 + lane] = g2[M - 1]; }
        __syncthreads();
        const double tla = valid ? a.tail_left[(long long)isys * nlines + line] : 0.0, tlb = valid ? a.tail_left[(long long)(isys + 1) * nlines + line] : 0.0;
        const double hra = valid ? a.head_right[(long long)isys * nlines + line] : 0.0, hrb = valid ? a.head_right[(long long)(isys + 1) * nlines + line] : 0.0;
        const double XSa = (s_x[lane] - sa.aS * tla) * sa.binv, XRa = (hra - sa.aSn * s_x[64 + lane]) * sa.binvn;
        const double XSb = (s_x2[lane] - sb.aS * tlb) * sb.binv, XRb = (hrb - sb.aSn * s_x2[64 + lane]) * sb.binvn;
        const double *Vra = sa.vw + row0, *Wra = sa.vw + kmax + row0, *Vrb = sb.vw + row0, *Wrb = sb.vw + kmax + row0;
#pragma unroll
        for (int p = 0; p < M; ++p) {
            f[p] = f[p] + Vra[p] * XSa + Wra[p] * XRa;
            g2[p] = g2[p] + Vrb[p] * XSb + Wrb[p] * XRb;
        }
    }
}

template <int M, int MODE, int PHASE, bool DUAL = false>
__global__ void __launch_bounds__(512) k_zslab(ZSlabArgs a) {
    __shared__ double s_yl[(DUAL ? 2 : 1) * 8 * 64];      // DUAL: the second system's rows behind the first's (z_solve2)
    __shared__ double s_r[(DUAL ? 2 : 1) * 8 * 64];
    __shared__ double s_x[(DUAL ? 2 : 1) * 2 * 64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: coefficient rows become scalar loads
    const int C = blockDim.x >> 6;
    long long tile = blockIdx.x;
    int fi = 0;
    if (MODE == MODE_BURGERS) {           // bid = x + 8 (f + nf y), tile = x + 8 y
        const long long q = blockIdx.x >> 3;
        fi = (int)(q % a.nf);
        tile = (blockIdx.x & 7) + 8 * (q / a.nf);
        if (tile * 64 >= a.nlines) return;          // whole workgroup, before any barrier
    }
    const long long line = tile * 64 + lane;
    const bool valid = line < a.nlines;
    const long long rs = a.nlines;
    const int row0 = w * M;
    const long long base = valid ? line : 0;
    const double *__restrict__ in0 = (MODE == MODE_BURGERS) ? a.fs[fi] : a.in0;
    double *__restrict__ out0 = (MODE == MODE_BURGERS) ? a.fo[fi] : a.out0;
    const double nu = (MODE == MODE_BURGERS) ? a.fnu[fi] : a.nu;
    const int msg0 = (MODE == MODE_BURGERS) ? 2 * fi : 0;       // first message row of this field

    // operand rows + 3-row halos; no wrap: the rows before / after the slab are the neighbours' planes (wave-uniform choice of the base pointer)
    const double *__restrict__ lo = (MODE == MODE_BURGERS) ? a.flo[fi] : a.lo0;
    const double *__restrict__ hi = (MODE == MODE_BURGERS) ? a.fhi[fi] : a.hi0;
    // Row addresses: one pointer per array, advanced by the plane stride from row to row (every (row0 + p) * rs as a product of its own was a
    // third of the instructions of this kernel and put it into 275 spilled SGPRs); only the three rows before the first chunk and after the last
    // one can lie in the neighbours' planes.
    const bool first_chunk = row0 == 0, last_chunk = row0 + M == a.kmax;
    auto load_rows = [&](const double *in, const double *l, const double *h, double (&v)[M + 6]) {
        // (wave-uniform row pointers, the lane's line as the index: scalar base + vector offset addressing)
        const double *pin = in + (long long)row0 * rs;                   // row row0
        const double *pl = first_chunk ? l : pin - 3 * rs;               // row row0 - 3
        const double *ph = last_chunk ? h : pin + (long long)M * rs;     // row row0 + M
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = valid ? pl[base] : 0.0; pl += rs; }
#pragma unroll
        for (int p = 0; p < M; ++p) { v[p + 3] = valid ? pin[base] : 0.0; pin += rs; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[M + 3 + k] = valid ? ph[base] : 0.0; ph += rs; }
    };
    double e[M + 6];
    load_rows(in0, lo, hi, e);
    if (MODE == MODE_P1 && a.in0b != nullptr) {
        double eb[M + 6];
        load_rows(a.in0b, a.lo0b, a.hi0b, eb);
#pragma unroll
        for (int p = 0; p < M + 6; ++p) e[p] = e[p] + eb[p] * a.scale;
    }
    double x1[M], x2[MODE == MODE_BURGERS ? M : 1];
#pragma unroll
    for (int p = 0; p < M; ++p) x1[p] = e[p + 4] - e[p + 2] + a.c2_1 * (e[p + 5] - e[p + 1]);
    if constexpr (MODE == MODE_BURGERS) {
#pragma unroll
        for (int p = 0; p < M; ++p)
            x2[p] = a.c0_2 * e[p + 3] + e[p + 4] + e[p + 2] + a.c2_2 * (e[p + 5] + e[p + 1]) + a.c3_2 * (e[p + 6] + e[p]);
    }
    double vl[(MODE == MODE_BURGERS && PHASE == 2) ? M : 1];
    if constexpr (MODE == MODE_BURGERS && PHASE == 2) {   // issued before the solves: its latency hides behind them
        const double *pv = a.vel + (long long)row0 * rs;
#pragma unroll
        for (int p = 0; p < M; ++p) { vl[p] = valid ? pv[base] : 0.0; pv += rs; }
    }
    // DUAL phase B: the old tendencies are asked for before the solves too (TLAB_ZSLAB_EARLY=0: after them, as the one-system form does)
    double oe[(MODE == MODE_BURGERS && PHASE == 2 && DUAL) ? M : 1];
    bool early = false;
    if constexpr (MODE == MODE_BURGERS && PHASE == 2 && DUAL) {
        early = a.acc && a.early;
        if (early) {
            const double *po = out0 + (long long)row0 * rs;
#pragma unroll
            for (int p = 0; p < M; ++p) { oe[p] = valid ? __builtin_nontemporal_load(po + base) : 0.0; po += rs; }
        }
    }
    if constexpr (MODE == MODE_BURGERS) {
        if constexpr (DUAL) z_solve2<M, PHASE>(x1, x2, a.y1, a.y2, a.kmax, w, C, lane, valid, line, a.nlines, msg0, a, s_yl, s_r, s_x);
        else {
            z_solve<M, PHASE>(x1, a.y1, a.kmax, w, C, lane, valid, line, a.nlines, msg0, a, s_yl, s_r, s_x);
            z_solve<M, PHASE>(x2, a.y2, a.kmax, w, C, lane, valid, line, a.nlines, msg0 + 1, a, s_yl, s_r, s_x);
        }
    } else {
        z_solve<M, PHASE>(x1, a.y1, a.kmax, w, C, lane, valid, line, a.nlines, msg0, a, s_yl, s_r, s_x);
    }
    if constexpr (PHASE == 2) {
        if (!valid) return;
        if constexpr (MODE == MODE_BURGERS) {
#pragma unroll
            for (int p = 0; p < M; ++p) x1[p] = nu * x2[p] - vl[p] * x1[p];     // opr_burgers.f90:513
        }
        if (MODE == MODE_P1 && a.fq != nullptr) {      // final-update epilogue (lane = (ix, j) in the plane, rows = k)
            const int j = (int)((line / a.fnx) % a.fny);
            const bool wall = (j == 0) || (j == a.fny - 1);
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {
                double h[8], qv[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    h[p] = out0[base + (long long)(row0 + p0 + p) * rs];
                    qv[p] = a.fq[base + (long long)(row0 + p0 + p) * rs];
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const double hv = wall ? 0.0 : h[p] - x1[p0 + p];
                    qv[p] = qv[p] + a.fdte * hv;
                    h[p] = a.fscale ? a.fkco * hv : hv;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    a.fq[base + (long long)(row0 + p0 + p) * rs] = qv[p];
                    out0[base + (long long)(row0 + p0 + p) * rs] = h[p];
                }
            }
            return;
        }
        if (MODE == MODE_BURGERS) {     // tendencies: read once, written once -> non-temporal (as in k_htile; 2 % of the launch)
            double *const po0 = out0 + (long long)row0 * rs;
            if (early) {
                if constexpr (DUAL) {
#pragma unroll
                    for (int p = 0; p < M; ++p) x1[p] = oe[p] + x1[p];
                }
            } else if (a.acc) {
                double o[M];
                const double *po = po0;
#pragma unroll
                for (int p = 0; p < M; ++p) { o[p] = __builtin_nontemporal_load(po + base); po += rs; }
#pragma unroll
                for (int p = 0; p < M; ++p) x1[p] = o[p] + x1[p];
            }
            if (a.ffin[fi]) {
                const int j = (int)((line / a.fnx) % a.fny);
                const bool wall = (j == 0) || (j == a.fny - 1);
                // (the operand rows are read again here, 8 at a time: keeping e[] alive through both solves costs the kernel its occupancy --
                // measured 0.54 against 0.38 ms per launch at 512 x 512 x 64)
                double *qo = const_cast<double *>(in0) + (long long)row0 * rs;
#pragma unroll
                for (int p0 = 0; p0 < M; p0 += 8) {
                    double qv[8];
                    double *qr = qo;
#pragma unroll
                    for (int p = 0; p < 8; ++p) { qv[p] = qr[base]; qr += rs; }
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const double hv = wall ? 0.0 : x1[p0 + p];
                        qo[base] = qv[p] + a.fdte * hv; qo += rs;
                        x1[p0 + p] = a.fscale ? a.fkco * hv : hv;
                    }
                }
            }
            {
                double *po = po0;
#pragma unroll
                for (int p = 0; p < M; ++p) { __builtin_nontemporal_store(x1[p], po + base); po += rs; }
            }
            return;
        }
        if (a.acc) {
            double o[M];
#pragma unroll
            for (int p = 0; p < M; ++p) o[p] = out0[base + (long long)(row0 + p) * rs];
#pragma unroll
            for (int p = 0; p < M; ++p) x1[p] = o[p] + x1[p];
        }
#pragma unroll
        for (int p = 0; p < M; ++p) out0[base + (long long)(row0 + p) * rs] = x1[p];
    }
}

struct ZSysHost {
    DeviceArray rowtab, ginv, vw;
    double cS = 0, aS = 0, binv = 0, aSn = 0, binvn = 0;
    ZSysDev dev() const { return ZSysDev{rowtab.p, ginv.p, vw.p, cS, aS, binv, aSn, binvn}; }
};

}  // namespace tlab

using namespace tlab;

struct tlab_zslab_plan {
    int nz = 0, kmax = 0, k0 = 0, M = 0, C = 0;
    double c2_1 = 0, c0_2 = 0, c2_2 = 0, c3_2 = 0;
    ZSysHost sys[2];
};

namespace {

struct Fail : std::runtime_error {
    int code;
    Fail(int c, const std::string &s) : std::runtime_error(s), code(c) {}
};

typedef long double ld;

// slab-level spikes of the slab that starts at global row k0: V = T_loc^-1 e_0, W = T_loc^-1 (-c_last e_last)
void slab_spikes(const TriDiag &G, int k0, int kmax, TriDiag &Tloc, std::vector<double> &V, std::vector<double> &W) {
    const int nz = G.n;
    Tloc.n = kmax;
    Tloc.periodic = false;
    Tloc.a.assign(kmax, 0.0); Tloc.b.assign(kmax, 0.0); Tloc.c.assign(kmax, 0.0);
    Tloc.b[0] = 1.0;
    for (int i = 1; i < kmax; ++i) {
        const int g = (k0 + i) % nz;
        Tloc.a[i] = G.a[g]; Tloc.b[i] = G.b[g]; Tloc.c[i] = (i < kmax - 1) ? G.c[g] : 0.0;
    }
    V.assign(kmax, 0.0); W.assign(kmax, 0.0);
    V[0] = 1.0;
    W[kmax - 1] = -G.c[(k0 + kmax - 1) % nz];
    tridiag_solve_direct(Tloc, V.data());
    tridiag_solve_direct(Tloc, W.data());
}

void build_system(tlab_zslab_plan &P, const TriDiag &G, ZSysHost &out) {
    const int nz = G.n, kmax = P.kmax, k0 = P.k0;
    const int kprev = ((k0 - kmax) % nz + nz) % nz, knext = (k0 + kmax) % nz;
    TriDiag Tm, Tp, Tn;
    std::vector<double> Vm, Wm, Vp, Wp, Vn, Wn;
    slab_spikes(G, k0, kmax, Tm, Vm, Wm);
    slab_spikes(G, kprev, kmax, Tp, Vp, Wp);
    slab_spikes(G, knext, kmax, Tn, Vn, Wn);
    // interface rows: alpha X_{r-1} + beta X_r + gamma X_{r+1}
    const double aS = G.a[k0], bS = G.b[k0], cS = G.c[k0];
    const double beta = bS + aS * Wp[kmax - 1] + cS * Vm[1];
    const double alpha = aS * Vp[kmax - 1], gamma = cS * Wm[1];
    const double aSn = G.a[knext], bSn = G.b[knext], cSn = G.c[knext];
    const double betan = bSn + aSn * Wm[kmax - 1] + cSn * Vn[1];
    const double alphan = aSn * Vm[kmax - 1], gamman = cSn * Wn[1];
    const double tol = 1e-19;
    if (std::fabs(alpha) > tol * std::fabs(beta) || std::fabs(gamma) > tol * std::fabs(beta) || std::fabs(alphan) > tol * std::fabs(betan) ||
        std::fabs(gamman) > tol * std::fabs(betan))
    {
        char buf[64];
        snprintf(buf, sizeof(buf), "%.2e", std::fabs(alpha / beta));
        throw Fail(TLAB_EUNSUPPORTED, std::string("z-slab operators: slab too thin, the coupling between slab separators (") + buf +
                                          ") is not below 1e-19; use the K-transpose path");
    }
    ChunkedTables ct;
    build_chunked(Tm, P.C, ct);
    if ((int)ct.ginv.size() != P.C * P.C) throw Fail(TLAB_EINVAL, "internal: dense separator inverse missing");
    std::vector<double> rowtab((size_t)5 * kmax);
    std::copy(ct.Lm.begin(), ct.Lm.end(), rowtab.begin());
    std::copy(ct.Dinv.begin(), ct.Dinv.end(), rowtab.begin() + kmax);
    std::copy(ct.Cm.begin(), ct.Cm.end(), rowtab.begin() + 2 * kmax);
    std::copy(ct.V.begin(), ct.V.end(), rowtab.begin() + 3 * kmax);
    std::copy(ct.W.begin(), ct.W.end(), rowtab.begin() + 4 * kmax);
    out.rowtab.upload(rowtab);
    out.ginv.upload(ct.ginv);
    std::vector<double> vw((size_t)2 * kmax);
    std::copy(Vm.begin(), Vm.end(), vw.begin());
    std::copy(Wm.begin(), Wm.end(), vw.begin() + kmax);
    out.vw.upload(vw);
    out.cS = cS; out.aS = aS; out.binv = (double)((ld)1 / (ld)beta);
    out.aSn = aSn; out.binvn = (double)((ld)1 / (ld)betan);
}

template <class F>
int guard(F &&f) {
    try {
        if (!tlab_device_ready()) throw Fail(TLAB_EHIP, "tlab_init has not been called (no CPU fallback exists)");
        f();
        return TLAB_OK;
    } catch (const Fail &e) {
        tlab_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        tlab_set_error(e.what());
        return TLAB_EINVAL;
    }
}

template <int M, int MODE>
void launch_m(int phase, int C, const ZSlabArgs &a, hipStream_t st) {
    const long long tiles = (a.nlines + 63) / 64;
    const long long nwg = (MODE == MODE_BURGERS) ? 8LL * a.nf * ((tiles + 7) / 8) : tiles;
    const dim3 grid((unsigned)nwg), block(64 * C);
    if (MODE == MODE_BURGERS && a.dual) {
        if (phase == 1) hipLaunchKernelGGL((k_zslab<M, MODE, 1, MODE == MODE_BURGERS>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_zslab<M, MODE, 2, MODE == MODE_BURGERS>), grid, block, 0, st, a);
        return;
    }
    if (phase == 1) hipLaunchKernelGGL((k_zslab<M, MODE, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_zslab<M, MODE, 2>), grid, block, 0, st, a);
}

void launch(const tlab_zslab_plan &P, int mode, int phase, const ZSlabArgs &a_in, hipStream_t st) {
    ZSlabArgs a = a_in;
    const long long rs3 = 3 * a.nlines, rsk = (long long)a.kmax * a.nlines;       // halos in place where the caller gave none
    if (a.in0 && !a.lo0) { a.lo0 = a.in0 - rs3; a.hi0 = a.in0 + rsk; }
    if (a.in0b && !a.lo0b) { a.lo0b = a.in0b - rs3; a.hi0b = a.in0b + rsk; }
    for (int f = 0; f < 4; ++f)
        if (a.fs[f] && !a.flo[f]) { a.flo[f] = a.fs[f] - rs3; a.fhi[f] = a.fs[f] + rsk; }
    const double pts = (double)a.nlines * a.kmax;
    const char *name = mode == MODE_P1 ? (phase == 1 ? "k_zslab<P1,A>" : "k_zslab<P1,B>") : (phase == 1 ? "k_zslab<BURGERS,A>" : "k_zslab<BURGERS,B>");
    double bpp = 8.0 * ((a.in0b && mode == MODE_P1) ? 2 : 1);
    if (phase == 2) bpp += 8.0 + (a.acc ? 8.0 : 0.0) + (a.fq ? 24.0 : 0.0);
    if (mode == MODE_BURGERS) {        // per field: operand (+ result, old result in phase B); the velocity once in phase B
        bpp = 0.0;
        for (int f = 0; f < a.nf; ++f) bpp += 8.0 + (phase == 2 ? 8.0 + (a.acc ? 8.0 : 0.0) : 0.0);
        if (phase == 2) bpp += 8.0;
    }
    ProfScope ps(name, st, pts * bpp);
    if (P.M == 32) {
        if (mode == MODE_P1) launch_m<32, MODE_P1>(phase, P.C, a, st);
        else launch_m<32, MODE_BURGERS>(phase, P.C, a, st);
    } else {
        if (mode == MODE_P1) launch_m<16, MODE_P1>(phase, P.C, a, st);
        else launch_m<16, MODE_BURGERS>(phase, P.C, a, st);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Fail(TLAB_EHIP, std::string("k_zslab launch: ") + hipGetErrorString(e));
}

ZSlabArgs base_args(const tlab_zslab_plan &P, int nx, int ny) {
    ZSlabArgs a{};
    a.nlines = (long long)nx * ny;
    a.kmax = P.kmax;
    a.c2_1 = P.c2_1; a.c0_2 = P.c0_2; a.c2_2 = P.c2_2; a.c3_2 = P.c3_2;
    a.y1 = P.sys[0].dev();
    a.y2 = P.sys[1].dev();
    a.nf = 1;
    static const int dual = [] { const char *e = getenv("TLAB_ZSLAB_DUAL"); return e ? atoi(e) : 1; }();
    a.dual = dual;
    static const int early = [] { const char *e = getenv("TLAB_ZSLAB_EARLY"); return e ? atoi(e) : 1; }();
    a.early = early;
    return a;
}

}  // namespace

int tlab_internal_zslab_partial_z(tlab_zslab_plan_t P, int phase, int nx, int ny, const double *u, const double *const *u_halo, const double *ub,
                                  const double *const *ub_halo, double scale, double *head, double *tail, const double *tail_left,
                                  const double *head_right, double *result, int acc);
int tlab_internal_zslab_burgers_z_n(tlab_zslab_plan_t P, int phase, int nx, int ny, int nf, const double *nu, const double *const *s,
                                    const double *const *s_lo, const double *const *s_hi, const double *vel, double *head, double *tail,
                                    const double *tail_left, const double *head_right, double *const *result, int acc, const int *fin, double dte,
                                    double kco, int scale);
int tlab_internal_zslab_gradient_final_z(tlab_zslab_plan_t P, int nx, int ny, const double *p, const double *const *p_halo, const double *tail_left,
                                         const double *head_right, double *q, double *h, double dte, double kco, int scale);

extern "C" {

int tlab_zslab_plan_create(tlab_zslab_plan_t *out, tlab_fdm_plan_t gz, int kmax, int koffset, int chunk) {
    return guard([&] {
        if (!out || !gz) throw Fail(TLAB_EINVAL, "tlab_zslab_plan_create: null argument");
        const int nz = gz->t.n;
        if (!gz->t.periodic) throw Fail(TLAB_EUNSUPPORTED, "z-slab operators: the decomposed direction must be periodic");
        if (kmax <= 0 || nz % kmax || koffset < 0 || koffset % kmax || koffset >= nz) throw Fail(TLAB_EINVAL, "tlab_zslab_plan_create: bad slab");
        auto P = std::make_unique<tlab_zslab_plan>();
        P->nz = nz; P->kmax = kmax; P->k0 = koffset;
        int M = chunk;
        if (M == 0) {
            static const int forced = [] { const char *e = getenv("TLAB_ZSLAB_M"); return e ? atoi(e) : 0; }();      // 16 / 32: experiments
            M = (kmax % 32 == 0 && kmax / 32 <= 8) ? 32 : 16;
            // two sub-chunks of 32 rows are two waves per workgroup at 256 registers (20 spilled in the Burgers phase B): slabs of 64 planes -- the
            // 512^3 box on 8 ranks -- take four of 16 rows (phase B 0.452 -> 0.407 ms per rank, phase A 0.157 -> 0.175); at 128 planes 32 rows win
            if (M == 32 && kmax / 32 <= 2 && kmax % 16 == 0) M = 16;
            if ((forced == 16 || forced == 32) && kmax % forced == 0 && kmax / forced <= 8) M = forced;
        }
        if ((M != 16 && M != 32) || kmax % M || kmax / M > 8 || kmax / M < 1)
            throw Fail(TLAB_EUNSUPPORTED, "z-slab operators: kmax must be a multiple of 16 or 32 with at most 8 sub-chunks");
        P->M = M; P->C = kmax / M;
        const StencilDev s1 = gz->stencil(1, 0), s2 = gz->stencil(2, 0);
        P->c2_1 = s1.c2;
        P->c0_2 = s2.c0; P->c2_2 = s2.c2; P->c3_2 = s2.c3;
        build_system(*P, gz->tridiag(1, 0), P->sys[0]);
        build_system(*P, gz->tridiag(2, 0), P->sys[1]);
        *out = P.release();
    });
}

int tlab_zslab_plan_destroy(tlab_zslab_plan_t p) {
    delete p;
    return TLAB_OK;
}

int tlab_zslab_partial_z(tlab_zslab_plan_t P, int phase, int nx, int ny, const double *u, const double *ub, double scale, double *head,
                         double *tail, const double *tail_left, const double *head_right, double *result, int acc) {
    return tlab_internal_zslab_partial_z(P, phase, nx, ny, u, nullptr, ub, nullptr, scale, head, tail, tail_left, head_right, result, acc);
}
}  // extern "C"

// the same with the halo planes of the operands in buffers of their own: halos = {lo, hi} of 3 planes each (nullptr: in place)
int tlab_internal_zslab_partial_z(tlab_zslab_plan_t P, int phase, int nx, int ny, const double *u, const double *const *u_halo, const double *ub,
                                  const double *const *ub_halo, double scale, double *head, double *tail, const double *tail_left,
                                  const double *head_right, double *result, int acc) {
    return guard([&] {
        if (!P || !u || nx < 1 || ny < 1 || (phase != 1 && phase != 2)) throw Fail(TLAB_EINVAL, "tlab_zslab_partial_z: bad arguments");
        ZSlabArgs a = base_args(*P, nx, ny);
        a.in0 = u; a.in0b = ub; a.scale = scale;
        if (u_halo) { a.lo0 = u_halo[0]; a.hi0 = u_halo[1]; }
        if (ub && ub_halo) { a.lo0b = ub_halo[0]; a.hi0b = ub_halo[1]; }
        if (phase == 1) {
            if (!head || !tail) throw Fail(TLAB_EINVAL, "tlab_zslab_partial_z: phase A needs head and tail");
            a.head = head; a.tail = tail;
        } else {
            if (!tail_left || !head_right || !result || result == u || result == ub) throw Fail(TLAB_EINVAL, "tlab_zslab_partial_z: phase B arguments");
            a.tail_left = tail_left; a.head_right = head_right; a.out0 = result; a.acc = acc;
        }
        launch(*P, MODE_P1, phase, a, tlab_current_stream());
    });
}

extern "C" {

int tlab_zslab_burgers_z(tlab_zslab_plan_t P, int phase, int nx, int ny, double nu, const double *s, const double *vel, double *head,
                         double *tail, const double *tail_left, const double *head_right, double *result, int acc) {
    return guard([&] {
        if (!P || !s || nx < 1 || ny < 1 || (phase != 1 && phase != 2)) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z: bad arguments");
        ZSlabArgs a = base_args(*P, nx, ny);
        a.in0 = s; a.nu = nu;
        a.nf = 1; a.fs[0] = s; a.fo[0] = result; a.fnu[0] = nu;
        if (phase == 1) {
            if (!head || !tail) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z: phase A needs head and tail");
            a.head = head; a.tail = tail;
        } else {
            if (!tail_left || !head_right || !result || !vel || result == s || result == vel) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z: phase B arguments");
            a.vel = vel; a.tail_left = tail_left; a.head_right = head_right; a.out0 = result; a.acc = acc;
        }
        launch(*P, MODE_BURGERS, phase, a, tlab_current_stream());
    });
}

int tlab_zslab_burgers_z_n(tlab_zslab_plan_t P, int phase, int nx, int ny, int nf, const double *nu, const double *const *s, const double *vel,
                           double *head, double *tail, const double *tail_left, const double *head_right, double *const *result, int acc) {
    return tlab_internal_zslab_burgers_z_n(P, phase, nx, ny, nf, nu, s, nullptr, nullptr, vel, head, tail, tail_left, head_right, result, acc, nullptr,
                                           0.0, 1.0, 0);
}
}  // extern "C"

int tlab_internal_zslab_burgers_z_n(tlab_zslab_plan_t P, int phase, int nx, int ny, int nf, const double *nu, const double *const *s,
                                    const double *const *s_lo, const double *const *s_hi, const double *vel, double *head, double *tail,
                                    const double *tail_left, const double *head_right, double *const *result, int acc, const int *fin, double dte,
                                    double kco, int scale) {
    return guard([&] {
        if (!P || !s || !nu || nf < 1 || nf > 4 || nx < 1 || ny < 1 || (phase != 1 && phase != 2)) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: bad arguments");
        ZSlabArgs a = base_args(*P, nx, ny);
        a.nf = nf;
        for (int f = 0; f < nf; ++f) {
            if (!s[f]) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: null operand");
            a.fs[f] = s[f]; a.fnu[f] = nu[f]; a.fo[f] = nullptr;
            if (s_lo && s_hi) { a.flo[f] = s_lo[f]; a.fhi[f] = s_hi[f]; }
        }
        if (phase == 1) {
            if (!head || !tail) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: phase A needs head and tail");
            a.head = head; a.tail = tail;
        } else {
            if (!tail_left || !head_right || !result || !vel) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: phase B arguments");
            for (int f = 0; f < nf; ++f) {
                if (!result[f] || result[f] == s[f] || result[f] == vel) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: null or aliased result");
                a.fo[f] = result[f];
            }
            a.vel = vel; a.tail_left = tail_left; a.head_right = head_right; a.acc = acc;
            for (int f = 0; f < nf; ++f) {
                a.ffin[f] = fin ? fin[f] : 0;
                if (a.ffin[f] && s[f] == vel) throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: the advecting velocity cannot be updated in place");
                // the in-place update of operand f is only ordered against the kernel's own reads of THAT field (the barriers of z_solve): it must not be
                // any other launch operand or result
                if (a.ffin[f])
                    for (int g2 = 0; g2 < nf; ++g2)
                        if ((g2 != f && s[g2] == s[f]) || result[g2] == s[f])
                            throw Fail(TLAB_EINVAL, "tlab_zslab_burgers_z_n: a field finished in place appears twice among the operands / results");
            }
            a.fdte = dte; a.fkco = kco; a.fscale = scale; a.fnx = nx; a.fny = ny;
        }
        launch(*P, MODE_BURGERS, phase, a, tlab_current_stream());
    });
}

extern "C" {

// phase 2 of d/dz p with the final update of w as its epilogue: h -= dp/dz; wall planes (Dirichlet); q += dte h; h *= kco
int tlab_zslab_gradient_final_z(tlab_zslab_plan_t P, int nx, int ny, const double *p, const double *tail_left, const double *head_right, double *q,
                                double *h, double dte, double kco, int scale) {
    return tlab_internal_zslab_gradient_final_z(P, nx, ny, p, nullptr, tail_left, head_right, q, h, dte, kco, scale);
}
}  // extern "C"

int tlab_internal_zslab_gradient_final_z(tlab_zslab_plan_t P, int nx, int ny, const double *p, const double *const *p_halo, const double *tail_left,
                                         const double *head_right, double *q, double *h, double dte, double kco, int scale) {
    return guard([&] {
        if (!P || !p || !tail_left || !head_right || !q || !h || q == h) throw Fail(TLAB_EINVAL, "tlab_zslab_gradient_final_z: bad arguments");
        ZSlabArgs a = base_args(*P, nx, ny);
        a.in0 = p; a.tail_left = tail_left; a.head_right = head_right; a.out0 = h;
        if (p_halo) { a.lo0 = p_halo[0]; a.hi0 = p_halo[1]; }
        a.fq = q; a.fdte = dte; a.fkco = kco; a.fscale = scale; a.fnx = nx; a.fny = ny;
        launch(*P, MODE_P1, 2, a, tlab_current_stream());
    });
}
