// HIP kernels (gfx950 / CDNA4, wave64) for Tlab's compact-FDM derivative operators.
//
// All kernels solve  A x = B u  per grid line, A tridiagonal (cyclic when periodic), B banded, with the
// chunked factorization of chunked.hpp so that a line is read once and written once:
//
//  k_xline  : derivative along the contiguous index (OPR_Partial_X / OPR_Burgers_X).  One WAVE per line:
//             lane l holds the M = n/64 consecutive points [l*M, (l+1)*M) in registers (coalesced 16-B loads),
//             stencil halos come from the neighbouring lanes, the 64-unknown separator system is solved by
//             parallel cyclic reduction with wave shuffles.  No LDS traffic for periodic (circulant) systems.
//  k_rtile  : derivative along a strided index (Y, Z).  One WORKGROUP per tile of 64 lines x n points:
//             lanes <-> 64 memory-contiguous lines (512-B coalesced rows), wave w holds rows [w*M, (w+1)*M) in
//             registers, coefficient rows are wave-uniform scalar loads, the P x P separator system goes through LDS.
//  k_generic: any n, any direction, one thread per line, three sweeps through the output array (fallback).
//
// The reference path these replace: TLab_Transpose + MatMul_* + TRIDSS/TRIDPSS.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_tables.hpp"
#include "kernels.hpp"
#include "profile.hpp"

namespace tlab {

// --------------------------------------------------------------------------------------------
// helpers
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

template <bool SYM>
__device__ __forceinline__ double stencil_interior(const StencilDev &s, double um3, double um2, double um1, double u0,
                                                   double up1, double up2, double up3) {
    if (SYM)  // MatMul_7d_sym / 5d_sym interior, fdm_matmul.f90:608-612 / :460-463 (c3 = 0)
        return s.c0 * u0 + up1 + um1 + s.c2 * (up2 + um2) + s.c3 * (up3 + um3);
    else      // MatMul_5d_antisym / 3d_antisym interior, fdm_matmul.f90:396-398 / :190-192 (c2 = 0)
        return up1 - um1 + s.c2 * (up2 - um2);
}

__device__ __forceinline__ double dense6(const double (&c)[6], double a0, double a1, double a2, double a3, double a4, double a5) {
    return a0 * c[0] + a1 * c[1] + a2 * c[2] + a3 * c[3] + a4 * c[4] + a5 * c[5];
}

// ============================================================================================
// k_xline : WPL waves per line, lane-chunked
// ============================================================================================
// WPL = 1: one wave per line (n = 64 M), every exchange between chunks is a wave shuffle.
// WPL = 2, 4: a line of n = 64 WPL M points is spread over WPL waves of the workgroup (lane gl = 64 wl + lane owns rows [gl M, (gl + 1) M)):
//   the register footprint per lane -- and with it the field pipelining -- stays that of the 512-point kernel for lines of 1024 and 2048
//   points (configs[3], configs[4] of BASELINE.json), where 16 / 32 rows per lane left no room to request the next operand ahead of the solves.
//   Inside a wave everything stays a shuffle; what crosses a wave boundary goes through LDS with one workgroup barrier: the stencil halos of
//   the two edge lanes, the previous chunk's last row for lane 0, and ONE value from each end of the wave's block of separator unknowns --
//   the separator system is reduced in two levels (chunked.hpp: PCR of the isolated 64 x 64 block by shuffles, 2 x 2 interface systems,
//   spike correction of the lanes next to the block ends).  Five barriers per transported field instead of one per reduction step.
//   Periodic lines only.
template <int WPL>
struct XCtx {
    int lane, gl, wl;            // lane of the wave / of the line, wave of the line
    double *eb, *hb;             // WPL > 1: this line's exchange buffers in LDS, eb[2][WPL][4] (parity-double-buffered) and hb[WPL][6]
    int par;
    unsigned *bar;               // WPL > 1, several lines per workgroup: arrival counter of this line in LDS (NULL: workgroup barrier)
    unsigned gen;
};
// The WPL waves of ONE line meet; the other lines of the workgroup go on.  A workgroup barrier ties all its lines together at every exchange (five per
// transported field) although they share nothing but read-only tables: measured with the same tables on one 512-thread workgroup against two of 256
// threads, 2.79 against 3.32 TB/s.  Arrival counter in LDS (LDS operations of a CU complete in order; the counter only grows).
template <int WPL>
__device__ __forceinline__ void xline_barrier(XCtx<WPL> &c) {
    if (c.bar == nullptr) { __syncthreads(); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    c.gen += WPL;
    if (c.lane == 0) __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - c.gen) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// v of the previous lane of the line (cyclic)
template <int WPL>
__device__ __forceinline__ double xprev(XCtx<WPL> &c, double v) {
    double r = shfl_d(v, (c.lane + 63) & 63);
    if constexpr (WPL > 1) {
        double *b = c.eb + c.par * (WPL * 4);
        if (c.lane == 63) b[c.wl * 4 + 0] = v;
        xline_barrier<WPL>(c);
        if (c.lane == 0) r = b[((c.wl + WPL - 1) & (WPL - 1)) * 4 + 0];
        c.par ^= 1;
    }
    return r;
}
// 3-point halos of the chunk from the neighbouring lanes
template <int M, int WPL>
__device__ __forceinline__ void xhalo(XCtx<WPL> &c, const double (&u)[M], double (&um)[3], double (&up)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        um[k] = shfl_d(u[M - 3 + k], (c.lane + 63) & 63);
        up[k] = shfl_d(u[k], (c.lane + 1) & 63);
    }
    if constexpr (WPL > 1) {
        // the previous use of hb (the halos of the previous field / line) lies at least one solve = two barriers back
        if (c.lane == 63) {
#pragma unroll
            for (int k = 0; k < 3; ++k) c.hb[c.wl * 6 + k] = u[M - 3 + k];
        }
        if (c.lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) c.hb[c.wl * 6 + 3 + k] = u[k];
        }
        xline_barrier<WPL>(c);
        if (c.lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) um[k] = c.hb[((c.wl + WPL - 1) & (WPL - 1)) * 6 + k];
        }
        if (c.lane == 63) {
#pragma unroll
            for (int k = 0; k < 3; ++k) up[k] = c.hb[((c.wl + 1) & (WPL - 1)) * 6 + 3 + k];
        }
    }
}

// CL: the 23 constants below live in LDS (staged once per workgroup and shared by its lines) instead of 46 VGPRs per system: the
// several-waves-per-line forms then fit two waves per SIMD
// LDS layout of the constants of one system: 17 rows per lane [XRL][64 WPL] -- 0-5 k1, 6-11 k2, 12 dinv, 13 vs, 14 ws, 15 a_s, 16 c_s -- followed by the six
// constants of the 2 x 2 interface systems, which are the same for all lanes of a wave: [6][WPL] -- wL, vF, dn, wLp, vFm, dp (chunked.cpp)
constexpr int XRL = 17;
// LDS pointers carry their address space in the type: the opaque copies below (asm) would otherwise degrade the generic pointers to flat loads
typedef __attribute__((address_space(3))) const double ldsd_t;
typedef __attribute__((address_space(3))) const float ldsf_t;
template <int WPL, bool CL = false>
struct XSys {                    // per-lane view of one chunked system
    const double *rowtab;        // global [5][n]
    ldsd_t *lds;                 // LV: [5][M][64 WPL] in LDS
    double k1[CL ? 1 : 6], k2[CL ? 1 : 6], dinv;   // PCR coefficients of this lane (WPL = 1: cyclic over the 64 chunks; WPL > 1: of the wave's isolated block)
    double a_s, c_s;             // separator-row couplings of this lane
    double vs, ws, wL, vF, dn, wLp, vFm, dp;      // WPL > 1: two-level reduction (chunked.hpp)
    ldsd_t *red, *redw;          // CL: the constants in LDS, [XRL][64 WPL] per lane and [6][WPL] per wave (xsys_init)
};

// LV = 0: every chunk has the same tables (scalar loads of chunk 0); 1: lane-variant tables [5][M][P] doubles in LDS; 2: lane-variant tables
// kept as chunk 0's value (scalar load) + a float difference in LDS -- where two systems of doubles do not fit (n >= 1024 with both systems).
// The reconstruction is exact when the chunks differ by less than 2^-29 relative (the 1e-13 wander of a "uniform" reference grid): the plan
// checks every entry on the host (xline_wide_ok, capi.cpp) and takes another kernel otherwise.
template <int M, int LV, int WPL>
__device__ __forceinline__ double xcoef(const double *rowtab, ldsd_t *lds, int tab, int p, int gl, int n) {
    constexpr int P = 64 * WPL;
    if (LV == 1) return lds[(tab * M + p) * P + gl];
    if (LV == 2) return rowtab[tab * n + p] + (double)((ldsf_t *)lds)[(tab * M + p) * P];
    return rowtab[tab * n + p];  // lane-invariant: chunk 0's row p, wave-uniform address -> scalar load
}
template <int M, int LV, int WPL, bool CL>
__device__ __forceinline__ void xsys_init(XSys<WPL, CL> &y, const SystemDev &sd, const double *lds, const double *red_lds, int gl, int n) {
    constexpr int P = 64 * WPL;
    y.rowtab = sd.rowtab;
    // float differences: per-lane pointers, opaque to the compiler -- behind 64 KB of tables every row of the constants would otherwise get an
    // address register of its own (148 spilled VGPRs in the two-system form); with doubles (1024 points) its own addressing is 3 % faster
    y.lds = LV == 2 ? (ldsd_t *)((ldsf_t *)lds + gl) : (ldsd_t *)lds;
    y.red = (ldsd_t *)red_lds;
    y.redw = (ldsd_t *)red_lds + XRL * P;
    if constexpr (CL && LV == 2) {
        y.red = (ldsd_t *)red_lds + gl;
        y.redw = (ldsd_t *)red_lds + XRL * P + (gl >> 6);
        asm volatile("" : "+v"(y.red), "+v"(y.redw));
    }
    if constexpr (!CL) {
        const int src = (LV || WPL > 1) ? gl : 0;
    #pragma unroll
        for (int s = 0; s < 6; ++s) {
            y.k1[s] = sd.red[s * P + src];
            y.k2[s] = sd.red[(6 + s) * P + src];
        }
        y.dinv = sd.red[12 * P + src];
        if constexpr (WPL > 1) {
            y.vs = sd.red[13 * P + src]; y.ws = sd.red[14 * P + src];
            y.wL = sd.red[15 * P + src]; y.vF = sd.red[16 * P + src]; y.dn = sd.red[17 * P + src];
            y.wLp = sd.red[18 * P + src]; y.vFm = sd.red[19 * P + src]; y.dp = sd.red[20 * P + src];
        }
        y.a_s = sd.rowtab[0 * n + (LV ? gl * M : 0)];
        y.c_s = sd.rowtab[2 * n + (LV ? gl * M : 0)];
    
    }
}

// f[0..M-1] (this lane's chunk of the right-hand side) -> solution, in place
template <int M, int LV, int WPL, bool CL>
__device__ __forceinline__ void xsolve(double (&f)[M], const XSys<WPL, CL> &y, XCtx<WPL> &c, int n) {
    constexpr int P = 64 * WPL;
    const int gl = c.gl, lane = c.lane;
#define XR(row) y.red[(row) * P + (LV == 2 ? 0 : gl)]
#define XW(q) y.redw[(q) * WPL + (LV == 2 ? 0 : c.wl)]
    ldsd_t *tl = y.lds;
    double g = 0.0;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        g = f[p] + xcoef<M, LV, WPL>(y.rowtab, tl, 0, p, gl, n) * g;
        f[p] = g;
    }
    double yn = 0.0;
#pragma unroll
    for (int p = M - 1; p >= 1; --p) {
        yn = f[p] * xcoef<M, LV, WPL>(y.rowtab, tl, 1, p, gl, n) + xcoef<M, LV, WPL>(y.rowtab, tl, 2, p, gl, n) * yn;
        f[p] = yn;
    }
    const double yLprev = xprev<WPL>(c, f[M - 1]);
    double r = f[0] - (CL ? XR(15) : y.a_s) * yLprev - (CL ? XR(16) : y.c_s) * f[1];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = 1 << s;
        const double rl = shfl_d(r, (lane - d) & 63);
        const double rr = shfl_d(r, (lane + d) & 63);
        r = r - (CL ? XR(s) : y.k1[CL ? 0 : s]) * rl - (CL ? XR(6 + s) : y.k2[CL ? 0 : s]) * rr;       // WPL > 1: the coefficients of neighbours outside the wave's block are zero
    }
    double X = r * (CL ? XR(12) : y.dinv), Xr;
    if constexpr (WPL == 1) {
        Xr = shfl_d(X, (lane + 1) & 63);
    } else {
        double *b = c.eb + c.par * (WPL * 4);
        if (lane == 0) b[c.wl * 4 + 1] = X;
        if (lane == 63) b[c.wl * 4 + 2] = X;
        xline_barrier<WPL>(c);
        const double YpL = b[((c.wl + WPL - 1) & (WPL - 1)) * 4 + 2], YnF = b[((c.wl + 1) & (WPL - 1)) * 4 + 1];
        const double myF = b[c.wl * 4 + 1], myL = b[c.wl * 4 + 2];
        c.par ^= 1;
        const double XL = (myL - (CL ? XW(0) : y.wL) * YnF) * (CL ? XW(2) : y.dn);          // last unknown of this wave and first one of the next: 2 x 2 interface system
        const double XnF = YnF - (CL ? XW(1) : y.vF) * XL;
        const double XpL = (YpL - (CL ? XW(3) : y.wLp) * myF) * (CL ? XW(5) : y.dp);        // last unknown of the previous wave (its interface with this one)
        X = X - (CL ? XR(13) : y.vs) * XpL - (CL ? XR(14) : y.ws) * XnF;
        Xr = shfl_d(X, (lane + 1) & 63);
        if (lane == 63) Xr = XnF;
    }
    f[0] = X;
#pragma unroll
    for (int p = 1; p < M; ++p) {
        f[p] = f[p] + xcoef<M, LV, WPL>(y.rowtab, tl, 3, p, gl, n) * X + xcoef<M, LV, WPL>(y.rowtab, tl, 4, p, gl, n) * Xr;
    }
#undef XR
#undef XW
}

// f = B u for this lane's chunk; um/up are the 3-point halos from the neighbouring lanes
template <int M, bool SYM>
__device__ __forceinline__ void xstencil(double (&f)[M], const double (&u)[M], const double (&um)[3], const double (&up)[3],
                                         const StencilDev &s, int lane) {
    double e[M + 6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e[k] = um[k]; e[M + 3 + k] = up[k]; }
#pragma unroll
    for (int p = 0; p < M; ++p) e[p + 3] = u[p];
#pragma unroll
    for (int p = 0; p < M; ++p) f[p] = stencil_interior<SYM>(s, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
    if (!s.periodic) {
        // boundary closures (one wave per line only): the six dense rows are computed wave-uniformly from broadcast values and
        // selected into the owning lane/register (rows 0..2 and n-3..n-1).
        constexpr int N = 64 * M;
        double ub[6], ut[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ub[k] = shfl_d(u[k % M], k / M);
            ut[k] = shfl_d(u[(N - 6 + k) % M], (N - 6 + k) / M);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double vb = dense6(s.bb[r], ub[0], ub[1], ub[2], ub[3], ub[4], ub[5]);
            const double vt = dense6(s.bt[r], ut[0], ut[1], ut[2], ut[3], ut[4], ut[5]);
            if (lane == r / M) f[r % M] = vb;
            if (lane == (N - 3 + r) / M) f[(N - 3 + r) % M] = vt;
        }
    }
}
template <int M, bool SYM>
__device__ __forceinline__ void xstencil_periodic(double (&f)[M], const double (&u)[M], const double (&um)[3], const double (&up)[3], const StencilDev &s) {
    double e[M + 6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e[k] = um[k]; e[M + 3 + k] = up[k]; }
#pragma unroll
    for (int p = 0; p < M; ++p) e[p + 3] = u[p];
#pragma unroll
    for (int p = 0; p < M; ++p) f[p] = stencil_interior<SYM>(s, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
}
// PER: the line is periodic (known on the host): no wall closures in the code -- their 2 x 36 coefficients are kernel arguments, and with them in
// play the one-wave forms spill scalar registers (a quarter of the instructions of the fused Burgers loop were v_readlane / v_accvgpr moves)
template <int M, bool SYM, int WPL, bool PER>
__device__ __forceinline__ void xsten(double (&f)[M], const double (&u)[M], const double (&um)[3], const double (&up)[3], const StencilDev &s, int lane) {
    if constexpr (WPL == 1 && !PER) xstencil<M, SYM>(f, u, um, up, s, lane);
    else xstencil_periodic<M, SYM>(f, u, um, up, s);
}

template <int M>
__device__ __forceinline__ void xload(double (&u)[M], const double *__restrict__ p) {
#pragma unroll
    for (int q = 0; q < M / 2; ++q) {
        const double2 v = reinterpret_cast<const double2 *>(p)[q];
        u[2 * q] = v.x;
        u[2 * q + 1] = v.y;
    }
}
template <int M>
__device__ __forceinline__ void xstore(double *__restrict__ p, const double (&u)[M]) {
#pragma unroll
    for (int q = 0; q < M / 2; ++q) reinterpret_cast<double2 *>(p)[q] = make_double2(u[2 * q], u[2 * q + 1]);
}

// LV / LV2: table form of the first- / second-derivative system (see xcoef)
// OCC = 2: two workgroups per CU (two waves per SIMD) asked of the compiler, and the lane-variant tables re-read from LDS where they are used instead of
// being kept in ~140 registers across the loop over lines and fields (512-point lines: 256 VGPRs + 102 AGPRs = one wave per SIMD otherwise)
template <int M, int MODE, int LV, int WPL, int LV2 = LV, int TPB = 256, bool CL = false, bool PER = false, int OCC = 1>
__global__ void __launch_bounds__(TPB, (OCC > 1 && TPB > 256) ? 1 : OCC) k_xline(XLineArgs a) {
    // (CL with one wave per line: stage_red leaves the rows of the two-level reduction empty)
    extern __shared__ double xlds[];
    constexpr bool NEED1 = (MODE != MODE_P2);
    constexpr bool NEED2 = (MODE != MODE_P1);
    constexpr int P = 64 * WPL, LPB = TPB / 64 / WPL;    // chunks per line, lines per workgroup
    constexpr int TAB = 5 * M * P;                        // table entries per system
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int lib = wib / WPL, wl = wib % WPL;            // line of the workgroup, wave of the line
    const int gl = wl * 64 + lane;
    const int n = P * M;

    // LDS: [system 1 tables][system 2 tables] (only those this MODE solves), then the exchange buffers of the LPB lines (WPL > 1)
    constexpr size_t TABW1 = LV == 1 ? (size_t)TAB : LV == 2 ? (size_t)(TAB / 2) : 0;       // doubles of system 1 / 2
    constexpr size_t TABW2 = LV2 == 1 ? (size_t)TAB : LV2 == 2 ? (size_t)(TAB / 2) : 0;
    const double *lds1 = xlds, *lds2 = xlds + (NEED1 ? TABW1 : 0);
    auto stage = [&](const double *rowtab, const double *dst, int lv) {      // lane-variant tables: [5][M][P] per system in LDS once per block
        if (lv == 1) {
            double *d = const_cast<double *>(dst);
            for (int idx = threadIdx.x; idx < TAB; idx += blockDim.x) {
                const int l = idx % P, p = (idx / P) % M, tab = idx / (P * M);
                d[idx] = rowtab[tab * n + l * M + p];
            }
        } else if (lv == 2) {                                                 // ... as float differences from chunk 0
            float *d = reinterpret_cast<float *>(const_cast<double *>(dst));
            for (int idx = threadIdx.x; idx < TAB; idx += blockDim.x) {
                const int l = idx % P, p = (idx / P) % M, tab = idx / (P * M);
                d[idx] = (float)(rowtab[tab * n + l * M + p] - rowtab[tab * n + p]);
            }
        }
    };
    if (NEED1) stage(a.y1.rowtab, lds1, LV);
    if (NEED2) stage(a.y2.rowtab, lds2, LV2);
    XCtx<WPL> cx;
    cx.lane = lane; cx.gl = gl; cx.wl = wl; cx.par = 0; cx.eb = nullptr; cx.hb = nullptr; cx.bar = nullptr; cx.gen = 0;
    constexpr size_t REDW = CL ? (size_t)XRL * P + 6 * WPL : 0;        // constants of one system
    double *red1 = xlds + ((NEED1 ? TABW1 : 0) + (NEED2 ? TABW2 : 0)), *red2 = red1 + (NEED1 ? REDW : 0);
    if constexpr (CL) {
        auto stage_red = [&](const SystemDev &sd, double *d, int lv) {
            for (int idx = threadIdx.x; idx < XRL * P; idx += blockDim.x) {
                const int l = idx % P, k = idx / P;
                // (one wave per line: the table has the 13 rows of the cyclic reduction only; rows 13, 14 and the interface constants belong to the two-level form)
                d[idx] = k < 13 ? sd.red[k * P + l] : k < 15 ? (WPL > 1 ? sd.red[k * P + l] : 0.0) : sd.rowtab[(k == 15 ? 0 : 2) * n + (lv ? l * M : 0)];
            }
            if constexpr (WPL > 1)
                for (int idx = threadIdx.x; idx < 6 * WPL; idx += blockDim.x) d[XRL * P + idx] = sd.red[(15 + idx / WPL) * P + (idx % WPL) * 64];
        };
        if (NEED1) stage_red(a.y1, red1, LV);
        if (NEED2) stage_red(a.y2, red2, LV2);
    }
    if constexpr (WPL > 1) {
        double *ex0 = xlds + ((NEED1 ? TABW1 : 0) + (NEED2 ? TABW2 : 0)) + (NEED1 ? REDW : 0) + (NEED2 ? REDW : 0);
        double *ex = ex0 + (size_t)lib * (14 * WPL);
        cx.eb = ex; cx.hb = ex + 8 * WPL;
        if constexpr (LPB > 1) {
            if (a.line_barriers) {
                unsigned *bars = reinterpret_cast<unsigned *>(ex0 + (size_t)LPB * (14 * WPL));
                if (threadIdx.x < LPB) bars[threadIdx.x] = 0u;
                cx.bar = bars + lib;
            }
        }
    }
    if (LV != 0 || LV2 != 0 || WPL > 1) __syncthreads();
    XSys<WPL, CL> y1, y2;
    if (NEED1) xsys_init<M, LV, WPL, CL>(y1, a.y1, lds1, red1, gl, n);
    if (NEED2) xsys_init<M, LV2, WPL, CL>(y2, a.y2, lds2, red2, gl, n);

    const long long stride = (long long)gridDim.x * LPB;
    // software pipeline over the lines of this workgroup: the operand of the NEXT line is requested before the solves of this one (with few
    // waves per CU -- one workgroup of 98-120 KB LDS at 2048 points -- nothing else hides the 1.5-2 us of a load from HBM)
    double pn[M], pbn[(MODE == MODE_P1) ? M : 1];
    bool have_pn = false;
    for (long long line0 = (long long)blockIdx.x * LPB; line0 < a.nlines; line0 += stride) {
        long long line = line0 + lib;
        const bool live = line < a.nlines;
        if (!live) {
            if constexpr (WPL == 1) continue;       // independent waves
            line = a.nlines - 1;                    // the waves of a workgroup meet at barriers: compute, do not store
        }
        const long long off = line * n + gl * M;
        const bool next = line0 + stride < a.nlines;            // workgroup-uniform
        long long nline = line0 + stride + lib;
        if (nline >= a.nlines) nline = a.nlines - 1;
        const long long noff = nline * n + gl * M;
        // many rows per lane: the lane-variant tables are loop-invariant and the compiler would keep all 2 x 5 x M of them in registers
        // (182 spilled VGPRs at M = 32); an opaque copy of the LDS pointers per line makes it re-read them where they are used
        if constexpr (LV == 2 || (LV == 1 && (M >= 16 || OCC > 1))) { if (NEED1) asm volatile("" : "+v"(y1.lds)); }
        if constexpr (LV2 == 2 || (LV2 == 1 && (M >= 16 || OCC > 1))) { if (NEED2) asm volatile("" : "+v"(y2.lds)); }
        if constexpr (MODE == MODE_BURGERS) {
            // the advecting velocity of the line is loaded once and serves every transported field (rhs_global_incompressible_1.f90:
            // 98-162 calls OPR_Burgers_X four times with the same u)
            double v[M];
            if (have_pn) {
#pragma unroll
                for (int p = 0; p < M; ++p) v[p] = pn[p];
            } else {
                xload<M>(v, a.in1 + off);
            }
            have_pn = false;
            // software pipeline over the fields: with one wave per SIMD nothing else hides the memory latency, so the operand of the NEXT
            // field and the old tendency of THIS one are requested before the two solves of this field start (2.9-3.0 -> 2.65-2.75 ms at 512^3;
            // requesting the next LINE's velocity during the last field as well did not add anything, measured)
            double un[M];
            bool have_next = false;
            for (int f = 0; f < a.nf; ++f) {
                const double *src = a.fs[f];
                double *dst = a.fo[f];
                const double nuf = a.ari ? a.fnu[f] * a.ari[line % a.ari_ny] : a.fnu[f];      // anelastic: ribackground(j) of this line on the diffusion term
                if constexpr (LV == 2 || (LV == 1 && (M >= 16 || OCC > 1))) asm volatile("" : "+v"(y1.lds));      // ... and per field
                if constexpr (LV2 == 2 || (LV2 == 1 && (M >= 16 || OCC > 1))) asm volatile("" : "+v"(y2.lds));
                double u[M];
                if (src == a.in1) {
#pragma unroll
                    for (int p = 0; p < M; ++p) u[p] = v[p];
                } else if (have_next) {
#pragma unroll
                    for (int p = 0; p < M; ++p) u[p] = un[p];
                } else {
                    xload<M>(u, src + off);
                }
                constexpr bool PIPE = (M <= 16) && !(CL && LV == 2 && LV2 == 2);     // 32 rows per lane: the extra line-sets would spill
                double o[M];
                if (PIPE && a.acc) xload<M>(o, dst + off);
                have_next = PIPE && (f + 1 < a.nf) && (a.fs[f + 1] != a.in1);
                if (have_next) xload<M>(un, a.fs[f + 1] + off);
                if (PIPE && M <= 8 && f + 1 == a.nf && next) { xload<M>(pn, a.in1 + noff); have_pn = true; }     // the next line's velocity
                double um[3], up[3];
                xhalo<M, WPL>(cx, u, um, up);
                double x1[M], x2[M];
                xsten<M, false, WPL, PER>(x1, u, um, up, a.s1, lane);
                xsolve<M, LV, WPL, CL>(x1, y1, cx, n);
                xsten<M, true, WPL, PER>(x2, u, um, up, a.s2, lane);
                xsolve<M, LV2, WPL, CL>(x2, y2, cx, n);
#pragma unroll
                for (int p = 0; p < M; ++p) x2[p] = nuf * x2[p] - v[p] * x1[p];      // opr_burgers.f90:513
                if (a.acc) {
                    if (!PIPE) xload<M>(o, dst + off);
#pragma unroll
                    for (int p = 0; p < M; ++p) x2[p] = o[p] + x2[p];
                }
                if constexpr (M <= 8) {
                    if (a.fdiv != nullptr && src == a.in1) {     // tendency of u complete: x term of the pressure forcing from the registers
                        double wq[M];
#pragma unroll
                        for (int p = 0; p < M; ++p) wq[p] = x2[p] + v[p] * a.fidte;
                        xhalo<M, WPL>(cx, wq, um, up);
                        xsten<M, false, WPL, PER>(x1, wq, um, up, a.s1, lane);
                        xsolve<M, LV, WPL, CL>(x1, y1, cx, n);
                        if (live) xstore<M>(a.fdiv + off, x1);
                    }
                    if (a.ffin[f]) {          // the tendency of this field is complete: wall planes, Runge-Kutta update, scaling (k_final_update's arithmetic)
                        const int j = (int)(line % a.fny);
                        const bool wall = (j == 0) || (j == a.fny - 1);
#pragma unroll
                        for (int p = 0; p < M; ++p) {
                            const double hv = wall ? 0.0 : x2[p];
                            u[p] = u[p] + a.fdte * hv;
                            x2[p] = a.fscale ? a.fkco * hv : hv;
                        }
                        if (live) xstore<M>(const_cast<double *>(src) + off, u);
                    }
                }
                if (live) xstore<M>(dst + off, x2);
            }
        } else {
            double u[M];
            if (have_pn) {
#pragma unroll
                for (int p = 0; p < M; ++p) u[p] = pn[p];
            } else {
                xload<M>(u, a.in0 + off);
            }
            if (MODE == MODE_P1 && a.in0b != nullptr) {   // operand = in0 + s * in0b
                double ub[M];
                if (have_pn) {
#pragma unroll
                    for (int p = 0; p < M; ++p) ub[p] = pbn[p];
                } else {
                    xload<M>(ub, a.in0b + off);
                }
#pragma unroll
                for (int p = 0; p < M; ++p) u[p] = u[p] + ub[p] * a.in0b_scale;
            }
            have_pn = false;
            if constexpr (M <= 8) {
                if (next) {
                    xload<M>(pn, a.in0 + noff);
                    if constexpr (MODE == MODE_P1) { if (a.in0b != nullptr) xload<M>(pbn, a.in0b + noff); }
                    have_pn = true;
                }
            }
            // old tendency / velocity of the epilogues: requested before the solve (one wave per SIMD: nothing else hides the latency)
            double h[MODE == MODE_P1 ? M : 1], qv[MODE == MODE_P1 ? M : 1];
            if constexpr (MODE == MODE_P1) {
                if (a.fq != nullptr) {
                    xload<M>(h, a.out0 + off);
                    xload<M>(qv, a.fq + off);
                } else if (a.acc) {
                    xload<M>(h, a.out0 + off);
                }
            }
            double um[3], up[3];
            xhalo<M, WPL>(cx, u, um, up);
            double x1[M], x2[M];
            if (NEED1) {
                xsten<M, false, WPL, PER>(x1, u, um, up, a.s1, lane);
                xsolve<M, LV, WPL, CL>(x1, y1, cx, n);
            }
            if (NEED2) {
                xsten<M, true, WPL, PER>(x2, u, um, up, a.s2, lane);
                xsolve<M, LV2, WPL, CL>(x2, y2, cx, n);
            }
            if (!live) continue;                     // (WPL > 1: after the last barrier of this line)
            if constexpr (MODE == MODE_P1) {
                if (a.fq != nullptr) {          // final-update epilogue: the line is (j, k) = (line % ny, line / ny)
                    const int j = (int)(line % a.fny);
                    const bool wall = (j == 0) || (j == a.fny - 1);
                    const double *wp = wall ? (j == 0 ? a.fpb : a.fpt) : nullptr;      // given wall tendencies (Neumann walls), or zero
                    if (wp != nullptr) wp += (line / a.fny) * n + gl * M;
#pragma unroll
                    for (int p = 0; p < M; ++p) {
                        const double hv = wall ? (wp ? wp[p] : 0.0) : h[p] - x1[p];
                        qv[p] = qv[p] + a.fdte * hv;
                        h[p] = a.fscale ? a.fkco * hv : hv;
                    }
                    xstore<M>(a.fq + off, qv);
                    xstore<M>(a.out0 + off, h);
                } else {
                    if (a.acc) {
#pragma unroll
                        for (int p = 0; p < M; ++p) x1[p] = (a.acc == 2) ? h[p] - x1[p] : h[p] + x1[p];
                    }
                    xstore<M>(a.out0 + off, x1);
                }
            } else if constexpr (MODE == MODE_P2) {
                xstore<M>(a.out0 + off, x2);
            } else {   // MODE_P2_P1
                xstore<M>(a.out0 + off, x2);
                xstore<M>(a.out1 + off, x1);
            }
        }
    }
}

// ============================================================================================
// k_rtile : one workgroup per 64-line tile, wave-chunked, data in registers
// ============================================================================================
template <int M, int MODE, int MAXT>
__global__ void __launch_bounds__(MAXT) k_rtile(RTileArgs a) {
    __shared__ double s_yl[16 * 64];
    __shared__ double s_r[16 * 64];
    constexpr bool SECOND = (MODE != MODE_P1);  // which operator this launch solves
    constexpr bool D1IN = (MODE == MODE_P2_D1IN || MODE == MODE_BURGERS_D1IN);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = blockDim.x >> 6;
    const int n = a.g.n;
    const long long rs = a.g.row_stride;
    const StencilDev &st = SECOND ? a.s2 : a.s1;
    const SystemDev &sy = SECOND ? a.y2 : a.y1;

    const int tiles_inner = (a.g.lines_inner + 63) >> 6;
    const long long outer = blockIdx.x / tiles_inner;
    const int l0 = (int)(blockIdx.x % tiles_inner) << 6;
    const bool valid = (l0 + lane) < a.g.lines_inner;
    const long long base = outer * a.g.outer_stride + l0 + lane;
    const int row0 = w * M;
    const bool per = st.periodic != 0;

    // ---- load this wave's M rows + 3-row halos (coalesced 512-B rows) ----
    double e[M + 6];
#pragma unroll
    for (int p = 0; p < M; ++p) e[p + 3] = valid ? a.in0[base + (long long)(row0 + p) * rs] : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int rl = row0 - 3 + k, rr = row0 + M + k;
        const bool okl = per || rl >= 0, okr = per || rr < n;
        if (rl < 0) rl += n;
        if (rr >= n) rr -= n;
        e[k] = (valid && okl) ? a.in0[base + (long long)rl * rs] : 0.0;
        e[M + 3 + k] = (valid && okr) ? a.in0[base + (long long)rr * rs] : 0.0;
    }
    if (MODE == MODE_P1 && a.in0b != nullptr) {   // operand = in0 + s * in0b (same rows, same halos)
#pragma unroll
        for (int p = 0; p < M; ++p)
            if (valid) e[p + 3] = e[p + 3] + a.in0b[base + (long long)(row0 + p) * rs] * a.in0b_scale;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int rl = row0 - 3 + k, rr = row0 + M + k;
            const bool okl = per || rl >= 0, okr = per || rr < n;
            if (rl < 0) rl += n;
            if (rr >= n) rr -= n;
            if (valid && okl) e[k] = e[k] + a.in0b[base + (long long)rl * rs] * a.in0b_scale;
            if (valid && okr) e[M + 3 + k] = e[M + 3 + k] + a.in0b[base + (long long)rr * rs] * a.in0b_scale;
        }
    }

    // ---- right-hand side ----
    double f[M];
    if (st.rowc != nullptr) {       // direct scheme (either derivative): per-row coefficients, wave-uniform -> scalar loads
        const double *rc = st.rowc + (long long)row0 * 5;
#pragma unroll
        for (int p = 0; p < M; ++p)
            f[p] = e[p + 1] * rc[p * 5 + 0] + e[p + 2] * rc[p * 5 + 1] + e[p + 3] * rc[p * 5 + 2] + e[p + 4] * rc[p * 5 + 3] + e[p + 5] * rc[p * 5 + 4];
    } else {
#pragma unroll
        for (int p = 0; p < M; ++p) f[p] = stencil_interior<SECOND>(st, e[p], e[p + 1], e[p + 2], e[p + 3], e[p + 4], e[p + 5], e[p + 6]);
    }
    if (!per) {
        if (w == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) f[r] = dense6(st.bb[r], e[3], e[4], e[5], e[6], e[7], e[8]);
        }
        if (w == P - 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r) f[M - 3 + r] = dense6(st.bt[r], e[M - 3], e[M - 2], e[M - 1], e[M], e[M + 1], e[M + 2]);
        }
    }
    if (D1IN && a.jc.j != nullptr) {
        // Jacobian correction f += A2 dx2 du (MatMul_3d_add, fdm_matmul.f90:126-153); du is read from in1
        const double *j1 = a.jc.j, *j2 = a.jc.j + n, *j3 = a.jc.j + 2 * n;
        double dprev, dcur, dnext;
        {
            int rm = row0 - 1;
            const bool ok = rm >= 0;
            if (rm < 0) rm = 0;
            dprev = (valid && ok) ? a.in1[base + (long long)rm * rs] : 0.0;
            dcur = valid ? a.in1[base + (long long)row0 * rs] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < M; ++p) {
            int rn = row0 + p + 1;
            const bool ok = rn < n;
            if (rn >= n) rn = n - 1;
            dnext = (valid && ok) ? a.in1[base + (long long)rn * rs] : 0.0;
            const int row = row0 + p;
            double add = dprev * j1[row] + dcur * j2[row] + dnext * j3[row];
            if (row == 0) {  // extended stencil: r1(1) multiplies du(3)
                const double d2v = valid ? a.in1[base + 2 * rs] : 0.0;
                add = dcur * j2[0] + dnext * j3[0] + d2v * j1[0];
            }
            if (row == n - 1) {  // extended stencil: r3(n) multiplies du(n-2)
                const double d3v = valid ? a.in1[base + (long long)(n - 3) * rs] : 0.0;
                add = d3v * j3[n - 1] + dprev * j1[n - 1] + dcur * j2[n - 1];
            }
            f[p] = f[p] + add;
            dprev = dcur;
            dcur = dnext;
        }
    }

    // ---- local Thomas solve of the interior rows; coefficient rows are wave-uniform (scalar loads) ----
    const double *Lm = sy.rowtab + row0, *Di = sy.rowtab + n + row0, *Cm = sy.rowtab + 2 * n + row0;
    const double *Vt = sy.rowtab + 3 * n + row0, *Wt = sy.rowtab + 4 * n + row0;
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
    // ---- separator system through LDS ----
    s_yl[w * 64 + lane] = f[M - 1];
    __syncthreads();
    const int wm = (w + P - 1) % P, wp = (w + 1) % P;
    const double yLprev = s_yl[wm * 64 + lane];
    s_r[w * 64 + lane] = f[0] - Lm[0] * yLprev - Cm[0] * f[1];
    __syncthreads();
    double X = 0.0, Xr = 0.0;
    for (int q = 0; q < P; ++q) {
        const double rq = s_r[q * 64 + lane];
        X += sy.red[w * P + q] * rq;
        Xr += sy.red[wp * P + q] * rq;
    }
    f[0] = X;
#pragma unroll
    for (int p = 1; p < M; ++p) f[p] = f[p] + Vt[p] * X + Wt[p] * Xr;

    // ---- epilogue ----
    if (MODE == MODE_BURGERS_D1IN) {
#pragma unroll
        for (int p = 0; p < M; ++p) {
            const long long idx = base + (long long)(row0 + p) * rs;
            if (valid) a.out0[idx] = a.nu * f[p] - a.in2[idx] * a.in1[idx];
        }
    } else if (MODE == MODE_P1 && a.fneu != 0) {
        // Neumann-final epilogue (y-direction lines, operand = the finished tendency h, f = its derivative by the Neumann variant of the system):
        // wall tendencies as k_neumann_planes forms them, then k_final_update's arithmetic.  h is read again (the operand registers are gone).
        if (valid) {
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {
                double h[8], qv[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    h[p] = a.in0[base + (long long)(row0 + p0 + p) * rs];
                    qv[p] = a.fq[base + (long long)(row0 + p0 + p) * rs];
                }
                if (p0 == 0 && w == 0)           // row 0: BOUNDARY_BCS_NEUMANN_Y's bottom value, or zero on a Dirichlet side
                    h[0] = (a.fneu & 1) ? ((h[1] * a.fcb[0] + h[2] * a.fcb[1]) + h[3] * a.fcb[2]) + a.fcb[3] * f[1] : 0.0;
                if (p0 == M - 8 && w == P - 1)   // row n-1
                    h[7] = (a.fneu & 2) ? ((h[4] * a.fct[0] + h[5] * a.fct[1]) + h[6] * a.fct[2]) + a.fct[3] * f[M - 2] : 0.0;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const double hv = h[p];
                    qv[p] = qv[p] + a.fdte * hv;
                    h[p] = a.fscale ? a.fkco * hv : hv;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    a.fq[base + (long long)(row0 + p0 + p) * rs] = qv[p];
                    a.out0[base + (long long)(row0 + p0 + p) * rs] = h[p];
                }
            }
        }
    } else if (MODE == MODE_P1 && a.fq != nullptr) {
        // final-update epilogue (z-direction lines: lane = (ix, j) inside the plane, rows = k)
        if (valid) {
            const int j = ((l0 + lane) / a.fnx) % a.fny;
            const bool wall = (j == 0) || (j == a.fny - 1);
            const double *wp = wall ? (j == 0 ? a.fpb : a.fpt) : nullptr;      // given wall tendencies (Neumann walls), or zero
            if (wp != nullptr) wp += (l0 + lane) % a.fnx;
            // 8 rows at a time: 16 loads in flight, then their stores (f[M] + h[M] + q[M] would not fit the 128 VGPRs of the 1024-thread launch)
#pragma unroll
            for (int p0 = 0; p0 < M; p0 += 8) {
                double h[8], qv[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    h[p] = a.out0[base + (long long)(row0 + p0 + p) * rs];
                    qv[p] = a.fq[base + (long long)(row0 + p0 + p) * rs];
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const double hv = wall ? (wp ? wp[(long long)(row0 + p0 + p) * a.fnx] : 0.0) : h[p] - f[p0 + p];
                    qv[p] = qv[p] + a.fdte * hv;
                    h[p] = a.fscale ? a.fkco * hv : hv;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    a.fq[base + (long long)(row0 + p0 + p) * rs] = qv[p];
                    a.out0[base + (long long)(row0 + p0 + p) * rs] = h[p];
                }
            }
        }
    } else {
        if (MODE == MODE_P1 && a.acc && valid) {   // all loads first: the compiler cannot move them across the stores itself
            double o[M];
#pragma unroll
            for (int p = 0; p < M; ++p) o[p] = a.out0[base + (long long)(row0 + p) * rs];
#pragma unroll
            for (int p = 0; p < M; ++p) f[p] = (a.acc == 2) ? o[p] - f[p] : o[p] + f[p];
        }
#pragma unroll
        for (int p = 0; p < M; ++p)
            if (valid) a.out0[base + (long long)(row0 + p) * rs] = f[p];
    }
}

// ============================================================================================
// k_generic : any n, one thread per line, single-chunk tables (P = 1), three sweeps through out
// ============================================================================================
template <bool SYM>
__device__ __forceinline__ double generic_rhs_row(const double *__restrict__ u, long long base, long long rs, int i, int n,
                                                  const StencilDev &s) {
    if (s.periodic) {
        auto at = [&](int k) {
            int r = i + k;
            if (r < 0) r += n;
            if (r >= n) r -= n;
            return u[base + (long long)r * rs];
        };
        return stencil_interior<SYM>(s, at(-3), at(-2), at(-1), at(0), at(1), at(2), at(3));
    }
    if (i < 3) return dense6(s.bb[i], u[base], u[base + rs], u[base + 2 * rs], u[base + 3 * rs], u[base + 4 * rs], u[base + 5 * rs]);
    if (i >= n - 3) {
        const long long b6 = base + (long long)(n - 6) * rs;
        return dense6(s.bt[i - (n - 3)], u[b6], u[b6 + rs], u[b6 + 2 * rs], u[b6 + 3 * rs], u[b6 + 4 * rs], u[b6 + 5 * rs]);
    }
    const long long c = base + (long long)i * rs;
    if (s.rowc != nullptr) {
        const double *rc = s.rowc + (long long)i * 5;
        return u[c - 2 * rs] * rc[0] + u[c - rs] * rc[1] + u[c] * rc[2] + u[c + rs] * rc[3] + u[c + 2 * rs] * rc[4];
    }
    return stencil_interior<SYM>(s, u[c - 3 * rs], u[c - 2 * rs], u[c - rs], u[c], u[c + rs], u[c + 2 * rs], u[c + 3 * rs]);
}

__device__ __forceinline__ double generic_jc_row(const double *__restrict__ du, long long base, long long rs, int i, int n, const double *j) {
    const double *j1 = j, *j2 = j + n, *j3 = j + 2 * n;
    if (i == 0) return du[base] * j2[0] + du[base + rs] * j3[0] + du[base + 2 * rs] * j1[0];
    if (i == n - 1)
        return du[base + (long long)(n - 3) * rs] * j3[i] + du[base + (long long)(n - 2) * rs] * j1[i] + du[base + (long long)(n - 1) * rs] * j2[i];
    const long long c = base + (long long)i * rs;
    return du[c - rs] * j1[i] + du[c] * j2[i] + du[c + rs] * j3[i];
}

template <bool SYM>
__global__ void __launch_bounds__(256) k_generic(GenericArgs a) {
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.g.nlines) return;
    const int n = a.g.n;
    const long long rs = a.g.row_stride;
    const long long base = (l / a.g.lines_inner) * a.g.outer_stride + (l % a.g.lines_inner) * (a.g.lines_inner == 1 ? 0 : 1);
    const double *Lm = a.y.rowtab, *Di = a.y.rowtab + n, *Cm = a.y.rowtab + 2 * n, *Vt = a.y.rowtab + 3 * n, *Wt = a.y.rowtab + 4 * n;
    const bool jc = a.jc.j != nullptr;

    double f0 = generic_rhs_row<SYM>(a.in0, base, rs, 0, n, a.s);
    if (jc) f0 += generic_jc_row(a.in1, base, rs, 0, n, a.jc.j);
    double g = 0.0;
    for (int i = 1; i < n; ++i) {
        double fi = generic_rhs_row<SYM>(a.in0, base, rs, i, n, a.s);
        if (jc) fi += generic_jc_row(a.in1, base, rs, i, n, a.jc.j);
        g = fi + Lm[i] * g;
        a.out0[base + (long long)i * rs] = g;
    }
    double yn = 0.0, y1 = 0.0, ylast = 0.0;
    for (int i = n - 1; i >= 1; --i) {
        yn = a.out0[base + (long long)i * rs] * Di[i] + Cm[i] * yn;
        a.out0[base + (long long)i * rs] = yn;
        if (i == n - 1) ylast = yn;
        if (i == 1) y1 = yn;
    }
    const double X = (f0 - Lm[0] * ylast - Cm[0] * y1) * a.y.red[0];
    a.out0[base] = X;
    for (int i = 1; i < n; ++i) {
        const long long idx = base + (long long)i * rs;
        a.out0[idx] = a.out0[idx] + Vt[i] * X + Wt[i] * X;
    }
}

// ============================================================================================
// pointwise / data-movement kernels
// ============================================================================================
// out = nu * out - vel * d1   (OPR_Burgers_1D epilogue, opr_burgers.f90:510-516)
// ============================================================================================
// k_penta1 : CompactJacobian6Penta first derivative, one thread per line, reference operation order
// ============================================================================================
__global__ void __launch_bounds__(256) k_penta1(PentaArgs a) {
    const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= a.g.nlines) return;
    const int n = a.g.n;
    const long long rs = a.g.row_stride;
    const long long base = (line / a.g.lines_inner) * a.g.outer_stride + (line % a.g.lines_inner);
    const double *u = a.in0 + base;
    double *f = a.out0 + base;
#define U(i) u[(long long)((i)-1) * rs]
#define F(i) f[(long long)((i)-1) * rs]
#define RI(i, k) a.rhs[((i)-1) + (size_t)n * ((k)-1)]
#define RB(j, c) a.rb[((j)-1) + 4 * (c)]
#define RT(r, c) a.rt[(r) + 5 * ((c)-1)]
    const double r6 = RI(5, 6), r7 = RI(5, 7);
    const int ibc = a.periodic ? -1 : a.ibc;
    const bool nb = (ibc == 1 || ibc == 3), nt = (ibc == 2 || ibc == 3);
    int nmin = 1, nmax = n;
    if (nb) { F(1) = 0.0; nmin = 2; }            // homogeneous Neumann (fdm_derivative.f90:238-246)
    if (nt) { F(n) = 0.0; nmax = n - 1; }
    // ---- MatMul_7d_antisym ----
    if (ibc == -1) {
        F(1) = U(2) - U(n) + r6 * (U(3) - U(n - 1)) + r7 * (U(4) - U(n - 2));
        F(2) = U(3) - U(1) + r6 * (U(4) - U(n)) + r7 * (U(5) - U(n - 1));
        F(3) = U(4) - U(2) + r6 * (U(5) - U(1)) + r7 * (U(6) - U(n));
        F(4) = U(5) - U(3) + r6 * (U(6) - U(2)) + r7 * (U(7) - U(1));
    } else if (nb) {
        const double f1 = 0.0;
        F(2) = f1 * RB(2, 3) + U(2) * RB(2, 4) + U(3) * RB(2, 5) + U(4) * RB(2, 6) + U(5) * RB(2, 7);
        F(3) = f1 * RB(3, 2) + U(2) * RB(3, 3) + U(3) * RB(3, 4) + U(4) * RB(3, 5) + U(5) * RB(3, 6) + U(6) * RB(3, 7);
        F(4) = f1 * RB(4, 1) + U(2) * RB(4, 2) + U(3) * RB(4, 3) + U(4) * RB(4, 4) + U(5) * RB(4, 5) + U(6) * RB(4, 6) + U(7) * RB(4, 7);
    } else {
        F(1) = U(1) * RI(1, 4) + U(2) * RI(1, 5) + U(3) * RI(1, 6) + U(4) * RI(1, 7) + U(5) * RI(1, 1);
        F(2) = U(1) * RI(2, 3) + U(2) * RI(2, 4) + U(3) * RI(2, 5) + U(4) * RI(2, 6) + U(5) * RI(2, 7);
        F(3) = U(1) * RI(3, 2) + U(2) * RI(3, 3) + U(3) * RI(3, 4) + U(4) * RI(3, 5) + U(5) * RI(3, 6) + U(6) * RI(3, 7);
        F(4) = U(1) * RI(4, 1) + U(2) * RI(4, 2) + U(3) * RI(4, 3) + U(4) * RI(4, 4) + U(5) * RI(4, 5) + U(6) * RI(4, 6) + U(7) * RI(4, 7);
    }
    for (int i = 5; i <= n - 4; ++i) F(i) = U(i + 1) - U(i - 1) + r6 * (U(i + 2) - U(i - 2)) + r7 * (U(i + 3) - U(i - 3));
    if (ibc == -1) {
        F(n - 3) = U(n - 2) - U(n - 4) + r6 * (U(n - 1) - U(n - 5)) + r7 * (U(n) - U(n - 6));
        F(n - 2) = U(n - 1) - U(n - 3) + r6 * (U(n) - U(n - 4)) + r7 * (U(1) - U(n - 5));
        F(n - 1) = U(n) - U(n - 2) + r6 * (U(1) - U(n - 3)) + r7 * (U(2) - U(n - 4));
        F(n) = U(1) - U(n - 1) + r6 * (U(2) - U(n - 2)) + r7 * (U(3) - U(n - 3));
    } else if (nt) {
        const double fn = 0.0;
        F(n - 3) = U(n - 6) * RT(1, 1) + U(n - 5) * RT(1, 2) + U(n - 4) * RT(1, 3) + U(n - 3) * RT(1, 4) + U(n - 2) * RT(1, 5) + U(n - 1) * RT(1, 6) + fn * RT(1, 7);
        F(n - 2) = U(n - 5) * RT(2, 1) + U(n - 4) * RT(2, 2) + U(n - 3) * RT(2, 3) + U(n - 2) * RT(2, 4) + U(n - 1) * RT(2, 5) + fn * RT(2, 6);
        F(n - 1) = U(n - 4) * RT(3, 1) + U(n - 3) * RT(3, 2) + U(n - 2) * RT(3, 3) + U(n - 1) * RT(3, 4) + fn * RT(3, 5);
    } else {
        F(n - 3) = U(n - 6) * RI(n - 3, 1) + U(n - 5) * RI(n - 3, 2) + U(n - 4) * RI(n - 3, 3) + U(n - 3) * RI(n - 3, 4) + U(n - 2) * RI(n - 3, 5) + U(n - 1) * RI(n - 3, 6) + U(n) * RI(n - 3, 7);
        F(n - 2) = U(n - 5) * RI(n - 2, 1) + U(n - 4) * RI(n - 2, 2) + U(n - 3) * RI(n - 2, 3) + U(n - 2) * RI(n - 2, 4) + U(n - 1) * RI(n - 2, 5) + U(n) * RI(n - 2, 6);
        F(n - 1) = U(n - 4) * RI(n - 1, 1) + U(n - 3) * RI(n - 1, 2) + U(n - 2) * RI(n - 1, 3) + U(n - 1) * RI(n - 1, 4) + U(n) * RI(n - 1, 5);
        F(n) = U(n - 4) * RI(n, 7) + U(n - 3) * RI(n, 1) + U(n - 2) * RI(n, 2) + U(n - 1) * RI(n, 3) + U(n) * RI(n, 4);
    }
    // ---- PENTADSS2 on rows nmin .. nmax with the LU columns of the variant (offset nmin, fdm_derivative.f90:270-272) ----
    const int ip = a.periodic ? 0 : a.ibc * 5;
    const double *A = a.lu + (size_t)n * (ip + 0) + (nmin - 1), *B = a.lu + (size_t)n * (ip + 1) + (nmin - 1), *C = a.lu + (size_t)n * (ip + 2) + (nmin - 1);
    const double *D = a.lu + (size_t)n * (ip + 3) + (nmin - 1), *E = a.lu + (size_t)n * (ip + 4) + (nmin - 1);
    const int m = nmax - nmin + 1;
#define G(i) f[(long long)((i) + nmin - 2) * rs]      /* 1-based inside the reduced system */
    G(m - 1) = G(m - 1) - G(m) * D[m - 2];
    for (int i = m - 2; i >= 1; --i) G(i) = G(i) - G(i + 1) * D[i - 1] - G(i + 2) * E[i - 1];
    G(1) = G(1) / C[0];
    G(2) = (G(2) - G(1) * B[1]) / C[1];
    for (int i = 3; i <= m; ++i) G(i) = (G(i) - G(i - 1) * B[i - 1] - G(i - 2) * A[i - 1]) / C[i - 1];
    if (a.periodic) {        // PENTADPSS :352-411 : Sherman-Morrison-Woodbury correction with the two stored vectors
        const double *Fv = a.lu + (size_t)n * 5, *Gv = a.lu + (size_t)n * 6;
        const double m1 = E[n - 1] * Fv[0] + A[0] * Fv[n - 2] + B[0] * Fv[n - 1] + 1.0;
        const double m2 = E[n - 1] * Gv[0] + A[0] * Gv[n - 2] + B[0] * Gv[n - 1];
        const double m3 = D[n - 1] * Fv[0] + E[n - 1] * Fv[1] + A[0] * Fv[n - 1];
        const double m4 = D[n - 1] * Gv[0] + E[n - 1] * Gv[1] + A[0] * Gv[n - 1] + 1.0;
        const double di = 1 / (m1 * m4 - m2 * m3);
        const double d11 = di * (m4 * E[n - 1] - m2 * D[n - 1]), d12 = di * (m4 * B[0] - m2 * A[0]), d13 = di * m4 * A[0], d14 = di * m2 * E[n - 1];
        const double d21 = di * (m1 * D[n - 1] - m3 * E[n - 1]), d22 = di * (m1 * A[0] - m3 * B[0]), d23 = di * m3 * A[0], d24 = di * m1 * E[n - 1];
        const double dummy1 = d11 * F(1) + d12 * F(n) + d13 * F(n - 1) - d14 * F(2);
        const double dummy2 = d21 * F(1) + d22 * F(n) - d23 * F(n - 1) + d24 * F(2);
        for (int i = 3; i <= n - 3; ++i) F(i) = F(i) - dummy1 * Fv[i - 1] - dummy2 * Gv[i - 1];
        for (int i = 1; i <= 2; ++i) F(i) = F(i) - dummy1 * Fv[i - 1] - dummy2 * Gv[i - 1];
        for (int i = n - 2; i <= n; ++i) F(i) = F(i) - dummy1 * Fv[i - 1] - dummy2 * Gv[i - 1];
    }
#undef U
#undef F
#undef RI
#undef RB
#undef RT
#undef G
}

__global__ void __launch_bounds__(256) k_burgers_epilogue(double *__restrict__ out, const double *__restrict__ vel,
                                                          const double *__restrict__ d1, double nu, long long ntot) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ntot; i += stride) out[i] = nu * out[i] - vel[i] * d1[i];
}

__global__ void __launch_bounds__(256) k_fill(double *__restrict__ out, double v, long long ntot) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ntot; i += stride) out[i] = v;
}

// b(j,i) = a(i,j) for Fortran a(nra,nca) -> b(nca,nra)  (TLab_Transpose, utils/tlab_transpose.f90:14-82); 64x64 LDS tiles
__global__ void __launch_bounds__(256) k_transpose(const double *__restrict__ a, double *__restrict__ b, int nra, int nca) {
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long i0 = (long long)blockIdx.x * 64, j0 = (long long)blockIdx.y * 64;
    for (int jj = ty; jj < 64; jj += 4) {
        const long long i = i0 + tx, j = j0 + jj;
        if (i < nra && j < nca) tile[jj][tx] = a[i + (long long)nra * j];
    }
    __syncthreads();
    for (int ii = ty; ii < 64; ii += 4) {
        const long long j = j0 + tx, i = i0 + ii;
        if (i < nra && j < nca) b[j + (long long)nca * i] = tile[tx][ii];
    }
}

// ============================================================================================
// host-side launchers
// ============================================================================================
static inline int imin(long long a, long long b) { return (int)(a < b ? a : b); }

template <int M, int LV, int WPL, int LV2 = LV, int TPB = 256, bool CL = false, bool PER = false, int OCC = 1>
static hipError_t launch_xline_m(int mode, const XLineArgs &a_in, hipStream_t st) {
    static const int lb = [] { const char *e = getenv("TLAB_XLINE_LINE_BARRIERS"); return e ? atoi(e) : 1; }();
    XLineArgs a = a_in;
    a.line_barriers = lb;
    constexpr int P = 64 * WPL, LPB = TPB / 64 / WPL;
    const long long blocks_needed = (a.nlines + LPB - 1) / LPB;
    static const int gmul = [] { const char *e = getenv("TLAB_XLINE_GRID"); return e ? atoi(e) : 8; }();      // persistent workgroups per CU-slot (experiments)
    const int grid = imin(blocks_needed, 256 * gmul * 256 / TPB);
    if (mode < 1 || mode > 4) return hipErrorInvalidValue;
    auto tabbytes = [](int lv) { return lv == 1 ? (size_t)5 * M * P * sizeof(double) : lv == 2 ? (size_t)5 * M * P * sizeof(float) : (size_t)0; };
    const size_t redbytes = CL ? ((size_t)XRL * P + 6 * WPL) * sizeof(double) : 0;
    const size_t lds = (mode != MODE_P2 ? tabbytes(LV) + redbytes : 0) + (mode != MODE_P1 ? tabbytes(LV2) + redbytes : 0) +
                       (WPL > 1 ? ((size_t)LPB * 14 * WPL + LPB) * sizeof(double) : 0);      // exchange buffers + one arrival counter per line
    const double pts = (double)a.nlines * P * M;
    static const char *names[5] = {"", "k_xline<P1>", "k_xline<P2>", "k_xline<P2_P1>", "k_xline<BURGERS>"};
    const double bpp[5] = {0, 16, 16, 24, 24};
    double bytes = pts * (bpp[mode] + (a.acc ? 8 : 0) + (a.in0b ? 8 : 0) + (a.fq ? 24 : 0));
    if (mode == MODE_BURGERS) {   // velocity once + per field: operand (unless it is the velocity), result, previous result when accumulating
        bytes = pts * 8;
        for (int f = 0; f < a.nf; ++f) bytes += pts * ((a.fs[f] == a.in1 ? 0 : 8) + 8 + (a.acc ? 8 : 0) + (a.ffin[f] ? 8 : 0));
        if (a.fdiv) bytes += pts * 8;       // epilogues: updated scalar, x term of the pressure forcing
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xline<M, MODE_P1, LV, WPL, LV2, TPB, CL, PER, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xline<M, MODE_P2, LV, WPL, LV2, TPB, CL, PER, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xline<M, MODE_P2_P1, LV, WPL, LV2, TPB, CL, PER, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xline<M, MODE_BURGERS, LV, WPL, LV2, TPB, CL, PER, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    ProfScope ps(names[mode], st, bytes);
    switch (mode) {
    case MODE_P1: hipLaunchKernelGGL((k_xline<M, MODE_P1, LV, WPL, LV2, TPB, CL, PER, OCC>), dim3(grid), dim3(TPB), lds, st, a); break;
    case MODE_P2: hipLaunchKernelGGL((k_xline<M, MODE_P2, LV, WPL, LV2, TPB, CL, PER, OCC>), dim3(grid), dim3(TPB), lds, st, a); break;
    case MODE_P2_P1: hipLaunchKernelGGL((k_xline<M, MODE_P2_P1, LV, WPL, LV2, TPB, CL, PER, OCC>), dim3(grid), dim3(TPB), lds, st, a); break;
    case MODE_BURGERS: hipLaunchKernelGGL((k_xline<M, MODE_BURGERS, LV, WPL, LV2, TPB, CL, PER, OCC>), dim3(grid), dim3(TPB), lds, st, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_penta1(const PentaArgs &a, hipStream_t st) {
    const double pts = (double)a.g.nlines * a.g.n;
    ProfScope ps("k_penta1", st, pts * 16.0);
    hipLaunchKernelGGL(k_penta1, dim3((unsigned)((a.g.nlines + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

bool xline_supported(int n) { return n == 256 || n == 512 || n == 1024; }      // n = 2048, and n = 1024 on several waves: see xline_chunks (capi.cpp)

// chunks = 64: one wave per line (n = 64 M); 128 / 256: two / four waves per line with 8 rows per lane (periodic lines of 1024 / 2048 points
// whose tables pass xline_wide_ok; float-difference tables when they are lane-variant)
hipError_t launch_xline(int mode, int n, int chunks, bool lane_variant, const XLineArgs &a, hipStream_t st) {
    // table forms on several waves per line: doubles in LDS where they fit (1024 points: both systems, 80 KB; 2048 points: one system, 80 KB),
    // the second-derivative system of the two-system modes at 2048 points as float differences (40 KB) -- the reconstruction costs a scalar
    // load, a conversion and an add per coefficient, which is why doubles are preferred
    // Two-system modes keep the per-lane constants of the separator reduction in LDS (CL) where the tables leave room: 254 -> ~245 VGPRs without
    // the 60-110 AGPRs, i.e. two waves per SIMD, on 512 threads (4 lines share the tables): fused 4-field Burgers at 1024 points 3.09 -> 3.41 TB/s.
    // Per-chunk tables in LDS beat scalar loads of chunk 0's rows even where every chunk has the same rows (measured with exactly uniform
    // tables: 512 points 4.55 against 3.87 TB/s, 1024 points 3.41 against 3.05), so only the 2048-point two-system forms -- whose per-chunk
    // tables (120 KB) leave no room for the constants -- take the scalar-load form when the tables allow it (3.12 against 2.46 TB/s).
    // Tables made by FDM_CreatePlan never do: its numerically derived "uniform" spacing wanders by ~1e-13 (see tlab_fdm_plan::system).
    static const int tpb = [] { const char *e = getenv("TLAB_XLINE_TPB"); return e ? atoi(e) : 0; }();
    const bool one_sys = (mode == MODE_P1 || mode == MODE_P2);       // ~150 VGPRs with the constants in registers: three waves per SIMD anyway
    if (chunks == 128 && n == 1024) {
        if (one_sys) return launch_xline_m<8, 1, 2, 1>(mode, a, st);
        // (float differences for both systems, measured at 1024 points: 2.79 TB/s on 512 threads, 3.32 on two 256-thread workgroups per CU -- their
        // barriers are independent --, against 3.50 for the doubles below, whose 115 KB allow one workgroup per CU only)
        return tpb == 256 ? launch_xline_m<8, 1, 2, 1, 256, true>(mode, a, st) : launch_xline_m<8, 1, 2, 1, 512, true>(mode, a, st);
    }
    if (chunks == 256 && n == 2048 && !lane_variant && !one_sys)
        return tpb == 256 ? launch_xline_m<8, 0, 4, 0, 256, true>(mode, a, st) : launch_xline_m<8, 0, 4, 0, 512, true>(mode, a, st);
    // 2048 points, two systems, lane-variant tables: both as float differences (2 x 40 KB) + the constants in LDS (70 KB): 234 VGPRs, two lines per
    // 512-thread workgroup, two waves per SIMD -- 2.79 TB/s against 2.67 for doubles + floats with the constants in registers (one wave per SIMD);
    // TLAB_XLINE_FF=0 keeps that form
    static const int ff = [] { const char *e = getenv("TLAB_XLINE_FF"); return e ? atoi(e) : 1; }();
    if (chunks == 256 && n == 2048 && !one_sys && ff) return launch_xline_m<8, 2, 4, 2, 512, true>(mode, a, st);
    if (chunks == 256 && n == 2048)
        // one system (80 KB of tables): two lines per 512-thread workgroup share them, i.e. 8 waves per CU instead of 4
    {
        const bool one = (mode == MODE_P1 || mode == MODE_P2);
        if (tpb == 256) return one ? launch_xline_m<8, 1, 4, 1, 256>(mode, a, st) : launch_xline_m<8, 1, 4, 2, 256>(mode, a, st);
        if (tpb == 512) return one ? launch_xline_m<8, 1, 4, 1, 512>(mode, a, st) : launch_xline_m<8, 1, 4, 2, 512>(mode, a, st);
        // two systems: 256 VGPRs + 70-110 AGPRs at 256 threads; on 512 threads the kernel spills (1.2 TB/s), and with chunk 0's rows read from LDS
        // instead of scalar loads (no spills) both forms run at 2.2 TB/s against 2.46 (measured): 256 threads, scalar loads
        return one ? launch_xline_m<8, 1, 4, 1, 512>(mode, a, st) : launch_xline_m<8, 1, 4, 2, 256>(mode, a, st);
    }
    if (chunks != 64) return hipErrorInvalidValue;
    switch (n) {
    case 256:
        if (a.s1.periodic && lane_variant) return launch_xline_m<4, 1, 1, 1, 256, false, true>(mode, a, st);
        return lane_variant ? launch_xline_m<4, 1, 1>(mode, a, st) : launch_xline_m<4, 0, 1>(mode, a, st);
    case 512:         // (scalar loads of common rows are slower: see above)
    {
        // fused Burgers on 512-point periodic lines: constants of the reduction AND tables in LDS, re-read where they are used, two workgroups per CU
        // (256 VGPRs, no AGPRs, two waves per SIMD) instead of everything in 256 + 102 registers and one wave per SIMD: 3.09 -> 2.94 ms on a box in
        // its fast state, 3.26 -> 3.20 on one in its slow state (A/B, profiles/r05/xline_occupancy.txt); TLAB_XLINE_OCC=1 keeps the old form
        static const int occ = [] { const char *e = getenv("TLAB_XLINE_OCC"); return e ? atoi(e) : 3; }();
        if (a.s1.periodic && occ == 2 && mode == MODE_BURGERS) return launch_xline_m<8, 1, 1, 1, 256, false, true, 2>(mode, a, st);
        if (a.s1.periodic && occ == 3 && mode == MODE_BURGERS) return launch_xline_m<8, 1, 1, 1, 256, true, true, 2>(mode, a, st);
        if (a.s1.periodic && occ == 4 && mode == MODE_BURGERS) return launch_xline_m<8, 1, 1, 1, 512, true, true, 2>(mode, a, st);      // one 8-wave workgroup per CU sharing the tables
        if (a.s1.periodic && occ == 5) return launch_xline_m<8, 1, 1, 1, 256, true, true, 2>(mode, a, st);                           // ... the derivative modes as well
        return a.s1.periodic ? launch_xline_m<8, 1, 1, 1, 256, false, true>(mode, a, st) : launch_xline_m<8, 1, 1>(mode, a, st);
    }
    case 1024: return lane_variant ? launch_xline_m<16, 1, 1>(mode, a, st) : launch_xline_m<16, 0, 1>(mode, a, st);
    case 2048: return lane_variant ? launch_xline_m<32, 2, 1>(mode, a, st) : launch_xline_m<32, 0, 1>(mode, a, st);     // the caller checked xline_wide_ok
    }
    return hipErrorInvalidValue;
}

// chunk length (rows per wave) used by the register-tile kernel for a line length n (0 = unsupported).
// M = 32 keeps the kernel under 128 VGPRs (16 waves per workgroup possible); TLAB_RTILE_M overrides for experiments.
static int g_rtile_forced = -1;
void rtile_force_chunk(int m) { g_rtile_forced = m; }
int rtile_chunk(int n) {
    if (g_rtile_forced < 0) {
        const char *e = getenv("TLAB_RTILE_M");
        g_rtile_forced = e ? atoi(e) : 0;
    }
    const int forced = g_rtile_forced;
    if (forced == 16 || forced == 32 || forced == 64)
        if (n % forced == 0 && n / forced <= 16) return forced;
    if (n % 32 == 0 && n / 32 <= 16) return 32;
    if (n % 64 == 0 && n / 64 <= 16) return 64;
    if (n % 16 == 0 && n / 16 <= 16) return 16;
    return 0;
}

template <int M, int MAXT>
static hipError_t launch_rtile_m(int mode, int P, long long tiles, const RTileArgs &a, hipStream_t st) {
    const dim3 grid((unsigned)tiles), block(64 * P);
    const double pts = (double)a.g.nlines * a.g.n;
    const char *name = mode == MODE_P1 ? "k_rtile<P1>" : mode == MODE_P2 ? "k_rtile<P2>" : mode == MODE_P2_D1IN ? "k_rtile<P2_D1IN>" : "k_rtile<BURGERS_D1IN>";
    const double bpp = mode == MODE_P1 || mode == MODE_P2 ? 16 : mode == MODE_P2_D1IN ? 24 : 32;   // operand reads + writes of this launch
    ProfScope ps(a.fneu ? "k_rtile<P1+neumann final>" : name, st, pts * (bpp + (a.acc ? 8 : 0) + (a.in0b ? 8 : 0) + (a.fq ? 24 : 0)));
    switch (mode) {
    case MODE_P1: hipLaunchKernelGGL((k_rtile<M, MODE_P1, MAXT>), grid, block, 0, st, a); break;
    case MODE_P2: hipLaunchKernelGGL((k_rtile<M, MODE_P2, MAXT>), grid, block, 0, st, a); break;
    case MODE_P2_D1IN: hipLaunchKernelGGL((k_rtile<M, MODE_P2_D1IN, MAXT>), grid, block, 0, st, a); break;
    case MODE_BURGERS_D1IN: hipLaunchKernelGGL((k_rtile<M, MODE_BURGERS_D1IN, MAXT>), grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_rtile(int mode, const RTileArgs &a, hipStream_t st) {
    const int n = a.g.n;
    const int M = rtile_chunk(n);
    if (M == 0) return hipErrorInvalidValue;
    const int P = n / M;
    const long long tiles_inner = (a.g.lines_inner + 63) / 64;
    const long long tiles = tiles_inner * (a.g.nlines / a.g.lines_inner);
    if (M == 64) return (P <= 8) ? launch_rtile_m<64, 512>(mode, P, tiles, a, st) : launch_rtile_m<64, 1024>(mode, P, tiles, a, st);
    if (M == 32) return (P <= 8) ? launch_rtile_m<32, 512>(mode, P, tiles, a, st) : launch_rtile_m<32, 1024>(mode, P, tiles, a, st);
    return (P <= 8) ? launch_rtile_m<16, 512>(mode, P, tiles, a, st) : launch_rtile_m<16, 1024>(mode, P, tiles, a, st);
}

hipError_t launch_generic(bool sym, const GenericArgs &a, hipStream_t st) {
    const int grid = (int)((a.g.nlines + 255) / 256);
    ProfScope ps("k_generic", st, (double)a.g.nlines * a.g.n * 16);
    if (sym) hipLaunchKernelGGL((k_generic<true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_generic<false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_burgers_epilogue(double *out, const double *vel, const double *d1, double nu, long long ntot, hipStream_t st) {
    const int grid = imin((ntot + 255) / 256, 256 * 8);
    ProfScope ps("k_burgers_epilogue", st, (double)ntot * 32);
    hipLaunchKernelGGL(k_burgers_epilogue, dim3(grid), dim3(256), 0, st, out, vel, d1, nu, ntot);
    return hipGetLastError();
}

hipError_t launch_fill(double *out, double v, long long ntot, hipStream_t st) {
    const int grid = imin((ntot + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st, out, v, ntot);
    return hipGetLastError();
}

hipError_t launch_transpose(const double *a, double *b, int nra, int nca, hipStream_t st) {
    dim3 grid((nra + 63) / 64, (nca + 63) / 64);
    ProfScope ps("k_transpose", st, (double)nra * nca * 16);
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, a, b, nra, nca);
    return hipGetLastError();
}

}  // namespace tlab
