// k_fftz : complex-to-complex FFT along a STRIDED index, lines contiguous across the batch (the z-transform of the spectral array:
// element (line l, point k) at l + nlines*k; OPR_Fourier_Z_Forward/Backward, opr_fourier.f90:343-428, dfftw_plan_many_dft :111-119).
//
// rocFFT serves this layout with a row kernel behind transposes when it is given as a 1-D strided batch (measured on the kx-pencils of
// the multi-GPU Poisson solver: 0.26 ms for 138 MB in + 138 MB out = 1.06 TB/s; its 2-D plans use a column kernel at 2.9 TB/s but
// also transform x).  Here a workgroup owns T = 8 neighbouring lines (128-B rows): Stockham autosort radix-8 passes (+ one radix-4 or
// radix-2 pass), the first pass reads HBM, the last one writes it, the passes in between exchange through one LDS buffer.  One read and
// one write of the array; twiddles from a table made in long double on the host.  Unnormalised in both directions like FFTW.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "fftz.hpp"
#include "profile.hpp"

namespace tlab {

struct cd {
    double x, y;
};
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// multiply by exp(SGN * i * pi/2) = SGN * i
template <int SGN>
__device__ __forceinline__ cd mul_pm_i(cd a) {
    return SGN > 0 ? cd{-a.y, a.x} : cd{a.y, -a.x};
}

// R-point DFT in place, kernel exp(SGN * 2 pi i j k / R), natural order
template <int SGN>
__device__ __forceinline__ void dft2(cd &a, cd &b) {
    const cd t = a - b;
    a = a + b;
    b = t;
}
template <int SGN>
__device__ __forceinline__ void dft4(cd &a0, cd &a1, cd &a2, cd &a3) {
    const cd t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_pm_i<SGN>(a1 - a3);
    a0 = t0 + t2; a1 = t1 + t3; a2 = t0 - t2; a3 = t1 - t3;
}
template <int SGN>
__device__ __forceinline__ void dft8(cd (&v)[8]) {
    cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4<SGN>(e0, e1, e2, e3);
    dft4<SGN>(o0, o1, o2, o3);
    const double h = 0.70710678118654752440;
    // W8^k = exp(SGN * 2 pi i k / 8)
    const cd w1 = {h, SGN * h}, w3 = {-h, SGN * h};
    o1 = cmul(o1, w1);
    o2 = mul_pm_i<SGN>(o2);
    o3 = cmul(o3, w3);
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

// lengths, radices and their running products are powers of two: shifts and masks instead of integer division (the compiler cannot know)
__device__ __forceinline__ int ilog2(int v) { return 31 - __builtin_clz(v); }

struct FftzArgs {
    const double2 *in;
    double2 *out;
    const double2 *tw;      // exp(-2 pi i k / n), k < n  (conjugated for the backward transform)
    long long nlines;       // = stride between consecutive points of a line
    int n;
    int npass;
    int radix[8];
};

// one Stockham pass for work item (column t, index j): radix R, Ns = product of the radices done before
template <int R, int SGN, int T>
__device__ __forceinline__ void fftz_pass(const FftzArgs &a, int t, int j, int Ns, bool first, bool last, long long col, bool colok, cd *lds) {
    const int n = a.n, m = n / R;
    cd v[R];
    if (first) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double2 w = make_double2(0.0, 0.0);
            if (colok) w = a.in[col + (long long)(j + r * m) * a.nlines];
            v[r] = {w.x, w.y};
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = lds[(j + r * m) * T + t];
    }
    // twiddles exp(SGN 2 pi i (j % Ns) r / (Ns R))
    const int lNs = ilog2(Ns), lR = ilog2(R), ln = ilog2(n);
    const int k0 = (j & (Ns - 1)) << (ln - lNs - lR);
#pragma unroll
    for (int r = 1; r < R; ++r) {
        const double2 w = a.tw[(k0 * r) & (n - 1)];
        v[r] = cmul(v[r], cd{w.x, SGN > 0 ? -w.y : w.y});
    }
    if (R == 8) dft8<SGN>(reinterpret_cast<cd(&)[8]>(v));
    else if (R == 4) dft4<SGN>(v[0], v[1 % R], v[2 % R], v[3 % R]);
    else dft2<SGN>(v[0], v[1 % R]);
    const int j0 = ((j >> lNs) << (lNs + lR)) + (j & (Ns - 1));
    if (!first) __syncthreads();          // every item of this pass has read its inputs from the buffer
    if (last) {
        if (colok) {
#pragma unroll
            for (int r = 0; r < R; ++r) a.out[col + (long long)(j0 + r * Ns) * a.nlines] = make_double2(v[r].x, v[r].y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) lds[(j0 + r * Ns) * T + t] = v[r];
        __syncthreads();
    }
}

// n = 8^a * {1,2,4} (16 .. 2048); every pass has n/R <= n/2 items per column; blockDim = T * n / 8 (all radix-8 passes: one item per thread; a final
// radix-4 / radix-2 pass has 2 / 4 items per thread).  The barriers inside fftz_pass are reached uniformly: the loops below have the
// same trip count for every thread.
template <int SGN, int T>
__global__ void __launch_bounds__(1024) k_fftz(FftzArgs a) {
    extern __shared__ double2 fz_lds[];
    cd *lds = reinterpret_cast<cd *>(fz_lds);
    const int t = threadIdx.x % T;
    const int jt = threadIdx.x / T;              // 0 .. n/8 - 1
    const long long col = (long long)blockIdx.x * T + t;
    const bool colok = col < a.nlines;
    int Ns = 1;
    for (int p = 0; p < a.npass; ++p) {
        const int R = a.radix[p];
        const bool first = p == 0, last = p == a.npass - 1;
        const int items = 8 / R;                  // items of this pass per thread
        if (R == 8) {
            fftz_pass<8, SGN, T>(a, t, jt, Ns, first, last, col, colok, lds);
        } else {
            // a thread owns items jt + q * (n/8): reads of all its items first, then one barrier, then the writes -> unroll by hand
            const int m8 = a.n / 8;
            if (R == 4) {
                cd keep[2][4];
                int j0s[2];
                for (int q = 0; q < 2; ++q) {
                    const int j = jt + q * m8, m = a.n / 4;
                    for (int r = 0; r < 4; ++r) {
                        if (first) {
                            double2 w = make_double2(0.0, 0.0);
                            if (colok) w = a.in[col + (long long)(j + r * m) * a.nlines];
                            keep[q][r] = {w.x, w.y};
                        } else {
                            keep[q][r] = lds[(j + r * m) * T + t];
                        }
                    }
                    const int k0 = (j & (Ns - 1)) * (a.n >> (ilog2(Ns) + 2));
                    for (int r = 1; r < 4; ++r) {
                        const double2 w = a.tw[(k0 * r) & (a.n - 1)];
                        keep[q][r] = cmul(keep[q][r], cd{w.x, SGN > 0 ? -w.y : w.y});
                    }
                    dft4<SGN>(keep[q][0], keep[q][1], keep[q][2], keep[q][3]);
                    j0s[q] = ((j >> ilog2(Ns)) << (ilog2(Ns) + 2)) + (j & (Ns - 1));
                }
                if (!first) __syncthreads();
                for (int q = 0; q < 2; ++q)
                    for (int r = 0; r < 4; ++r) {
                        if (last) {
                            if (colok) a.out[col + (long long)(j0s[q] + r * Ns) * a.nlines] = make_double2(keep[q][r].x, keep[q][r].y);
                        } else {
                            lds[(j0s[q] + r * Ns) * T + t] = keep[q][r];
                        }
                    }
                if (!last) __syncthreads();
            } else {
                cd keep[4][2];
                int j0s[4];
                for (int q = 0; q < 4; ++q) {
                    const int j = jt + q * m8, m = a.n / 2;
                    for (int r = 0; r < 2; ++r) {
                        if (first) {
                            double2 w = make_double2(0.0, 0.0);
                            if (colok) w = a.in[col + (long long)(j + r * m) * a.nlines];
                            keep[q][r] = {w.x, w.y};
                        } else {
                            keep[q][r] = lds[(j + r * m) * T + t];
                        }
                    }
                    const int k0 = (j & (Ns - 1)) * (a.n >> (ilog2(Ns) + 1));
                    const double2 w = a.tw[k0 & (a.n - 1)];
                    keep[q][1] = cmul(keep[q][1], cd{w.x, SGN > 0 ? -w.y : w.y});
                    dft2<SGN>(keep[q][0], keep[q][1]);
                    j0s[q] = ((j >> ilog2(Ns)) << (ilog2(Ns) + 1)) + (j & (Ns - 1));
                }
                if (!first) __syncthreads();
                for (int q = 0; q < 4; ++q)
                    for (int r = 0; r < 2; ++r) {
                        if (last) {
                            if (colok) a.out[col + (long long)(j0s[q] + r * Ns) * a.nlines] = make_double2(keep[q][r].x, keep[q][r].y);
                        } else {
                            lds[(j0s[q] + r * Ns) * T + t] = keep[q][r];
                        }
                    }
                if (!last) __syncthreads();
            }
            (void)items;
        }
        Ns *= R;
    }
}

// ================================================================================================
// k_fftx_r2c : real-to-complex FFT of contiguous lines (OPR_Fourier_X_Forward: dfftw_plan_many_dft_r2c, opr_fourier.f90:163-166), in ONE pass
// over the data.  rocFFT does this transform as a half-length complex FFT + a separate even/odd post-processing kernel (0.34 + 0.38 ms at
// 512^3, two reads and two writes of the field); here the n reals of a line are read as m = n/2 complex numbers z_j = x_2j + i x_2j+1, the
// m-point Stockham transform runs through LDS exactly like k_fftz (radix-8 passes + one radix-4 / radix-2 pass, m/8 threads per line), and the
// same threads then form X_k = E_k + w^k O_k, X_(m-k) = conj(E_k - w^k O_k) with E = (Z_k + conj Z_(m-k))/2, O = -i (Z_k - conj Z_(m-k))/2
// from the LDS copy and write the n/2+1 outputs of the line.  Unnormalised, exp(-i ...), like FFTW.
// ================================================================================================
struct FftxArgs {
    const double *in;
    double2 *out;
    const double2 *tw;      // exp(-2 pi i k / n), k < n
    long long nlines;
    int n, m;               // real length, m = n/2
    int npass;
    int radix[8];
    int tl, lines;          // threads per line (m/8), lines per workgroup
    // the complex side in the slab <-> kx-pencil PACK layout instead of (m+1) contiguous values per line (tlab_pencil_repack_blocks folded into the
    // transform): element (line, kx) at kxoff[kx] + line * kxw[kx] complex values; NULL: contiguous
    const long long *kxoff;
    const int *kxw;
};

template <int R>
__device__ __forceinline__ void fftx_pass(const FftxArgs &a, const double2 *__restrict__ zin, bool ok, cd *lds, int jt, int Ns, bool first) {
    constexpr int Q = 8 / R;                 // items of this pass per thread
    const int M = a.m, m = M / R;
    cd v[Q][R];
    int j0s[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = jt + q * a.tl;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (first) {
                double2 w = make_double2(0.0, 0.0);
                if (ok) w = zin[j + r * m];
                v[q][r] = {w.x, w.y};
            } else {
                v[q][r] = lds[j + r * m];
            }
        }
        const int k0 = (j & (Ns - 1)) << (ilog2(M) - ilog2(Ns) - ilog2(R));
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const double2 w = a.tw[2 * ((k0 * r) & (M - 1))];      // exp(-2 pi i (k0 r) / m) from the table of n = 2 m
            v[q][r] = cmul(v[q][r], cd{w.x, w.y});
        }
        if (R == 8) dft8<-1>(reinterpret_cast<cd(&)[8]>(v[q]));
        else if (R == 4) dft4<-1>(v[q][0], v[q][1 % R], v[q][2 % R], v[q][3 % R]);
        else dft2<-1>(v[q][0], v[q][1 % R]);
        j0s[q] = ((j >> ilog2(Ns)) << (ilog2(Ns) + ilog2(R))) + (j & (Ns - 1));
    }
    if (!first) __syncthreads();          // every item of this pass has read its inputs from the buffer
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) lds[j0s[q] + r * Ns] = v[q][r];
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_fftx_r2c(FftxArgs a) {
    extern __shared__ double2 fx_lds[];
    const int jt = threadIdx.x & (a.tl - 1), ll = threadIdx.x >> ilog2(a.tl);      // tl = m/8: a power of two
    const long long line = (long long)blockIdx.x * a.lines + ll;
    const bool ok = line < a.nlines;
    cd *lds = reinterpret_cast<cd *>(fx_lds) + (size_t)ll * a.m;
    const double2 *zin = reinterpret_cast<const double2 *>(a.in + (ok ? line : 0) * a.n);
    int Ns = 1;
    for (int p = 0; p < a.npass; ++p) {
        const int R = a.radix[p];
        if (R == 8) fftx_pass<8>(a, zin, ok, lds, jt, Ns, p == 0);
        else if (R == 4) fftx_pass<4>(a, zin, ok, lds, jt, Ns, p == 0);
        else fftx_pass<2>(a, zin, ok, lds, jt, Ns, p == 0);
        Ns *= R;
    }
    if (!ok) return;
    double2 *out = a.out + line * (a.m + 1);
    const int M = a.m;
    if (a.kxw != nullptr) {      // same values, scattered into the pack buffer
        for (int k = jt; k <= M / 2; k += a.tl) {
            const cd A = lds[k], Zc = lds[(M - k) & (M - 1)];
            const cd B = {Zc.x, -Zc.y};
            const cd E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
            const cd D = A - B;
            const cd O = {0.5 * D.y, -0.5 * D.x};
            const double2 w = a.tw[k];
            const cd T = cmul(cd{w.x, w.y}, O);
            a.out[a.kxoff[k] + line * a.kxw[k]] = make_double2(E.x + T.x, E.y + T.y);
            if (k != M - k) a.out[a.kxoff[M - k] + line * a.kxw[M - k]] = make_double2(E.x - T.x, -(E.y - T.y));
        }
        return;
    }
    for (int k = jt; k <= M / 2; k += a.tl) {
        const cd A = lds[k], Zc = lds[(M - k) & (M - 1)];
        const cd B = {Zc.x, -Zc.y};
        const cd E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
        const cd D = A - B;
        const cd O = {0.5 * D.y, -0.5 * D.x};
        const double2 w = a.tw[k];
        const cd T = cmul(cd{w.x, w.y}, O);
        out[k] = make_double2(E.x + T.x, E.y + T.y);
        if (k != M - k) out[M - k] = make_double2(E.x - T.x, -(E.y - T.y));
    }
}

// k_fftx_c2r : the inverse (dfftw_plan_many_dft_c2r, opr_fourier.f90:167-170; unnormalised), same structure backwards: the threads of a line form
// Z_k = (X_k + conj X_(m-k)) + i (X_k - conj X_(m-k)) conj(w^k) in LDS (the imaginary parts of X_0 and X_m are ignored, like FFTW), run the m-point
// transform with exp(+i ...), and x_2j + i x_2j+1 = z_j goes out as the line of n reals.
// FINAL: the line is not written; it is the pressure-gradient component g of k_final_update (pointwise.hip), whose arithmetic follows in the same
// registers -- h = h - g, zero on the wall rows, q += dte h, h *= kco -- so that dp/dy is never stored (rhs.cpp: the v equation after OPR_Poisson).
struct FftxFinal {
    double *q, *h;
    double dte, kco;
    int scale, ny;
};

template <int R>
__device__ __forceinline__ void fftx_pass_inv(const FftxArgs &a, cd *lds, int jt, int Ns) {
    constexpr int Q = 8 / R;
    const int M = a.m, m = M / R;
    cd v[Q][R];
    int j0s[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = jt + q * a.tl;
#pragma unroll
        for (int r = 0; r < R; ++r) v[q][r] = lds[j + r * m];
        const int k0 = (j & (Ns - 1)) << (ilog2(M) - ilog2(Ns) - ilog2(R));
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const double2 w = a.tw[2 * ((k0 * r) & (M - 1))];
            v[q][r] = cmul(v[q][r], cd{w.x, -w.y});
        }
        if (R == 8) dft8<+1>(reinterpret_cast<cd(&)[8]>(v[q]));
        else if (R == 4) dft4<+1>(v[q][0], v[q][1 % R], v[q][2 % R], v[q][3 % R]);
        else dft2<+1>(v[q][0], v[q][1 % R]);
        j0s[q] = ((j >> ilog2(Ns)) << (ilog2(Ns) + ilog2(R))) + (j & (Ns - 1));
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) lds[j0s[q] + r * Ns] = v[q][r];
    __syncthreads();
}

template <bool FINAL>
__global__ void __launch_bounds__(256) k_fftx_c2r(FftxArgs a, FftxFinal fin) {
    extern __shared__ double2 fx_lds[];
    const int jt = threadIdx.x & (a.tl - 1), ll = threadIdx.x >> ilog2(a.tl);      // tl = m/8: a power of two
    const long long line = (long long)blockIdx.x * a.lines + ll;
    const bool ok = line < a.nlines;
    cd *lds = reinterpret_cast<cd *>(fx_lds) + (size_t)ll * a.m;
    const int M = a.m;
    if (ok) {
        const double2 *X = reinterpret_cast<const double2 *>(a.in) + line * (M + 1);
        const double2 *XP = reinterpret_cast<const double2 *>(a.in);
        for (int k = jt; k <= M / 2; k += a.tl) {
            double2 xa, xb;
            if (a.kxw != nullptr) { xa = XP[a.kxoff[k] + line * a.kxw[k]]; xb = XP[a.kxoff[M - k] + line * a.kxw[M - k]]; }
            else { xa = X[k]; xb = X[M - k]; }
            if (k == 0) { xa.y = 0.0; xb.y = 0.0; }
            const cd E = {xa.x + xb.x, xa.y - xb.y};                 // X_k + conj X_(m-k)
            const cd D = {xa.x - xb.x, xa.y + xb.y};                 // X_k - conj X_(m-k)
            const double2 w = a.tw[k];
            const cd O = cmul(D, cd{w.x, -w.y});
            if (k < M) lds[k % M] = {E.x - O.y, E.y + O.x};          // E + i O
            if (k != 0 && k != M - k) lds[M - k] = {E.x + O.y, -E.y + O.x};      // conj(E) + i conj(O)
        }
    }
    __syncthreads();
    int Ns = 1;
    for (int p = 0; p < a.npass; ++p) {
        const int R = a.radix[p];
        if (R == 8) fftx_pass_inv<8>(a, lds, jt, Ns);
        else if (R == 4) fftx_pass_inv<4>(a, lds, jt, Ns);
        else fftx_pass_inv<2>(a, lds, jt, Ns);
        Ns *= R;
    }
    if (!ok) return;
    if (!FINAL) {
        double2 *out = a.out + line * M;
        for (int j = jt; j < M; j += a.tl) out[j] = make_double2(lds[j].x, lds[j].y);
    } else {
        const int jy = a.nlines < (1LL << 31) ? (int)((unsigned)line % (unsigned)fin.ny) : (int)(line % fin.ny);      // (32-bit where it fits: cheaper)
        const bool wall = jy == 0 || jy == fin.ny - 1;
        double2 *q2 = reinterpret_cast<double2 *>(fin.q) + line * M, *h2 = reinterpret_cast<double2 *>(fin.h) + line * M;
        for (int j = jt; j < M; j += a.tl) {
            const double2 hv0 = h2[j], qv = q2[j];
            double hx = hv0.x - lds[j].x, hy = hv0.y - lds[j].y;
            if (wall) { hx = 0.0; hy = 0.0; }
            q2[j] = make_double2(qv.x + fin.dte * hx, qv.y + fin.dte * hy);
            h2[j] = fin.scale ? make_double2(fin.kco * hx, fin.kco * hy) : make_double2(hx, hy);
        }
    }
}

bool FftxPlan::supported(int n) {
    if (n % 2 || n < 128 || n > 2048) return false;
    int m = n / 2;
    while (m % 8 == 0) m /= 8;
    return m == 1 || m == 2 || m == 4;
}

FftxPlan::FftxPlan(int n_, long long nlines_) : n(n_), nlines(nlines_) {
    if (!supported(n)) throw std::runtime_error("FftxPlan: unsupported length");
    int m = n / 2;
    while (m % 8 == 0) { radix.push_back(8); m /= 8; }
    if (m > 1) radix.push_back(m);
    std::vector<double> tw((size_t)2 * n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        tw[2 * k] = (double)cosl(two_pi * k / n);
        tw[2 * k + 1] = (double)(-sinl(two_pi * k / n));
    }
    if (hipMalloc((void **)&d_tw, tw.size() * sizeof(double)) != hipSuccess) throw std::runtime_error("FftxPlan: hipMalloc");
    if (hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("FftxPlan: hipMemcpy");
}

FftxPlan::~FftxPlan() {
    if (d_tw) (void)hipFree(d_tw);
}

void FftxPlan::exec(const double *in, double *out, hipStream_t st, const long long *kxoff, const int *kxw) const {
    FftxArgs a{};
    a.kxoff = kxoff; a.kxw = kxw;
    a.in = in;
    a.out = reinterpret_cast<double2 *>(out);
    a.tw = reinterpret_cast<const double2 *>(d_tw);
    a.nlines = nlines; a.n = n; a.m = n / 2; a.npass = (int)radix.size();
    for (size_t p = 0; p < radix.size(); ++p) a.radix[p] = radix[p];
    a.tl = a.m / 8;
    a.lines = a.tl >= 256 ? 1 : 256 / a.tl;
    const unsigned grid = (unsigned)((nlines + a.lines - 1) / a.lines);
    const size_t lds = (size_t)a.lines * a.m * sizeof(double2);
    ProfScope ps("k_fftx_r2c", st, (double)nlines * (n * 8.0 + (a.m + 1) * 16.0));
    hipLaunchKernelGGL(k_fftx_r2c, dim3(grid), dim3((unsigned)(a.tl * a.lines)), lds, st, a);
    if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_fftx_r2c launch failed");
}

bool FftzPlan::supported(int n) {
    if (n < 16 || n > 2048) return false;
    int m = n;
    while (m % 8 == 0) m /= 8;
    return m == 1 || m == 2 || m == 4;
}

FftzPlan::FftzPlan(int n_, long long nlines_) : n(n_), nlines(nlines_) {
    if (!supported(n)) throw std::runtime_error("FftzPlan: unsupported length");
    int m = n;
    while (m % 8 == 0) { radix.push_back(8); m /= 8; }
    if (m > 1) radix.push_back(m);
    std::vector<double> tw((size_t)2 * n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        tw[2 * k] = (double)cosl(two_pi * k / n);
        tw[2 * k + 1] = (double)(-sinl(two_pi * k / n));
    }
    if (hipMalloc((void **)&d_tw, tw.size() * sizeof(double)) != hipSuccess) throw std::runtime_error("FftzPlan: hipMalloc");
    if (hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("FftzPlan: hipMemcpy");
}

FftzPlan::~FftzPlan() {
    if (d_tw) (void)hipFree(d_tw);
}

template <int T>
static void fftz_launch(int dir, const FftzArgs &a, hipStream_t st) {
    const unsigned grid = (unsigned)((a.nlines + T - 1) / T);
    const unsigned block = (unsigned)(T * a.n / 8);
    const size_t lds = (size_t)a.n * T * sizeof(double2);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fftz<-1, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fftz<+1, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        attr_done = true;
    }
    if (dir > 0) hipLaunchKernelGGL((k_fftz<-1, T>), dim3(grid), dim3(block), lds, st, a);      // forward: exp(-i ...)
    else hipLaunchKernelGGL((k_fftz<+1, T>), dim3(grid), dim3(block), lds, st, a);
}

void FftzPlan::exec(int dir, const double *in, double *out, hipStream_t st) const {
    FftzArgs a{};
    a.in = reinterpret_cast<const double2 *>(in);
    a.out = reinterpret_cast<double2 *>(out);
    a.tw = reinterpret_cast<const double2 *>(d_tw);
    a.nlines = nlines; a.n = n; a.npass = (int)radix.size();
    for (size_t p = 0; p < radix.size(); ++p) a.radix[p] = radix[p];
    ProfScope ps("k_fftz", st, (double)nlines * n * 32.0);
    static const int t16 = [] { const char *e = getenv("TLAB_FFTZ_T"); return e ? atoi(e) : 8; }();      // experiment: 16 lines (256-B rows), 1024 threads, one workgroup per CU
    if (n <= 512 && t16 == 16) fftz_launch<16>(dir, a, st);
    else if (n <= 1024) fftz_launch<8>(dir, a, st);       // 8 neighbouring lines per workgroup (128-B rows), n * 8 * 16 B of LDS
    else fftz_launch<4>(dir, a, st);                 // n = 2048: 4 lines (64-B rows) to stay within 1024 threads / 128 KB
    if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_fftz launch failed");
}

void FftxPlan::launch_inverse(const double *in, double *out, const double *q, const double *h, double dte, double kco, int scale, int ny,
                              hipStream_t st, const long long *kxoff, const int *kxw) const {
    FftxArgs a{};
    a.kxoff = kxoff; a.kxw = kxw;
    a.in = in;
    a.out = reinterpret_cast<double2 *>(out);
    a.tw = reinterpret_cast<const double2 *>(d_tw);
    a.nlines = nlines; a.n = n; a.m = n / 2; a.npass = (int)radix.size();
    for (size_t p = 0; p < radix.size(); ++p) a.radix[p] = radix[p];
    a.tl = a.m / 8;
    a.lines = a.tl >= 256 ? 1 : 256 / a.tl;
    const unsigned grid = (unsigned)((nlines + a.lines - 1) / a.lines);
    const size_t lds = (size_t)a.lines * a.m * sizeof(double2);
    FftxFinal f{const_cast<double *>(q), const_cast<double *>(h), dte, kco, scale, ny};
    if (q) {
        ProfScope ps("k_fftx_c2r<final>", st, (double)nlines * ((a.m + 1) * 16.0 + n * 32.0));
        hipLaunchKernelGGL(k_fftx_c2r<true>, dim3(grid), dim3((unsigned)(a.tl * a.lines)), lds, st, a, f);
    } else {
        ProfScope ps("k_fftx_c2r", st, (double)nlines * ((a.m + 1) * 16.0 + n * 8.0));
        hipLaunchKernelGGL(k_fftx_c2r<false>, dim3(grid), dim3((unsigned)(a.tl * a.lines)), lds, st, a, f);
    }
    if (hipGetLastError() != hipSuccess) throw std::runtime_error("k_fftx_c2r launch failed");
}

}  // namespace tlab
