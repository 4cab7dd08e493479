// Pointwise kernels of the RHS assembly and of the low-storage Runge-Kutta update:
// tools/dns/rhs_global_incompressible_1.f90:106-112,197-201,255-260,348-352,373-375 and tools/dns/time.f90:272-297,645-664.
// All are streaming kernels with 16-B accesses per lane, grid-strided over at most 2048 workgroups.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"
#include "profile.hpp"

namespace tlab {

static inline int pw_grid(long long n2) {
    const long long b = (n2 + 255) / 256;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

// h += a + b + c            (hq(ij,1) = hq(ij,1) + tmp1(ij) + tmp7(ij) + tmp8(ij))
__global__ void __launch_bounds__(256) k_add3(double *__restrict__ h, const double *__restrict__ a, const double *__restrict__ b,
                                              const double *__restrict__ c, long long n) {
    const long long n2 = n >> 1, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 hv = reinterpret_cast<double2 *>(h)[i];
        const double2 av = reinterpret_cast<const double2 *>(a)[i], bv = reinterpret_cast<const double2 *>(b)[i],
                      cv = reinterpret_cast<const double2 *>(c)[i];
        hv.x = hv.x + av.x + bv.x + cv.x;
        hv.y = hv.y + av.y + bv.y + cv.y;
        reinterpret_cast<double2 *>(h)[i] = hv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) h[n - 1] = h[n - 1] + a[n - 1] + b[n - 1] + c[n - 1];
}

// o1 = h1 + q1*s ; o2 = h2 + q2*s ; o3 = h3 + q3*s      (tmp = hq + q/dte, rhs_global_incompressible_1.f90:197-201)
__global__ void __launch_bounds__(256) k_axpy3(double *__restrict__ o1, double *__restrict__ o2, double *__restrict__ o3,
                                               const double *__restrict__ h1, const double *__restrict__ h2, const double *__restrict__ h3,
                                               const double *__restrict__ q1, const double *__restrict__ q2, const double *__restrict__ q3,
                                               double s, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        o1[i] = h1[i] + q1[i] * s;
        o2[i] = h2[i] + q2[i] * s;
        o3[i] = h3[i] + q3[i] * s;
    }
}
// ... and the anelastic density weight on top, o = (h + q s) * w(j)  (Thermo_Anelastic_WEIGHT_INPLACE with rbackground, :211-214): k_axpy3 + 3 x k_weight_y in one pass
__global__ void __launch_bounds__(256) k_axpy3w(double *__restrict__ o1, double *__restrict__ o2, double *__restrict__ o3,
                                                const double *__restrict__ h1, const double *__restrict__ h2, const double *__restrict__ h3,
                                                const double *__restrict__ q1, const double *__restrict__ q2, const double *__restrict__ q3,
                                                double s, const double *__restrict__ w, int nx, int ny, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double wj = w[(i / nx) % ny];
        const double t1 = h1[i] + q1[i] * s, t2 = h2[i] + q2[i] * s, t3 = h3[i] + q3[i] * s;
        o1[i] = t1 * wj;
        o2[i] = t2 * wj;
        o3[i] = t3 * wj;
    }
}

// a = a + b + c             (tmp1 = tmp1 + tmp2 + tmp3, :257-259)
__global__ void __launch_bounds__(256) k_sum3(double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, long long n) {
    const long long n2 = n >> 1, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 av = reinterpret_cast<double2 *>(a)[i];
        const double2 bv = reinterpret_cast<const double2 *>(b)[i], cv = reinterpret_cast<const double2 *>(c)[i];
        av.x = av.x + bv.x + cv.x;
        av.y = av.y + bv.y + cv.y;
        reinterpret_cast<double2 *>(a)[i] = av;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) a[n - 1] = a[n - 1] + b[n - 1] + c[n - 1];
}

// h1 -= a ; h2 -= b ; h3 -= c     (hq = hq - grad p, :348-352)
__global__ void __launch_bounds__(256) k_sub3(double *__restrict__ h1, double *__restrict__ h2, double *__restrict__ h3,
                                              const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        h1[i] = h1[i] - a[i];
        h2[i] = h2[i] - b[i];
        h3[i] = h3[i] - c[i];
    }
}

// q = q + dte*h  and then  h = kco*h  (time.f90:645-664 and :272-297; kco == 1 leaves h untouched as after the last substep)
__global__ void __launch_bounds__(256) k_rk_update(double *__restrict__ q, double *__restrict__ h, double dte, double kco, int scale,
                                                   long long n) {
    const long long n2 = n >> 1, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 qv = reinterpret_cast<double2 *>(q)[i], hv = reinterpret_cast<double2 *>(h)[i];
        qv.x = qv.x + dte * hv.x;
        qv.y = qv.y + dte * hv.y;
        reinterpret_cast<double2 *>(q)[i] = qv;
        if (scale) {
            hv.x = kco * hv.x;
            hv.y = kco * hv.y;
            reinterpret_cast<double2 *>(h)[i] = hv;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        q[n - 1] = q[n - 1] + dte * h[n - 1];
        if (scale) h[n - 1] = kco * h[n - 1];
    }
}

// hb(i,k) = f(i,1,k) ; ht(i,k) = f(i,ny,k)    (BcsFlowJmin%ref(:,:,2) = p_bcs(:,1,:), :279-280)
__global__ void __launch_bounds__(256) k_get_wall_planes(const double *__restrict__ f, double *__restrict__ hb, double *__restrict__ ht,
                                                          int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)(i % nx);
    const long long k = i / nx;
    hb[i] = f[ix + (long long)nx * (0 + (long long)ny * k)];
    ht[i] = f[ix + (long long)nx * ((ny - 1) + (long long)ny * k)];
}

// f(:,1,:) = vb ; f(:,ny,:) = vt   (p_bcs(:,1,:) = BcsFlowJmin%ref = 0 for Dirichlet walls, :373-375)
__global__ void __launch_bounds__(256) k_fill_wall_planes(double *__restrict__ f, double vb, double vt, int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)(i % nx);
    const long long k = i / nx;
    f[ix + (long long)nx * (0 + (long long)ny * k)] = vb;
    f[ix + (long long)nx * ((ny - 1) + (long long)ny * k)] = vt;
}

// f(:,1,:) = pb(:,:) ; f(:,ny,:) = pt(:,:)   (p_bcs(:,1,:) = BcsFlowJmin%ref(:,:,iq), :373-375; a null plane stands for ref = 0)
__global__ void __launch_bounds__(256) k_set_wall_planes_opt(double *__restrict__ f, const double *__restrict__ pb, const double *__restrict__ pt,
                                                         int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)(i % nx);
    const long long k = i / nx;
    f[ix + (long long)nx * (0 + (long long)ny * k)] = pb ? pb[i] : 0.0;
    f[ix + (long long)nx * ((ny - 1) + (long long)ny * k)] = pt ? pt[i] : 0.0;
}

// BOUNDARY_BCS_NEUMANN_Y (tools/dns/boundary_bcs.f90:368-473), last step: with du = the y-derivative of u computed under the
// Neumann variant ibc (zero at the chosen walls), the wall value that makes du/dy vanish there is
//   bcs_hb = u(2) r_b(1,.) + u(3) r_b(1,.) + u(4) r_b(1,.) + lu(1, ip+idl+1) du(2)       (fdm_matmul.f90:384 + boundary_bcs.f90:452)
//   bcs_ht = u(n-3) r_t(.,.) + u(n-2) r_t(.,.) + u(n-1) r_t(.,.) + lu(n, ip+idl-1) du(n-1)   (fdm_matmul.f90:410 + boundary_bcs.f90:457)
// cb/ct = the three stencil coefficients in that order followed by the LHS coefficient.
struct NeumannCoef { double cb[4], ct[4]; };
__global__ void __launch_bounds__(256) k_neumann_planes(const double *__restrict__ u, const double *__restrict__ du, NeumannCoef c, int do_b,
                                                        int do_t, double *__restrict__ hb, double *__restrict__ ht, int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)(i % nx);
    const long long k = i / nx;
    const long long base = ix + (long long)nx * ny * k;
#define AT(a, j) a[base + (long long)nx * (j)]
    if (do_b) hb[i] = ((AT(u, 1) * c.cb[0] + AT(u, 2) * c.cb[1]) + AT(u, 3) * c.cb[2]) + c.cb[3] * AT(du, 1);
    if (do_t) ht[i] = ((AT(u, ny - 4) * c.ct[0] + AT(u, ny - 3) * c.ct[1]) + AT(u, ny - 2) * c.ct[2]) + c.ct[3] * AT(du, ny - 2);
#undef AT
}

// fused tail of the substep for one velocity component (rhs_global_incompressible_1.f90:348-352, :373-375; time.f90:645-664, :272-297):
//   h = h - g (pressure gradient); h = 0 on the wall planes j = 1, ny; q = q + dte*h; h = kco*h (if scale)
// gw != NULL: h = h - g gw(j), the anelastic form (Thermo_Anelastic_WEIGHT_SUBTRACT with ribackground, :326-329)
__global__ void __launch_bounds__(256) k_final_update(double *__restrict__ q, double *__restrict__ h, const double *__restrict__ g,
                                                      const double *__restrict__ pb, const double *__restrict__ pt, double dte,
                                                      double kco, int scale, int nx, int ny, long long n, const double *__restrict__ gw) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int j = (int)((i / nx) % ny);
        double hv = g ? (gw ? h[i] - g[i] * gw[j] : h[i] - g[i]) : h[i];
        if (j == 0) hv = pb ? pb[(i % nx) + (long long)nx * (i / ((long long)nx * ny))] : 0.0;
        else if (j == ny - 1) hv = pt ? pt[(i % nx) + (long long)nx * (i / ((long long)nx * ny))] : 0.0;
        q[i] = q[i] + dte * hv;
        h[i] = scale ? kco * hv : hv;
    }
}

// h += a ; o = a + s*b   (fallbacks of the fused accumulate entry points)
__global__ void __launch_bounds__(256) k_add1(double *__restrict__ h, const double *__restrict__ a, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) h[i] = h[i] + a[i];
}
__global__ void __launch_bounds__(256) k_axpy1(double *__restrict__ o, const double *__restrict__ a, const double *__restrict__ b, double s, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) o[i] = a[i] + b[i] * s;
}

// ---- z-slab <-> kx-pencil repacking around the all-to-all of the Poisson solver (tlab_amd/parallel.py) ----------------------
// slab a(nxh, ny, kmax) complex, x fastest; buffer = for every peer p the block [kmax][ny][nxl_p] of its kx range [ioff_p, ioff_p + nxl_p),
// blocks one after the other.  dir = +1: a -> buffer (before sending), dir = -1: buffer -> a (after receiving).  Pure index work.
struct PencilMap { int nproc; int ioff[17]; long long base[17]; };    // block p = kx range [ioff[p], ioff[p+1]); base[p] = its first complex element in the buffer
__global__ void __launch_bounds__(256) k_pencil_repack(double2 *__restrict__ a, double2 *__restrict__ buf, PencilMap m, int nxh, int ny, int kmax,
                                                       int dir) {
    const long long n = (long long)nxh * ny * kmax, stride = (long long)gridDim.x * blockDim.x;
    __shared__ unsigned char s_blk[4096];             // block of every kx (when the row fits)
    __shared__ int s_off[17];
    __shared__ long long s_base[17];
    if (threadIdx.x <= m.nproc) { s_off[threadIdx.x] = m.ioff[threadIdx.x]; s_base[threadIdx.x] = m.base[threadIdx.x]; }
    const bool table = nxh <= 4096;
    if (table) {
        for (int i = threadIdx.x; i < nxh; i += blockDim.x) {
            int p = 0;
            while (p + 1 < m.nproc && i >= m.ioff[p + 1]) ++p;
            s_blk[i] = (unsigned char)p;
        }
    }
    __syncthreads();
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int i = (int)(e % nxh);
        const long long jk = e / nxh;                 // j + ny * k
        int p = 0;
        if (table) p = s_blk[i];
        else while (p + 1 < m.nproc && i >= m.ioff[p + 1]) ++p;
        const int nxl = s_off[p + 1] - s_off[p];      // s_off[nproc] = nxh
        const long long b = s_base[p] + jk * nxl + (i - s_off[p]);
        if (dir > 0) buf[b] = a[e];
        else a[e] = buf[b];
    }
}

// ---- reductions (TIME_COURANT tools/dns/time.f90:365-548, MINMAX utils/minmax.f90:6) -----------------------------------------
// per-block partial (min, max) of a[i] (mode 0) or of |u|/dx(i) + |v|/dy(j) + |w|/dz(k) (mode 1); part: [2][gridDim.x]
__global__ void __launch_bounds__(256) k_minmax_partial(const double *__restrict__ a, const double *__restrict__ v, const double *__restrict__ w,
                                                        const double *__restrict__ odx, const double *__restrict__ ody,
                                                        const double *__restrict__ odz, int mode, int nx, int ny, int nz, int koff, int zon,
                                                        double *__restrict__ part) {
    __shared__ double smn[4], smx[4];
    const long long n = (long long)nx * ny * nz, stride = (long long)gridDim.x * blockDim.x;
    double mn = 1.0e300, mx = -1.0e300;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double val = a[i];
        if (mode == 1) {
            const int ix = (int)(i % nx), j = (int)((i / nx) % ny), k = (int)(i / ((long long)nx * ny));
            val = fabs(a[i]) * odx[ix] + fabs(v[i]) * ody[j];
            if (zon) val += fabs(w[i]) * odz[k + koff];      // the GLOBAL z%size > 1 (time.f90:402), not the slab depth
        }
        mn = fmin(mn, val); mx = fmax(mx, val);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, d, 64));
        mx = fmax(mx, __shfl_xor(mx, d, 64));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smn[wv] = mn; smx[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
        part[gridDim.x + blockIdx.x] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    }
}
// a = value (hq = 0.0_wp of TIME_RUNGEKUTTA, time.f90:212-216) ; a = alpha * a (the tendency scaling, time.f90:272-297)
__global__ void __launch_bounds__(256) k_scale(double *__restrict__ a, double alpha, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = alpha * a[i];
}
__global__ void __launch_bounds__(256) k_negate(double *__restrict__ a, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = -a[i];
}

#define CHECK_LAUNCH() hipGetLastError()

hipError_t launch_minmax_partial(const double *a, const double *v, const double *w, const double *odx, const double *ody, const double *odz,
                                 int mode, int nx, int ny, int nz, int koff, int zon, double *part, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(k_minmax_partial, dim3(nblocks), dim3(256), 0, st, a, v, w, odx, ody, odz, mode, nx, ny, nz, koff, zon, part);
    return CHECK_LAUNCH();
}
// base == nullptr: the blocks follow each other in the buffer; otherwise base[p] = first complex element of block p (any order, no overlap)
hipError_t launch_pencil_repack(double *a, double *buf, int nxh, int ny, int kmax, int nproc, const int *ioff, const long long *base, int dir,
                                hipStream_t st) {
    if (nproc < 1 || nproc > 16) return hipErrorInvalidValue;
    PencilMap m;
    m.nproc = nproc;
    long long acc = 0;
    for (int p = 0; p < nproc; ++p) {
        m.ioff[p] = ioff[p];
        m.base[p] = base ? base[p] : acc;
        acc += (long long)((p + 1 < nproc ? ioff[p + 1] : nxh) - ioff[p]) * ny * kmax;
    }
    m.ioff[nproc] = nxh; m.base[nproc] = acc;
    const long long n = (long long)nxh * ny * kmax;
    ProfScope ps("k_pencil_repack", st, (double)n * 32.0);
    hipLaunchKernelGGL(k_pencil_repack, dim3(pw_grid(n)), dim3(256), 0, st, reinterpret_cast<double2 *>(a), reinterpret_cast<double2 *>(buf), m, nxh, ny,
                       kmax, dir);
    return CHECK_LAUNCH();
}
// Thermo_Anelastic_WEIGHT_{INPLACE,OUTPLACE,SUBTRACT} (thermodynamics/thermo_anelastic.f90:377-448): a(i,j,k) weighted by a profile of y
//   mode 0: out = in * w(j)      mode 1: out = out - in * w(j)
__global__ void __launch_bounds__(256) k_weight_y(double *__restrict__ out, const double *__restrict__ in, const double *__restrict__ w, int nx,
                                                  int ny, long long n, int mode) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double wj = w[(i / nx) % ny];
        out[i] = mode ? out[i] - in[i] * wj : in[i] * wj;
    }
}
hipError_t launch_weight_y(double *out, const double *in, const double *w, int nx, int ny, long long n, int mode, hipStream_t st) {
    ProfScope ps("k_weight_y", st, (double)n * (mode ? 24.0 : 16.0));
    hipLaunchKernelGGL(k_weight_y, dim3(pw_grid(n)), dim3(256), 0, st, out, in, w, nx, ny, n, mode);
    return CHECK_LAUNCH();
}
// OPR_Burgers_1D with rhoinv%active (physics/opr_burgers.f90:504-507): out = (nu d2) * ribackground(j) - vel * d1
__global__ void __launch_bounds__(256) k_burgers_epilogue_anelastic(double *__restrict__ out, const double *__restrict__ vel,
                                                                    const double *__restrict__ d1, double nu, const double *__restrict__ ri,
                                                                    int nx, int ny, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (nu * out[i]) * ri[(i / nx) % ny] - vel[i] * d1[i];
}
hipError_t launch_burgers_epilogue_anelastic(double *out, const double *vel, const double *d1, double nu, const double *ri, int nx, int ny,
                                             long long n, hipStream_t st) {
    ProfScope ps("k_burgers_epilogue_anelastic", st, (double)n * 32.0);
    hipLaunchKernelGGL(k_burgers_epilogue_anelastic, dim3(pw_grid(n)), dim3(256), 0, st, out, vel, d1, nu, ri, nx, ny, n);
    return CHECK_LAUNCH();
}
// BOUNDARY_BCS_SURFACE_Y (tools/dns/boundary_bcs.f90:478-546): ref(i,k) += cpl (sign diff t(i,j,k) - diff avg), avg = AVG1V2D of plane javg of t
// (utils/averages.f90:114-137).  One workgroup sums the plane (deterministic order, not the reference's serial one), a second kernel adds.
__global__ void __launch_bounds__(1024) k_plane_sum(const double *__restrict__ t, int j, int nx, int ny, int nz, double *__restrict__ out) {
    __shared__ double part[1024];
    double acc = 0.0;
    const long long np = (long long)nx * nz;
    for (long long q = threadIdx.x; q < np; q += blockDim.x) acc += t[(q % nx) + (long long)nx * (j + (long long)ny * (q / nx))];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / (double)np;
}
__global__ void __launch_bounds__(256) k_surface_flux(double *__restrict__ ref, const double *__restrict__ t, int j, double sign, double diff, double cpl,
                                                      const double *__restrict__ avg, int nx, int ny, int nz) {
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)nx * nz) return;
    const double hfx = sign * diff * t[(q % nx) + (long long)nx * (j + (long long)ny * (q / nx))];
    const double hfx_avg = diff * avg[0];
    ref[q] = ref[q] + cpl * (hfx - hfx_avg);
}
// the same with the plane average as a kernel argument (z-slab driver: the all-reduced value lives on the host)
__global__ void __launch_bounds__(256) k_surface_flux_v(double *__restrict__ ref, const double *__restrict__ t, int j, double sign, double diff, double cpl,
                                                        double avg, int nx, int ny, int nz) {
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)nx * nz) return;
    const double hfx = sign * diff * t[(q % nx) + (long long)nx * (j + (long long)ny * (q / nx))];
    const double hfx_avg = diff * avg;
    ref[q] = ref[q] + cpl * (hfx - hfx_avg);
}
hipError_t launch_surface_flux(double *ref, const double *t, int j, int javg, double sign, double diff, double cpl, double *avg_scratch, int nx, int ny,
                               int nz, hipStream_t st) {
    hipLaunchKernelGGL(k_plane_sum, dim3(1), dim3(1024), 0, st, t, javg, nx, ny, nz, avg_scratch);
    hipLaunchKernelGGL(k_surface_flux, dim3(pw_grid((long long)nx * nz)), dim3(256), 0, st, ref, t, j, sign, diff, cpl, avg_scratch, nx, ny, nz);
    return CHECK_LAUNCH();
}
// the two halves on their own (z-slab driver: the plane average is an all-reduce over the ranks between them)
hipError_t launch_plane_avg(const double *t, int j, int nx, int ny, int nz, double *avg, hipStream_t st) {
    hipLaunchKernelGGL(k_plane_sum, dim3(1), dim3(1024), 0, st, t, j, nx, ny, nz, avg);
    return CHECK_LAUNCH();
}
hipError_t launch_surface_flux_avg(double *ref, const double *t, int j, double sign, double diff, double cpl, double avg, int nx, int ny, int nz,
                                   hipStream_t st) {
    hipLaunchKernelGGL(k_surface_flux_v, dim3(pw_grid((long long)nx * nz)), dim3(256), 0, st, ref, t, j, sign, diff, cpl, avg, nx, ny, nz);
    return CHECK_LAUNCH();
}
hipError_t launch_scale(double *a, double alpha, long long n, hipStream_t st) {
    hipLaunchKernelGGL(k_scale, dim3(pw_grid(n)), dim3(256), 0, st, a, alpha, n);
    return CHECK_LAUNCH();
}
hipError_t launch_negate(double *a, long long n, hipStream_t st) {
    hipLaunchKernelGGL(k_negate, dim3(pw_grid(n)), dim3(256), 0, st, a, n);
    return CHECK_LAUNCH();
}

hipError_t launch_add1(double *h, const double *a, long long n, hipStream_t st) {
    ProfScope ps("k_add1", st, (double)n * 24);
    hipLaunchKernelGGL(k_add1, dim3(pw_grid(n)), dim3(256), 0, st, h, a, n);
    return CHECK_LAUNCH();
}
hipError_t launch_axpy1(double *o, const double *a, const double *b, double s, long long n, hipStream_t st) {
    ProfScope ps("k_axpy1", st, (double)n * 24);
    hipLaunchKernelGGL(k_axpy1, dim3(pw_grid(n)), dim3(256), 0, st, o, a, b, s, n);
    return CHECK_LAUNCH();
}

hipError_t launch_final_update(double *q, double *h, const double *g, const double *pb, const double *pt, double dte, double kco, int scale,
                               int nx, int ny, int nz, hipStream_t st, const double *gw) {
    const long long n = (long long)nx * ny * nz;
    ProfScope ps("k_final_update", st, (double)n * (g ? 40 : 32));
    hipLaunchKernelGGL(k_final_update, dim3(pw_grid(n)), dim3(256), 0, st, q, h, g, pb, pt, dte, kco, scale, nx, ny, n, gw);
    return CHECK_LAUNCH();
}

hipError_t launch_add3(double *h, const double *a, const double *b, const double *c, long long n, hipStream_t st) {
    ProfScope ps("k_add3", st, (double)n * 40);
    hipLaunchKernelGGL(k_add3, dim3(pw_grid(n / 2)), dim3(256), 0, st, h, a, b, c, n);
    return CHECK_LAUNCH();
}
hipError_t launch_axpy3(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3, const double *q1,
                        const double *q2, const double *q3, double s, long long n, hipStream_t st) {
    ProfScope ps("k_axpy3", st, (double)n * 72);
    hipLaunchKernelGGL(k_axpy3, dim3(pw_grid(n)), dim3(256), 0, st, o1, o2, o3, h1, h2, h3, q1, q2, q3, s, n);
    return CHECK_LAUNCH();
}
hipError_t launch_axpy3w(double *o1, double *o2, double *o3, const double *h1, const double *h2, const double *h3, const double *q1,
                         const double *q2, const double *q3, double s, const double *w, int nx, int ny, long long n, hipStream_t st) {
    ProfScope ps("k_axpy3w", st, (double)n * 72);
    hipLaunchKernelGGL(k_axpy3w, dim3(pw_grid(n)), dim3(256), 0, st, o1, o2, o3, h1, h2, h3, q1, q2, q3, s, w, nx, ny, n);
    return CHECK_LAUNCH();
}
hipError_t launch_sum3(double *a, const double *b, const double *c, long long n, hipStream_t st) {
    ProfScope ps("k_sum3", st, (double)n * 32);
    hipLaunchKernelGGL(k_sum3, dim3(pw_grid(n / 2)), dim3(256), 0, st, a, b, c, n);
    return CHECK_LAUNCH();
}
hipError_t launch_sub3(double *h1, double *h2, double *h3, const double *a, const double *b, const double *c, long long n, hipStream_t st) {
    ProfScope ps("k_sub3", st, (double)n * 72);
    hipLaunchKernelGGL(k_sub3, dim3(pw_grid(n)), dim3(256), 0, st, h1, h2, h3, a, b, c, n);
    return CHECK_LAUNCH();
}
hipError_t launch_rk_update(double *q, double *h, double dte, double kco, int scale, long long n, hipStream_t st) {
    ProfScope ps("k_rk_update", st, (double)n * (scale ? 32 : 24));
    hipLaunchKernelGGL(k_rk_update, dim3(pw_grid(n / 2)), dim3(256), 0, st, q, h, dte, kco, scale, n);
    return CHECK_LAUNCH();
}
hipError_t launch_get_wall_planes(const double *f, double *hb, double *ht, int nx, int ny, int nz, hipStream_t st) {
    hipLaunchKernelGGL(k_get_wall_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, f, hb, ht, nx, ny, nz);
    return CHECK_LAUNCH();
}
hipError_t launch_fill_wall_planes(double *f, double vb, double vt, int nx, int ny, int nz, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_wall_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, f, vb, vt, nx, ny, nz);
    return CHECK_LAUNCH();
}

hipError_t launch_set_wall_planes(double *f, const double *pb, const double *pt, int nx, int ny, int nz, hipStream_t st) {
    hipLaunchKernelGGL(k_set_wall_planes_opt, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, f, pb, pt, nx, ny, nz);
    return CHECK_LAUNCH();
}
// Weighted sums over the K rows next to each wall: ob[ix, k] = sum_{j < K} wb[j] a[ix, j, k], ot[ix, k] = sum_{j < K} wt[j] a[ix, ny-1-j, k], for one or
// two fields at once (a2 / ob2 / ot2 may be NULL).  With wb, wt = the row of the Neumann operator that BOUNDARY_BCS_NEUMANN_Y applies to a finished
// tendency (its wall value is a linear functional of the line whose weights decay like 0.38^j), this is that wall value without the y-derivative
// pass over the whole field.
__global__ void __launch_bounds__(256) k_wall_weighted(const double *__restrict__ a1, const double *__restrict__ a2, const double *__restrict__ wb,
                                                       const double *__restrict__ wt, int K, double *__restrict__ ob1, double *__restrict__ ot1,
                                                       double *__restrict__ ob2, double *__restrict__ ot2, int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)((unsigned long long)i % (unsigned)nx);
    const long long k = i / nx;
    const long long base = ix + (long long)nx * ny * k;
    double sb1 = 0.0, st1 = 0.0, sb2 = 0.0, st2 = 0.0;
    for (int j = 0; j < K; ++j) {
        const double cb = wb ? wb[j] : 0.0, ct = wt ? wt[j] : 0.0;
        const long long lo = base + (long long)nx * j, hi = base + (long long)nx * (ny - 1 - j);
        sb1 += cb * a1[lo]; st1 += ct * a1[hi];
        if (a2) { sb2 += cb * a2[lo]; st2 += ct * a2[hi]; }
    }
    ob1[i] = sb1; ot1[i] = st1;
    if (a2) { ob2[i] = sb2; ot2[i] = st2; }
}
hipError_t launch_wall_weighted(const double *a1, const double *a2, const double *wb, const double *wt, int K, double *ob1, double *ot1, double *ob2,
                                double *ot2, int nx, int ny, int nz, hipStream_t st) {
    ProfScope ps("k_wall_weighted", st, (double)nx * nz * K * (a2 ? 32.0 : 16.0));
    hipLaunchKernelGGL(k_wall_weighted, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, a1, a2, wb, wt, K, ob1, ot1, ob2, ot2, nx, ny, nz);
    return CHECK_LAUNCH();
}
// Wall planes of a field whose interior was finished with zero wall tendencies (the Burgers epilogue's Dirichlet treatment) although its walls are
// Neumann ones: sb / st = the weighted sums of k_wall_weighted over the STORED tendencies, i.e. kco times the wall tendency when the tendencies were
// scaled (the functional is linear) -- h(wall) = s, q(wall) += dte s / kco (or dte s unscaled)
__global__ void __launch_bounds__(256) k_wall_fix(double *__restrict__ q, double *__restrict__ h, const double *__restrict__ sb,
                                                  const double *__restrict__ st, double dte, double kco, int scale, int nx, int ny, int nz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * nz) return;
    const int ix = (int)((unsigned long long)i % (unsigned)nx);
    const long long k = i / nx;
    const long long lo = ix + (long long)nx * ny * k, hi = lo + (long long)nx * (ny - 1);
    if (sb) { const double s = sb[i]; h[lo] = s; q[lo] = q[lo] + dte * (scale ? s / kco : s); }
    if (st) { const double s = st[i]; h[hi] = s; q[hi] = q[hi] + dte * (scale ? s / kco : s); }
}
hipError_t launch_wall_fix(double *q, double *h, const double *sb, const double *st, double dte, double kco, int scale, int nx, int ny, int nz,
                           hipStream_t stream) {
    hipLaunchKernelGGL(k_wall_fix, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, stream, q, h, sb, st, dte, kco, scale, nx, ny, nz);
    return CHECK_LAUNCH();
}
// o = a - b (planes)
__global__ void __launch_bounds__(256) k_sub2(double *__restrict__ o, const double *__restrict__ a, const double *__restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] - b[i];
}
hipError_t launch_sub2(double *o, const double *a, const double *b, long long n, hipStream_t st) {
    hipLaunchKernelGGL(k_sub2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, o, a, b, n);
    return CHECK_LAUNCH();
}

// The one strided copy of an I- / K-transposition (TLabMPI_Trp_Exec*, base/tlab_mpi_transpose.f90:232-256, :301-325): strided array S and wire format W
//     S[(q m + r) + (m P) o]   <->   W[q (m c) + r + m o]        q = peer < P, r < m, o < c          (to_wire: W = S, else S = W)
// (the same kernel as k_trp_copy of comm.hip, here for the native pencil driver of the operator library: bit-exact index work)
template <int VEC>
__global__ void __launch_bounds__(256) k_trp_copy_core(double *__restrict__ S, double *__restrict__ W, long long m, int P, long long c, int to_wire) {
    const long long mv = m / VEC, total = mv * c * P, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i % mv, o = (i / mv) % c, q = i / (mv * c);
        const long long s = (q * mv + r) + (mv * P) * o;
        if (VEC == 2) {
            double2 *S2 = reinterpret_cast<double2 *>(S), *W2 = reinterpret_cast<double2 *>(W);
            if (to_wire) W2[i] = S2[s];
            else S2[s] = W2[i];
        } else {
            if (to_wire) W[i] = S[s];
            else S[s] = W[i];
        }
    }
}
hipError_t launch_trp_copy(double *S, double *W, long long m, int P, long long c, int to_wire, hipStream_t st) {
    const long long total = m * c * P;
    const bool vec = (m % 2 == 0) && ((((size_t)S) | ((size_t)W)) % 16 == 0);
    const long long work = vec ? total / 2 : total;
    const int grid = (int)std::min<long long>(4096, std::max<long long>(1, (work + 255) / 256));
    ProfScope ps("k_trp_copy", st, (double)total * 16.0);
    if (vec) hipLaunchKernelGGL(k_trp_copy_core<2>, dim3(grid), dim3(256), 0, st, S, W, m, P, c, to_wire);
    else hipLaunchKernelGGL(k_trp_copy_core<1>, dim3(grid), dim3(256), 0, st, S, W, m, P, c, to_wire);
    return CHECK_LAUNCH();
}

// up to 48 device-to-device copies in ONE launch (the single-process loopback transport of the slab driver: P x P blocks per all-to-all, the
// ring messages of all ranks): one kernel instead of dozens of hipMemcpyAsync calls, whose host cost made that diagnostic host-bound
struct CopyBlocks { int n; const double *src[48]; double *dst[48]; long long cnt[48]; };
__global__ void __launch_bounds__(256) k_copy_blocks(CopyBlocks c) {
    const int b = blockIdx.y;
    const double *__restrict__ s = c.src[b];
    double *__restrict__ d = c.dst[b];
    const long long n = c.cnt[b], stride = (long long)gridDim.x * blockDim.x;
    if ((((size_t)s | (size_t)d) & 15) == 0) {
        const double2 *s2 = reinterpret_cast<const double2 *>(s);
        double2 *d2 = reinterpret_cast<double2 *>(d);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += stride) d2[i] = s2[i];
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) d[n - 1] = s[n - 1];
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
    }
}
hipError_t launch_copy_blocks(int n, const double *const *src, double *const *dst, const long long *cnt, hipStream_t st) {
    for (int b0 = 0; b0 < n; b0 += 48) {
        CopyBlocks c;
        c.n = std::min(48, n - b0);
        long long mx = 0;
        for (int b = 0; b < 48; ++b) {
            const bool in = b < c.n;
            c.src[b] = in ? src[b0 + b] : nullptr; c.dst[b] = in ? dst[b0 + b] : nullptr; c.cnt[b] = in ? cnt[b0 + b] : 0;
            mx = std::max(mx, c.cnt[b]);
        }
        if (mx == 0) continue;
        const unsigned gx = (unsigned)std::min<long long>((mx / 2 + 255) / 256 + 1, 2048 / c.n + 1);
        long long total = 0;
        for (int b = 0; b < c.n; ++b) total += c.cnt[b];
        // (the loopback transports' stand-in for an exchange: bench.py reports its share of a `--loopback` / `--decomp` line as `exchange_standin`)
        ProfScope ps("k_copy_blocks", st, (double)total * 16.0);
        hipLaunchKernelGGL(k_copy_blocks, dim3(gx, (unsigned)c.n), dim3(256), 0, st, c);
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    }
    return hipSuccess;
}
hipError_t launch_neumann_planes(const double *u, const double *du, const double *cb, const double *ct, int do_b, int do_t, double *hb,
                                 double *ht, int nx, int ny, int nz, hipStream_t st) {
    NeumannCoef c;
    for (int i = 0; i < 4; ++i) { c.cb[i] = cb[i]; c.ct[i] = ct[i]; }
    hipLaunchKernelGGL(k_neumann_planes, dim3((unsigned)(((long long)nx * nz + 255) / 256)), dim3(256), 0, st, u, du, c, do_b, do_t, hb, ht, nx, ny, nz);
    return CHECK_LAUNCH();
}

}  // namespace tlab
