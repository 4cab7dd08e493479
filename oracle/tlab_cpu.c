/* tlab_cpu.c -- C + OpenMP restatement of the reference's CPU algorithm for the Navier-Stokes RHS hot path.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY.  Nothing under tlab_amd/ links or calls this file; it is used by tests/ (as a second
 * checker beside the numpy oracle) and by bench.py's `cpu_baseline` leg, which times it on the host cores of the GPU box
 * (kind: "port").  It is NOT a fallback of the product path.
 *
 * Parity status: PINNED through the numpy oracle (oracle/tlab_oracle*.py, itself bitwise against oracle/_ref = the reference's own
 * Fortran) and the golden fixtures generated from the reference (tests/test_cpu_baseline.py: <= 1e-14).
 *
 * It follows the reference's CPU structure, not the GPU design: explicit local transposes around the x and y operators
 * (TLab_Transpose), separate right-hand-side pass (MatMul_*) and Thomas sweeps (TRIDSS / TRIDPSS) over lines-fastest arrays
 * u(nlines, n), the transposed velocity threaded from the OPR_B_SELF calls to the OPR_B_U_IN calls, per-mode pentadiagonal systems
 * factorized at initialisation (OPR_Elliptic_Initialize) and the three homogeneous solutions recomputed on every call
 * (OPR_ODE2_Factorize_NN).  Threading: the reference partitions the line index statically over OpenMP threads
 * (TLab_OMP_PARTITION, base/tlab_openmp.f90:12-62); the same here, plus the mode loop of the Poisson solver, which the reference
 * distributes over MPI ranks.  Citations are path:line under the reference's src/.
 *
 * Coefficient tables come from the caller (numpy oracle plans = FDM_CreatePlan / FDM_Int1_Initialize): initialisation is not on the hot path.
 * Default schemes only: CompactJacobian6 (tridiagonal LHS, 5-diagonal antisymmetric RHS) and CompactJacobian6Hyper (7-diagonal symmetric RHS).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BCS_DD 0
#define BCS_ND 1
#define BCS_DN 2
#define BCS_NN 3
#define BCS_PERIODIC (-1)

typedef struct {
    int n, periodic, need_1der;
    const double *rhs1;   /* der1%rhs(n,5)  column-major                        fdm_derivative.f90:16-29 */
    const double *rhs_b1; /* der1%rhs_b(4,0:7) as [row-1][col], row-major       */
    const double *rhs_t1; /* der1%rhs_t(0:4,7) as [row][col-1], row-major       */
    const double *lu1;    /* der1%lu(n,5) periodic | (n,20) otherwise, column-major */
    const double *rhs2;   /* der2%rhs(n,7+3) column-major: 7 diagonals + 3 Jacobian-correction diagonals */
    const double *lu2;    /* der2%lu(n,5) periodic | (n,3), column-major (used by OPR_Partial; Burgers passes its own) */
} cpu_fdm_t;

/* ---------------------------------------------------------------------------------------------------------------------------
 * TLab_OMP_PARTITION   base/tlab_openmp.f90:12-62 : static partition of len items over the threads of the enclosing region */
static void omp_partition(int len, int *l0, int *l1) {
#ifdef _OPENMP
    const int nt = omp_get_num_threads(), it = omp_get_thread_num();
#else
    const int nt = 1, it = 0;
#endif
    const int q = len / nt, r = len % nt;
    *l0 = it * q + (it < r ? it : r);
    *l1 = *l0 + q + (it < r ? 1 : 0);
}

/* TLab_Transpose(a, nra, nca, ma, b, mb)   utils/tlab_transpose.f90:14-82 : b(j,i) = a(i,j), 64 x 64 blocks, OpenMP over column blocks */
void tlabcpu_transpose(const double *a, int nra, int nca, double *b) {
    const int jb = 64, kb = 64;
#pragma omp parallel for schedule(static)
    for (int k0 = 0; k0 < nca; k0 += kb) {
        const int k1 = k0 + kb < nca ? k0 + kb : nca;
        for (int j0 = 0; j0 < nra; j0 += jb) {
            const int j1 = j0 + jb < nra ? j0 + jb : nra;
            for (int j = j0; j < j1; ++j)
                for (int k = k0; k < k1; ++k) b[(size_t)j * nca + k] = a[(size_t)k * nra + j];
        }
    }
}
/* TLab_Transpose_COMPLEX   utils/tlab_transpose.f90:148-210 (16-byte elements) */
static void transpose_complex(const double *a, int nra, int nca, double *b) {
#pragma omp parallel for schedule(static)
    for (int k0 = 0; k0 < nca; k0 += 64) {
        const int k1 = k0 + 64 < nca ? k0 + 64 : nca;
        for (int j0 = 0; j0 < nra; j0 += 64) {
            const int j1 = j0 + 64 < nra ? j0 + 64 : nra;
            for (int j = j0; j < j1; ++j)
                for (int k = k0; k < k1; ++k) {
                    b[2 * ((size_t)j * nca + k)] = a[2 * ((size_t)k * nra + j)];
                    b[2 * ((size_t)j * nca + k) + 1] = a[2 * ((size_t)k * nra + j) + 1];
                }
        }
    }
}

/* lines-fastest access u(l, i) of u(len, n) */
#define U(i) (u + (size_t)(i) * len)
#define F(i) (f + (size_t)(i) * len)
#define R(i, c) rhs[(size_t)(c) * n + (i)]

/* MatMul_5d_antisym   fdm/fdm_matmul.f90:359-419 */
static void matmul_5d_antisym(const double *rhs, const double *rb, const double *rt, int n, const double *u, double *f, int len, int l0, int l1,
                              int ibc) {
    const double r5 = R(3, 4);
#define RB(r, c) rb[(r) * 8 + (c)]
#define RT(r, c) rt[(r) * 7 + (c)]
    if (ibc == BCS_PERIODIC) {
        for (int l = l0; l < l1; ++l) {
            F(0)[l] = U(1)[l] - U(n - 1)[l] + r5 * (U(2)[l] - U(n - 2)[l]);
            F(1)[l] = U(2)[l] - U(0)[l] + r5 * (U(3)[l] - U(n - 1)[l]);
            F(2)[l] = U(3)[l] - U(1)[l] + r5 * (U(4)[l] - U(0)[l]);
        }
    } else if (ibc == BCS_ND || ibc == BCS_NN) {
        for (int l = l0; l < l1; ++l) {
            F(1)[l] = F(0)[l] * RB(1, 2) + U(1)[l] * RB(1, 3) + U(2)[l] * RB(1, 4) + U(3)[l] * RB(1, 5);
            F(2)[l] = F(0)[l] * RB(2, 1) + U(1)[l] * RB(2, 2) + U(2)[l] * RB(2, 3) + U(3)[l] * RB(2, 4) + U(4)[l] * RB(2, 5);
        }
    } else {
        for (int l = l0; l < l1; ++l) {
            F(0)[l] = U(0)[l] * R(0, 2) + U(1)[l] * R(0, 3) + U(2)[l] * R(0, 4) + U(3)[l] * R(0, 0);
            F(1)[l] = U(0)[l] * R(1, 1) + U(1)[l] * R(1, 2) + U(2)[l] * R(1, 3) + U(3)[l] * R(1, 4);
            F(2)[l] = U(0)[l] * R(2, 0) + U(1)[l] * R(2, 1) + U(2)[l] * R(2, 2) + U(3)[l] * R(2, 3) + U(4)[l] * R(2, 4);
        }
    }
    for (int i = 3; i < n - 3; ++i) {
        const double *um2 = U(i - 2), *um1 = U(i - 1), *up1 = U(i + 1), *up2 = U(i + 2);
        double *fi = F(i);
        for (int l = l0; l < l1; ++l) fi[l] = up1[l] - um1[l] + r5 * (up2[l] - um2[l]);
    }
    if (ibc == BCS_PERIODIC) {
        for (int l = l0; l < l1; ++l) {
            F(n - 3)[l] = U(n - 2)[l] - U(n - 4)[l] + r5 * (U(n - 1)[l] - U(n - 5)[l]);
            F(n - 2)[l] = U(n - 1)[l] - U(n - 3)[l] + r5 * (U(0)[l] - U(n - 4)[l]);
            F(n - 1)[l] = U(0)[l] - U(n - 2)[l] + r5 * (U(1)[l] - U(n - 3)[l]);
        }
    } else if (ibc == BCS_DN || ibc == BCS_NN) {
        for (int l = l0; l < l1; ++l) {
            F(n - 3)[l] = U(n - 5)[l] * RT(1, 0) + U(n - 4)[l] * RT(1, 1) + U(n - 3)[l] * RT(1, 2) + U(n - 2)[l] * RT(1, 3) + F(n - 1)[l] * RT(1, 4);
            F(n - 2)[l] = U(n - 4)[l] * RT(2, 0) + U(n - 3)[l] * RT(2, 1) + U(n - 2)[l] * RT(2, 2) + F(n - 1)[l] * RT(2, 3);
        }
    } else {
        for (int l = l0; l < l1; ++l) {
            F(n - 3)[l] = U(n - 5)[l] * R(n - 3, 0) + U(n - 4)[l] * R(n - 3, 1) + U(n - 3)[l] * R(n - 3, 2) + U(n - 2)[l] * R(n - 3, 3) + U(n - 1)[l] * R(n - 3, 4);
            F(n - 2)[l] = U(n - 4)[l] * R(n - 2, 0) + U(n - 3)[l] * R(n - 2, 1) + U(n - 2)[l] * R(n - 2, 2) + U(n - 1)[l] * R(n - 2, 3);
            F(n - 1)[l] = U(n - 4)[l] * R(n - 1, 4) + U(n - 3)[l] * R(n - 1, 0) + U(n - 2)[l] * R(n - 1, 1) + U(n - 1)[l] * R(n - 1, 2);
        }
    }
#undef RB
#undef RT
}

/* MatMul_7d_sym   fdm/fdm_matmul.f90:562-642 (ibc: BCS_PERIODIC or BCS_DD, as FDM_Der2_Solve calls it, fdm_derivative.f90:427-435) */
static void matmul_7d_sym(const double *rhs, int n, const double *u, double *f, int len, int l0, int l1, int ibc) {
    const double r4 = R(3, 3), r6 = R(3, 5), r7 = R(3, 6);
    if (ibc == BCS_PERIODIC) {
        for (int l = l0; l < l1; ++l) {
            F(0)[l] = r4 * U(0)[l] + U(1)[l] + U(n - 1)[l] + r6 * (U(2)[l] + U(n - 2)[l]) + r7 * (U(3)[l] + U(n - 3)[l]);
            F(1)[l] = r4 * U(1)[l] + U(2)[l] + U(0)[l] + r6 * (U(3)[l] + U(n - 1)[l]) + r7 * (U(4)[l] + U(n - 2)[l]);
            F(2)[l] = r4 * U(2)[l] + U(3)[l] + U(1)[l] + r6 * (U(4)[l] + U(0)[l]) + r7 * (U(5)[l] + U(n - 1)[l]);
        }
    } else {
        for (int l = l0; l < l1; ++l) {
            F(0)[l] = U(0)[l] * R(0, 3) + U(1)[l] * R(0, 4) + U(2)[l] * R(0, 5) + U(3)[l] * R(0, 6) + U(4)[l] * R(0, 0);
            F(1)[l] = U(0)[l] * R(1, 2) + U(1)[l] * R(1, 3) + U(2)[l] * R(1, 4) + U(3)[l] * R(1, 5) + U(4)[l] * R(1, 6);
            F(2)[l] = U(0)[l] * R(2, 1) + U(1)[l] * R(2, 2) + U(2)[l] * R(2, 3) + U(3)[l] * R(2, 4) + U(4)[l] * R(2, 5) + U(5)[l] * R(2, 6);
        }
    }
    for (int i = 3; i < n - 3; ++i) {
        const double *u0 = U(i), *um1 = U(i - 1), *up1 = U(i + 1), *um2 = U(i - 2), *up2 = U(i + 2), *um3 = U(i - 3), *up3 = U(i + 3);
        double *fi = F(i);
        for (int l = l0; l < l1; ++l) fi[l] = r4 * u0[l] + up1[l] + um1[l] + r6 * (up2[l] + um2[l]) + r7 * (up3[l] + um3[l]);
    }
    if (ibc == BCS_PERIODIC) {
        for (int l = l0; l < l1; ++l) {
            F(n - 3)[l] = r4 * U(n - 3)[l] + U(n - 2)[l] + U(n - 4)[l] + r6 * (U(n - 1)[l] + U(n - 5)[l]) + r7 * (U(0)[l] + U(n - 6)[l]);
            F(n - 2)[l] = r4 * U(n - 2)[l] + U(n - 1)[l] + U(n - 3)[l] + r6 * (U(0)[l] + U(n - 4)[l]) + r7 * (U(1)[l] + U(n - 5)[l]);
            F(n - 1)[l] = r4 * U(n - 1)[l] + U(0)[l] + U(n - 2)[l] + r6 * (U(1)[l] + U(n - 3)[l]) + r7 * (U(2)[l] + U(n - 4)[l]);
        }
    } else {
        for (int l = l0; l < l1; ++l) {
            F(n - 3)[l] = U(n - 6)[l] * R(n - 3, 0) + U(n - 5)[l] * R(n - 3, 1) + U(n - 4)[l] * R(n - 3, 2) + U(n - 3)[l] * R(n - 3, 3) + U(n - 2)[l] * R(n - 3, 4) + U(n - 1)[l] * R(n - 3, 5);
            F(n - 2)[l] = U(n - 5)[l] * R(n - 2, 0) + U(n - 4)[l] * R(n - 2, 1) + U(n - 3)[l] * R(n - 2, 2) + U(n - 2)[l] * R(n - 2, 3) + U(n - 1)[l] * R(n - 2, 4);
            F(n - 1)[l] = U(n - 5)[l] * R(n - 1, 6) + U(n - 4)[l] * R(n - 1, 0) + U(n - 3)[l] * R(n - 1, 1) + U(n - 2)[l] * R(n - 1, 2) + U(n - 1)[l] * R(n - 1, 3);
        }
    }
}

/* MatMul_3d_add   fdm/fdm_matmul.f90:126-153 : f += B u (Jacobian correction of the second derivative, fdm_derivative.f90:437-440) */
static void matmul_3d_add(const double *rhs, int n, const double *u, double *f, int len, int l0, int l1) {
    for (int l = l0; l < l1; ++l) F(0)[l] = F(0)[l] + U(0)[l] * R(0, 1) + U(1)[l] * R(0, 2) + U(2)[l] * R(0, 0);
    for (int i = 1; i < n - 1; ++i) {
        const double r1 = R(i, 0), r2 = R(i, 1), r3 = R(i, 2);
        const double *um1 = U(i - 1), *u0 = U(i), *up1 = U(i + 1);
        double *fi = F(i);
        for (int l = l0; l < l1; ++l) fi[l] = fi[l] + um1[l] * r1 + u0[l] * r2 + up1[l] * r3;
    }
    for (int l = l0; l < l1; ++l) F(n - 1)[l] = F(n - 1)[l] + U(n - 3)[l] * R(n - 1, 2) + U(n - 2)[l] * R(n - 1, 0) + U(n - 1)[l] * R(n - 1, 1);
}
#undef R

/* TRIDSS   utils/linear3.f90:56-150 : a, b, c of nmax entries (pre-factored); f(len, nmax) */
static void tridss(int nmax, const double *a, const double *b, const double *c, double *f, int len, int l0, int l1) {
    for (int i = 1; i < nmax; ++i) {
        const double ai = a[i];
        double *fi = F(i);
        const double *fm = F(i - 1);
        for (int l = l0; l < l1; ++l) fi[l] = fi[l] + ai * fm[l];
    }
    for (int l = l0; l < l1; ++l) F(nmax - 1)[l] = F(nmax - 1)[l] * b[nmax - 1];
    for (int i = nmax - 2; i >= 0; --i) {
        const double ci = c[i], bi = b[i];
        double *fi = F(i);
        const double *fp = F(i + 1);
        for (int l = l0; l < l1; ++l) fi[l] = (fi[l] + ci * fp[l]) * bi;
    }
}
/* TRIDPSS   utils/linear3.f90:321-442 : wrk(len) */
static void tridpss(int nmax, const double *a, const double *b, const double *c, const double *d, const double *e, double *f, double *wrk, int len,
                    int l0, int l1) {
    for (int l = l0; l < l1; ++l) F(0)[l] = F(0)[l] * b[0];
    for (int i = 1; i < nmax - 1; ++i) {
        const double ai = a[i], bi = b[i];
        double *fi = F(i);
        const double *fm = F(i - 1);
        for (int l = l0; l < l1; ++l) fi[l] = fi[l] * bi + ai * fm[l];
    }
    for (int l = l0; l < l1; ++l) wrk[l] = 0.0;
    for (int i = 0; i < nmax - 1; ++i) {
        const double di = d[i];
        const double *fi = F(i);
        for (int l = l0; l < l1; ++l) wrk[l] = wrk[l] + di * fi[l];
    }
    for (int l = l0; l < l1; ++l) F(nmax - 1)[l] = (F(nmax - 1)[l] - wrk[l]) * b[nmax - 1];
    for (int l = l0; l < l1; ++l) F(nmax - 2)[l] = e[nmax - 2] * F(nmax - 1)[l] + F(nmax - 2)[l];
    for (int i = nmax - 3; i >= 0; --i) {
        const double ci = c[i], ei = e[i];
        double *fi = F(i);
        const double *fp = F(i + 1), *fn = F(nmax - 1);
        for (int l = l0; l < l1; ++l) fi[l] = fi[l] + ci * fp[l] + ei * fn[l];
    }
}

/* FDM_Der1_Solve   fdm/fdm_derivative.f90:218-278 */
static void der1_solve(const cpu_fdm_t *g, int ibc, const double *u, double *f, double *wrk, int len, int l0, int l1) {
    const int n = g->n;
    int ibc_loc = ibc, nmin = 0, nmax = n;
    const int ip = ibc * 5;
    if (g->periodic) ibc_loc = BCS_PERIODIC;
    if (ibc_loc == BCS_ND || ibc_loc == BCS_NN) {
        for (int l = l0; l < l1; ++l) F(0)[l] = 0.0;
        nmin += 1;
    }
    if (ibc_loc == BCS_DN || ibc_loc == BCS_NN) {
        for (int l = l0; l < l1; ++l) F(n - 1)[l] = 0.0;
        nmax -= 1;
    }
    matmul_5d_antisym(g->rhs1, g->rhs_b1, g->rhs_t1, n, u, f, len, l0, l1, ibc_loc);
    const double *lu = g->lu1;
    if (g->periodic)
        tridpss(n, lu, lu + n, lu + 2 * (size_t)n, lu + 3 * (size_t)n, lu + 4 * (size_t)n, f, wrk, len, l0, l1);
    else
        tridss(nmax - nmin, lu + (size_t)ip * n + nmin, lu + (size_t)(ip + 1) * n + nmin, lu + (size_t)(ip + 2) * n + nmin, f + (size_t)nmin * len, len,
               l0, l1);
}
/* FDM_Der2_Solve   fdm/fdm_derivative.f90:413-459 : lu is an argument (OPR_Burgers passes the diffusivity-scaled one) */
static void der2_solve(const cpu_fdm_t *g, const double *lu, const double *u, const double *du, double *f, double *wrk, int len, int l0, int l1) {
    const int n = g->n;
    matmul_7d_sym(g->rhs2, n, u, f, len, l0, l1, g->periodic ? BCS_PERIODIC : BCS_DD);
    if (g->need_1der) matmul_3d_add(g->rhs2 + 7 * (size_t)n, n, du, f, len, l0, l1);
    if (g->periodic)
        tridpss(n, lu, lu + n, lu + 2 * (size_t)n, lu + 3 * (size_t)n, lu + 4 * (size_t)n, f, wrk, len, l0, l1);
    else
        tridss(n, lu, lu + n, lu + 2 * (size_t)n, f, len, l0, l1);
}
#undef U
#undef F

/* ---------------------------------------------------------------------------------------------------------------------------
 * OPR_Partial_X/Y/Z(type, nx, ny, nz, bcs, g, u, result, tmp1)   operators/opr_partial.f90:31-150, 266-377, 154-262 (serial branch)
 * type 1 = OPR_P1, 2 = OPR_P2, 3 = OPR_P2_P1 (tmp1 = first derivative).  wrk3d: nx*ny*nz, wrk2d: nlines doubles. */
int tlabcpu_opr_partial(int dir, int type, int nx, int ny, int nz, int ibc, const cpu_fdm_t *g, const double *u, double *result, double *tmp1,
                        double *wrk3d, double *wrk2d) {
    const size_t ntot = (size_t)nx * ny * nz;
    const int n = dir == 1 ? nx : dir == 2 ? ny : nz;
    if (g->n != n) return -1;
    if (n == 1 && dir != 1) {                              /* 2-D guard, opr_partial.f90:175-177, 287-289 */
        memset(result, 0, ntot * sizeof(double));
        if (type == 3) memset(tmp1, 0, ntot * sizeof(double));
        return 0;
    }
    const int len = (int)(ntot / n);
    const double *ul = u;      /* lines-fastest operand */
    double *rl = result;       /* lines-fastest result */
    double *dl = tmp1;         /* lines-fastest first derivative (P2, P2_P1) */
    if (dir == 1) {            /* :87 local transpose (nx, nyz) -> (nyz, nx) */
        tlabcpu_transpose(u, nx, ny * nz, result);
        ul = result; rl = wrk3d;
    } else if (dir == 2) {     /* :303 (nxy, nz) -> (nz, nxy), viewed as (nx*nz, ny) */
        tlabcpu_transpose(u, nx * ny, nz, result);
        ul = result; rl = wrk3d;
    }
    if (type != 1 && dir != 3 && tmp1 == NULL) return -2;
#pragma omp parallel
    {
        int l0, l1;
        omp_partition(len, &l0, &l1);
        if (type == 1) {
            der1_solve(g, ibc, ul, rl, wrk2d, len, l0, l1);
        } else {
            if (type == 3 || g->need_1der) der1_solve(g, ibc, ul, dl, wrk2d, len, l0, l1);      /* :96, :100 */
            der2_solve(g, g->lu2, ul, dl, rl, wrk2d, len, l0, l1);                              /* :97, :101 */
        }
    }
    if (dir == 1) {            /* :133-134 transpose back */
        if (type == 3) { tlabcpu_transpose(dl, ny * nz, nx, result); memcpy(dl, result, ntot * sizeof(double)); }
        tlabcpu_transpose(rl, ny * nz, nx, result);
    } else if (dir == 2) {
        if (type == 3) { tlabcpu_transpose(dl, nz, nx * ny, result); memcpy(dl, result, ntot * sizeof(double)); }
        tlabcpu_transpose(rl, nz, nx * ny, result);
    }
    return 0;
}

/* OPR_Burgers_X/Y/Z(ivel, is, nx, ny, nz, bcs, s, u, result, tmp1, u_t)   physics/opr_burgers.f90:190-273, 277-355, 359-431
 * + OPR_Burgers_1D :439-521.  lu2d = fdmDiffusion(ig)%lu(:,:,is) (:90-114).  ivel = 0 (OPR_B_SELF): the transposed operand left in tmp1
 * is the velocity; 1 (OPR_B_U_IN): u_t is the transposed velocity of an earlier SELF call (for dir = 3: the field itself). */
int tlabcpu_opr_burgers(int dir, int ivel, int nx, int ny, int nz, int ibc, const cpu_fdm_t *g, const double *lu2d, const double *s,
                        double *result, double *tmp1, const double *u_t, double *wrk3d, double *wrk2d) {
    const size_t ntot = (size_t)nx * ny * nz;
    const int n = dir == 1 ? nx : dir == 2 ? ny : nz;
    if (g->n != n) return -1;
    if (n == 1) { memset(result, 0, ntot * sizeof(double)); return 0; }
    const int len = (int)(ntot / n);
    const double *sl, *vel;
    double *rl, *dsdx;
    if (dir == 1) {
        tlabcpu_transpose(s, nx, ny * nz, tmp1);            /* :250 */
        sl = tmp1; rl = wrk3d; dsdx = result;
    } else if (dir == 2) {
        tlabcpu_transpose(s, nx * ny, nz, tmp1);            /* :316 */
        sl = tmp1; rl = wrk3d; dsdx = result;
    } else {
        sl = s; rl = result; dsdx = tmp1;                   /* :406-414: no local transpose */
    }
    vel = ivel == 0 ? sl : u_t;                             /* :236-240 */
#pragma omp parallel
    {
        int l0, l1;
        omp_partition(len, &l0, &l1);
        der1_solve(g, ibc, sl, dsdx, wrk2d, len, l0, l1);                     /* :471 */
        der2_solve(g, lu2d, sl, dsdx, rl, wrk2d, len, l0, l1);               /* :472 */
        for (int i = 0; i < n; ++i) {                                         /* :503-517 result = result - u * dsdx */
            double *r = rl + (size_t)i * len;
            const double *v = vel + (size_t)i * len, *d = dsdx + (size_t)i * len;
            for (int l = l0; l < l1; ++l) r[l] = r[l] - v[l] * d[l];
        }
    }
    if (dir == 1) tlabcpu_transpose(rl, ny * nz, nx, result);               /* :261 */
    else if (dir == 2) tlabcpu_transpose(rl, nz, nx * ny, result);          /* :343 */
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * FFTs.  The reference calls FFTW (un-vendored: dfftw_execute_dft_r2c / _dft / _dft_c2r, operators/opr_fourier.f90:266,322,355,422).
 * Here: iterative radix-2 Stockham autosort for powers of two, a plain DFT otherwise (test sizes).  Unnormalised, like FFTW. */
typedef struct { int n; int pow2; double *tw; } fft_plan_t;     /* tw[2k] + i tw[2k+1] = exp(-2 pi i k / n) */
static void fft_plan_init(fft_plan_t *p, int n) {
    p->n = n;
    p->pow2 = (n & (n - 1)) == 0;
    p->tw = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    for (int k = 0; k < n; ++k) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        p->tw[2 * k] = cos(a); p->tw[2 * k + 1] = sin(a);
    }
}
/* x (n complex, interleaved) -> X; sign = -1 forward (exp(-i..)), +1 backward.  y: scratch of n complex.  Result in x. */
static void fft_exec(const fft_plan_t *p, double *x, double *y, int sign) {
    const int n = p->n;
    if (n == 1) return;
    if (!p->pow2) {
        for (int k = 0; k < n; ++k) {
            double sr = 0.0, si = 0.0;
            for (int j = 0; j < n; ++j) {
                const int t = (int)(((long long)j * k) % n);
                const double wr = p->tw[2 * t], wi = sign < 0 ? p->tw[2 * t + 1] : -p->tw[2 * t + 1];
                sr += x[2 * j] * wr - x[2 * j + 1] * wi;
                si += x[2 * j] * wi + x[2 * j + 1] * wr;
            }
            y[2 * k] = sr; y[2 * k + 1] = si;
        }
        memcpy(x, y, sizeof(double) * 2 * (size_t)n);
        return;
    }
    double *a = x, *b = y;
    int l = n / 2, m = 1;                 /* Stockham: n = 2 l m */
    while (l >= 1) {
        for (int j = 0; j < l; ++j) {
            const int t = j * m;          /* w = exp(-+ 2 pi i j / (2 l)) = tw[j * m * ... ]: 2 l m = n -> j/(2l) = j m / n */
            const double wr = p->tw[2 * t], wi = sign < 0 ? p->tw[2 * t + 1] : -p->tw[2 * t + 1];
            for (int k = 0; k < m; ++k) {
                const double c0r = a[2 * (k + j * m)], c0i = a[2 * (k + j * m) + 1];
                const double c1r = a[2 * (k + j * m + l * m)], c1i = a[2 * (k + j * m + l * m) + 1];
                b[2 * (k + 2 * j * m)] = c0r + c1r;
                b[2 * (k + 2 * j * m) + 1] = c0i + c1i;
                const double dr = c0r - c1r, di = c0i - c1i;
                b[2 * (k + 2 * j * m + m)] = dr * wr - di * wi;
                b[2 * (k + 2 * j * m + m) + 1] = dr * wi + di * wr;
            }
        }
        double *t2 = a; a = b; b = t2;
        l /= 2; m *= 2;
    }
    if (a != x) memcpy(x, a, sizeof(double) * 2 * (size_t)n);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Poisson solver.  Tables per regular mode m (built by FDM_Int1_Initialize at start-up, operators/opr_elliptic.f90:205-209):
 *   lhs[bc][m][5][ny]  LU-factorized pentadiagonal system of  u' +- lambda u = f  (rows 2..ny-1 factorized by PENTADFS, rows 1 and ny as built)
 *   rb[bc][m][3][4] = fdmi%rhs_b(1:3, 0:3),  rt[bc][m][3][4] = fdmi%rhs_t(0:2, 1:4)
 * bc = 0: BCS_MIN with +lambda (fdm_int1(BCS_MIN,...)), 1: BCS_MAX with -lambda.  rhs(ny,3): fdmi%rhs, the same for all modes (opr_elliptic.f90:213-217). */
typedef struct {
    int nx, ny, nz, nxh;
    long long nreg, nsing;          /* regular / singular modes */
    const int *reg_mode;            /* [nreg] flat mode index k * nxh + i */
    const int *sing_mode;           /* [nsing] */
    const double *lam;              /* [nreg] sqrt(lambda) */
    const double *lhs[2];           /* [nreg][5][ny] */
    const double *rb[2], *rt[2];    /* [nreg][3][4] */
    const double *slhs[2];          /* singular modes (lambda = 0): [nsing][5][ny] */
    const double *srb[2], *srt[2];
    const double *rhs[2];           /* [3][ny] (diagonal-major) of the BCS_MIN / BCS_MAX integral operators */
    double norm;
    fft_plan_t fx, fz;
} cpu_poisson_t;

cpu_poisson_t *tlabcpu_poisson_create(int nx, int ny, int nz, long long nreg, const int *reg_mode, const double *lam, const double *lhs_min,
                                      const double *lhs_max, const double *rb_min, const double *rb_max, const double *rt_min, const double *rt_max,
                                      long long nsing, const int *sing_mode, const double *slhs_min, const double *slhs_max, const double *srb_min,
                                      const double *srb_max, const double *srt_min, const double *srt_max, const double *rhs_min,
                                      const double *rhs_max) {
    cpu_poisson_t *P = (cpu_poisson_t *)calloc(1, sizeof(cpu_poisson_t));
    P->nx = nx; P->ny = ny; P->nz = nz; P->nxh = nx / 2 + 1;
    P->nreg = nreg; P->reg_mode = reg_mode; P->lam = lam;
    P->lhs[0] = lhs_min; P->lhs[1] = lhs_max; P->rb[0] = rb_min; P->rb[1] = rb_max; P->rt[0] = rt_min; P->rt[1] = rt_max;
    P->nsing = nsing; P->sing_mode = sing_mode;
    P->slhs[0] = slhs_min; P->slhs[1] = slhs_max; P->srb[0] = srb_min; P->srb[1] = srb_max; P->srt[0] = srt_min; P->srt[1] = srt_max;
    P->rhs[0] = rhs_min; P->rhs[1] = rhs_max;
    P->norm = 1.0 / ((double)nx * (double)nz);       /* opr_elliptic.f90:130 */
    fft_plan_init(&P->fx, nx);
    fft_plan_init(&P->fz, nz);
    return P;
}
void tlabcpu_poisson_destroy(cpu_poisson_t *P) {
    if (!P) return;
    free(P->fx.tw); free(P->fz.tw); free(P);
}

/* one first-order integral system of one mode: lhs[5][n], rb[3][4], rt[3][4], rhs[3][n]; NL lines interleaved: f[NL*j + l] */
typedef struct { const double *lhs, *rb, *rt, *rhs; int n, bc_min; } int1_t;

/* FDM_Int1_Solve   fdm/fdm_integral.f90:219-314 (pentadiagonal case) = MatMul_3d with BCS_BOTH (fdm_matmul.f90:70-121) + PENTADSS
 * (utils/linear5.f90:76-131) + recovery of the far boundary value (:265-311).  res carries the boundary value on entry (res(:,1) for BCS_MIN,
 * res(:,n) for BCS_MAX).  du (may be NULL): derivative at the boundary where the value was given (:283-290, :303-310). */
#define DEF_INT1_SOLVE(NL)                                                                                                            \
    static void int1_solve_##NL(const int1_t *s, const double *f, double *res, double *du) {                                           \
        const int n = s->n;                                                                                                            \
        const double *a = s->lhs, *b = a + n, *c = b + n, *d = c + n, *e = d + n;                                                      \
        const double *r1 = s->rhs, *r2 = r1 + n;                                                                                       \
        const double *rb = s->rb, *rt = s->rt;                                                                                         \
        double bcs_b[NL], bcs_t[NL];                                                                                                   \
        for (int l = 0; l < NL; ++l) {                                                                                                 \
            if (s->bc_min) res[NL * (n - 1) + l] = f[NL * (n - 1) + l];                                                               \
            else res[l] = f[l];                                                                                                        \
        }                                                                                                                              \
        for (int l = 0; l < NL; ++l) {                                                                                                 \
            const double r0 = res[l], rn = res[NL * (n - 1) + l];                                                                      \
            bcs_b[l] = r0 * rb[0 * 4 + 2] + f[NL * 1 + l] * rb[0 * 4 + 3] + f[NL * 2 + l] * rb[0 * 4 + 1];                           \
            res[NL * 1 + l] = r0 * rb[1 * 4 + 1] + f[NL * 1 + l] * rb[1 * 4 + 2] + f[NL * 2 + l] * rb[1 * 4 + 3];                    \
            res[NL * 2 + l] = r0 * rb[2 * 4 + 0] + f[NL * 1 + l] * rb[2 * 4 + 1] + f[NL * 2 + l] * rb[2 * 4 + 2] + f[NL * 3 + l] * rb[2 * 4 + 3]; \
            for (int i = 3; i < n - 3; ++i) res[NL * i + l] = f[NL * (i - 1) + l] * r1[i] + f[NL * i + l] * r2[i] + f[NL * (i + 1) + l]; \
            res[NL * (n - 3) + l] = f[NL * (n - 4) + l] * rt[0 * 4 + 0] + f[NL * (n - 3) + l] * rt[0 * 4 + 1] + f[NL * (n - 2) + l] * rt[0 * 4 + 2] + rn * rt[0 * 4 + 3]; \
            res[NL * (n - 2) + l] = f[NL * (n - 3) + l] * rt[1 * 4 + 0] + f[NL * (n - 2) + l] * rt[1 * 4 + 1] + rn * rt[1 * 4 + 2];  \
            bcs_t[l] = f[NL * (n - 3) + l] * rt[2 * 4 + 2] + f[NL * (n - 2) + l] * rt[2 * 4 + 0] + rn * rt[2 * 4 + 1];               \
        }                                                                                                                              \
        /* PENTADSS on rows 2..n-1 (0-based 1..n-2): g = res + NL, coefficient index i = row + 1 */                                   \
        {                                                                                                                              \
            double *g = res + NL;                                                                                                      \
            const int m = n - 2;                                                                                                       \
            const double *A = a + 1, *B = b + 1, *C = c + 1, *D = d + 1, *E = e + 1;                                                   \
            for (int l = 0; l < NL; ++l) g[NL * 1 + l] = g[NL * 1 + l] + g[l] * B[1];                                                  \
            for (int i = 2; i < m; ++i)                                                                                                \
                for (int l = 0; l < NL; ++l) g[NL * i + l] = g[NL * i + l] + g[NL * (i - 1) + l] * B[i] + g[NL * (i - 2) + l] * A[i]; \
            for (int l = 0; l < NL; ++l) g[NL * (m - 1) + l] = g[NL * (m - 1) + l] * C[m - 1];                                         \
            for (int l = 0; l < NL; ++l) g[NL * (m - 2) + l] = (g[NL * (m - 2) + l] + g[NL * (m - 1) + l] * D[m - 2]) * C[m - 2];     \
            for (int i = m - 3; i >= 0; --i)                                                                                           \
                for (int l = 0; l < NL; ++l)                                                                                           \
                    g[NL * i + l] = (g[NL * i + l] + g[NL * (i + 1) + l] * D[i] + g[NL * (i + 2) + l] * E[i]) * C[i];                  \
        }                                                                                                                              \
        /* lhs(row, diag) = s->lhs[diag * n + row]; idl = 3, ndl = 5, idr = 2 of the integral operator */                             \
        if (!s->bc_min) {                                                                                                              \
            for (int l = 0; l < NL; ++l) {                                                                                             \
                double v = bcs_b[l];                                                                                                   \
                v = v + d[0] * res[NL * 1 + l];                                                                                        \
                v = v + e[0] * res[NL * 2 + l];                                                                                        \
                v = v + a[0] * res[NL * 3 + l];                                                                                        \
                res[l] = v;                                                                                                            \
                if (du) {                                                                                                              \
                    double w = c[n - 1] * res[NL * (n - 1) + l];                                                                       \
                    w = w + b[n - 1] * res[NL * (n - 2) + l];                                                                          \
                    w = w + a[n - 1] * res[NL * (n - 3) + l];                                                                          \
                    w = w + e[n - 1] * res[NL * (n - 4) + l];                                                                          \
                    w = w + r1[n - 1] * f[NL * (n - 2) + l];                                                                           \
                    du[l] = w;                                                                                                         \
                }                                                                                                                      \
            }                                                                                                                          \
        } else {                                                                                                                       \
            const double *r3 = r2 + n;                                                                                                 \
            for (int l = 0; l < NL; ++l) {                                                                                             \
                double v = bcs_t[l];                                                                                                   \
                v = v + b[n - 1] * res[NL * (n - 2) + l];                                                                              \
                v = v + a[n - 1] * res[NL * (n - 3) + l];                                                                              \
                v = v + e[n - 1] * res[NL * (n - 4) + l];                                                                              \
                res[NL * (n - 1) + l] = v;                                                                                             \
                if (du) {                                                                                                              \
                    double w = c[0] * res[l];                                                                                          \
                    w = w + d[0] * res[NL * 1 + l];                                                                                    \
                    w = w + e[0] * res[NL * 2 + l];                                                                                    \
                    w = w + a[0] * res[NL * 3 + l];                                                                                    \
                    w = w + r3[0] * f[NL * 1 + l];                                                                                     \
                    du[l] = w;                                                                                                         \
                }                                                                                                                      \
            }                                                                                                                          \
        }                                                                                                                              \
    }
DEF_INT1_SOLVE(1)
DEF_INT1_SOLVE(2)
DEF_INT1_SOLVE(3)

/* OPR_ODE2_Factorize_NN   operators/opr_odes.f90:265-386.  f(2, n) (Re, Im), bcs[2][2] = (bottom, top) x (Re, Im); u, v out.
 * w: scratch of 12 n doubles. */
static void ode2_factorize_nn(const int1_t *fmin, const int1_t *fmax, double lam, double *f, const double *bcs, double *u, double *v, double *w) {
    const int n = fmin->n;
    double *f1 = w, *h2 = w + 3 * (size_t)n, *h1 = w + 6 * (size_t)n;     /* (3, n) each: f1; (v1, em, -); (u1, sp, ep) */
    double du0_n[2], der[3];
    f[2 * (n - 1)] = 0.0; f[2 * (n - 1) + 1] = 0.0;                       /* :302-305 v^(0): v' + lambda v = f, v_1 = 0 */
    v[0] = 0.0; v[1] = 0.0;
    int1_solve_2(fmin, f, v, NULL);
    memset(f1, 0, sizeof(double) * 3 * (size_t)n);                         /* :308-313 v^(1), e^(-) */
    f1[3 * (n - 1) + 0] = 1.0;
    h2[0] = 0.0; h2[1] = 1.0; h2[2] = 0.0;
    int1_solve_3(fmin, f1, h2, NULL);
    u[2 * (n - 1)] = 0.0; u[2 * (n - 1) + 1] = 0.0;                       /* :316-318 u^(0): u' - lambda u = v, u_n = 0 */
    int1_solve_2(fmax, v, u, du0_n);
    for (int i = 0; i < n; ++i) h2[3 * i + 2] = 0.0;                       /* :321-326 u^(1), s^(+), e^(+) */
    h1[3 * (n - 1) + 0] = 0.0; h1[3 * (n - 1) + 1] = 0.0; h1[3 * (n - 1) + 2] = 1.0;
    int1_solve_3(fmax, h2, h1, der);
#define V1(i) h2[3 * (i) + 0]
#define EM(i) h2[3 * (i) + 1]
#define U1(i) h1[3 * (i) + 0]
#define SP(i) h1[3 * (i) + 1]
#define EP(i) h1[3 * (i) + 2]
    const double du1_n = der[0], dsp_n = der[1], dep_n = der[2];
    double a11 = 1.0 + lam * SP(0), a21 = EM(n - 1), a31 = dsp_n;          /* :329-348 */
    double a12 = lam * EP(0), a22 = lam, a32 = dep_n;
    double a13 = lam * U1(0), a23 = V1(n - 1), a33 = du1_n;
    a12 = a12 / a11;
    a22 = a22 - a21 * a12;
    a32 = a32 - a31 * a12;
    a13 = a13 / a11;
    a23 = (a23 - a21 * a13) / a22;
    a33 = a33 - a31 * a13 - a32 * a23;
    for (int l = 0; l < 2; ++l) {                                           /* :350-367 */
        const double bb = bcs[l], bt = bcs[2 + l];
        double v0 = (bb - lam * u[l]) / a11;
        double un = (bt - v[2 * (n - 1) + l] - a21 * v0) / a22;
        const double fn = (bt - du0_n[l] - a31 * v0 - a32 * un) / a33;
        un = un - a23 * fn;
        v0 = v0 - a12 * un - a13 * fn;
        v[l] = v0; u[2 * (n - 1) + l] = un;
        int i = n - 1;
        v[2 * i + l] = v[2 * i + l] + fn * V1(i) + v0 * EM(i) + lam * u[2 * i + l];
        for (i = n - 2; i >= 1; --i) {
            u[2 * i + l] = u[2 * i + l] + fn * U1(i) + v0 * SP(i) + un * EP(i);
            v[2 * i + l] = v[2 * i + l] + fn * V1(i) + v0 * EM(i) + lam * u[2 * i + l];
        }
        i = 0;
        u[l] = u[l] + fn * U1(0) + v0 * SP(0) + un * EP(0);
        v[l] = v[l] + lam * u[l];
    }
}
/* OPR_ODE2_Factorize_NN_Sing -> _DN_Sing   operators/opr_odes.f90:165-183, 37-96 (lambda = 0 systems; bcs_b = 0) */
static void ode2_factorize_nn_sing(const int1_t *fmin, const int1_t *fmax, double *f, const double *bcs, double *u, double *v, double *w) {
    const int n = fmin->n;
    double *f1 = w, *v1 = w + n, *u1 = w + 2 * (size_t)n;
    double du0_n[2], du1_n[1];
    f[0] = 0.0; f[1] = 0.0;
    v[2 * (n - 1)] = bcs[2]; v[2 * (n - 1) + 1] = bcs[3];
    int1_solve_2(fmax, f, v, NULL);
    memset(f1, 0, sizeof(double) * n); f1[0] = 1.0;
    memset(v1, 0, sizeof(double) * n);
    int1_solve_1(fmax, f1, v1, NULL);
    u[0] = 0.0; u[1] = 0.0;                                  /* bcs_b = 0: p pinned at the bottom (:179-180) */
    int1_solve_2(fmin, v, u, du0_n);
    memset(u1, 0, sizeof(double) * n);
    int1_solve_1(fmin, v1, u1, du1_n);
    const double fac = 1.0 / (du1_n[0] - v1[0]);
    for (int l = 0; l < 2; ++l) {
        const double cc = (v[l] - du0_n[l]) * fac;
        for (int i = 0; i < n; ++i) {
            u[2 * i + l] = u[2 * i + l] + cc * u1[i];
            v[2 * i + l] = v[2 * i + l] + cc * v1[i];
        }
    }
}

/* OPR_Poisson_FourierXZ_Factorize(nx, ny, nz, BCS_NN, p, tmp1, tmp2, bcs_hb, bcs_ht, dpdy)   operators/opr_elliptic.f90:263-364
 * tmp1, tmp2, wrk3d: (nx+2)*ny*nz doubles each.  p in: forcing; out: solution.  dpdy out. */
int tlabcpu_opr_poisson(const cpu_poisson_t *P, double *p, double *tmp1, double *tmp2, double *wrk3d, const double *bcs_hb, const double *bcs_ht,
                        double *dpdy) {
    const int nx = P->nx, ny = P->ny, nz = P->nz, nxh = P->nxh;
    const size_t nxy = (size_t)nx * ny;
    /* :285-286 boundary data into the forcing */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; ++k) {
        memcpy(p + k * nxy, bcs_hb + (size_t)k * nx, sizeof(double) * nx);
        memcpy(p + k * nxy + (size_t)(ny - 1) * nx, bcs_ht + (size_t)k * nx, sizeof(double) * nx);
    }
    /* OPR_Fourier_X_Forward :288 (r2c per x line -> (nxh, ny, nz) complex in tmp2) */
#pragma omp parallel
    {
        double *x = (double *)malloc(sizeof(double) * 4 * (size_t)(nx > nz ? nx : nz)), *y = x + 2 * (size_t)(nx > nz ? nx : nz);
#pragma omp for schedule(static)
        for (long long line = 0; line < (long long)ny * nz; ++line) {
            const double *src = p + (size_t)line * nx;
            for (int i = 0; i < nx; ++i) { x[2 * i] = src[i]; x[2 * i + 1] = 0.0; }
            fft_exec(&P->fx, x, y, -1);
            memcpy(tmp2 + 2 * (size_t)line * nxh, x, sizeof(double) * 2 * nxh);
        }
        /* OPR_Fourier_Z_Forward :290 (c2c, stride nxh*ny) -> tmp1 ; :295 normalisation */
        const size_t stride = (size_t)nxh * ny;
#pragma omp for schedule(static)
        for (long long line = 0; line < (long long)stride; ++line) {
            for (int k = 0; k < nz; ++k) { x[2 * k] = tmp2[2 * (line + k * stride)]; x[2 * k + 1] = tmp2[2 * (line + k * stride) + 1]; }
            if (nz > 1) fft_exec(&P->fz, x, y, -1);
            for (int k = 0; k < nz; ++k) { tmp1[2 * (line + k * stride)] = x[2 * k] * P->norm; tmp1[2 * (line + k * stride) + 1] = x[2 * k + 1] * P->norm; }
        }
        free(x);
    }
    /* :301 TLab_Transpose_COMPLEX (nxh, ny*nz) -> (ny*nz, nxh): every mode's y line contiguous, f(2*ny, nz, nxh) */
    transpose_complex(tmp1, nxh, ny * nz, tmp2);
    /* :308-333 mode loop; u -> wrk3d (p^), v -> tmp1 (dp^/dy) */
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * 12 * (size_t)ny);
#pragma omp for schedule(static)
        for (long long r = 0; r < P->nreg; ++r) {
            const int m = P->reg_mode[r], k = m / nxh, i = m % nxh;
            const size_t off = 2 * ((size_t)ny * k + (size_t)ny * nz * i);
            double *f = tmp2 + off, *u = wrk3d + off, *v = tmp1 + off;
            const double bcs[4] = {f[0], f[1], f[2 * (ny - 1)], f[2 * (ny - 1) + 1]};           /* :310-311 */
            int1_t smin = {P->lhs[0] + (size_t)r * 5 * ny, P->rb[0] + (size_t)r * 12, P->rt[0] + (size_t)r * 12, P->rhs[0], ny, 1};
            int1_t smax = {P->lhs[1] + (size_t)r * 5 * ny, P->rb[1] + (size_t)r * 12, P->rt[1] + (size_t)r * 12, P->rhs[1], ny, 0};
            ode2_factorize_nn(&smin, &smax, P->lam[r], f, bcs, u, v, w);                        /* :318-319 */
        }
#pragma omp for schedule(static)
        for (long long r = 0; r < P->nsing; ++r) {
            const int m = P->sing_mode[r], k = m / nxh, i = m % nxh;
            const size_t off = 2 * ((size_t)ny * k + (size_t)ny * nz * i);
            double *f = tmp2 + off, *u = wrk3d + off, *v = tmp1 + off;
            const double bcs[4] = {f[0], f[1], f[2 * (ny - 1)], f[2 * (ny - 1) + 1]};
            int1_t smin = {P->slhs[0] + (size_t)r * 5 * ny, P->srb[0] + (size_t)r * 12, P->srt[0] + (size_t)r * 12, P->rhs[0], ny, 1};
            int1_t smax = {P->slhs[1] + (size_t)r * 5 * ny, P->srb[1] + (size_t)r * 12, P->srt[1] + (size_t)r * 12, P->rhs[1], ny, 0};
            ode2_factorize_nn_sing(&smin, &smax, f, bcs, u, v, w);                              /* :316 */
        }
        free(w);
    }
    /* :335-336 transpose back, :341-356 inverse transforms of p and dp/dy */
    for (int which = 0; which < 2; ++which) {
        double *spec = which == 0 ? wrk3d : tmp1, *out = which == 0 ? p : dpdy;
        if (which == 1 && !dpdy) break;
        transpose_complex(spec, ny * nz, nxh, tmp2);
#pragma omp parallel
        {
            double *x = (double *)malloc(sizeof(double) * 4 * (size_t)(nx > nz ? nx : nz)), *y = x + 2 * (size_t)(nx > nz ? nx : nz);
            const size_t stride = (size_t)nxh * ny;
#pragma omp for schedule(static)
            for (long long line = 0; line < (long long)stride; ++line) {
                if (nz > 1) {
                    for (int k = 0; k < nz; ++k) { x[2 * k] = tmp2[2 * (line + k * stride)]; x[2 * k + 1] = tmp2[2 * (line + k * stride) + 1]; }
                    fft_exec(&P->fz, x, y, +1);
                    for (int k = 0; k < nz; ++k) { tmp2[2 * (line + k * stride)] = x[2 * k]; tmp2[2 * (line + k * stride) + 1] = x[2 * k + 1]; }
                }
            }
#pragma omp for schedule(static)
            for (long long line = 0; line < (long long)ny * nz; ++line) {
                const double *src = tmp2 + 2 * (size_t)line * nxh;          /* c2r: Hermitian extension of the nx/2+1 coefficients */
                for (int i = 0; i < nxh; ++i) { x[2 * i] = src[2 * i]; x[2 * i + 1] = src[2 * i + 1]; }
                x[1] = 0.0;
                if (nx % 2 == 0) x[2 * (nx / 2) + 1] = 0.0;
                for (int i = nxh; i < nx; ++i) { x[2 * i] = src[2 * (nx - i)]; x[2 * i + 1] = -src[2 * (nx - i) + 1]; }
                fft_exec(&P->fx, x, y, +1);
                double *dst = out + (size_t)line * nx;
                for (int i = 0; i < nx; ++i) dst[i] = x[2 * i];
            }
            free(x);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * RHS_GLOBAL_INCOMPRESSIBLE_1   tools/dns/rhs_global_incompressible_1.f90:15-405 (convective form, no-slip walls / Dirichlet scalars,
 * no buffer zone, no IBM, no anelastic terms) + the update of TIME_SUBSTEP_INCOMPRESSIBLE_EXPLICIT (tools/dns/time.f90:645-664) and the
 * tendency scaling of TIME_RUNGEKUTTA (:261-298). */
typedef struct {
    int nx, ny, nz, nscal;
    const cpu_fdm_t *g[3];
    const double *lu2d[3][8];      /* fdmDiffusion(ig)%lu(:,:,is), is = 0 (visc), 1..nscal (visc/schmidt) */
    const cpu_poisson_t *poisson;
} cpu_dns_t;

#define PW for (size_t i = 0; i < n; ++i)
int tlabcpu_time_substep(const cpu_dns_t *D, double dte, double kco, int scale, double *const *q, double *const *s, double *const *hq,
                         double *const *hs, double *const *txc, double *wrk3d, double *wrk2d, double *bcs_hb, double *bcs_ht) {
    const int nx = D->nx, ny = D->ny, nz = D->nz;
    const size_t n = (size_t)nx * ny * nz;
    double *u = q[0], *v = q[1], *w = q[2];
    double *tmp1 = txc[0], *tmp2 = txc[1], *tmp3 = txc[2], *tmp4 = txc[3], *tmp5 = txc[4], *tmp6 = txc[5], *tmp7 = txc[6], *tmp8 = txc[7], *tmp9 = txc[8];
    const cpu_fdm_t *gx = D->g[0], *gy = D->g[1], *gz = D->g[2];
#define BX(ivel, is, a, r, t, ut) tlabcpu_opr_burgers(1, ivel, nx, ny, nz, 0, gx, D->lu2d[0][is], a, r, t, ut, wrk3d, wrk2d)
#define BY(ivel, is, a, r, t, ut) tlabcpu_opr_burgers(2, ivel, nx, ny, nz, 0, gy, D->lu2d[1][is], a, r, t, ut, wrk3d, wrk2d)
#define BZ(ivel, is, a, r, t, ut) tlabcpu_opr_burgers(3, ivel, nx, ny, nz, 0, gz, D->lu2d[2][is], a, r, t, ut, wrk3d, wrk2d)
    /* :98-100 the three SELF calls leave the transposed velocities in tmp4, tmp5, tmp6 */
    BX(0, 0, u, tmp1, tmp4, NULL);
    BY(0, 0, v, tmp2, tmp5, NULL);
    BZ(0, 0, w, tmp3, tmp6, w);
    const double *w_t = nz > 1 ? w : w;                     /* opr_burgers.f90:406-414: with npro_k = 1 the "transposed" w is w itself */
    /* :103-112 u equation */
    BY(1, 0, u, tmp7, tmp9, tmp5);
    BZ(1, 0, u, tmp8, tmp9, w_t);
#pragma omp parallel for schedule(static)
    PW hq[0][i] = hq[0][i] + tmp1[i] + tmp7[i] + tmp8[i];
    /* :115-124 v equation */
    BX(1, 0, v, tmp7, tmp9, tmp4);
    BZ(1, 0, v, tmp8, tmp9, w_t);
#pragma omp parallel for schedule(static)
    PW hq[1][i] = hq[1][i] + tmp2[i] + tmp7[i] + tmp8[i];
    /* :127-136 w equation */
    BX(1, 0, w, tmp7, tmp9, tmp4);
    BY(1, 0, w, tmp8, tmp9, tmp5);
#pragma omp parallel for schedule(static)
    PW hq[2][i] = hq[2][i] + tmp3[i] + tmp7[i] + tmp8[i];
    /* :149-162 scalars */
    for (int is = 0; is < D->nscal; ++is) {
        BX(1, is + 1, s[is], tmp1, tmp9, tmp4);
        BY(1, is + 1, s[is], tmp2, tmp9, tmp5);
        BZ(1, is + 1, s[is], tmp3, tmp9, w_t);
#pragma omp parallel for schedule(static)
        PW hs[is][i] = hs[is][i] + tmp1[i] + tmp2[i] + tmp3[i];
    }
    /* :188-201 forcing of the pressure equation */
    const double dummy = 1.0 / dte;
#pragma omp parallel for schedule(static)
    PW {
        tmp2[i] = hq[1][i] + v[i] * dummy;
        tmp3[i] = hq[0][i] + u[i] * dummy;
        tmp4[i] = hq[2][i] + w[i] * dummy;
    }
    /* :228-230, :257-259 */
    tlabcpu_opr_partial(2, 1, nx, ny, nz, 0, gy, tmp2, tmp1, NULL, wrk3d, wrk2d);
    tlabcpu_opr_partial(1, 1, nx, ny, nz, 0, gx, tmp3, tmp2, NULL, wrk3d, wrk2d);
    tlabcpu_opr_partial(3, 1, nx, ny, nz, 0, gz, tmp4, tmp3, NULL, wrk3d, wrk2d);
#pragma omp parallel for schedule(static)
    PW tmp1[i] = tmp1[i] + tmp2[i] + tmp3[i];
    /* :272-281 Neumann data of the pressure = wall planes of hq(:,2) */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; ++k) {
        memcpy(bcs_hb + (size_t)k * nx, hq[1] + (size_t)k * nx * ny, sizeof(double) * nx);
        memcpy(bcs_ht + (size_t)k * nx, hq[1] + (size_t)k * nx * ny + (size_t)(ny - 1) * nx, sizeof(double) * nx);
    }
    /* :284 */
    tlabcpu_opr_poisson(D->poisson, tmp1, tmp2, tmp4, wrk3d, bcs_hb, bcs_ht, tmp3);
    /* :319-320, :348-352 */
    tlabcpu_opr_partial(1, 1, nx, ny, nz, 0, gx, tmp1, tmp2, NULL, wrk3d, wrk2d);
    tlabcpu_opr_partial(3, 1, nx, ny, nz, 0, gz, tmp1, tmp4, NULL, wrk3d, wrk2d);
#pragma omp parallel for schedule(static)
    PW {
        hq[0][i] = hq[0][i] - tmp2[i];
        hq[1][i] = hq[1][i] - tmp3[i];
        hq[2][i] = hq[2][i] - tmp4[i];
    }
    /* :360-396 wall planes of the tendencies (Dirichlet: zero) */
    for (int f = 0; f < 3 + D->nscal; ++f) {
        double *h = f < 3 ? hq[f] : hs[f - 3];
#pragma omp parallel for schedule(static)
        for (int k = 0; k < nz; ++k) {
            memset(h + (size_t)k * nx * ny, 0, sizeof(double) * nx);
            memset(h + (size_t)k * nx * ny + (size_t)(ny - 1) * nx, 0, sizeof(double) * nx);
        }
    }
    /* time.f90:645-664 update, :261-298 scaling */
    for (int f = 0; f < 3 + D->nscal; ++f) {
        double *a = f < 3 ? q[f] : s[f - 3], *h = f < 3 ? hq[f] : hs[f - 3];
#pragma omp parallel for schedule(static)
        PW a[i] = a[i] + dte * h[i];
        if (scale) {
#pragma omp parallel for schedule(static)
            PW h[i] = kco * h[i];
        }
    }
    return 0;
}

/* threads of the following calls (0: leave) */
void tlabcpu_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
/* first touch of a freshly mapped array with the static partition the operators use, so that its pages land on the NUMA nodes of the
 * threads that will work on them (the reference does the same implicitly: every MPI rank allocates its own slab) */
void tlabcpu_fill(double *a, long long n, double value) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; ++i) a[i] = value;
}

int tlabcpu_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
