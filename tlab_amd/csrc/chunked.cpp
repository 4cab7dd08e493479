// Host precomputation for the chunked Thomas solve (see chunked.hpp).  Init-time only.
#include "chunked.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

namespace tlab {

typedef long double ld;

static inline int wrap(int i, int n) { return ((i % n) + n) % n; }

void build_chunked(const TriDiag &T, int P, ChunkedTables &t) {
    const int n = T.n;
    if (P < 1 || n % P != 0) throw std::runtime_error("chunked: n not divisible by the number of chunks");
    const int m = n / P;
    if (m < 2) throw std::runtime_error("chunked: chunk length < 2");
    t.n = n; t.P = P; t.m = m; t.periodic = T.periodic;
    t.Lm.assign(n, 0.0); t.Dinv.assign(n, 0.0); t.Cm.assign(n, 0.0); t.V.assign(n, 0.0); t.W.assign(n, 0.0);
    t.alpha.assign(P, 0.0); t.beta.assign(P, 0.0); t.gamma.assign(P, 0.0);

    std::vector<ld> a(n), b(n), c(n);
    for (int i = 0; i < n; ++i) { a[i] = T.a[i]; b[i] = T.b[i]; c[i] = T.c[i]; }
    if (!T.periodic) { a[0] = 0; c[n - 1] = 0; }

    std::vector<ld> VL(P), WL(P), VF(P), WF(P);  // spikes at the last / first interior row of each chunk
    std::vector<ld> d(m), v(m), w(m), l(m);
    for (int j = 0; j < P; ++j) {
        const int s = j * m;
        const int ni = m - 1;  // interior rows s+1 .. s+ni
        // local LU of the interior block
        for (int p = 1; p <= ni; ++p) {
            const int i = s + p;
            if (p == 1) { l[p] = 0; d[p] = b[i]; }
            else { l[p] = a[i] / d[p - 1]; d[p] = b[i] - l[p] * c[i - 1]; }
            if (d[p] == 0) throw std::runtime_error("chunked: zero pivot in interior block");
        }
        // spikes: v = T_I^{-1} (a_{s+1} e_1), w = T_I^{-1} (c_{s+ni} e_ni)
        for (int p = 1; p <= ni; ++p) { v[p] = (p == 1) ? a[s + 1] : (ld)0; w[p] = (p == ni) ? c[s + ni] : (ld)0; }
        for (int p = 2; p <= ni; ++p) { v[p] -= l[p] * v[p - 1]; w[p] -= l[p] * w[p - 1]; }
        v[ni] /= d[ni]; w[ni] /= d[ni];
        for (int p = ni - 1; p >= 1; --p) {
            v[p] = (v[p] - c[s + p] * v[p + 1]) / d[p];
            w[p] = (w[p] - c[s + p] * w[p + 1]) / d[p];
        }
        for (int p = 1; p <= ni; ++p) {
            const int i = s + p;
            t.Lm[i] = (double)(-l[p]);
            t.Dinv[i] = (double)(1 / d[p]);
            t.Cm[i] = (p == ni) ? 0.0 : (double)(-c[i] / d[p]);
            t.V[i] = (double)(-v[p]);
            t.W[i] = (double)(-w[p]);
        }
        VF[j] = -v[1]; WF[j] = -w[1]; VL[j] = -v[ni]; WL[j] = -w[ni];
        // separator row: couplings kept in the Lm / Cm slots
        t.Lm[s] = (double)a[s];
        t.Cm[s] = (double)c[s];
    }
    // separator system (cyclic when periodic)
    std::vector<ld> al(P), be(P), ga(P);
    for (int j = 0; j < P; ++j) {
        const int s = j * m;
        const int jm = wrap(j - 1, P);
        // x_{s-1} = yL_{j-1} + VL_{j-1} X_{j-1} + WL_{j-1} X_j ; x_{s+1} = yF_j + VF_j X_j + WF_j X_{j+1}
        al[j] = a[s] * VL[jm];
        be[j] = b[s] + a[s] * WL[jm] + c[s] * VF[j];
        ga[j] = c[s] * WF[j];
        if (!T.periodic && j == 0) al[j] = 0;
        if (!T.periodic && j == P - 1) ga[j] = 0;
        t.alpha[j] = (double)al[j]; t.beta[j] = (double)be[j]; t.gamma[j] = (double)ga[j];
    }
    if (P == 1) {  // single chunk: X_0 couples to itself through both spikes when periodic
        be[0] += al[0] + ga[0];
        al[0] = ga[0] = 0;
    }

    // dense matrix of the separator system
    std::vector<ld> M((size_t)P * P, 0);
    for (int j = 0; j < P; ++j) {
        M[(size_t)j * P + j] += be[j];
        if (P > 1) {
            M[(size_t)j * P + wrap(j - 1, P)] += al[j];
            M[(size_t)j * P + wrap(j + 1, P)] += ga[j];
        }
    }
    // (1) dense inverse by Gauss-Jordan with partial pivoting
    if (P <= 32) {
        std::vector<ld> A(M), I((size_t)P * P, 0);
        for (int j = 0; j < P; ++j) I[(size_t)j * P + j] = 1;
        for (int k = 0; k < P; ++k) {
            int piv = k;
            for (int r = k + 1; r < P; ++r)
                if (fabsl(A[(size_t)r * P + k]) > fabsl(A[(size_t)piv * P + k])) piv = r;
            if (A[(size_t)piv * P + k] == 0) throw std::runtime_error("chunked: singular separator system");
            if (piv != k)
                for (int q = 0; q < P; ++q) { std::swap(A[(size_t)k * P + q], A[(size_t)piv * P + q]); std::swap(I[(size_t)k * P + q], I[(size_t)piv * P + q]); }
            ld inv = 1 / A[(size_t)k * P + k];
            for (int q = 0; q < P; ++q) { A[(size_t)k * P + q] *= inv; I[(size_t)k * P + q] *= inv; }
            for (int r = 0; r < P; ++r) {
                if (r == k) continue;
                ld f = A[(size_t)r * P + k];
                if (f == 0) continue;
                for (int q = 0; q < P; ++q) { A[(size_t)r * P + q] -= f * A[(size_t)k * P + q]; I[(size_t)r * P + q] -= f * I[(size_t)k * P + q]; }
            }
        }
        t.ginv.resize((size_t)P * P);
        for (size_t q = 0; q < (size_t)P * P; ++q) t.ginv[q] = (double)I[q];
    }
    // (2) parallel cyclic reduction schedule, simulated on the dense matrix
    t.pcr_steps = 0;
    if (P == 1) t.pcr_dinv.assign(1, (double)(1 / M[0]));
    if (P >= 2 && (P & (P - 1)) == 0) {
        int steps = 0;
        while ((1 << steps) < P) ++steps;
        t.pcr_steps = steps;
        t.pcr_k1.assign((size_t)steps * P, 0.0);
        t.pcr_k2.assign((size_t)steps * P, 0.0);
        t.pcr_dinv.assign(P, 0.0);
        std::vector<ld> A(M), B((size_t)P * P);
        for (int s = 0; s < steps; ++s) {
            const int dd = 1 << s;
            for (int j = 0; j < P; ++j) {
                ld k1 = 0, k2 = 0;
                int jl = j - dd, jr = j + dd;
                bool hasl, hasr;
                if (T.periodic) { jl = wrap(jl, P); jr = wrap(jr, P); hasl = hasr = true; }
                else { hasl = jl >= 0; hasr = jr < P; }
                if (hasl && hasr && jl == jr) {  // both neighbours are the same equation (d == P/2, cyclic)
                    k1 = A[(size_t)j * P + jl] / A[(size_t)jl * P + jl];
                    hasr = false;
                } else {
                    if (hasl) k1 = A[(size_t)j * P + jl] / A[(size_t)jl * P + jl];
                    if (hasr) k2 = A[(size_t)j * P + jr] / A[(size_t)jr * P + jr];
                }
                for (int q = 0; q < P; ++q) {
                    ld val = A[(size_t)j * P + q];
                    if (hasl) val -= k1 * A[(size_t)jl * P + q];
                    if (hasr) val -= k2 * A[(size_t)jr * P + q];
                    B[(size_t)j * P + q] = val;
                }
                // the entries just eliminated are zero by construction; clear their rounding residue
                if (hasl) B[(size_t)j * P + jl] = (jl == j) ? B[(size_t)j * P + jl] : (ld)0;
                if (hasr) B[(size_t)j * P + jr] = (jr == j) ? B[(size_t)j * P + jr] : (ld)0;
                t.pcr_k1[(size_t)s * P + j] = (double)k1;
                t.pcr_k2[(size_t)s * P + j] = (double)k2;
            }
            A.swap(B);
        }
        for (int j = 0; j < P; ++j) {
            ld off = 0;
            for (int q = 0; q < P; ++q)
                if (q != j) off = std::fmax(off, fabsl(A[(size_t)j * P + q]));
            if (off > 1e-13L * fabsl(A[(size_t)j * P + j]))
                throw std::runtime_error("chunked: PCR did not diagonalise the separator system (off-diagonal " + std::to_string((double)off) + ")");
            t.pcr_dinv[j] = (double)(1 / A[(size_t)j * P + j]);
        }
    }
    // (3) two-level reduction for W = P / 64 waves (see chunked.hpp)
    t.tl_waves = 0;
    t.tl.clear();
    if (P > 64 && P % 64 == 0) {
        const int W = P / 64, B = 64;
        std::vector<double> tl((size_t)21 * P, 0.0);
        std::vector<ld> vs((size_t)P), ws((size_t)P);
        bool ok = true;
        for (int w = 0; w < W && ok; ++w) {
            const int j0 = w * B;
            // PCR schedule of the isolated block (no wrap-around: out-of-range neighbours carry zero coefficients)
            std::vector<ld> A((size_t)B * B, 0), Bm((size_t)B * B);
            for (int l = 0; l < B; ++l) {
                A[(size_t)l * B + l] = be[j0 + l];
                if (l > 0) A[(size_t)l * B + l - 1] = al[j0 + l];
                if (l < B - 1) A[(size_t)l * B + l + 1] = ga[j0 + l];
            }
            for (int s = 0; s < 6; ++s) {
                const int dd = 1 << s;
                for (int l = 0; l < B; ++l) {
                    const int jl = l - dd, jr = l + dd;
                    const bool hasl = jl >= 0, hasr = jr < B;
                    ld k1 = 0, k2 = 0;
                    if (hasl) k1 = A[(size_t)l * B + jl] / A[(size_t)jl * B + jl];
                    if (hasr) k2 = A[(size_t)l * B + jr] / A[(size_t)jr * B + jr];
                    for (int q = 0; q < B; ++q) {
                        ld val = A[(size_t)l * B + q];
                        if (hasl) val -= k1 * A[(size_t)jl * B + q];
                        if (hasr) val -= k2 * A[(size_t)jr * B + q];
                        Bm[(size_t)l * B + q] = val;
                    }
                    if (hasl) Bm[(size_t)l * B + jl] = 0;
                    if (hasr) Bm[(size_t)l * B + jr] = 0;
                    tl[(size_t)s * P + j0 + l] = (double)k1;
                    tl[(size_t)(6 + s) * P + j0 + l] = (double)k2;
                }
                A.swap(Bm);
            }
            for (int l = 0; l < B; ++l) {
                ld off = 0;
                for (int q = 0; q < B; ++q)
                    if (q != l) off = std::fmax(off, fabsl(A[(size_t)l * B + q]));
                if (off > 1e-13L * fabsl(A[(size_t)l * B + l])) ok = false;
                tl[(size_t)12 * P + j0 + l] = (double)(1 / A[(size_t)l * B + l]);
            }
            // spikes of the block: S_w^{-1} e_first al_first, S_w^{-1} e_last ga_last (Thomas, long double)
            std::vector<ld> dd(B), r1(B, 0), r2(B, 0);
            r1[0] = al[j0]; r2[B - 1] = ga[j0 + B - 1];
            dd[0] = be[j0];
            for (int l = 1; l < B; ++l) {
                const ld m2 = al[j0 + l] / dd[l - 1];
                dd[l] = be[j0 + l] - m2 * ga[j0 + l - 1];
                r1[l] -= m2 * r1[l - 1]; r2[l] -= m2 * r2[l - 1];
            }
            r1[B - 1] /= dd[B - 1]; r2[B - 1] /= dd[B - 1];
            for (int l = B - 2; l >= 0; --l) {
                r1[l] = (r1[l] - ga[j0 + l] * r1[l + 1]) / dd[l];
                r2[l] = (r2[l] - ga[j0 + l] * r2[l + 1]) / dd[l];
            }
            for (int l = 0; l < B; ++l) { vs[j0 + l] = r1[l]; ws[j0 + l] = r2[l]; }
            if (fabsl(r1[B - 1]) > 1e-30L || fabsl(r2[0]) > 1e-30L) ok = false;      // coupling across a whole block must vanish
        }
        if (ok) {
            for (int w = 0; w < W; ++w) {
                const int j0 = w * B, wn = (w + 1) % W, wp = (w + W - 1) % W;
                const ld wL = ws[j0 + B - 1], vF = vs[(size_t)wn * B], wLp = ws[(size_t)wp * B + B - 1], vFm = vs[j0];
                for (int l = 0; l < B; ++l) {
                    const int j = j0 + l;
                    tl[(size_t)13 * P + j] = (double)vs[j];
                    tl[(size_t)14 * P + j] = (double)ws[j];
                    tl[(size_t)15 * P + j] = (double)wL;
                    tl[(size_t)16 * P + j] = (double)vF;
                    tl[(size_t)17 * P + j] = (double)(1 / (1 - wL * vF));
                    tl[(size_t)18 * P + j] = (double)wLp;
                    tl[(size_t)19 * P + j] = (double)vFm;
                    tl[(size_t)20 * P + j] = (double)(1 / (1 - wLp * vFm));
                }
            }
            t.tl.swap(tl);
            t.tl_waves = W;
        }
    }
}

void chunked_solve_host(const ChunkedTables &t, double *f, bool use_pcr) {
    const int n = t.n, P = t.P, m = t.m;
    std::vector<double> y(n), r(P), X(P);
    for (int j = 0; j < P; ++j) {
        const int s = j * m;
        double g = 0.0;
        for (int p = 1; p < m; ++p) { g = f[s + p] + t.Lm[s + p] * g; y[s + p] = g; }
        double yn = 0.0;
        for (int p = m - 1; p >= 1; --p) { yn = y[s + p] * t.Dinv[s + p] + t.Cm[s + p] * yn; y[s + p] = yn; }
    }
    for (int j = 0; j < P; ++j) {
        const int s = j * m;
        const int jm = wrap(j - 1, P);
        double yL = y[jm * m + m - 1];  // multiplied by a_s = 0 when there is no left neighbour
        r[j] = f[s] - t.Lm[s] * yL - t.Cm[s] * y[s + 1];
    }
    if (use_pcr && t.tl_waves > 0) {        // the two-level reduction exactly as k_xline does it on several waves per line
        const int W = t.tl_waves, B = 64;
        auto TL = [&](int q, int j) { return t.tl[(size_t)q * P + j]; };
        std::vector<double> Y(r), Y2(P);
        for (int s = 0; s < 6; ++s) {
            const int dd = 1 << s;
            for (int j = 0; j < P; ++j) {
                const int w = j / B, l = j % B;
                const double rl = Y[w * B + ((l - dd) & (B - 1))], rr = Y[w * B + ((l + dd) & (B - 1))];      // wrap inside the wave: zero coefficients there
                Y2[j] = Y[j] - TL(s, j) * rl - TL(6 + s, j) * rr;
            }
            Y.swap(Y2);
        }
        for (int j = 0; j < P; ++j) Y[j] = Y[j] * TL(12, j);
        for (int j = 0; j < P; ++j) {
            const int w = j / B, wn = (w + 1) % W, wp = (w + W - 1) % W;
            const double myF = Y[w * B], myL = Y[w * B + B - 1], YpL = Y[wp * B + B - 1], YnF = Y[wn * B];
            const double XL = (myL - TL(15, j) * YnF) * TL(17, j), XnF = YnF - TL(16, j) * XL;
            const double XpL = (YpL - TL(18, j) * myF) * TL(20, j);
            X[j] = Y[j] - TL(13, j) * XpL - TL(14, j) * XnF;
        }
    } else if (use_pcr) {
        if (t.pcr_steps == 0 && P > 1) throw std::runtime_error("chunked: no PCR tables");
        std::vector<double> r2(P);
        for (int s = 0; s < t.pcr_steps; ++s) {
            const int dd = 1 << s;
            for (int j = 0; j < P; ++j)
                r2[j] = r[j] - t.pcr_k1[(size_t)s * P + j] * r[wrap(j - dd, P)] - t.pcr_k2[(size_t)s * P + j] * r[wrap(j + dd, P)];
            r.swap(r2);
        }
        for (int j = 0; j < P; ++j) X[j] = r[j] * t.pcr_dinv[j];
    } else {
        for (int j = 0; j < P; ++j) {
            double acc = 0.0;
            for (int q = 0; q < P; ++q) acc += t.ginv[(size_t)j * P + q] * r[q];
            X[j] = acc;
        }
    }
    for (int j = 0; j < P; ++j) {
        const int s = j * m;
        const double Xl = X[j], Xr = X[wrap(j + 1, P)];
        f[s] = Xl;
        for (int p = 1; p < m; ++p) f[s + p] = y[s + p] + t.V[s + p] * Xl + t.W[s + p] * Xr;
    }
}

void tridiag_solve_direct(const TriDiag &T, double *f) {
    const int n = T.n;
    std::vector<ld> A((size_t)n * n, 0), x(n);
    for (int i = 0; i < n; ++i) {
        A[(size_t)i * n + i] += T.b[i];
        if (i > 0) A[(size_t)i * n + i - 1] += T.a[i];
        else if (T.periodic) A[(size_t)i * n + n - 1] += T.a[i];
        if (i < n - 1) A[(size_t)i * n + i + 1] += T.c[i];
        else if (T.periodic) A[(size_t)i * n + 0] += T.c[i];
        x[i] = f[i];
    }
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int r = k + 1; r < n; ++r)
            if (fabsl(A[(size_t)r * n + k]) > fabsl(A[(size_t)piv * n + k])) piv = r;
        if (piv != k) {
            for (int q = 0; q < n; ++q) std::swap(A[(size_t)k * n + q], A[(size_t)piv * n + q]);
            std::swap(x[k], x[piv]);
        }
        for (int r = k + 1; r < n; ++r) {
            ld fct = A[(size_t)r * n + k] / A[(size_t)k * n + k];
            if (fct == 0) continue;
            for (int q = k; q < n; ++q) A[(size_t)r * n + q] -= fct * A[(size_t)k * n + q];
            x[r] -= fct * x[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        ld s = x[k];
        for (int q = k + 1; q < n; ++q) s -= A[(size_t)k * n + q] * x[q];
        x[k] = s / A[(size_t)k * n + k];
    }
    for (int i = 0; i < n; ++i) f[i] = (double)x[i];
}

}  // namespace tlab
