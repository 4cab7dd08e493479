// Chunked (partitioned) Thomas factorization of one tridiagonal system, precomputed on the host at plan time.
//
// Why: the reference solves A x = f by a serial Thomas sweep along the line (utils/linear3.f90 TRIDSS/TRIDPSS),
// which on a GPU would need the whole line's intermediate to be stored and re-read (32 B/point instead of 16).
// Because A is the same for every line, everything that depends on A only is precomputed here, and the kernels
// solve each line in P independent chunks that live in registers, coupled through a P x P "separator" system:
//
//   chunk j = rows [j*m, (j+1)*m); its first row s_j is the separator, rows s_j+1 .. s_j+m-1 the interior.
//   interior:  x_i = y_i + V_i X_j + W_i X_{j+1},  y = local Thomas solve of the interior block (tables Lm, Dinv, Cm)
//   separator: alpha_j X_{j-1} + beta_j X_j + gamma_j X_{j+1} = f_s - a_s yL_{j-1} - c_s yF_j   (cyclic if periodic)
//
// The separator system is solved either by parallel cyclic reduction across the 64 lanes of a wave (P == 64, tables
// pcr_k1/pcr_k2/pcr_dinv) or with its dense inverse (P <= 16, table ginv).  Exact in exact arithmetic; in fp64 it
// differs from the serial sweep by rounding only (~1e-16 relative, tests bound it by 1e-12 as north_star asks).
#pragma once
#include <vector>

namespace tlab {

struct TriDiag {           // a_i x_{i-1} + b_i x_i + c_i x_{i+1} = f_i ; a_0 / c_{n-1} wrap when periodic, else ignored
    int n = 0;
    bool periodic = false;
    std::vector<double> a, b, c;
};

struct ChunkedTables {
    int n = 0, P = 0, m = 0;
    bool periodic = false;
    // per-row tables, each of length n.  Interior rows: Lm, Dinv, Cm, V, W as in the header comment.
    // Separator rows reuse the slots: Lm = a_s, Cm = c_s (Dinv, V, W unused = 0).
    std::vector<double> Lm, Dinv, Cm, V, W;
    // separator system
    std::vector<double> alpha, beta, gamma;      // (P)
    std::vector<double> ginv;                    // (P*P) row-major dense inverse (filled when P <= 32)
    std::vector<double> pcr_k1, pcr_k2;          // (nsteps*P): r_j -= k1*r_{j-d} + k2*r_{j+d}, d = 2^s (filled when P is a power of 2)
    std::vector<double> pcr_dinv;                // (P)
    int pcr_steps = 0;
    // Two-level reduction for P = 64 W chunks handled by W waves of 64 lanes (k_xline on several waves per line): every wave reduces the
    // isolated 64 x 64 block of its own separator unknowns with wave shuffles (PCR schedule of that block), the blocks are coupled through
    // one value at each end (the coupling across a whole block is ~(4e-4)^63: zero), and the block spikes correct the lanes next to the ends.
    //   tl[q][P], q = 0-5 k1, 6-11 k2, 12 dinv, 13 vs, 14 ws (x = y - vs X_prev_last - ws X_next_first),
    //   15 wL = ws[63] of the lane's wave, 16 vF = vs[0] of the next wave, 17 1/(1 - wL vF), 18-20 the same for the interface with the previous wave.
    std::vector<double> tl;
    int tl_waves = 0;                            // 0: not available (P not a multiple of 64 > 64, or the neglected couplings are not negligible)
};

// Builds the tables; throws std::runtime_error if n is not divisible by P, m < 2, or a pivot vanishes.
void build_chunked(const TriDiag &T, int P, ChunkedTables &out);

// Scalar emulation of the device algorithm (debug/tests only): f (n) in, x out.  use_pcr selects the PCR path.
void chunked_solve_host(const ChunkedTables &t, double *f, bool use_pcr);      // P = 128, 256 with use_pcr: the two-level reduction (tl) when available

// Reference direct solve (Gaussian elimination on the cyclic tridiagonal, long double) for self-checks.
void tridiag_solve_direct(const TriDiag &T, double *f);

}  // namespace tlab
