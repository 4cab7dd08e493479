// Strided complex FFT along z (fftz.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace tlab {

class FftzPlan {
public:
    static bool supported(int n);                       // n = 8^a * {1, 2, 4}, 16 <= n <= 2048
    FftzPlan(int n, long long nlines);                  // element (line l, point k) at l + nlines * k, complex interleaved
    ~FftzPlan();
    FftzPlan(const FftzPlan &) = delete;
    FftzPlan &operator=(const FftzPlan &) = delete;
    void exec(int dir, const double *in, double *out, hipStream_t st) const;      // dir > 0 forward (exp(-i)), else backward; in == out allowed

private:
    int n;
    long long nlines;
    std::vector<int> radix;
    double *d_tw = nullptr;
};

// real -> complex FFT of contiguous lines in one pass (fftz.hip: k_fftx_r2c); out-of-place, line l: n reals at in + l n -> n/2+1 complex at out + l (n/2+1)
class FftxPlan {
public:
    static bool supported(int n);                       // n/2 = 8^a * {1, 2, 4}, 128 <= n <= 2048
    FftxPlan(int n, long long nlines);
    ~FftxPlan();
    FftxPlan(const FftxPlan &) = delete;
    FftxPlan &operator=(const FftxPlan &) = delete;
    // kxoff, kxw (device, n/2+1 entries each; both NULL: contiguous lines): the complex side lives in the slab <-> kx-pencil pack layout, element
    // (line, kx) at kxoff[kx] + line * kxw[kx] complex values (tlab_pencil_repack_blocks folded into the transform)
    void exec(const double *in, double *out, hipStream_t st, const long long *kxoff = nullptr, const int *kxw = nullptr) const;
    // the inverse (n/2+1 complex -> n reals per line, unnormalised)
    void exec_inverse(const double *in, double *out, hipStream_t st, const long long *kxoff = nullptr, const int *kxw = nullptr) const {
        launch_inverse(in, out, nullptr, nullptr, 0.0, 0.0, 0, 0, st, kxoff, kxw);
    }
    // the inverse as the pressure-gradient operand g of the final update of one velocity component (pointwise.hip: k_final_update with zero wall
    // planes): h = h - g, h = 0 on the rows j = 0, ny-1 of every x-y plane, q += dte h, h *= kco (if scale); g itself is not stored
    void exec_inverse_final(const double *in, double *q, double *h, double dte, double kco, int scale, int ny, hipStream_t st,
                            const long long *kxoff = nullptr, const int *kxw = nullptr) const {
        launch_inverse(in, nullptr, q, h, dte, kco, scale, ny, st, kxoff, kxw);
    }

private:
    void launch_inverse(const double *in, double *out, const double *q, const double *h, double dte, double kco, int scale, int ny, hipStream_t st,
                        const long long *kxoff, const int *kxw) const;
    int n;
    long long nlines;
    std::vector<int> radix;
    double *d_tw = nullptr;
};

}  // namespace tlab
