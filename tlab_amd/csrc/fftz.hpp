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

}  // namespace tlab
