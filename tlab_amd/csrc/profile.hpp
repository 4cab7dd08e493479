// Optional in-library kernel timing with HIP events on the library's stream (bench.py's "roofline" object needs the average
// duration of a kernel measured live over the timed region, on the stream the kernel is launched on).
#pragma once
#include <hip/hip_runtime.h>

namespace tlab {

bool prof_enabled(const char *tag);      // timing on, and the tag passes the filter (tlab_profile_filter)
void prof_begin(const char *tag, hipStream_t st, double bytes);   // bytes = algorithmic (compulsory operand) bytes of this launch
void prof_end(hipStream_t st);

struct ProfScope {
    hipStream_t st;
    bool on;
    ProfScope(const char *tag, hipStream_t s, double bytes) : st(s), on(prof_enabled(tag)) {
        if (on) prof_begin(tag, st, bytes);
    }
    ~ProfScope() {
        if (on) prof_end(st);
    }
};

}  // namespace tlab
