#include "profile.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tlab_amd.h"

namespace tlab {

namespace {
struct Rec {
    const char *tag;
    hipEvent_t a, b;
    double bytes;
};
bool g_on = false;
std::string g_only;      // empty: every launch is timed
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open_a = nullptr;
const char *g_open_tag = nullptr;
double g_open_bytes = 0.0;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_enabled(const char *tag) { return g_on && (g_only.empty() || (tag && g_only == tag)); }

void prof_begin(const char *tag, hipStream_t st, double bytes) {
    g_open_a = get_event();
    g_open_tag = tag;
    g_open_bytes = bytes;
    (void)hipEventRecord(g_open_a, st);
}

void prof_end(hipStream_t st) {
    hipEvent_t b = get_event();
    (void)hipEventRecord(b, st);
    g_recs.push_back(Rec{g_open_tag, g_open_a, b, g_open_bytes});
}

}  // namespace tlab

using namespace tlab;

extern "C" {

int tlab_profile_enable(int on) {
    g_on = on != 0;
    return TLAB_OK;
}

int tlab_profile_filter(const char *tag) {
    g_only = tag ? tag : "";
    return TLAB_OK;
}

int tlab_profile_reset(void) {
    for (auto &r : g_recs) {
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return TLAB_OK;
}

// Writes one line per kernel tag: "tag calls total_ms total_algorithmic_bytes\n".  Synchronises the device first.
int tlab_profile_report(char *buf, int nbuf) {
    if (!buf || nbuf < 2) return TLAB_EINVAL;
    (void)hipDeviceSynchronize();
    struct Acc { long long calls = 0; double ms = 0.0, bytes = 0.0; };
    std::map<std::string, Acc> acc;
    for (auto &r : g_recs) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        Acc &a = acc[r.tag];
        a.calls++;
        a.ms += ms;
        a.bytes += r.bytes;
    }
    std::string out;
    char line[512];
    for (auto &kv : acc) {
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\t%.0f\n", kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.bytes);
        out += line;
    }
    if ((int)out.size() + 1 > nbuf) return TLAB_EINVAL;
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

}  // extern "C"
