// Plan objects behind the C ABI: host tables (reference layout) + lazily built device tables.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/tlab_amd.h"
#include "chunked.hpp"
#include "device_tables.hpp"
#include "fdm_schemes.hpp"

namespace tlab {

struct DeviceArray {
    double *p = nullptr;
    size_t n = 0;
    DeviceArray() = default;
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    ~DeviceArray();
    void upload(const std::vector<double> &h);
    void alloc(size_t count);       // uninitialised scratch of count doubles (grows or shrinks to exactly that)
};

struct SystemEntry {
    ChunkedTables host;
    DeviceArray rowtab, red;
    bool lane_invariant = false;
    bool chunk_invariant = false;      // interior chunks 1 .. P-2 bitwise equal (k_ptile's compact tables)
    DeviceArray band;                  // the five central (cyclic) diagonals of the dense separator inverse, when the rest is negligible (SystemDev::band)
    bool band_ok = false;
    SystemDev dev() const { return SystemDev{rowtab.p, red.p, lane_invariant ? 1 : 0, chunk_invariant ? 1 : 0, band_ok ? band.p : nullptr}; }
};

}  // namespace tlab

// the opaque handle of include/tlab_amd.h
struct tlab_fdm_plan {
    tlab::FdmTables t;
    // (which = 1|2, ibc, P) -> chunked system; built on first use
    std::map<std::tuple<int, int, int>, std::unique_ptr<tlab::SystemEntry>> systems;
    std::unique_ptr<tlab::DeviceArray> jc;   // [3][n] Jacobian-correction diagonals (non-uniform grids)
    std::unique_ptr<tlab::DeviceArray> rowc2;  // [n][5] per-row RHS of a direct second-derivative scheme
    std::unique_ptr<tlab::DeviceArray> rowc1[4];   // the same for a direct first derivative, one per Neumann variant (rows 4 and n-3 differ)
    std::unique_ptr<tlab::DeviceArray> penta_ws;              // ... and two transposed copies of a field for its x direction
    std::unique_ptr<tlab::DeviceArray> penta_rhs, penta_lu;   // CompactJacobian6Penta first derivative: g%der1%rhs (n,7) and g%der1%lu on the device
    struct PentaTile { tlab::DeviceArray rows, blocks, smw; };
    std::map<int, std::unique_ptr<PentaTile>> penta_tile;     // ... and the tables of k_pentatile per Neumann variant (pentatile_build)
    tlab_filter_t interp[4] = {nullptr, nullptr, nullptr, nullptr};      // interpolatory operators P1 VP, P1 PV, P0 VP, P0 PV as periodic compact-filter objects (capi.cpp)
    int wide_ok[3] = {-1, -1, -1};             // x lines on 64 / 128 / 256 chunks: float-difference tables exact? (-1 = not checked yet; capi.cpp xline_wide_ok)

    tlab::TriDiag tridiag(int which, int ibc) const;            // tridiagonal matrix of the variant (wall rows -> identity)
    tlab::StencilDev stencil(int which, int ibc);               // RHS operator of the variant
    tlab::SystemEntry &system(int which, int ibc, int P);       // cached chunked factorization on the device
    tlab::JacCorrDev jaccorr();
};
