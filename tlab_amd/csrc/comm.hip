// libtlab_amd_comm.so -- pencil transpositions of the hot path over RCCL (include/tlab_amd_comm.h).
//
// Reference: TLabMPI_Trp_PlanI / PlanK and TLabMPI_Trp_Exec{I,K}_{Forward,Backward} (base/tlab_mpi_transpose.f90:205-553), which move the data with
// MPI_ISEND/IRECV or MPI_ALLTOALLW on derived vector datatypes (:557-610).  RCCL has neither derived datatypes nor alltoallw; here every
// transposition has exactly ONE strided side (see below), served by one HIP copy kernel, and the exchange itself is a grouped ncclSend / ncclRecv of
// contiguous blocks straight from / into the caller's array on the other side -- one staging buffer per plan, one extra pass over the local data.
//
//   strided array S and wire format W share one index pattern (m = inner length in doubles, c = outer count, P = ranks of the direction):
//       S[(q m + r) + (m P) o]   <->   W[q (m c) + r + m o]          q = peer, r < m, o < c
//   I (x pencils): m = imax e, c = nlines ; S = b(imax P, nlines), W = blocks of my lines' x-segments      (tlab_mpi_transpose.f90:232-256)
//   K (z pencils): m = nlines e, c = kmax ; S = a(npage, kmax),    W = blocks (nlines, kmax) per peer      (:301-325)
//   e = 1 (real) or 2 (complex).  Forward I unpacks (W -> S), backward I packs; forward K packs (S -> W), backward K unpacks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/tlab_amd.h"
#include "../../include/tlab_amd_comm.h"

extern hipStream_t tlab_current_stream();
extern void tlab_set_error(const std::string &s);
extern bool tlab_device_ready();

namespace {

struct Fail {
    int code;
    std::string msg;
};
void hipc(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Fail{TLAB_EHIP, std::string(what) + ": " + hipGetErrorString(e)};
}
void ncc(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw Fail{TLAB_EHIP, std::string(what) + ": " + ncclGetErrorString(r)};
}

// to_wire != 0: W[i] = S[..]; else S[..] = W[i].  i runs over the wire order, so the W side is a plain stream and the S side is contiguous in
// runs of m doubles.  VEC = 2: 16-byte accesses (m even, bases 16-byte aligned).
template <int VEC>
__global__ void __launch_bounds__(256) k_trp_copy(double *__restrict__ S, double *__restrict__ W, long long m, int P, long long c, int to_wire) {
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

// TransposeType{I,K} = single (tlab_mpi_transpose.f90:106-122, :362-371, :473-482): the data travel as fp32.  The same index pattern with a float wire;
// strided = 0: S is read / written in wire order (the side that is blocked by peer as it stands).  to_wire: W[i] = (float) S[..]; else S[..] = (double) W[i].
template <int VEC>
__global__ void __launch_bounds__(256) k_trp_copy_f32(double *__restrict__ S, float *__restrict__ W, long long m, int P, long long c, int to_wire, int strided) {
    const long long mv = m / VEC, total = mv * c * P, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i % mv, o = (i / mv) % c, q = i / (mv * c);
        const long long s = strided ? (q * mv + r) + (mv * P) * o : i;
        if (VEC == 2) {
            double2 *S2 = reinterpret_cast<double2 *>(S);
            float2 *W2 = reinterpret_cast<float2 *>(W);
            if (to_wire) { const double2 v = S2[s]; W2[i] = make_float2((float)v.x, (float)v.y); }
            else { const float2 v = W2[i]; S2[s] = make_double2((double)v.x, (double)v.y); }
        } else {
            if (to_wire) W[i] = (float)S[s];
            else S[s] = (double)W[i];
        }
    }
}

}  // namespace

struct tlab_comm {
    ncclComm_t world = nullptr, cx = nullptr, cz = nullptr;
    int nranks = 1, rank = 0, npro_i = 1, npro_k = 1, pro_i = 0, pro_k = 0;
    hipStream_t stream = nullptr;      // the library's communication stream
    // slab transport (tlab_comm_slab_transport): events of the exchanges in flight, device scratch of the all-reduce
    static constexpr int NEV = 64;
    hipEvent_t ev_start[NEV] = {}, ev_done[NEV] = {};
    int next_ticket = 0;
    double *red = nullptr;
};

struct tlab_trp_plan {
    tlab_comm *comm = nullptr;
    ncclComm_t nc = nullptr;
    int dir = 0, P = 1, rank = 0, e = 1;
    long long m = 0, c = 0, blk = 0, local = 0, nlines = 0;
    double *stage = nullptr;
    bool single = false;               // fp32 on the wire (real plans only): stage = send floats | receive floats
    hipStream_t stream = nullptr;      // comm->stream, or an own one for plans without a communicator
    bool own_stream = false;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    // the transposition in flight
    bool pending = false, pending_unpack = false;
    double *pending_out = nullptr;
};

namespace {

void copy_strided(const tlab_trp_plan *p, double *S, double *W, int to_wire, hipStream_t st) {
    const long long total = p->m * p->c * p->P;
    const bool vec = (p->m % 2 == 0) && ((reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
    const long long work = vec ? total / 2 : total;
    const int grid = (int)std::min<long long>(4096, std::max<long long>(1, (work + 255) / 256));
    if (vec) hipLaunchKernelGGL(k_trp_copy<2>, dim3(grid), dim3(256), 0, st, S, W, p->m, p->P, p->c, to_wire);
    else hipLaunchKernelGGL(k_trp_copy<1>, dim3(grid), dim3(256), 0, st, S, W, p->m, p->P, p->c, to_wire);
    hipc(hipGetLastError(), "k_trp_copy");
}
bool strided_on_send(const tlab_trp_plan *p, int forward) { return (p->dir == 1 && !forward) || (p->dir == 3 && forward); }
void copy_f32(const tlab_trp_plan *p, double *S, float *W, int to_wire, int strided, hipStream_t st) {
    const long long total = p->m * p->c * p->P;
    const bool vec = (p->m % 2 == 0) && (reinterpret_cast<uintptr_t>(S) % 16 == 0) && (reinterpret_cast<uintptr_t>(W) % 8 == 0);
    const long long work = vec ? total / 2 : total;
    const int grid = (int)std::min<long long>(4096, std::max<long long>(1, (work + 255) / 256));
    if (vec) hipLaunchKernelGGL(k_trp_copy_f32<2>, dim3(grid), dim3(256), 0, st, S, W, p->m, p->P, p->c, to_wire, strided);
    else hipLaunchKernelGGL(k_trp_copy_f32<1>, dim3(grid), dim3(256), 0, st, S, W, p->m, p->P, p->c, to_wire, strided);
    hipc(hipGetLastError(), "k_trp_copy_f32");
}

int guard(const Fail &f) {
    tlab_set_error(f.msg);
    return f.code;
}

}  // namespace

// ---- tlab_slab_transport over RCCL (include/tlab_amd.h: the exchanges of the z-slab driver, csrc/slab.cpp) ----------------------------------
// Every exchange is one grouped ncclSend / ncclRecv on the communication stream: it starts after the work enqueued on the caller's stream so far
// (event), and the caller's stream takes it up again at `wait` (event) -- in between the x / y operators run beside it.  Exchanges are issued in
// program order on every rank, so the in-order communication stream matches them up.
namespace {

ncclComm_t zcomm(tlab_comm *c) { return c->cz ? c->cz : c->world; }

int slab_begin(tlab_comm *c, hipStream_t cur) {
    const int t = c->next_ticket;
    c->next_ticket = (t + 1) % tlab_comm::NEV;
    if (!c->ev_start[t]) {
        hipc(hipEventCreateWithFlags(&c->ev_start[t], hipEventDisableTiming), "hipEventCreate");
        hipc(hipEventCreateWithFlags(&c->ev_done[t], hipEventDisableTiming), "hipEventCreate");
    }
    hipc(hipEventRecord(c->ev_start[t], cur), "hipEventRecord");
    hipc(hipStreamWaitEvent(c->stream, c->ev_start[t], 0), "hipStreamWaitEvent");
    return t;
}

int slab_ring_start(void *ctx, void *stream, int nmsg, const long long *count, double *const *to_left, double *const *to_right, double *const *from_right,
                    double *const *from_left) {
    try {
        tlab_comm *c = static_cast<tlab_comm *>(ctx);
        const int P = c->npro_k, me = c->pro_k, left = (me + P - 1) % P, right = (me + 1) % P;
        const int t = slab_begin(c, (hipStream_t)stream);
        ncclComm_t nc = zcomm(c);
        ncc(ncclGroupStart(), "ncclGroupStart");
        ncclResult_t r = ncclSuccess;      // a failing call must not leave the group open: it is closed first, the error raised afterwards
        for (int i = 0; i < nmsg && r == ncclSuccess; ++i) r = ncclSend(to_left[i], (size_t)count[i], ncclDouble, left, nc, c->stream);
        for (int i = 0; i < nmsg && r == ncclSuccess; ++i) r = ncclSend(to_right[i], (size_t)count[i], ncclDouble, right, nc, c->stream);
        for (int i = 0; i < nmsg && r == ncclSuccess; ++i) r = ncclRecv(from_right[i], (size_t)count[i], ncclDouble, right, nc, c->stream);
        for (int i = 0; i < nmsg && r == ncclSuccess; ++i) r = ncclRecv(from_left[i], (size_t)count[i], ncclDouble, left, nc, c->stream);
        const ncclResult_t e = ncclGroupEnd();
        ncc(r, "ncclSend / ncclRecv (ring)");
        ncc(e, "ncclGroupEnd");
        hipc(hipEventRecord(c->ev_done[t], c->stream), "hipEventRecord");
        return t;
    } catch (const Fail &f) {
        tlab_set_error(f.msg);
        return f.code;
    }
}

int slab_alltoallv_start(void *ctx, void *stream, double *const *send, const long long *scount, double *const *recv, const long long *rcount) {
    try {
        tlab_comm *c = static_cast<tlab_comm *>(ctx);
        const int P = c->npro_k, me = c->pro_k;
        const int t = slab_begin(c, (hipStream_t)stream);
        ncclComm_t nc = zcomm(c);
        long long so = 0, ro = 0, my_so = 0, my_ro = 0;
        ncc(ncclGroupStart(), "ncclGroupStart");
        ncclResult_t r = ncclSuccess;
        for (int q = 0; q < P; ++q) {
            if (q == me) { my_so = so; my_ro = ro; }
            else if (r == ncclSuccess) {
                if (scount[q] > 0) r = ncclSend(send[0] + so, (size_t)scount[q], ncclDouble, q, nc, c->stream);
                if (r == ncclSuccess && rcount[q] > 0) r = ncclRecv(recv[0] + ro, (size_t)rcount[q], ncclDouble, q, nc, c->stream);
            }
            so += scount[q];
            ro += rcount[q];
        }
        const ncclResult_t e = ncclGroupEnd();
        ncc(r, "ncclSend / ncclRecv (all-to-all)");
        ncc(e, "ncclGroupEnd");
        if (scount[me] != rcount[me]) throw Fail{TLAB_EINVAL, "slab transport: own block sizes differ"};
        if (scount[me] > 0)
            hipc(hipMemcpyAsync(recv[0] + my_ro, send[0] + my_so, (size_t)scount[me] * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync (own block)");
        hipc(hipEventRecord(c->ev_done[t], c->stream), "hipEventRecord");
        return t;
    } catch (const Fail &f) {
        tlab_set_error(f.msg);
        return f.code;
    }
}

int slab_wait(void *ctx, void *stream, int ticket) {
    try {
        tlab_comm *c = static_cast<tlab_comm *>(ctx);
        if (ticket < 0 || ticket >= tlab_comm::NEV || !c->ev_done[ticket]) throw Fail{TLAB_EINVAL, "slab transport: unknown ticket"};
        hipc(hipStreamWaitEvent((hipStream_t)stream, c->ev_done[ticket], 0), "hipStreamWaitEvent");
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.msg);
        return f.code;
    }
}

int slab_allreduce(void *ctx, double *values, int n, int op) {
    try {
        tlab_comm *c = static_cast<tlab_comm *>(ctx);
        if (n < 1 || n > 64) throw Fail{TLAB_EINVAL, "slab transport: all-reduce of 1..64 values"};
        if (!c->red) hipc(hipMalloc((void **)&c->red, 64 * sizeof(double)), "hipMalloc");
        // on the communication stream like every other operation of this communicator (one stream per communicator: no ordering left to RCCL),
        // behind the work enqueued on the caller's stream so far
        hipStream_t cur = tlab_current_stream(), cs = c->stream;
        const int t = slab_begin(c, cur);
        hipc(hipMemcpyAsync(c->red, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, cs), "hipMemcpyAsync");
        ncc(ncclAllReduce(c->red, c->red, (size_t)n, ncclDouble, op == 0 ? ncclMax : (op == 1 ? ncclMin : ncclSum), zcomm(c), cs), "ncclAllReduce");
        hipc(hipMemcpyAsync(values, c->red, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, cs), "hipMemcpyAsync");
        hipc(hipEventRecord(c->ev_done[t], cs), "hipEventRecord");
        hipc(hipStreamSynchronize(cs), "hipStreamSynchronize");
        return TLAB_OK;
    } catch (const Fail &f) {
        tlab_set_error(f.msg);
        return f.code;
    }
}

}  // namespace

extern "C" {

int tlab_comm_slab_transport(tlab_comm_t c, tlab_slab_transport *out) {
    if (!c || !out) { tlab_set_error("tlab_comm_slab_transport: null argument"); return TLAB_EINVAL; }
    if (c->npro_i != 1) { tlab_set_error("tlab_comm_slab_transport: the slab driver is the 1 x npro_k decomposition (ims_npro_i = 1)"); return TLAB_EUNSUPPORTED; }
    out->ctx = c;
    out->nranks = c->npro_k; out->nlocal = 1; out->first = c->pro_k;
    out->ring_start = slab_ring_start; out->alltoallv_start = slab_alltoallv_start; out->wait = slab_wait; out->allreduce = slab_allreduce;
    out->destroy = nullptr;       // the communicator stays the caller's (tlab_comm_destroy)
    return TLAB_OK;
}

// ---- tlab_pencil_transport over RCCL (include/tlab_amd.h: the exchanges of the native x/z pencil driver, csrc/pencil.cpp) ------------------------------
// MPI_Alltoallv inside the world, ims_comm_x or ims_comm_z communicator as one grouped ncclSend / ncclRecv on the communication stream, ordered against
// the caller's stream by events like the slab transport (same tickets, same wait).
namespace {
int pencil_alltoallv_start(void *ctx, void *stream, int which, double *const *send, const long long *scount, double *const *recv, const long long *rcount) {
    try {
        tlab_comm *c = static_cast<tlab_comm *>(ctx);
        if (which < 0 || which > 2) throw Fail{TLAB_EINVAL, "pencil transport: communicator 0 (world), 1 (x) or 2 (z)"};
        ncclComm_t nc = which == 0 ? c->world : (which == 1 ? c->cx : c->cz);
        const int S = which == 0 ? c->nranks : (which == 1 ? c->npro_i : c->npro_k);
        const int me = which == 0 ? c->rank : (which == 1 ? c->pro_i : c->pro_k);
        const int t = slab_begin(c, (hipStream_t)stream);
        long long so = 0, ro = 0, my_so = 0, my_ro = 0;
        ncclResult_t r = ncclSuccess, e = ncclSuccess;
        if (S > 1) {
            if (!nc) throw Fail{TLAB_EINVAL, "pencil transport: no communicator for this direction"};
            ncc(ncclGroupStart(), "ncclGroupStart");
        }
        for (int q = 0; q < S; ++q) {
            if (q == me) { my_so = so; my_ro = ro; }
            else if (r == ncclSuccess) {
                if (scount[q] > 0) r = ncclSend(send[0] + so, (size_t)scount[q], ncclDouble, q, nc, c->stream);
                if (r == ncclSuccess && rcount[q] > 0) r = ncclRecv(recv[0] + ro, (size_t)rcount[q], ncclDouble, q, nc, c->stream);
            }
            so += scount[q];
            ro += rcount[q];
        }
        if (S > 1) e = ncclGroupEnd();
        ncc(r, "ncclSend / ncclRecv (pencil all-to-all)");
        ncc(e, "ncclGroupEnd");
        if (scount[me] != rcount[me]) throw Fail{TLAB_EINVAL, "pencil transport: own block sizes differ"};
        if (scount[me] > 0)
            hipc(hipMemcpyAsync(recv[0] + my_ro, send[0] + my_so, (size_t)scount[me] * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync (own block)");
        hipc(hipEventRecord(c->ev_done[t], c->stream), "hipEventRecord");
        return t;
    } catch (const Fail &f) {
        tlab_set_error(f.msg);
        return f.code;
    }
}
}  // namespace

int tlab_comm_pencil_transport(tlab_comm_t c, tlab_pencil_transport *out) {
    if (!c || !out) { tlab_set_error("tlab_comm_pencil_transport: null argument"); return TLAB_EINVAL; }
    out->ctx = c;
    out->npro_i = c->npro_i; out->npro_k = c->npro_k; out->nlocal = 1; out->first = c->rank;
    out->alltoallv_start = pencil_alltoallv_start; out->wait = slab_wait; out->allreduce = slab_allreduce;
    out->destroy = nullptr;       // the communicator stays the caller's (tlab_comm_destroy)
    return TLAB_OK;
}

int tlab_comm_get_unique_id(void *id_bytes) {
    try {
        if (!id_bytes) throw Fail{TLAB_EINVAL, "tlab_comm_get_unique_id: null buffer"};
        static_assert(sizeof(ncclUniqueId) == TLAB_COMM_ID_BYTES, "ncclUniqueId size");
        ncclUniqueId id;
        ncc(ncclGetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id_bytes, &id, sizeof(id));
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_comm_init(tlab_comm_t *out, const void *id_bytes, int nranks, int rank, int npro_i, int npro_k) {
    try {
        if (!out || !id_bytes) throw Fail{TLAB_EINVAL, "tlab_comm_init: null argument"};
        if (!tlab_device_ready()) throw Fail{TLAB_EHIP, "tlab_comm_init: tlab_init has not been called"};
        if (nranks < 1 || rank < 0 || rank >= nranks || npro_i < 1 || npro_k < 1 || npro_i * npro_k != nranks)
            throw Fail{TLAB_EINVAL, "tlab_comm_init: nranks must equal npro_i * npro_k (tlab_mpi_procs.f90:58-66)"};
        auto *c = new tlab_comm();
        c->nranks = nranks; c->rank = rank; c->npro_i = npro_i; c->npro_k = npro_k;
        c->pro_i = rank % npro_i;               // tlab_mpi_procs.f90:76-86
        c->pro_k = rank / npro_i;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof(id));
        ncc(ncclCommInitRank(&c->world, nranks, id, rank), "ncclCommInitRank");
        // ims_comm_x: equal ims_pro_k, ordered by ims_pro_i ; ims_comm_z: equal ims_pro_i, ordered by ims_pro_k
        if (npro_i > 1) ncc(ncclCommSplit(c->world, c->pro_k, c->pro_i, &c->cx, nullptr), "ncclCommSplit (x)");
        if (npro_k > 1) ncc(ncclCommSplit(c->world, c->pro_i, c->pro_k, &c->cz, nullptr), "ncclCommSplit (z)");
        hipc(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
        *out = c;
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_comm_destroy(tlab_comm_t c) {
    if (!c) return TLAB_OK;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (int i = 0; i < tlab_comm::NEV; ++i) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->red) (void)hipFree(c->red);
    if (c->cx) (void)ncclCommDestroy(c->cx);
    if (c->cz) (void)ncclCommDestroy(c->cz);
    if (c->world) (void)ncclCommDestroy(c->world);
    delete c;
    return TLAB_OK;
}

int tlab_comm_info(tlab_comm_t c, int what) {
    if (!c) return TLAB_EINVAL;
    switch (what) {
        case 0: return c->rank;
        case 1: return c->nranks;
        case 2: return c->pro_i;
        case 3: return c->npro_i;
        case 4: return c->pro_k;
        case 5: return c->npro_k;
        // what RCCL itself reports for the world communicator (not what the caller passed to tlab_comm_init): ranks, this rank, its device
        case 6: { int v = -1; return (c->world && ncclCommCount(c->world, &v) == ncclSuccess) ? v : TLAB_EHIP; }
        case 7: { int v = -1; return (c->world && ncclCommUserRank(c->world, &v) == ncclSuccess) ? v : TLAB_EHIP; }
        case 8: { int v = -1; return (c->world && ncclCommCuDevice(c->world, &v) == ncclSuccess) ? v : TLAB_EHIP; }
    }
    return TLAB_EINVAL;
}

int tlab_comm_allreduce_max(tlab_comm_t c, double *v, int n) {
    try {
        if (!c || !v || n < 1) throw Fail{TLAB_EINVAL, "tlab_comm_allreduce_max: bad arguments"};
        if (c->nranks > 1) ncc(ncclAllReduce(v, v, (size_t)n, ncclDouble, ncclMax, c->world, tlab_current_stream()), "ncclAllReduce");
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_plan_create(tlab_trp_plan_t *out, tlab_comm_t comm, int dir, int nmax, int npage, int elem_doubles, int rank_dir, int npro_dir) {
    try {
        if (!out) throw Fail{TLAB_EINVAL, "tlab_trp_plan_create: null argument"};
        if (!tlab_device_ready()) throw Fail{TLAB_EHIP, "tlab_trp_plan_create: tlab_init has not been called"};
        if ((dir != 1 && dir != 3) || nmax < 1 || npage < 1 || (elem_doubles != 1 && elem_doubles != 2))
            throw Fail{TLAB_EINVAL, "tlab_trp_plan_create: dir = 1 (I) or 3 (K), elem_doubles = 1 or 2"};
        auto *p = new tlab_trp_plan();
        p->dir = dir; p->e = elem_doubles; p->comm = comm;
        if (comm) {
            p->P = dir == 1 ? comm->npro_i : comm->npro_k;
            p->rank = dir == 1 ? comm->pro_i : comm->pro_k;
            p->nc = dir == 1 ? comm->cx : comm->cz;
            p->stream = comm->stream;
        } else {
            if (npro_dir < 1 || rank_dir < 0 || rank_dir >= npro_dir) { delete p; throw Fail{TLAB_EINVAL, "tlab_trp_plan_create: rank_dir of npro_dir"}; }
            p->P = npro_dir; p->rank = rank_dir;
            hipc(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking), "hipStreamCreate");
            p->own_stream = true;
        }
        if (npage % p->P != 0) {               // tlab_mpi_transpose.f90:223, :292
            delete p;
            throw Fail{TLAB_EINVAL, "tlab_trp_plan_create: npage must be a multiple of the number of ranks of the direction"};
        }
        p->nlines = npage / p->P;
        if (dir == 1) { p->m = (long long)nmax * p->e; p->c = p->nlines; }
        else { p->m = p->nlines * p->e; p->c = nmax; }
        p->blk = p->m * p->c;
        p->local = p->blk * p->P;
        hipc(hipMalloc((void **)&p->stage, (size_t)p->local * sizeof(double)), "hipMalloc (staging buffer)");
        hipc(hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming), "hipEventCreate");
        hipc(hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming), "hipEventCreate");
        *out = p;
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_plan_destroy(tlab_trp_plan_t p) {
    if (!p) return TLAB_OK;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->stage) (void)hipFree(p->stage);
    if (p->ev_ready) (void)hipEventDestroy(p->ev_ready);
    if (p->ev_done) (void)hipEventDestroy(p->ev_done);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return TLAB_OK;
}

int tlab_trp_plan_set_wire(tlab_trp_plan_t p, int single) {
    try {
        if (!p) throw Fail{TLAB_EINVAL, "tlab_trp_plan_set_wire: null plan"};
        if (p->pending) throw Fail{TLAB_EINVAL, "tlab_trp_plan_set_wire: a transposition of this plan is in flight"};
        if (single && p->e != 1) throw Fail{TLAB_EUNSUPPORTED, "tlab_trp_plan_set_wire: complex plans always travel in double precision (tlab_mpi_transpose.f90:386-399)"};
        p->single = single != 0;
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_plan_info(tlab_trp_plan_t p, int what) {
    if (!p) return TLAB_EINVAL;
    switch (what) {
        case 0: return (int)p->nlines;
        case 1: return p->P;
        case 2: return p->rank;
        case 3: return (int)p->blk;
        case 4: return (int)p->local;
        case 5: return p->single ? 4 : 8;      // bytes per real on the wire
    }
    return TLAB_EINVAL;
}

int tlab_trp_pack(tlab_trp_plan_t p, int forward, const double *in, double *sendbuf) {
    try {
        if (!p || !in || !sendbuf || in == sendbuf) throw Fail{TLAB_EINVAL, "tlab_trp_pack: bad arguments"};
        hipStream_t st = tlab_current_stream();
        if (p->single) copy_f32(p, const_cast<double *>(in), reinterpret_cast<float *>(sendbuf), 1, strided_on_send(p, forward) ? 1 : 0, st);
        else if (strided_on_send(p, forward)) copy_strided(p, const_cast<double *>(in), sendbuf, 1, st);
        else hipc(hipMemcpyAsync(sendbuf, in, (size_t)p->local * sizeof(double), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_unpack(tlab_trp_plan_t p, int forward, const double *recvbuf, double *out) {
    try {
        if (!p || !recvbuf || !out || recvbuf == out) throw Fail{TLAB_EINVAL, "tlab_trp_unpack: bad arguments"};
        hipStream_t st = tlab_current_stream();
        if (p->single) copy_f32(p, out, reinterpret_cast<float *>(const_cast<double *>(recvbuf)), 0, strided_on_send(p, forward) ? 0 : 1, st);
        else if (!strided_on_send(p, forward)) copy_strided(p, out, const_cast<double *>(recvbuf), 0, st);
        else hipc(hipMemcpyAsync(out, recvbuf, (size_t)p->local * sizeof(double), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_start(tlab_trp_plan_t p, int forward, const double *in, double *out) {
    try {
        if (!p || !in || !out || in == out) throw Fail{TLAB_EINVAL, "tlab_trp_start: bad arguments (in and out must differ)"};
        if (p->pending) throw Fail{TLAB_EINVAL, "tlab_trp_start: a transposition of this plan is still in flight (tlab_trp_wait)"};
        if (p->P > 1 && !p->nc) throw Fail{TLAB_EINVAL, "tlab_trp_start: the plan has no communicator (use tlab_trp_pack / tlab_trp_unpack)"};
        hipStream_t cur = tlab_current_stream(), cs = p->stream;
        const bool pack = strided_on_send(p, forward);
        if (p->single) {
            // fp32 wire: BOTH sides go through the staging buffer (send floats in its first half, received floats in its second), the conversions
            // ride on the pack / unpack passes.  Every block -- the own one included -- is rounded to fp32, as in the reference, which converts the
            // whole array before the exchange (tlab_mpi_transpose.f90:362-371).
            float *sf = reinterpret_cast<float *>(p->stage), *rf = sf + p->local;
            copy_f32(p, const_cast<double *>(in), sf, 1, pack ? 1 : 0, cur);
            hipc(hipEventRecord(p->ev_ready, cur), "hipEventRecord");
            hipc(hipStreamWaitEvent(cs, p->ev_ready, 0), "hipStreamWaitEvent");
            hipc(hipMemcpyAsync(rf + (size_t)p->rank * p->blk, sf + (size_t)p->rank * p->blk, (size_t)p->blk * sizeof(float), hipMemcpyDeviceToDevice, cs),
                 "hipMemcpyAsync (own block)");
            if (p->P > 1) {
                ncc(ncclGroupStart(), "ncclGroupStart");
                ncclResult_t r = ncclSuccess;
                for (int q = 0; q < p->P && r == ncclSuccess; ++q) {
                    if (q == p->rank) continue;
                    r = ncclSend(sf + (size_t)q * p->blk, (size_t)p->blk, ncclFloat, q, p->nc, cs);
                    if (r == ncclSuccess) r = ncclRecv(rf + (size_t)q * p->blk, (size_t)p->blk, ncclFloat, q, p->nc, cs);
                }
                const ncclResult_t e = ncclGroupEnd();      // (always closed, also when a call inside the group failed)
                ncc(r, "ncclSend / ncclRecv");
                ncc(e, "ncclGroupEnd");
            }
            hipc(hipEventRecord(p->ev_done, cs), "hipEventRecord");
            p->pending = true; p->pending_unpack = !pack; p->pending_out = out;
            return TLAB_OK;
        }
        const double *src = in;       // blocked by peer
        double *dst = out;            // blocked by peer
        if (pack) { copy_strided(p, const_cast<double *>(in), p->stage, 1, cur); src = p->stage; }
        else dst = p->stage;
        hipc(hipEventRecord(p->ev_ready, cur), "hipEventRecord");          // in (and the packed copy) are ready; earlier readers of out are done
        hipc(hipStreamWaitEvent(cs, p->ev_ready, 0), "hipStreamWaitEvent");
        const size_t bytes = (size_t)p->blk * sizeof(double);
        hipc(hipMemcpyAsync(dst + (size_t)p->rank * p->blk, src + (size_t)p->rank * p->blk, bytes, hipMemcpyDeviceToDevice, cs), "hipMemcpyAsync (own block)");
        if (p->P > 1) {
            ncc(ncclGroupStart(), "ncclGroupStart");
            ncclResult_t r = ncclSuccess;
            for (int q = 0; q < p->P && r == ncclSuccess; ++q) {
                if (q == p->rank) continue;
                r = ncclSend(src + (size_t)q * p->blk, (size_t)p->blk, ncclDouble, q, p->nc, cs);
                if (r == ncclSuccess) r = ncclRecv(dst + (size_t)q * p->blk, (size_t)p->blk, ncclDouble, q, p->nc, cs);
            }
            const ncclResult_t e = ncclGroupEnd();      // (always closed, also when a call inside the group failed: no open group is left behind)
            ncc(r, "ncclSend / ncclRecv");
            ncc(e, "ncclGroupEnd");
        }
        hipc(hipEventRecord(p->ev_done, cs), "hipEventRecord");
        p->pending = true; p->pending_unpack = !pack; p->pending_out = out;
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_wait(tlab_trp_plan_t p) {
    try {
        if (!p) throw Fail{TLAB_EINVAL, "tlab_trp_wait: null plan"};
        if (!p->pending) return TLAB_OK;
        hipStream_t cur = tlab_current_stream();
        hipc(hipStreamWaitEvent(cur, p->ev_done, 0), "hipStreamWaitEvent");
        if (p->single) copy_f32(p, p->pending_out, reinterpret_cast<float *>(p->stage) + p->local, 0, p->pending_unpack ? 1 : 0, cur);
        else if (p->pending_unpack) copy_strided(p, p->pending_out, p->stage, 0, cur);
        p->pending = false;
        return TLAB_OK;
    } catch (const Fail &f) { return guard(f); }
}

int tlab_trp_exec(tlab_trp_plan_t p, int forward, const double *in, double *out) {
    const int rc = tlab_trp_start(p, forward, in, out);
    return rc != TLAB_OK ? rc : tlab_trp_wait(p);
}

}  // extern "C"
