/* tlab_amd_comm.h -- C ABI of the pencil transpositions of the MI355X-native Tlab hot path (libtlab_amd_comm.so).
 *
 * Replaces the MPI layer under the x- and z-directional operators: TLabMPI_Initialize's cartesian communicators
 * (base/tlab_mpi_procs.f90:17-116), the transposition plans TLabMPI_Trp_PlanI / PlanK (base/tlab_mpi_transpose.f90:205-339) and
 * TLabMPI_Trp_Exec{I,K}_{Forward,Backward} (:343-553), real and complex (the MPI_DOUBLE_COMPLEX plans of operators/opr_fourier.f90:86-88,
 * 132-134).  One process per GPU; the exchange is an RCCL grouped ncclSend / ncclRecv among the ranks of the direction's communicator
 * (RCCL has no alltoallw: the derived vector datatypes of the reference become a HIP pack or unpack kernel on the strided side), on a
 * stream the library owns, so that a caller can run other operators while a transposition is in flight (tlab_trp_start / tlab_trp_wait).
 *
 * The library is separate from libtlab_amd.so so that the operator library carries no RCCL dependency; it links libtlab_amd.so
 * (stream, error text) and librccl.  Conventions as in tlab_amd.h: int return codes (0 = ok), device pointers, fp64.
 *
 * Rank layout of the reference (tlab_mpi_procs.f90:76-86): ims_pro_i = mod(ims_pro, ims_npro_i), ims_pro_k = ims_pro / ims_npro_i;
 * ims_comm_x = the ranks of equal ims_pro_k, ims_comm_z = the ranks of equal ims_pro_i.
 */
#ifndef TLAB_AMD_COMM_H
#define TLAB_AMD_COMM_H

#include <stddef.h>

#include "tlab_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TLAB_COMM_ID_BYTES 128          /* sizeof(ncclUniqueId) */

typedef struct tlab_comm *tlab_comm_t;
typedef struct tlab_trp_plan *tlab_trp_plan_t;

/* Rank 0 makes the identifier (ncclGetUniqueId) and the host distributes its TLAB_COMM_ID_BYTES bytes to every rank by whatever it has
 * (MPI_Bcast in a Fortran host, a file, torch.distributed's store). */
int tlab_comm_get_unique_id(void *id_bytes);
/* TLabMPI_Initialize (tlab_mpi_procs.f90:17-116): world communicator of nranks = npro_i * npro_k ranks + the two direction communicators
 * (ncclCommSplit).  The calling process must already have selected its GPU (tlab_init). */
int tlab_comm_init(tlab_comm_t *out, const void *id_bytes, int nranks, int rank, int npro_i, int npro_k);
int tlab_comm_destroy(tlab_comm_t comm);
int tlab_comm_info(tlab_comm_t comm, int what);   /* 0 ims_pro, 1 ims_npro, 2 ims_pro_i, 3 ims_npro_i, 4 ims_pro_k, 5 ims_npro_k;
                                                    * from RCCL itself: 6 ncclCommCount, 7 ncclCommUserRank, 8 ncclCommCuDevice of the world communicator */
/* MPI_ALLREDUCE(.., MPI_MAX, ..) of TIME_COURANT (tools/dns/time.f90:522) on n device doubles, in place, on the current stream */
int tlab_comm_allreduce_max(tlab_comm_t comm, double *dev_values, int n);

/* The exchanges of the native z-slab driver (tlab_slab_dns_create, include/tlab_amd.h) over this communicator's z direction: ring neighbours and
 * all-to-all-v as grouped ncclSend / ncclRecv on the library's communication stream, MPI_MAX / MPI_MIN as ncclAllReduce.  Needs npro_i = 1.
 * The struct refers to comm, which must outlive the driver made from it. */
int tlab_comm_slab_transport(tlab_comm_t comm, tlab_slab_transport *out);
/* ... and of the native x/z pencil driver (tlab_pencil_dns_create): MPI_Alltoallv inside the world / ims_comm_x / ims_comm_z communicators. */
int tlab_comm_pencil_transport(tlab_comm_t comm, tlab_pencil_transport *out);

/* TLabMPI_Trp_PlanI (dir = 1) / TLabMPI_Trp_PlanK (dir = 3)   base/tlab_mpi_transpose.f90:205-286, 290-339
 *   dir = 1: nmax = imax, npage = jmax*kmax:  local a(imax, npage)  <->  b(imax*npro_i, nlines), nlines = npage / npro_i
 *   dir = 3: nmax = kmax, npage = imax*jmax:  local a(npage, kmax)  <->  b(nlines, kmax*npro_k), nlines = npage / npro_k
 * elem_doubles = 1 (real fields) or 2 (complex: the nx/2+1 layout of the Poisson solver).  npage must be a multiple of the number of ranks
 * of the direction (:223, :292).  comm may be NULL for a plan that is only packed / unpacked by the caller (rank_dir of npro_dir given
 * explicitly: a GPU-aware MPI_Alltoall between tlab_trp_pack and tlab_trp_unpack, or the single-process loopback of the tests);
 * with a communicator rank_dir / npro_dir are taken from it and the arguments are ignored. */
int tlab_trp_plan_create(tlab_trp_plan_t *out, tlab_comm_t comm, int dir, int nmax, int npage, int elem_doubles, int rank_dir, int npro_dir);
int tlab_trp_plan_destroy(tlab_trp_plan_t plan);
int tlab_trp_plan_info(tlab_trp_plan_t plan, int what);   /* 0 nlines, 1 npro, 2 rank, 3 reals per peer block, 4 local reals, 5 bytes per real on the wire */
/* [Parallel] TransposeTypeI / TransposeTypeK = single (tlab_mpi_transpose.f90:106-122; applied in the real Exec routines :362-371, :473-482): the
 * data of a REAL plan travel as fp32 -- the pack pass converts, the unpack pass converts back, the result is the transposition of the field rounded
 * to single precision (every block, the own one included, exactly as the reference, which converts the whole array first): half the bytes over
 * xGMI.  Complex plans always travel in double precision (:386-399): TLAB_EUNSUPPORTED.  With single the wire format of tlab_trp_pack /
 * tlab_trp_unpack is npro blocks of floats (the buffers are still passed as double pointers; half of their bytes are used). */
int tlab_trp_plan_set_wire(tlab_trp_plan_t plan, int single);

/* TLabMPI_Trp_Exec{I,K}_Forward (forward != 0) / _Backward (forward == 0): a -> b (forward) resp. b -> a, bit-exact index work.
 * exec = start + wait.  start: pack (if this direction's send side is strided) on the current stream, then the grouped exchange on the
 * library's communication stream; wait: the current stream waits for the exchange and runs the unpack (if the receive side is strided).
 * Between the two the caller may enqueue independent work on its own stream.  in and out must not alias; one transposition in flight per plan. */
int tlab_trp_exec(tlab_trp_plan_t plan, int forward, const double *in, double *out);
int tlab_trp_start(tlab_trp_plan_t plan, int forward, const double *in, double *out);
int tlab_trp_wait(tlab_trp_plan_t plan);

/* The two halves on their own.  Wire format: npro blocks of (local doubles / npro) doubles, block p = what goes to / comes from rank p of the
 * direction, in the element order of the contiguous side (I: (imax, nlines) of the lines of p; K: (nlines, kmax) of the planes of the sender).
 *   tlab_trp_pack  : local array -> send buffer (a plain copy where the send side is already blocked by peer)
 *   tlab_trp_unpack: receive buffer -> local array */
int tlab_trp_pack(tlab_trp_plan_t plan, int forward, const double *in, double *sendbuf);
int tlab_trp_unpack(tlab_trp_plan_t plan, int forward, const double *recvbuf, double *out);

#ifdef __cplusplus
}
#endif
#endif /* TLAB_AMD_COMM_H */
