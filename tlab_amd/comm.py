"""ctypes binding of libtlab_amd_comm.so (include/tlab_amd_comm.h): RCCL communicators and the pencil transpositions
TLabMPI_Trp_Exec{I,K}_{Forward,Backward} (base/tlab_mpi_transpose.f90:205-553) behind the C ABI.  No torch import: the library works on raw
device pointers (tlab_malloc or any other device allocation)."""
import ctypes
import os

from .lib import TlabError, load as load_core, c_int, c_vp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ID_BYTES = 128

SIGNATURES = {
    "tlab_comm_get_unique_id": (c_int, [c_vp]),
    "tlab_comm_init": (c_int, [ctypes.POINTER(c_vp), c_vp, c_int, c_int, c_int, c_int]),
    "tlab_comm_destroy": (c_int, [c_vp]),
    "tlab_comm_info": (c_int, [c_vp, c_int]),
    "tlab_comm_allreduce_max": (c_int, [c_vp, c_vp, c_int]),
    "tlab_comm_slab_transport": (c_int, [c_vp, c_vp]),
    "tlab_comm_pencil_transport": (c_int, [c_vp, c_vp]),
    "tlab_trp_plan_create": (c_int, [ctypes.POINTER(c_vp), c_vp, c_int, c_int, c_int, c_int, c_int, c_int]),
    "tlab_trp_plan_destroy": (c_int, [c_vp]),
    "tlab_trp_plan_info": (c_int, [c_vp, c_int]),
    "tlab_trp_plan_set_wire": (c_int, [c_vp, c_int]),
    "tlab_trp_exec": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_trp_start": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_trp_wait": (c_int, [c_vp]),
    "tlab_trp_pack": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tlab_trp_unpack": (c_int, [c_vp, c_int, c_vp, c_vp]),
}


def lib_path():
    return os.path.join(_HERE, "libtlab_amd_comm.so")


def load():
    """Loads libtlab_amd.so first (the comm library links it), then the comm library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        load_core()
        if not os.path.exists(lib_path()):
            raise TlabError("%s not found: make -C tlab_amd/csrc" % lib_path())
        L = ctypes.CDLL(lib_path())
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def check(code, what):
    if code != 0:
        msg = load_core().tlab_last_error()
        raise TlabError("%s failed (%d): %s" % (what, code, msg.decode() if msg else ""))


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    check(load().tlab_comm_get_unique_id(buf), "tlab_comm_get_unique_id")
    return buf.raw


class NativeComm:
    """TLabMPI_Initialize: world + ims_comm_x + ims_comm_z over RCCL."""

    def __init__(self, id_bytes, nranks, rank, npro_i, npro_k):
        self._h = c_vp(0)
        self._id = ctypes.create_string_buffer(bytes(id_bytes), ID_BYTES)
        check(load().tlab_comm_init(ctypes.byref(self._h), self._id, nranks, rank, npro_i, npro_k), "tlab_comm_init")

    def info(self, what):
        return load().tlab_comm_info(self._h, what)

    def allreduce_max(self, dev_ptr, n):
        check(load().tlab_comm_allreduce_max(self._h, c_vp(dev_ptr), n), "tlab_comm_allreduce_max")

    def close(self):
        if self._h:
            load().tlab_comm_destroy(self._h)
            self._h = c_vp(0)


class TrpPlan:
    """TLabMPI_Trp_PlanI (dir = 1) / PlanK (dir = 3)."""

    def __init__(self, comm, dir, nmax, npage, elem_doubles=1, rank_dir=0, npro_dir=1):
        self._h = c_vp(0)
        check(load().tlab_trp_plan_create(ctypes.byref(self._h), comm._h if comm is not None else None, dir, nmax, npage, elem_doubles, rank_dir,
                                          npro_dir), "tlab_trp_plan_create")

    def info(self, what):
        return load().tlab_trp_plan_info(self._h, what)

    def set_wire(self, single):
        """[Parallel] TransposeTypeI / TransposeTypeK = single: fp32 on the wire (real plans only)."""
        check(load().tlab_trp_plan_set_wire(self._h, int(bool(single))), "tlab_trp_plan_set_wire")

    def exec(self, forward, src, dst):
        check(load().tlab_trp_exec(self._h, int(forward), c_vp(src), c_vp(dst)), "tlab_trp_exec")

    def start(self, forward, src, dst):
        check(load().tlab_trp_start(self._h, int(forward), c_vp(src), c_vp(dst)), "tlab_trp_start")

    def wait(self):
        check(load().tlab_trp_wait(self._h), "tlab_trp_wait")

    def pack(self, forward, src, sendbuf):
        check(load().tlab_trp_pack(self._h, int(forward), c_vp(src), c_vp(sendbuf)), "tlab_trp_pack")

    def unpack(self, forward, recvbuf, dst):
        check(load().tlab_trp_unpack(self._h, int(forward), c_vp(recvbuf), c_vp(dst)), "tlab_trp_unpack")

    def close(self):
        if self._h:
            load().tlab_trp_plan_destroy(self._h)
            self._h = c_vp(0)
