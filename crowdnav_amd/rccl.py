"""Minimal ctypes view of RCCL (librccl.so.1) for the C-ABI seam: create / destroy the communicator that
cn_gather_records (include/crowdnav_amd.h) all-gathers the episode records over.  A host that already runs
torch.distributed does not need this (crowdnav_amd.distributed uses its 'nccl' backend = the same RCCL); a C or ctypes
consumer of libcrowdnav_amd.so (seam 2) owns its communicator exactly like this:

    uid = rccl.get_unique_id() on rank 0, shipped to the other ranks by any side channel (file, socket, MPI, a store)
    comm = rccl.comm_init_rank(world, uid, rank)          # one rank per GPU, after hipSetDevice(local_rank)
    blocks_all = engine.gather_records_rccl(comm, world, engine.rollout_records())
    rccl.comm_destroy(comm)
"""
import ctypes as C

UNIQUE_ID_BYTES = 128  # NCCL_UNIQUE_ID_BYTES (rccl.h)


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * UNIQUE_ID_BYTES)]


_rccl = None


def lib():
    """librccl.so.1 — in a PyTorch-ROCm process the copy torch has already mapped (same SONAME), else ROCm's."""
    global _rccl
    if _rccl is None:
        last = None
        for name in ('librccl.so.1', '/opt/rocm/lib/librccl.so.1'):
            try:
                _rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError as err:
                last = err
        if _rccl is None:
            raise ImportError('librccl.so.1 not found: %s' % last)
        _rccl.ncclGetErrorString.restype = C.c_char_p
        _rccl.ncclGetErrorString.argtypes = [C.c_int]
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return _rccl


def _check(status, what):
    if status != 0:
        raise RuntimeError('%s failed: %s' % (what, lib().ncclGetErrorString(status).decode()))


def get_unique_id():
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
    return C.string_at(C.addressof(uid), UNIQUE_ID_BYTES)


def comm_init_rank(world, unique_id, rank):
    """ncclCommInitRank on the CURRENT HIP device; returns the ncclComm_t as an int."""
    if len(unique_id) != UNIQUE_ID_BYTES:
        raise ValueError('unique id must be %d bytes' % UNIQUE_ID_BYTES)
    uid = _UniqueId()
    C.memmove(C.addressof(uid), unique_id, UNIQUE_ID_BYTES)
    comm = C.c_void_p()
    _check(lib().ncclCommInitRank(C.byref(comm), int(world), uid, int(rank)), 'ncclCommInitRank')
    return comm.value


def comm_destroy(comm):
    _check(lib().ncclCommDestroy(C.c_void_p(int(comm))), 'ncclCommDestroy')
