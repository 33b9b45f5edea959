"""RCCL (the collective library torch.distributed's "nccl" backend wraps on ROCm) bound directly with ctypes, for ONE purpose:
the gradient all-reduce of a bucket segment enqueued ON THE CALLER'S STREAM -- so that it can be a node of the step's hipGraph, on
the branch of the network whose gradients it exchanges (cc_amd/trainer.py, pipeline "per_network").

Why not torch.distributed.all_reduce there: ProcessGroupNCCL runs every collective on a stream of its own and hands completion
to a watchdog thread through HIP events.  Issued inside a capture from the autograd engine's worker thread (where a network's
backward node runs) the work is still queued for the watchdog, whose event query then fails with "operation not permitted on
an event last recorded in a capturing stream" and takes the process down (tools/rccl_capture_probe.py, profiles/r06_rccl_capture_probe.txt).
A collective that is just a kernel on the caller's stream needs none of that machinery: no side stream, no events, no watchdog,
no host-side wait (the `work.wait()` that blocks the host under AMD_DIRECT_DISPATCH=1).

torch.distributed stays the control plane: the communicator's unique id travels through the default process group
(any backend), start-up broadcasts and the bench's timing reductions use it as before.  The library handle is the librccl.so
torch itself has loaded, so there is one RCCL in the process.
"""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128      # rccl.h:40
ncclSum, ncclFloat32 = 0, 7     # rccl.h:448,466


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _library():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path if os.path.isfile(path) else "librccl.so")
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
        for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy, lib.ncclGetVersion):
            f.restype = ctypes.c_int
        _lib = lib
    return _lib


def _check(code, what):
    if code != 0:
        raise RuntimeError("RCCL %s failed: %s (%d)" % (what, _library().ncclGetErrorString(code).decode(), code))


def version():
    v = ctypes.c_int(0)
    _check(_library().ncclGetVersion(ctypes.byref(v)), "ncclGetVersion")
    return v.value


class Communicator:
    """One RCCL communicator over the ranks of the default torch.distributed group, one rank per GPU (the current HIP device)."""

    def __init__(self, device):
        assert dist.is_available() and dist.is_initialized(), "rccl.Communicator needs the default process group (unique id exchange)"
        lib = _library()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        box = [bytes(uid.internal)] if self.rank == 0 else [None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        ctypes.memmove(ctypes.addressof(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_sum_(self, t):
        """t (contiguous fp32 on this communicator's device) <- sum over the ranks, enqueued on the CURRENT torch stream and ordered
        like any kernel of that stream; inside a stream capture it becomes a node of the graph.  Every rank must issue the
        collectives of one communicator in the same order."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(_library().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), ncclFloat32, ncclSum, self.comm, stream), "ncclAllReduce")

    def destroy(self):
        if self.comm:
            _library().ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()
