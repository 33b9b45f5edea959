"""cc_amd/rccl.py without a GPU: the ctypes binding resolves the RCCL entry points it uses in the librccl.so torch ships, the
struct passed by value has the ABI's size, and the module is plumbing only (no collective can be issued without a device)."""
import ctypes

import pytest
import torch


def test_rccl_binding_loads_and_matches_the_header():
    from cc_amd import rccl
    lib = rccl._library()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclGetVersion", "ncclGetErrorString"):
        assert getattr(lib, name) is not None
    assert ctypes.sizeof(rccl._UniqueId) == rccl.NCCL_UNIQUE_ID_BYTES == 128           # rccl.h:40-43
    assert (rccl.ncclSum, rccl.ncclFloat32) == (0, 7)                                   # rccl.h:448,466
    v = rccl.version()
    assert v >= 21800, v                                                                # graph capture of collectives on the caller's stream
    assert "success" in lib.ncclGetErrorString(0).decode().lower() or lib.ncclGetErrorString(0)


def test_communicator_needs_a_process_group():
    from cc_amd import rccl
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        pytest.skip("a process group is up")
    with pytest.raises(AssertionError):
        rccl.Communicator("cuda:0")
