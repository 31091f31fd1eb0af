"""The ctypes RCCL binding (r3d_amd/rccl.py) resolves, without a GPU: the librccl.so beside torch loads and exports every
entry point the binding declares, with the argument layouts the header (rccl.h, RCCL 2.26) gives them."""
import ctypes

import pytest


def test_librccl_loads_and_exports_the_bound_symbols():
    rccl = pytest.importorskip("r3d_amd.rccl")
    lib = rccl._load()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclReduceScatter",
                 "ncclAllGather", "ncclAllToAll", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    assert ctypes.sizeof(rccl._UniqueId) == 128                  # NCCL_UNIQUE_ID_BYTES
    assert lib.ncclAllReduce.argtypes[2] is ctypes.c_size_t and len(lib.ncclAllReduce.argtypes) == 7
    assert len(lib.ncclAllToAll.argtypes) == 6 and len(lib.ncclAllGather.argtypes) == 6
    assert b"success" in lib.ncclGetErrorString(0).lower() or lib.ncclGetErrorString(0)


def test_comm_needs_an_initialised_group():
    rccl = pytest.importorskip("r3d_amd.rccl")
    with pytest.raises(RuntimeError):
        rccl.RcclComm()
