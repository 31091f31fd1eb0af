"""RCCL called directly on the step's own HIP stream (ctypes onto the librccl.so that PyTorch already loaded).

Why not torch.distributed for the exchanges of the step: ProcessGroupNCCL runs every collective on its private stream and
joins it to the caller's stream with two events.  On MI355X each such cross-queue join costs 10-20 us of dead time, and a
0.29 ms training step has five exchanges (measured with one rank: 0.30 ms -> 0.56 ms per step from the joins alone).
A collective enqueued on the launch stream itself is just one more kernel in the stream -- and is captured into the
step's hipGraph like any other launch, so the whole multi-GPU step replays as ONE graph.

The communicator is bootstrapped over an existing torch.distributed group (any backend): rank 0's ncclUniqueId is
broadcast as 128 bytes, then every rank calls ncclCommInitRank.  One process per GPU, as everywhere in this package.

Reference: the reference's only multi-GPU mechanism is nn.DataParallel (main_darai.py:129-133) -- a gather/scatter
through device 0 per step.  This replaces that exchange; the mathematics (sum of per-replica gradients) is the same.
"""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_FLOAT32, _NCCL_FLOAT64, _NCCL_INT64, _NCCL_UINT8 = 7, 8, 4, 1
_NCCL_SUM = 0
_DTYPES = {torch.float32: _NCCL_FLOAT32, torch.float64: _NCCL_FLOAT64, torch.int64: _NCCL_INT64,
           torch.uint8: _NCCL_UINT8}


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        raise RuntimeError(f"librccl.so not found beside torch ({path})")
    L = ctypes.CDLL(path)
    vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ncclGetErrorString.restype = ctypes.c_char_p
    L.ncclGetErrorString.argtypes = [i]
    L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i, _UniqueId, i]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclCommCount.argtypes = [vp, ctypes.POINTER(i)]
    L.ncclCommCount.restype = i
    L.ncclAllReduce.argtypes = [vp, vp, sz, i, i, vp, vp]
    L.ncclReduceScatter.argtypes = [vp, vp, sz, i, i, vp, vp]
    L.ncclAllGather.argtypes = [vp, vp, sz, i, vp, vp]
    L.ncclAllToAll.argtypes = [vp, vp, sz, i, vp, vp]
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclReduceScatter",
              "ncclAllGather", "ncclAllToAll", "ncclGroupStart", "ncclGroupEnd"):
        getattr(L, f).restype = i
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"RCCL {what} failed: {_load().ncclGetErrorString(rc).decode()} ({rc})")


def _stream(stream):
    return ctypes.c_void_p((stream if stream is not None else torch.cuda.current_stream()).cuda_stream)


class _Group:
    def __enter__(self):
        _check(_load().ncclGroupStart(), "ncclGroupStart")
        return self

    def __exit__(self, *exc):
        _check(_load().ncclGroupEnd(), "ncclGroupEnd")
        return False


class RcclComm:
    """One RCCL communicator over the ranks of a torch.distributed group.  Every method enqueues on the current (or
    given) torch stream and returns at once; nothing here synchronises the device or allocates."""

    def __init__(self, process_group=None, device=None):
        if not dist.is_initialized():
            raise RuntimeError("RcclComm needs an initialised torch.distributed group to exchange the unique id")
        L = _load()
        self.pg = process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        on_gpu = dist.get_backend(process_group) == "nccl"
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=self.device if on_gpu else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                       group=process_group)
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().tolist()), 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def nranks(self):
        """Ranks RCCL itself reports for this communicator (ncclCommCount) -- what bench.py's N > 1 line records."""
        n = ctypes.c_int(0)
        _check(_load().ncclCommCount(self._comm, ctypes.byref(n)), "ncclCommCount")
        return int(n.value)

    def close(self):
        if getattr(self, "_comm", None):
            _load().ncclCommDestroy(self._comm)
            self._comm = None

    def group(self):
        """with comm.group(): ...  -- the collectives enqueued inside are submitted together (ncclGroupStart/End), so
        RCCL can launch them as one kernel instead of one per call."""
        return _Group()

    @staticmethod
    def _ok(*tensors):
        for t in tensors:
            if not (t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES):
                raise ValueError("RCCL operands must be contiguous device tensors of a supported dtype")

    def all_reduce(self, t, stream=None):
        """In-place sum over the ranks."""
        self._ok(t)
        if t.dtype == torch.float32:          # the gradient buckets: through the C ABI (r3d_allreduce_flat, include/r3d_hip.h)
            from . import _lib as _abi
            _abi.check(_abi.load().r3d_allreduce_flat(t.data_ptr(), t.numel(), self._comm, _stream(stream)),
                       "r3d_allreduce_flat")
            return
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], _NCCL_SUM, self._comm,
                                     _stream(stream)), "ncclAllReduce")

    def reduce_scatter_inplace(self, full, stream=None):
        """full = world equal blocks; afterwards block `rank` of it holds the sum over the ranks of that block (the other
        blocks are left as they were).  Returns that block."""
        self._ok(full)
        if full.numel() % self.world:
            raise ValueError("reduce_scatter: the buffer does not split into equal blocks")
        n = full.numel() // self.world
        mine = full.view(-1)[self.rank * n:(self.rank + 1) * n]
        _check(_load().ncclReduceScatter(full.data_ptr(), mine.data_ptr(), n, _DTYPES[full.dtype], _NCCL_SUM, self._comm,
                                         _stream(stream)), "ncclReduceScatter")
        return mine

    def all_gather(self, out, inp, stream=None):
        """out (world equal blocks, ordered by rank) <- every rank's inp."""
        self._ok(out, inp)
        if out.numel() != inp.numel() * self.world or out.dtype != inp.dtype:
            raise ValueError("all_gather: out must hold world blocks of inp's size and dtype")
        _check(_load().ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), _DTYPES[inp.dtype], self._comm,
                                     _stream(stream)), "ncclAllGather")

    def all_to_all(self, recv, send, stream=None):
        """Block j of send goes to rank j; block i of recv comes from rank i (equal blocks, out of place)."""
        self._ok(recv, send)
        if recv.numel() != send.numel() or send.numel() % self.world or recv.data_ptr() == send.data_ptr():
            raise ValueError("all_to_all: distinct buffers of equal size, divisible by the world size")
        _check(_load().ncclAllToAll(send.data_ptr(), recv.data_ptr(), send.numel() // self.world, _DTYPES[send.dtype],
                                    self._comm, _stream(stream)), "ncclAllToAll")
