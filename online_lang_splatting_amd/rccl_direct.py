"""RCCL's C API on the CALLER'S HIP stream.

`torch.distributed` issues a collective on a stream of its own: every call is an event recorded on the caller's stream, a wait
on the process group's stream, the collective, and the way back — about 10 us per collective even in a group of one rank
(profiles/r5_exchange_overhead.json), plus a fifth stream that takes a hardware queue from four lanes.  Against a 0.23 ms frame
of a surface map that is the exchange's whole local cost.  Here the collectives of the frame-sharded step are enqueued with
`ncclAllReduce` / `ncclReduceScatter` / `ncclAllGather` directly on the lane's stream: no hop, no event pair, no extra stream.

MEASURED (profiles/r6_exchange_via.json; one rank over RCCL, every collective issued, four frames in flight): this path is
SLOWER than torch.distributed's — room map 3 340 / 3 590 fps (sparse / two-phase dense) against 4 010 / 4 170, volume 2 050 / 2 070
against 2 360 / 2 350; no exchange: 4 580 / 2 430.  On the lane's own stream a collective is a dependent RCCL kernel of 25 - 40 us
between the lane's backward and its next forward; on torch's side stream it runs BESIDE that forward's binning chain, and the
lane waits for it only before its next backward rewrites the bucket.  The hop costs less than the serialisation.  The module
stays as the measured alternative (bench.py --exchange-via rccl) and for callers whose stream has nothing else to do.

The communicator is RCCL's own (`ncclCommInitRank`); its unique id travels through the process group `torch.distributed` has
already built (any backend: one small broadcast at set-up), which is the only thing this module needs from it.  The library is
the librccl.so PyTorch itself loaded (one RCCL per process).  Frame sharding is one process per GPU: rank r of the group drives
the current device.

    comm = DirectComm.from_process_group()        # collective over the group; None when RCCL is not usable
    comm.all_reduce(t, "sum")                     # in place, on torch's current stream
    GradientBucket.direct_comm = comm             # the exchanges of frame_shard.py then go through it
"""
import ctypes as C
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128
_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
           torch.float64: 8}
_OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3}
_lib = None


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * NCCL_UNIQUE_ID_BYTES)]   # (c_ubyte: a c_char array reads back truncated at the first NUL)


def rccl():
    """librccl.so as PyTorch loaded it (the wheel's own copy), with the prototypes used here; raises when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    err = None
    for p in cands:
        try:
            L = C.CDLL(p)
            break
        except OSError as e:  # noqa: PERF203
            err = e
    else:
        raise ImportError(f"librccl.so not found ({err})")
    vp = C.c_void_p
    L.ncclGetUniqueId.argtypes, L.ncclGetUniqueId.restype = [C.POINTER(_UniqueId)], C.c_int
    L.ncclCommInitRank.argtypes, L.ncclCommInitRank.restype = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int], C.c_int
    L.ncclCommDestroy.argtypes, L.ncclCommDestroy.restype = [vp], C.c_int
    L.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclAllReduce.restype = C.c_int
    L.ncclReduceScatter.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    L.ncclReduceScatter.restype = C.c_int
    L.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.ncclAllGather.restype = C.c_int
    L.ncclGetErrorString.argtypes, L.ncclGetErrorString.restype = [C.c_int], C.c_char_p
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"RCCL {what}: {rccl().ncclGetErrorString(rc).decode()}")


class DirectComm:
    """One RCCL communicator over the ranks of a torch.distributed group; collectives on torch's current stream."""

    def __init__(self, comm, rank, world, device):
        self.comm, self.rank, self.world, self.device = comm, rank, world, torch.device(device)

    @classmethod
    def from_process_group(cls, group=None, device=None):
        """Collective over `group` (every rank calls it).  The unique id is created by the group's rank 0 and broadcast as a
        CPU byte tensor through the group (gloo) or as a device tensor (nccl)."""
        import torch.distributed as dist
        L = rccl()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        uid = _UniqueId()
        if rank == 0:
            _check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            raw = torch.frombuffer(bytearray(C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES)), dtype=torch.uint8).clone()
            on_gpu = dist.get_backend(group) == "nccl"
            t = raw.to(device) if on_gpu else raw
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if rank != 0:
                C.memmove(C.byref(uid), t.cpu().numpy().tobytes(), NCCL_UNIQUE_ID_BYTES)
        comm = C.c_void_p()
        with torch.cuda.device(device):
            _check(L.ncclCommInitRank(C.byref(comm), world, uid, rank), "ncclCommInitRank")
        return cls(comm, rank, world, device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce(self, t, op="sum"):
        """in place"""
        assert t.is_cuda and t.is_contiguous()
        _check(rccl().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], _OPS[op], self.comm,
                                    self._stream()), "ncclAllReduce")

    def reduce_scatter(self, out, inp, op="sum"):
        """out (inp.numel() / world elements; may be a slice of inp: in place when it is rank r's chunk) = this rank's chunk
        of the reduction of inp over the ranks"""
        assert out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()
        assert out.numel() * self.world == inp.numel()
        _check(rccl().ncclReduceScatter(inp.data_ptr(), out.data_ptr(), out.numel(), _DTYPES[inp.dtype], _OPS[op], self.comm,
                                        self._stream()), "ncclReduceScatter")

    def all_gather(self, out, inp):
        """out (world x inp.numel()) = the ranks' inp in rank order (inp may be rank r's chunk of out: in place)"""
        assert out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()
        assert inp.numel() * self.world == out.numel()
        _check(rccl().ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), _DTYPES[inp.dtype], self.comm,
                                    self._stream()), "ncclAllGather")

    def destroy(self):
        if self.comm:
            rccl().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
