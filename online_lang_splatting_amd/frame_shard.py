"""Pre-allocated rasterizer workspace and the frame-sharded multi-GPU step.

Two things the reference does not have:

* `RasterWorkspace` — one forward+backward of the rasterizer with every buffer allocated once
  (geometry / image / binning state sized for a capacity of instances, outputs, gradients) and
  NO host synchronisation: it drives `olsr_forward_async` / `olsr_backward` of include/olsr.h.
  The reference re-allocates and zero-fills ~20 tensors per call and blocks on a D2H copy of
  the instance count (DGR/rasterize_points.cu:170-184,386-398; CR/rasterizer_impl.cu:454-455).

* `FrameShardedStep` — the mapping loop of utils/slam_backend.py:499-670 renders up to 12
  viewpoints of the SAME Gaussians and sums their losses before one backward.  Views are
  independent, so view v goes to rank v mod world; each rank accumulates its views' gradients
  into one flat fp32 buffer [P x (3 xyz + 3M sh + 1 opacity + 3 scale + 4 rot + F lang)] and ONE
  sum all-reduce (RCCL over xGMI) per optimisation step makes every rank hold the total, after
  which all ranks apply the identical update.  Side reductions needed by densification are
  computed per view BEFORE reducing where they are not linear (||means2D.grad||,
  gaussian_model.py:965-969).  Pose gradients stay on the owning rank.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Sequence

import torch

from . import _abi
from ._lib import check, lib


def views_of_rank(num_views: int, rank: int, world: int) -> List[int]:
    """View v is rendered by rank v mod world."""
    return [v for v in range(num_views) if v % world == rank]


@dataclass
class GradLayout:
    """Column layout of the flat per-Gaussian gradient buffer."""
    M: int
    F: int

    @property
    def fields(self):
        return [("means3D", 3), ("sh", 3 * self.M), ("opacity", 1), ("scales", 3), ("rotations", 4),
                ("language", self.F)]

    @property
    def width(self):
        return sum(w for _, w in self.fields)

    def slices(self):
        out, c = {}, 0
        for name, w in self.fields:
            out[name] = slice(c, c + w)
            c += w
        return out


class GradientBucket:
    """Flat [P, width] fp32 gradient buffer + the side buffers of the densification bookkeeping."""

    # True: issue the collectives also in a group of ONE rank (they are identities there).  Only tests set it: it lets a
    # one-GPU box execute every RCCL call of the exchanges (dtype / op / shape support of the "nccl" backend).
    exchange_single_rank = False
    # True: sparse_all_reduce_capped runs its torch formulation (the CPU specification) on GPU buckets too - the A/B leg of
    # the exchange-overhead measurement and of tests/test_gpu_sparse_exchange.py, never the default
    capped_torch_formulation = False

    # rccl_direct.DirectComm, or None.  Set (by the caller that built it, on every rank alike), the exchanges below that a
    # frame-sharded step issues per frame — all_reduce(), reduce_scatter_all_gather(), sparse_all_reduce_capped() — enqueue
    # their collectives through RCCL's C API on the CURRENT stream instead of through torch.distributed, which hops to a stream
    # of its own and back (an event pair and ~10 us per collective even in a group of one rank, and a fifth stream beside four
    # lanes).  Same buffers, same operations, same results.
    direct_comm = None

    def _direct(self, group=None):
        return self.direct_comm if (self.direct_comm is not None and group is None and self.flat.is_cuda) else None

    @classmethod
    def _multi(cls, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(group) > 1 or cls.exchange_single_rank

    def __init__(self, P, layout: GradLayout, device, track_rows=False):
        """track_rows: keep a bit per row "may be non-zero" (olsr_grad_bucket.row_mask, include/olsr.h) so that the backward's
        overwrite of a step's first view stores only the rows that carry a gradient or carried one before, instead of all P
        (98 % of them zeros at config 3).  The mask follows every write this class makes to `flat`; code that writes `flat`
        or `sum_storage` DIRECTLY must call rows_unknown() afterwards (or leave track_rows off).  GPU buckets only."""
        self.layout = layout
        # one storage for everything that is SUM-reduced, so the step's exchange is a single large all-reduce
        # (+ one small MAX all-reduce): [P x width gradients | P x 2 densification statistics]
        # (row P is a spare row that stays zero: the fill target of the capacity-bound sparse exchange)
        off = ((P + 1) * layout.width + 3) // 4 * 4  # keep the statistics 16-byte aligned
        # (padded to a multiple of 64 floats: the two-phase dense exchange cuts the storage into `world` equal chunks in place)
        self.sum_storage = torch.zeros((off + 2 * P + 63) // 64 * 64, dtype=torch.float32, device=device)
        self.flat = self.sum_storage[: P * layout.width].view(P, layout.width)
        self.flat_ext = self.sum_storage[: (P + 1) * layout.width].view(P + 1, layout.width)
        # xyz_gradient_accum, denom (gaussian_model.py:965-969): sum-reducible once the norm is taken per view
        self.densify = self.sum_storage[off:off + 2 * P].view(P, 2)
        self.max_radii = torch.zeros(P, dtype=torch.int32, device=device)  # max-reducible
        self._sl = layout.slices()
        # all ones = unknown: the first overwrite is dense (the storage is zero now, but "unknown" is the safe start)
        self.row_mask = (torch.full(((P + 63) // 64,), -1, dtype=torch.int64, device=device)
                         if (track_rows and torch.device(device).type == "cuda") else None)

    def rows_unknown(self):
        """`flat` was written by something other than the backward's fused accumulation: every row may be non-zero."""
        if self.row_mask is not None:
            self.row_mask.fill_(-1)

    def rows_merge(self, other):
        """`other`'s rows were added into this bucket's `flat`."""
        if self.row_mask is not None:
            if other.row_mask is not None:
                self.row_mask.bitwise_or_(other.row_mask)
            else:
                self.row_mask.fill_(-1)

    def add_bucket(self, other):
        """self += other (gradient rows and densification statistics SUM, max_radii MAX): the sum of the lane buckets of a
        step.  On the GPU one launch of olsr_bucket_add that reads and writes only the rows `other`'s row mask flags
        (FrameLanes' buckets keep one); on CPU tensors the torch formulation — the kernel's specification."""
        if self.flat.is_cuda:
            P, width = self.flat.shape
            check(lib().olsr_bucket_add(
                P, width, self.flat.data_ptr(), self.densify.data_ptr(), self.max_radii.data_ptr(),
                self.row_mask.data_ptr() if self.row_mask is not None else None, other.flat.data_ptr(),
                other.densify.data_ptr(), other.max_radii.data_ptr(),
                other.row_mask.data_ptr() if other.row_mask is not None else None,
                C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)))
            if self.row_mask is not None and other.row_mask is None:
                self.rows_unknown()
            return
        self.sum_storage.add_(other.sum_storage)
        self.rows_merge(other)
        torch.maximum(self.max_radii, other.max_radii, out=self.max_radii)

    def zero_(self):
        self.flat.zero_()
        self.densify.zero_()
        self.max_radii.zero_()
        if self.row_mask is not None:
            self.row_mask.zero_()

    def view(self, name):
        return self.flat[:, self._sl[name]]

    def accumulate(self, grads: Dict[str, torch.Tensor], radii: torch.Tensor, first: bool = False):
        """Add one view's gradients (first=True: overwrite instead, so zero_() can be skipped) (names of the C-ABI / reference backward outputs).  On the GPU this
        is one fused kernel (olsr_accumulate_gradients); the torch formulation below is the CPU path of
        the gloo tests and the specification the kernel is tested against."""
        if self.flat.is_cuda:
            P = self.flat.shape[0]

            def p(t):
                return t.data_ptr() if t is not None and t.numel() > 0 else None
            check(lib().olsr_accumulate_gradients(
                P, self.layout.M, self.layout.F, 1 if first else 0, p(grads["dL_dmeans3D"]), p(grads.get("dL_dsh")),
                p(grads["dL_dopacity"]), p(grads["dL_dscales"]), p(grads["dL_drotations"]),
                p(grads.get("dL_dlanguage")), p(grads["dL_dmeans2D"]), radii.data_ptr(), self.flat.data_ptr(),
                self.densify.data_ptr(), self.max_radii.data_ptr(),
                C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)))
            self.rows_unknown()
            return
        if first:
            self.zero_()
        self.view("means3D").add_(grads["dL_dmeans3D"])
        if self.layout.M > 0:
            self.view("sh").add_(grads["dL_dsh"].reshape(grads["dL_dsh"].shape[0], -1))
        self.view("opacity").add_(grads["dL_dopacity"].reshape(-1, 1))
        self.view("scales").add_(grads["dL_dscales"])
        self.view("rotations").add_(grads["dL_drotations"])
        if self.layout.F > 0:
            self.view("language").add_(grads["dL_dlanguage"])
        vis = radii > 0
        self.densify[:, 0].add_(torch.norm(grads["dL_dmeans2D"][:, :2], dim=-1) * vis)
        self.densify[:, 1].add_(vis.to(torch.float32))
        torch.maximum(self.max_radii, radii.to(torch.int32), out=self.max_radii)

    def all_reduce(self, group=None, async_op=False):
        """The one exchange step.  backend 'nccl' is RCCL on ROCm; 'gloo' in the CPU tests.
        async_op=True returns the pending work handles (call .wait() on each before the buffers are read or
        written again): the collective then overlaps whatever the issuing stream does next — e.g. the forward
        of this lane's next frame, which does not touch the bucket."""
        import torch.distributed as dist
        if not self._multi(group):
            return []
        dc = self._direct(group)
        if dc is not None and not async_op:
            dc.all_reduce(self.sum_storage, "sum")
            dc.all_reduce(self.max_radii, "max")
            self.rows_unknown()
            return []
        works = [dist.all_reduce(self.sum_storage, op=dist.ReduceOp.SUM, group=group, async_op=async_op),
                 dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=group, async_op=async_op)]
        self.rows_unknown()  # (other ranks' rows arrive)
        return works if async_op else []

    # ---- owner-applies exchange: rank r owns the Gaussians [r0, r1) ---------------------
    @staticmethod
    def owned_rows(P, rank, world):
        """Contiguous, equal-sized (up to padding) row ranges: rank r owns [r * ceil(P / world), ...) clipped to P."""
        per = (P + world - 1) // world
        return min(rank * per, P), min((rank + 1) * per, P)

    def reduce_scatter(self, rank, world, group=None):
        """Sum over ranks of the gradient rows this rank owns, left in self.flat[r0:r1] (other rows keep this rank's
        own partial sums).  On xGMI every GPU has a direct link to each of its 7 peers: a reduce-scatter moves
        (G-1)/G of the buffer once over all links concurrently, where a ring all-reduce moves 2 (G-1)/G of it around
        one ring.  RCCL: reduce_scatter_tensor; gloo (CPU tests) has no reduce-scatter: all-reduce + slice."""
        import torch.distributed as dist
        P, width = self.flat.shape
        r0, r1 = self.owned_rows(P, rank, world)
        if not self._multi(group):
            return r0, r1
        self.rows_unknown()
        if dist.get_backend(group) == "nccl":
            per = (P + world - 1) // world
            pad = per * world - P
            src = self.flat if pad == 0 else torch.cat([self.flat, self.flat.new_zeros(pad, width)])
            out = torch.empty(per, width, dtype=self.flat.dtype, device=self.flat.device)
            dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=group)
            self.flat[r0:r1].copy_(out[: r1 - r0])
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        # the densification statistics and radii are small: plain all-reduces
        dist.all_reduce(self.densify, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=group)
        return r0, r1

    def reduce_scatter_all_gather(self, rank, world, group=None):
        """The dense all-reduce in two direct phases: reduce-scatter of the gradient rows (rank r receives the total of
        the rows it owns), then all-gather of those totals — every rank ends with the full sum, like all_reduce(), but each
        phase sends 1/G of the buffer to every peer concurrently over the G-1 direct xGMI links instead of around rings.
        (With an optimiser the second phase carries the updated PARAMETER rows instead: FrameShardedStep.optimizer_step.)"""
        import torch.distributed as dist
        P, width = self.flat.shape
        n = self.sum_storage.numel()
        if self._multi(group) and dist.get_backend(group) == "nccl" and n % world == 0:
            # Both phases IN PLACE on the one SUM storage [gradient rows | spare row | statistics | padding]: rank r reduces
            # chunk r (ncclReduceScatter with recvbuff = sendbuff + r x chunk), then the chunks are gathered where they lie —
            # three collectives (with the MAX of the radii) and no staging copy.  Round 4 went through row-aligned staging
            # buffers (a 58 MB out-of-place receive, its copy back, a clone for the gather) and four collectives; every rank
            # wants the whole sum here, so the chunks need not respect row ownership.
            chunk = n // world
            mine = self.sum_storage[rank * chunk:(rank + 1) * chunk]
            dc = self._direct(group)
            if dc is not None:   # (RCCL's C API on this stream: no hop to the process group's stream)
                dc.reduce_scatter(mine, self.sum_storage, "sum")
                dc.all_gather(self.sum_storage, mine)
                dc.all_reduce(self.max_radii, "max")
            else:
                dist.reduce_scatter_tensor(mine, self.sum_storage, op=dist.ReduceOp.SUM, group=group)
                dist.all_gather_into_tensor(self.sum_storage, mine, group=group)
                dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=group)
            self.rows_unknown()
            return self.owned_rows(P, rank, world)
        r0, r1 = self.reduce_scatter(rank, world, group)
        if not self._multi(group) or dist.get_backend(group) != "nccl":
            return r0, r1  # (gloo: reduce_scatter() already all-reduced)
        per = (P + world - 1) // world
        if per * world == P:
            dist.all_gather_into_tensor(self.flat, self.flat[r0:r1].clone(), group=group)
        else:
            mine = self.flat.new_zeros(per, width)
            mine[: r1 - r0] = self.flat[r0:r1]
            full = self.flat.new_empty(per * world, width)
            dist.all_gather_into_tensor(full, mine, group=group)
            self.flat.copy_(full[:P])
        return r0, r1

    # the sparse exchange is two collectives and three local passes (flags, pack, unpack) where the dense one is one
    # collective: it has to save at least a quarter of the payload to be the faster one
    SPARSE_MARGIN = 0.75

    def sparse_pays(self, rows, capacity=None):
        """Choosing the exchange from the data: does exchanging `rows` packed gradient rows (the union
        of the ranks' non-zero rows; `capacity` rows travel in the capacity-bound form) move fewer bytes than the dense
        bucket (by the margin above)?  Payloads per rank: sparse = 8 P (flags + radii, int32) + (rows x width + 2 P) x 4;
        dense = the bucket, the statistics and the radii.  On the i.i.d. volume of SURVEY 8(d) a view leaves 2 % of the rows live and sparse wins by
        an order of magnitude; on a surface map (scene.make_room_scene) one view leaves 20 % live and the union over a
        12-view window two thirds — there the dense two-phase exchange is the smaller one.
        Returns (pays, sparse_bytes, dense_bytes)."""
        P, width = self.flat.shape
        n = int(rows if capacity is None else capacity)
        sparse_bytes = 8 * P + (n * width + 2 * P) * 4
        dense_bytes = self.sum_storage.numel() * 4 + 4 * P
        return sparse_bytes < self.SPARSE_MARGIN * dense_bytes, sparse_bytes, dense_bytes

    # ---- sparse exchange (the parity-preserving half) ------------------------------
    def sparse_all_reduce(self, group=None, auto=False):
        """Exchange only the gradient rows that are non-zero on at least one rank: every other row is zero everywhere, so
        its sum is the zero it already holds.  Saturation ends most tile lists early, so only the front layer of Gaussians
        receives gradients at all (config 3: 2 % of the visible ones per view) — far fewer rows than the ones a rank merely
        SAW, which is what round 2 exchanged.  On the GPU: the two collectives and the library launches of
        sparse_all_reduce_capped, with the packed buffer sized EXACTLY from the union's row count, which is read back (one
        host synchronisation per exchange; a mapping step has one exchange).  On CPU tensors (the specification): all-gather
        of the per-rank bitmasks of non-zero rows (P / 8 bytes), all-reduce (SUM) of the packed [n_active, width] rows, and
        the small side buffers — the two densification statistics (SUM, [P, 2]) and max_radii (MAX, [P]) — dense.
        Same values as all_reduce().
        auto: once the union's size is known (it is identical on every rank) the rows travel packed only if that is the
        smaller payload (sparse_pays), else the bucket is all-reduced densely — the flags have told every rank the same thing,
        so all ranks take the same branch; max_radii is already reduced by then and the row mask holds the exact union.
        Returns dict(active_rows, bytes_dense, bytes_sparse, chosen) — the bytes each rank contributes to the wire."""
        import torch.distributed as dist
        P, width = self.flat.shape
        dense_bytes = self.sum_storage.numel() * 4 + self.max_radii.numel() * 4
        if self.flat.is_cuda and not self.capped_torch_formulation:
            # the product path: the library's launches around TWO collectives (the capacity-bound form's, include/olsr.h), the
            # packed buffer sized exactly from a count that is read back - this exchange's one host synchronisation
            L, dev = lib(), self.flat.device
            multi = self._multi(group)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            mask_p = self.row_mask.data_ptr() if self.row_mask is not None else None
            imax = torch.empty(2 * P, dtype=torch.int32, device=dev)
            scratch = torch.empty(max(1, int(L.olsr_sparse_exchange_scratch_ints(P))), dtype=torch.int32, device=dev)
            status = torch.zeros(2, dtype=torch.int32, device=dev)
            check(L.olsr_sparse_exchange_mask(P, width, self.flat.data_ptr(), mask_p, self.max_radii.data_ptr(),
                                              imax.data_ptr(), stream))
            if multi:
                dist.all_reduce(imax, op=dist.ReduceOp.MAX, group=group)
            check(L.olsr_sparse_exchange_pack(P, width, 1, self.flat.data_ptr(), imax.data_ptr(), self.max_radii.data_ptr(),
                                              mask_p, self.densify.data_ptr(), None, None, scratch.data_ptr(),
                                              status.data_ptr(), stream))                      # count only
            n = int(status[0].item())                                                           # identical on every rank
            pays, sparse_b, dense_b = self.sparse_pays(n)
            if not multi:
                return dict(active_rows=n, bytes_dense=dense_bytes, bytes_sparse=0,
                            chosen=("sparse" if (pays or not auto) else "dense"))
            if auto and not pays:
                # the dense leg: gradient rows and statistics in one SUM all-reduce (the radii travelled with the flags; the
                # row mask is the union the count-only pack left: exact for the summed bucket too)
                dist.all_reduce(self.sum_storage, op=dist.ReduceOp.SUM, group=group)
                return dict(active_rows=n, bytes_dense=dense_bytes, bytes_sparse=int(8 * P) + dense_b - 4 * P, chosen="dense")
            cap = max(n, 1)
            idx = torch.empty(cap, dtype=torch.int32, device=dev)
            fsum = torch.empty(cap * width + 2 * P, dtype=torch.float32, device=dev)
            check(L.olsr_sparse_exchange_pack(P, width, cap, self.flat.data_ptr(), imax.data_ptr(), self.max_radii.data_ptr(),
                                              mask_p, self.densify.data_ptr(), idx.data_ptr(), fsum.data_ptr(),
                                              scratch.data_ptr(), status.data_ptr(), stream))
            dist.all_reduce(fsum, op=dist.ReduceOp.SUM, group=group)
            check(L.olsr_sparse_exchange_unpack(P, width, cap, idx.data_ptr(), fsum.data_ptr(), self.flat.data_ptr(),
                                                self.densify.data_ptr(), stream))
            return dict(active_rows=n, bytes_dense=dense_bytes, bytes_sparse=int(8 * P + (n * width + 2 * P) * 4), chosen="sparse")
        # CPU tensors (the gloo tests): torch operations, four collectives - the specification of the path above
        nonzero = (self.flat != 0).any(dim=1)
        if not self._multi(group):
            n_ = int(nonzero.sum())
            return dict(active_rows=n_, bytes_dense=dense_bytes, bytes_sparse=0,
                        chosen=("sparse" if (self.sparse_pays(n_)[0] or not auto) else "dense"))
        world = dist.get_world_size(group)
        self.rows_unknown()
        nb = (P + 7) // 8
        bits = torch.zeros(nb * 8, dtype=torch.uint8, device=nonzero.device)
        bits[:P] = nonzero
        weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=nonzero.device)
        mine = (bits.view(nb, 8).to(torch.int32) * weights).sum(dim=1).to(torch.uint8)       # P / 8 bytes
        every = torch.empty(world * nb, dtype=torch.uint8, device=nonzero.device)
        dist.all_gather_into_tensor(every, mine, group=group)                                 # bitmasks of all ranks
        union = every.view(world, nb)[0].clone()
        for r in range(1, world):
            union |= every.view(world, nb)[r]
        anyrow = ((union.view(nb, 1).to(torch.int32) // weights) % 2).reshape(-1)[:P]
        idx = torch.nonzero(anyrow, as_tuple=False).reshape(-1)            # identical on every rank, ascending
        if auto and not self.sparse_pays(int(idx.numel()))[0]:
            dist.all_reduce(self.sum_storage, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=group)
            return dict(active_rows=int(idx.numel()), bytes_dense=dense_bytes, bytes_sparse=int(nb) + dense_bytes, chosen="dense")
        packed = self.flat.index_select(0, idx)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        self.flat.index_copy_(0, idx, packed)
        dist.all_reduce(self.densify, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=group)
        return dict(active_rows=int(idx.numel()), bytes_dense=dense_bytes,
                    bytes_sparse=int(nb + idx.numel() * width * 4 + P * 12), chosen="sparse")

    def sparse_all_reduce_capped(self, capacity, group=None):
        """The sparse exchange WITHOUT a host synchronisation, for callers that keep several frames in flight
        (bench.py's weak-scaling mode): the packed buffer has a fixed `capacity` of rows instead of the exact count, and the
        step is TWO collectives (each costs a launch and a latency on every rank, whatever its size):
        1. all-reduce (MAX) of one int32 buffer [flags of this rank's non-zero gradient rows | max_radii] (8 P bytes) —
           afterwards every rank holds the same union of rows, and max_radii is done;
        2. all-reduce (SUM) of one fp32 buffer [the union's rows, packed in ascending order into capacity x width (unused
           slots are zero and belong to no row) | the two densification statistics], scattered back afterwards.
        The local work around the collectives is four launches of the library (olsr_sparse_exchange_mask / _pack / _unpack,
        include/olsr.h); with track_rows the backward's row mask tells the first of them which rows to look at and the second
        leaves the union in it, so the mask stays exact across the exchange (a dense collective has to mark it unknown).
        Everything is enqueued on the current stream.  Returns a device int32[2] {rows in the union, overflow flag}: with more
        than `capacity` rows in the union only the first `capacity` of them were exchanged - the step's gradients are then
        incomplete and the caller must repeat it with a larger capacity (or densely), the same contract as an instance
        overflow of olsr_forward_async.  Same values as all_reduce() otherwise."""
        import torch.distributed as dist
        P, width = self.flat.shape
        cap = int(capacity)
        dev = self.flat.device
        multi = self._multi(group)
        if self.flat.is_cuda and not self.capped_torch_formulation:
            # the product path: four launches of the library around the two collectives (include/olsr.h, "the capacity-bound
            # sparse exchange"); the row mask the backward keeps says which rows to look at, and comes back as the union
            st = getattr(self, "_capped", None)
            if st is None or st["cap"] != cap:
                st = dict(cap=cap, idx=torch.empty(cap, dtype=torch.int32, device=dev),
                          fsum=torch.empty(cap * width + 2 * P, dtype=torch.float32, device=dev),
                          imax=torch.empty(2 * P, dtype=torch.int32, device=dev),
                          scratch=torch.empty(max(1, int(lib().olsr_sparse_exchange_scratch_ints(P))), dtype=torch.int32,
                                              device=dev),
                          status=torch.zeros(2, dtype=torch.int32, device=dev))
                self._capped = st
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            mask_p = self.row_mask.data_ptr() if self.row_mask is not None else None
            check(lib().olsr_sparse_exchange_mask(P, width, self.flat.data_ptr(), mask_p, self.max_radii.data_ptr(),
                                                  st["imax"].data_ptr(), stream))
            dc = self._direct(group)
            if multi and dc is not None:
                dc.all_reduce(st["imax"], "max")
            elif multi:
                dist.all_reduce(st["imax"], op=dist.ReduceOp.MAX, group=group)  # the union, identical on every rank; the radii
            check(lib().olsr_sparse_exchange_pack(P, width, cap, self.flat.data_ptr(), st["imax"].data_ptr(),
                                                  self.max_radii.data_ptr(), mask_p, self.densify.data_ptr(),
                                                  st["idx"].data_ptr(), st["fsum"].data_ptr(), st["scratch"].data_ptr(),
                                                  st["status"].data_ptr(), stream))
            if multi:
                if dc is not None:
                    dc.all_reduce(st["fsum"], "sum")
                else:
                    dist.all_reduce(st["fsum"], op=dist.ReduceOp.SUM, group=group)
                check(lib().olsr_sparse_exchange_unpack(P, width, cap, st["idx"].data_ptr(), st["fsum"].data_ptr(),
                                                        self.flat.data_ptr(), self.densify.data_ptr(), stream))
            return st["status"]
        # CPU tensors (the gloo tests): the same exchange in torch operations - the specification of the kernels above
        st = getattr(self, "_capped", None)
        if st is None or st["cap"] != cap:
            st = dict(cap=cap, idx=torch.empty(cap + 1, dtype=torch.int64, device=dev),
                      fsum=torch.empty(cap * width + 2 * P, dtype=torch.float32, device=dev),
                      imax=torch.empty(2 * P, dtype=torch.int32, device=dev),
                      arange=torch.arange(P, dtype=torch.int64, device=dev),
                      status=torch.zeros(2, dtype=torch.int32, device=dev))
            self._capped = st
        imax, fsum = st["imax"], st["fsum"]
        mask, packed, tail = imax[:P], fsum[: cap * width].view(cap, width), fsum[cap * width:]
        mask.copy_((self.flat != 0).any(dim=1))
        if multi:
            imax[P:].copy_(self.max_radii)
            dist.all_reduce(imax, op=dist.ReduceOp.MAX, group=group)       # the union, identical on every rank; the radii
            self.max_radii.copy_(imax[P:])
        pos = torch.cumsum(mask, 0, dtype=torch.int32)                     # 1-based rank of every row of the union
        slot = torch.where((mask != 0) & (pos <= cap), pos - 1, cap).to(torch.int64)
        st["idx"].fill_(P)
        st["idx"].scatter_(0, slot, st["arange"])                          # (slot `cap` collects the rows left out)
        idx = st["idx"][:cap]
        torch.index_select(self.flat_ext, 0, idx, out=packed)
        if multi:
            self.rows_unknown()
            tail.copy_(self.densify.reshape(-1))
            dist.all_reduce(fsum, op=dist.ReduceOp.SUM, group=group)
            self.flat_ext.index_copy_(0, idx, packed)                      # (fill slots write zeros to the spare row)
            self.densify.copy_(tail.view(P, 2))
        st["status"][0:1] = pos[-1:]
        st["status"][1:2] = (pos[-1:] > cap).to(torch.int32)
        return st["status"]

    def exchange_bytes(self, exchange, capacity=None):
        """Payload bytes one rank hands to the collectives of one step, by exchange mode (the wire volume per GPU is
        this times the algorithm's factor, e.g. 2 (G-1)/G for a ring all-reduce)."""
        P, width = self.flat.shape
        side = P * 8 + P * 4
        if exchange == "sparse":  # (capacity-bound form: [mask | radii] as int32, then [packed rows | statistics] as fp32)
            return 8 * P + (int(capacity) * width + 2 * P) * 4
        return self.sum_storage.numel() * 4 + P * 4


class FusedAdam:
    """torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) of the reference's GaussianModel
    (gaussian_splatting/scene/gaussian_model.py:393-440) as ONE kernel over the gradient bucket
    (olsr_adam_step, include/olsr.h): same arithmetic, same dense semantics, parameters updated in place."""

    def __init__(self, P, layout: GradLayout, device, betas=(0.9, 0.999), eps=1e-15):
        self.layout, self.betas, self.eps = layout, betas, eps
        self.exp_avg = torch.zeros(P, layout.width, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(P, layout.width, dtype=torch.float32, device=device)
        self.step_count = 0
        self.use_row_masks = True  # (False: read every gradient row, the A/B leg of tests and bench)

    def step(self, bucket, params: Dict[str, torch.Tensor], lrs: Dict[str, float], rows=None):
        """params: means3D [P,3], shs [P,M,3], opacities [P(,1)], scales [P,3], rotations [P,4], language [P,F]
        (contiguous fp32 on the GPU, updated in place); lrs: xyz, sh_dc, sh_rest, opacity, scale, rotation, language.
        rows = (r0, r1): update only that contiguous range of Gaussians (the rows a rank owns after a
        reduce-scatter); the moments of the other rows are left alone."""
        self.step_count += 1
        hp = _abi.OlsrAdamParams(lr_xyz=lrs["xyz"], lr_sh_dc=lrs["sh_dc"], lr_sh_rest=lrs["sh_rest"], lr_opacity=lrs["opacity"],
                                 lr_scale=lrs["scale"], lr_rotation=lrs["rotation"], lr_language=lrs.get("language", 0.0),
                                 beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, step=self.step_count)
        for k, t in params.items():
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError(f"FusedAdam: {k} must be a contiguous fp32 tensor on the GPU")
        # `bucket` may be a LIST of buckets (the lane buckets of a FrameLanes): their rows are summed on the fly, in list
        # order — the bits a sum into the first one would give, without its passes over the buckets (olsr_adam_step_sum)
        buckets = list(bucket) if isinstance(bucket, (list, tuple)) else [bucket]
        bucket = buckets[0]
        P = bucket.flat.shape[0]
        r0, r1 = (0, P) if rows is None else rows
        if r1 <= r0:
            return
        M, F, W = self.layout.M, self.layout.F, self.layout.width
        per_row = dict(means3D=3, shs=3 * M, opacities=1, scales=3, rotations=4, language=F)

        def p(name):
            t = params.get(name)
            return t.data_ptr() + 4 * r0 * per_row[name] if t is not None and t.numel() > 0 else None
        stream = C.c_void_p(torch.cuda.current_stream(bucket.flat.device).cuda_stream)
        # rows a bucket's row mask proves zero are not read (olsr_adam_step_masked: the update stays dense, the bits are
        # torch.optim.Adam's); a row range that does not start on a mask word (64 rows) takes the unmasked step
        masked = self.use_row_masks and r0 % 64 == 0 and any(b.row_mask is not None for b in buckets)
        if len(buckets) > 1 or masked:
            flats = (C.c_void_p * len(buckets))(*[b.flat.data_ptr() + 4 * r0 * W for b in buckets])
            masks = None
            if masked:
                masks = (C.c_void_p * len(buckets))(*[(b.row_mask.data_ptr() + 8 * (r0 // 64)) if b.row_mask is not None else None
                                                      for b in buckets])
            check(lib().olsr_adam_step_masked(r1 - r0, M, F, C.byref(hp), len(buckets), flats, masks, p("means3D"), p("shs"),
                                              p("opacities"), p("scales"), p("rotations"), p("language"),
                                              self.exp_avg.data_ptr() + 4 * r0 * W,
                                              self.exp_avg_sq.data_ptr() + 4 * r0 * W, stream))
            return
        check(lib().olsr_adam_step(r1 - r0, M, F, C.byref(hp), bucket.flat.data_ptr() + 4 * r0 * W, p("means3D"),
                                   p("shs"), p("opacities"), p("scales"), p("rotations"), p("language"),
                                   self.exp_avg.data_ptr() + 4 * r0 * W, self.exp_avg_sq.data_ptr() + 4 * r0 * W,
                                   stream))


def _raise_on_sync_error(status, where):
    if status == 2:
        raise RuntimeError(f"olsr {where}: device-side synchronisation error (OLSR_STATUS_SYNC_ERROR): a look-back of the "
                           "frame's radix sort / row compaction never received a predecessor's counts; the frame's "
                           "synchronisation words were overwritten mid-frame.  Images of that frame are invalid, gradients zero")


class RasterWorkspace:
    """Allocation-free, sync-free forward+backward through the C-ABI (one view at a time)."""

    def __init__(self, P, W, H, F, M, capacity, device, tile=15, bwd_mode=_abi.BWD_REFERENCE, row_capacity=None,
                 binning=_abi.BINNING_ELLIPSE, flags=0, depth_cut=False, rows_in_forward=None, carry_order=False):
        self.P, self.W, self.H, self.F, self.M = int(P), int(W), int(H), int(F), int(M)
        self.capacity, self.tile, self.bwd_mode = int(capacity), int(tile), int(bwd_mode)
        self.binning_mode = int(binning)
        self.flags = int(flags)  # _abi.FLAG_* (e.g. FLAG_FWD_ACCUM_MFMA)
        # rows of the backward scratch: live (instance, slot) pairs, at most 4 per instance; one per
        # instance covers ordinary scenes several times over (config 3 needs 0.24), overflow is reported
        self.row_capacity = int(row_capacity) if row_capacity is not None else self.capacity
        # the forward's last launch also compacts the backward's rows (olsr_scene.backward_row_capacity, include/olsr.h): one
        # launch less per frame; False keeps the compaction in the backward (a forward that is never followed by one)
        # None: on for a workspace that renders alone (isolated frame + 0.7 %, tracking iteration - 1.5 %), off with
        # OLSR_FLAG_FRAMES_IN_FLIGHT: the merged launch is made of sixteen-wave blocks, and beside another lane's composite those
        # are placed late — the tile order's light blocks too, which the separate launch starts as four waves (measured: headline
        # - 1 %, 12-view mapping iteration - 1 %).  OLSR_ROWS_IN_FORWARD=0 / 1 overrides (the A/B switch of the measurement).
        if rows_in_forward is None:
            env = os.environ.get("OLSR_ROWS_IN_FORWARD")
            rows_in_forward = (env != "0") if env is not None else not (self.flags & _abi.FLAG_FRAMES_IN_FLIGHT)
        self.rows_in_forward = bool(rows_in_forward) and self.row_capacity > 0
        self.device = torch.device(device)
        L = lib()
        u8 = dict(dtype=torch.uint8, device=self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.geom = torch.empty(L.olsr_geometry_bytes(P, F), **u8)
        self.img = torch.empty(L.olsr_image_bytes(W, H, tile), **u8)
        self.binning = torch.empty(L.olsr_binning_bytes(self.capacity, F), **u8)
        self.scratch = torch.empty(L.olsr_backward_scratch_bytes(self.row_capacity, F), **u8)
        self.bwd_status = torch.zeros(2, **i32)  # {live rows, overflow flag}
        self.out = dict(color=torch.empty(3, H, W, **f32), language=torch.empty(F, H, W, **f32),
                        depth=torch.empty(1, H, W, **f32), opacity=torch.empty(1, H, W, **f32),
                        radii=torch.empty(P, **i32), n_touched=torch.empty(P, **i32))
        self.num_rendered = torch.zeros(2, **i32)  # {R, overflow flag}, stays on the device
        # launch-order hint of the forward composite: heaviest tile first per XCD chunk, as measured on this
        # workspace's previous frame (consecutive SLAM frames load the tiles alike); identity before the first
        ntiles = ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
        self.tile_order = torch.arange(ntiles, **i32)
        # per-tile depth cut-offs (include/olsr.h; opt-in, for sequences of nearly identical views such as the tracking
        # iterations of a frame): in = what the previous forward left, out = this forward's; +inf = no cut.  A forward whose
        # status is OLSR_STATUS_CUT_MISS (forward_status() == 3) must be repeated — the array has been repaired by then
        # per-tile depth cut-offs (include/olsr.h): [0, tiles) the cut-offs in force, [tiles, 2 tiles) library scratch
        self._depth_cut_buf = torch.full((2 * ntiles,), float("inf"), **f32) if depth_cut else None
        self.depth_cut = self._depth_cut_buf[:ntiles] if depth_cut else None
        # Carried depth order (include/olsr.h, csrc/k_order_carry.hip; opt-in, for sequences of nearly identical views of the
        # same Gaussians): the forward repairs the order its previous frame left here instead of sorting from scratch (two
        # launches instead of five dependent ones) and falls back to the radix passes on the device when it cannot prove
        # the result — the lists never depend on the array's content, so it may be swapped (one array per VIEW: a caller that
        # cycles through views assigns ws.depth_order_carry before set_scene, MappingStep does) or hold zeros.
        self.depth_order_carry = torch.zeros(P, **i32) if carry_order else None
        self.grads = dict(dL_dmeans2D=torch.empty(P, 3, **f32), dL_dconic=torch.empty(P, 4, **f32),
                          dL_dopacity=torch.empty(P, 1, **f32), dL_dcolors=torch.empty(P, 3, **f32),
                          dL_dlanguage=torch.empty(P, F, **f32), dL_ddepths=torch.empty(P, 1, **f32),
                          dL_dmeans3D=torch.empty(P, 3, **f32), dL_dcov3D=torch.empty(P, 6, **f32),
                          dL_dsh=torch.empty(P, M, 3, **f32), dL_dscales=torch.empty(P, 3, **f32),
                          dL_drotations=torch.empty(P, 4, **f32), dL_dtau=torch.empty(P, 6, **f32),
                          dL_dtau_sum=torch.empty(6, **f32))
        self._scene = None
        self._keep = None

    def state_bytes(self):
        return self.geom.numel() + self.img.numel() + self.binning.numel() + self.scratch.numel()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_scene(self, *, bg, means3D, opacities, scales, rotations, shs, language, viewmatrix, projmatrix,
                  projmatrix_raw, campos, tanfovx, tanfovy, sh_degree, scale_modifier=1.0, colors_precomp=None,
                  cov3D_precomp=None, activations=0):
        """Bind (borrow) the input tensors of the next forward/backward.  All must be contiguous fp32 on the
        workspace device.  `activations` (_abi.ACT_* bits): opacities / scales / rotations are the RAW parameters
        (GaussianModel._opacity, _scaling, _rotation); the kernels apply sigmoid / exp / normalize and the
        backward returns gradients with respect to the raw parameters."""
        keep = [bg, means3D, shs, colors_precomp, language, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                projmatrix, projmatrix_raw, campos]
        for t in keep:
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError("RasterWorkspace inputs must be contiguous fp32 tensors on the GPU")
        self._keep = keep
        self._scene = _abi.make_scene(
            P=self.P, D=sh_degree, M=self.M if shs is not None else 0, F=self.F, width=self.W, height=self.H,
            tile=self.tile, prefiltered=False, debug=False, bwd_mode=self.bwd_mode, tan_fovx=tanfovx,
            tan_fovy=tanfovy, scale_modifier=scale_modifier, binning=self.binning_mode, activations=activations,
            flags=self.flags, background=bg,
            means3D=means3D, shs=shs,
            colors_precomp=colors_precomp, language_precomp=language, opacities=opacities, scales=scales,
            rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix,
            projmatrix_raw=projmatrix_raw, cam_pos=campos, tile_depth_cut=self._depth_cut_buf,
            backward_row_capacity=self.row_capacity if self.rows_in_forward else 0,
            depth_order_carry=self.depth_order_carry)

    def forward(self):
        o = self.out
        check(lib().olsr_forward_async(
            C.byref(self._scene), self.geom.data_ptr(), self.binning.data_ptr(), self.capacity, self.img.data_ptr(),
            o["color"].data_ptr(), o["language"].data_ptr() if self.F > 0 else None, o["depth"].data_ptr(),
            o["opacity"].data_ptr(), o["radii"].data_ptr(), o["n_touched"].data_ptr(), self.num_rendered.data_ptr(),
            self.tile_order.data_ptr(), self._stream()))
        return o

    def forward_loss(self, gt_image, gt_depth, gt_language=None, exposure=None, grad_mask=None, *, tracking=False,
                     alpha=0.95, rgb_boundary_threshold=0.01, lamda_lang=1.0, initialization=False, skip_images=True):
        """The forward with the mapping (tracking=False) or tracking loss evaluated in the composite kernel's epilogue
        (olsr_forward_async_loss): no separate loss kernel, the rendered images make no round trip.  Targets as
        losses.mapping_loss / losses.tracking_loss take them, already float32 and contiguous on the workspace's device.
        skip_images: the images are not written at all (self.out keeps whatever it held).
        Returns dict(loss[4], dL_dimage, dL_ddepth, dL_dlanguage or None, dL_dexposure[2]) — workspace-owned buffers that
        the next call overwrites; hand the three cotangents to backward()."""
        f32 = dict(dtype=torch.float32, device=self.device)
        if getattr(self, "_fl", None) is None:
            L = lib()
            self._fl = dict(loss=torch.empty(4, **f32), dL_dimage=torch.empty(3, self.H, self.W, **f32),
                            dL_ddepth=torch.empty(1, self.H, self.W, **f32), dL_dexposure=torch.empty(2, **f32),
                            dL_dlanguage=torch.empty(self.F, self.H, self.W, **f32) if self.F > 0 else None,
                            scratch=torch.empty(L.olsr_fused_loss_scratch_bytes(self.W, self.H, self.tile),
                                                dtype=torch.uint8, device=self.device))
        fl = self._fl
        lang = (not tracking) and self.F > 0 and gt_language is not None
        for name, t in (("gt_image", gt_image), ("gt_depth", gt_depth), ("gt_language", gt_language if lang else None),
                        ("exposure", exposure), ("grad_mask", grad_mask)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()
                                  or t.device != self.device):
                raise RuntimeError(f"forward_loss: {name} must be a contiguous float32 tensor on {self.device}")
        if tuple(gt_image.shape) != (3, self.H, self.W) or gt_depth.numel() != self.H * self.W:
            raise RuntimeError("forward_loss: gt_image must be [3,H,W] and gt_depth [H,W] of the workspace's size")
        if grad_mask is not None and grad_mask.numel() != self.H * self.W:
            raise RuntimeError("forward_loss: grad_mask must be [H,W]")
        if lang and (gt_language.dim() != 3 or gt_language.shape[0] != self.F):
            raise RuntimeError("forward_loss: gt_language must be [F,h,w]")
        p = _abi.OlsrLossParams(width=self.W, height=self.H, F=self.F if lang else 0,
                                lang_width=gt_language.shape[2] if lang else 0,
                                lang_height=gt_language.shape[1] if lang else 0, initialization=int(bool(initialization)),
                                alpha=float(alpha), rgb_boundary_threshold=float(rgb_boundary_threshold),
                                lamda_lang=float(lamda_lang))
        ptr = lambda t: t.data_ptr() if t is not None and t.numel() > 0 else None  # noqa: E731
        lf = _abi.OlsrLossFusion(params=p, tracking=int(bool(tracking)), skip_images=int(bool(skip_images)),
                                 gt_image=ptr(gt_image), gt_depth=ptr(gt_depth), gt_language=ptr(gt_language) if lang else None,
                                 exposure=ptr(exposure), grad_mask=ptr(grad_mask), dL_dimage=ptr(fl["dL_dimage"]),
                                 dL_ddepth=ptr(fl["dL_ddepth"]), dL_dlanguage=ptr(fl["dL_dlanguage"]) if lang else None,
                                 loss=ptr(fl["loss"]), dL_dexposure=ptr(fl["dL_dexposure"]), scratch=ptr(fl["scratch"]))
        o = self.out
        check(lib().olsr_forward_async_loss(
            C.byref(self._scene), self.geom.data_ptr(), self.binning.data_ptr(), self.capacity, self.img.data_ptr(),
            o["color"].data_ptr(), o["language"].data_ptr() if self.F > 0 else None, o["depth"].data_ptr(),
            o["opacity"].data_ptr(), o["radii"].data_ptr(), o["n_touched"].data_ptr(), self.num_rendered.data_ptr(),
            self.tile_order.data_ptr(), C.byref(lf), self._stream()))
        return dict(loss=fl["loss"], dL_dimage=fl["dL_dimage"], dL_ddepth=fl["dL_ddepth"],
                    dL_dlanguage=fl["dL_dlanguage"] if lang else None, dL_dexposure=fl["dL_dexposure"])

    def backward(self, dL_dcolor, dL_dlanguage, dL_ddepth, bucket=None, first=False, bucket_only=False,
                 pose_only=False):
        """Backward of the last forward.  With `bucket` (a GradientBucket) the per-Gaussian backward kernel also
        writes (first=True) or adds this view's gradients into the bucket — the fused form of
        bucket.accumulate(); `bucket_only` additionally skips the separate per-Gaussian gradient arrays
        (a mapping step only consumes the bucket and dL_dtau_sum).  `pose_only`: tracking — only dL_dtau_sum
        ([rho | theta] of the camera) is produced, no per-Gaussian array is written."""
        g = self.grads

        def p(t):
            return t.data_ptr() if t is not None and t.numel() > 0 else None
        bk = None
        if bucket is not None:
            bk = _abi.OlsrGradBucket(flat=bucket.flat.data_ptr(), densify=bucket.densify.data_ptr(),
                                     max_radii=bucket.max_radii.data_ptr(), assign=1 if first else 0,
                                     row_mask=bucket.row_mask.data_ptr() if bucket.row_mask is not None else None)
            if bucket_only:
                g = {k: (v if k == "dL_dtau_sum" else None) for k, v in g.items()}
        if pose_only:
            g = {k: (v if k == "dL_dtau_sum" else None) for k, v in g.items()}
        check(lib().olsr_backward(
            C.byref(self._scene), self.out["radii"].data_ptr(), self.geom.data_ptr(), self.capacity,
            self.binning.data_ptr(), self.img.data_ptr(), _abi.ALLOC_FN(0), None, self.scratch.data_ptr(),
            self.row_capacity, p(dL_dcolor), p(dL_dlanguage), p(dL_ddepth),
            p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]), p(g["dL_dcolors"]), p(g["dL_dlanguage"]),
            p(g["dL_ddepths"]), p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]),
            p(g["dL_drotations"]), p(g["dL_dtau"]), p(g["dL_dtau_sum"]), C.byref(bk) if bk is not None else None,
            self.bwd_status.data_ptr(), self._stream()))
        return g

    def backward_status(self):
        """(live rows L, overflow) of the last backward — synchronises; call outside timed regions.  Raises on a device-side
        synchronisation error (OLSR_STATUS_SYNC_ERROR, include/olsr.h): the gradients of that backward are zeros."""
        st = self.bwd_status.cpu()
        _raise_on_sync_error(int(st[1]), "backward")
        return int(st[0]), bool(st[1])

    def rendered(self):
        """(R, overflow) — synchronises; call outside timed regions.  Raises on a device-side synchronisation error."""
        r = self.num_rendered.cpu()
        _raise_on_sync_error(int(r[1]), "forward")
        return int(r[0]), int(r[1]) == 1

    def forward_status(self):
        """OLSR_STATUS_* of the last forward (0 fine, 1 capacity overflow, 2 synchronisation error, 3 a depth cut-off was
        missed: repeat the forward) — synchronises."""
        return int(self.num_rendered.cpu()[1])

    def carry_missed(self):
        """True when the last forward could not repair its carried depth order and took the radix passes (the result is the
        same either way) — synchronises; for tests and benchmarks."""
        from . import _C
        return bool(int(_C.state_field("geometry", self.geom, "carry_miss", P=self.P, F=self.F, dtype=torch.int32, count=1).cpu()[0]))

    def reset_depth_cut(self):
        if self.depth_cut is not None:
            self.depth_cut.fill_(float("inf"))


class FrameLanes:
    """Several frames in flight on one GPU: `n` independent RasterWorkspaces, each with its own HIP
    stream and gradient bucket.  The views of a mapping step are independent, and the ~40 small
    binning kernels of one frame leave most of the 256 CUs idle — a second and third frame on other
    streams fill them (measured on config 3: 705 -> 913 -> 1005 frames/s for 1 / 2 / 3 lanes)."""

    def __init__(self, n, P, W, H, F, M, capacity, device, track_rows=True, **kw):
        """track_rows: the lanes' buckets keep a row mask (GradientBucket): the first view a lane renders in a step rewrites
        only the gradient rows that changed.
        With more than one lane every workspace carries OLSR_FLAG_FRAMES_IN_FLIGHT (include/olsr.h): the radix sorts then run
        as four-wave workgroups, which get onto the CUs beside another lane's composite kernel (+ 2 % frames/s with four
        lanes at config 3; with ONE frame in flight the 1024-thread shape is 11 % faster, and a single lane keeps it)."""
        self.device = torch.device(device)
        self.lanes = []
        if int(n) > 1:
            kw = dict(kw, flags=int(kw.get("flags", 0)) | _abi.FLAG_FRAMES_IN_FLIGHT)
        for i in range(max(1, int(n))):
            ws = RasterWorkspace(P, W, H, F, M, capacity, device, **kw)
            stream = torch.cuda.current_stream(self.device) if i == 0 else self.lane_stream(self.device, i)
            self.lanes.append((ws, GradientBucket(P, GradLayout(M, F), device, track_rows=track_rows), stream))
        self._next = 0

    # Lane i of EVERY FrameLanes of a process runs on the same HIP stream (round 6).  A process has four hardware queues to give
    # (streams beyond them share queues, and a four-lane loop then runs 12 % slower until the process
    # ends); objects that each created their own streams — a benchmark's second scene, a mapping step beside a headline loop —
    # pushed later loops past that.  Sets of lanes are used one after the other, so sharing the streams orders nothing that
    # was concurrent.
    _lane_streams: Dict = {}

    @classmethod
    def lane_stream(cls, device, i):
        key = (torch.device(device).index or 0, int(i))
        if key not in cls._lane_streams:
            cls._lane_streams[key] = torch.cuda.Stream(torch.device(device))
        return cls._lane_streams[key]

    def __len__(self):
        return len(self.lanes)

    def next_lane(self):
        lane = self.lanes[self._next % len(self.lanes)]
        self._next += 1
        return lane

    def synchronize(self):
        for _, _, st in self.lanes:
            st.synchronize()


class FrameShardedStep:
    """One optimisation step's worth of rasterization, sharded by viewpoint over the ranks of a
    torch.distributed group (one process per GPU).  `cameras` is the full list of views of the step
    (identical on every rank); this rank renders views_of_rank(...).

    exchange:
      "all_reduce"      one SUM all-reduce of the flat bucket (+ the statistics); every rank then holds the total
                        and applies the identical update — what north_star names;
      "reduce_scatter"  owner-applies: rank r receives the summed rows of the Gaussians it
                        owns, steps Adam on those rows only (`optimizer_step`), and the updated parameter rows are
                        all-gathered — 2 x (G-1)/G of the PARAMETER bytes + (G-1)/G of the gradient bytes per GPU,
                        all of it spread over the G-1 direct xGMI links;
      "sparse"          all-reduce of the gradient rows that are non-zero on some rank (GradientBucket.sparse_all_reduce).
      "auto"            chosen per step from the data: the ranks learn the union of their non-zero rows (8 P bytes of flags
                        and radii), and the rows travel packed only when that is the smaller payload — else the bucket is
                        all-reduced densely (GradientBucket.sparse_pays; `wire["chosen"]` says which).  A view of the
                        i.i.d. volume leaves 2 % of the rows live, a view of a surface map 20 %, a 12-view window of it 67 %.
    A capacity overflow on ANY rank (instances or gradient rows: that view contributed zeros) is surfaced: the
    per-rank flag travels with the MAX all-reduce and `run` raises OverflowError on every rank, with the capacity
    that would have sufficed, so the caller can regrow its workspace and repeat the step."""

    def __init__(self, workspace, rank=0, world=1, group=None, exchange="all_reduce"):
        """workspace: a RasterWorkspace (this rank's views are rendered one after the other) or a FrameLanes (they are
        rendered `len(lanes)` at a time on the lanes' streams, each lane accumulating into its own bucket; the lane
        buckets are summed — in lane order, a fixed order — before the exchange)."""
        assert exchange in ("all_reduce", "reduce_scatter", "sparse", "auto")
        if isinstance(workspace, FrameLanes):
            self.lanes = list(workspace.lanes)
        else:
            self.lanes = [(workspace, GradientBucket(workspace.P, GradLayout(workspace.M, workspace.F), workspace.device),
                           None)]
        self.ws = self.lanes[0][0]
        ws = self.ws
        self.rank, self.world, self.group, self.exchange = rank, world, group, exchange
        self.bucket = self.lanes[0][1]
        self.pose_grads: Dict[int, torch.Tensor] = {}
        self.owned = GradientBucket.owned_rows(ws.P, rank, world)
        self.wire = None  # dict from the sparse exchange
        # per lane: max over its views {R, live rows} and the overflow flag (stay on the device, on the lane's stream)
        # launch-order hints of the forward composite, one per VIEW index (created on first use): consecutive steps of a
        # mapping call render the same window of keyframes, and a view's own previous order is worth 25-35 % of its forward
        # composite against the order of whatever the lane rendered last (slam_iterations.MappingStep)
        self.view_hints: Dict[int, torch.Tensor] = {}
        self._need = [torch.zeros(2, dtype=torch.int32, device=ws.device) for _ in self.lanes]
        self._ovf = [torch.zeros(1, dtype=torch.int32, device=ws.device) for _ in self.lanes]

    def run(self, gaussians: Dict[str, torch.Tensor], cameras: Sequence[Dict], cotangents, sh_degree=0):
        """gaussians: bg, means3D, opacities, scales, rotations, shs, language.
        cameras[v]: viewmatrix, projmatrix, projmatrix_raw, campos (device tensors), tanfovx, tanfovy.
        cotangents(v, outputs) -> (dL_dcolor, dL_dlanguage, dL_ddepth): the caller's loss gradient (called on the
        lane's stream)."""
        import torch.distributed as dist
        dev = self.ws.device
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        self.pose_grads.clear()
        mine = views_of_rank(len(cameras), self.rank, self.world)
        L = len(self.lanes)
        used = []
        # the overflow bookkeeping is cleared on the caller's stream, BEFORE the lanes are ordered behind it: the final fold
        # below reads every lane's words on that stream, also those of lanes this step gives no view
        for i in range(L):
            self._need[i].zero_()
            self._ovf[i].zero_()
        # (the tile-order hints of views seen for the first time are created here, on the caller's stream, BEFORE the lanes are
        #  ordered behind it: created inside the loop below they were written on this stream and read on a lane's)
        for v in mine:
            if v not in self.view_hints:
                self.view_hints[v] = torch.arange(self.ws.tile_order.numel(), dtype=torch.int32, device=self.ws.device)
        for i, (ws, bucket, stream) in enumerate(self.lanes):
            st = stream if stream is not None else main
            if st != main:
                st.wait_stream(main)  # the parameters this step renders from were written on the caller's stream
        for n_done, v in enumerate(mine):
            i = n_done % L
            ws, bucket, stream = self.lanes[i]
            st = stream if stream is not None else main
            first = i not in used
            if first:
                used.append(i)
            cam = cameras[v]
            with torch.cuda.stream(st):
                ws.tile_order = self.view_hints[v]
                ws.set_scene(sh_degree=sh_degree, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"],
                             projmatrix_raw=cam["projmatrix_raw"], campos=cam["campos"], tanfovx=cam["tanfovx"],
                             tanfovy=cam["tanfovy"], **gaussians)
                out = ws.forward()
                dc, dl, dd = cotangents(v, out)
                # the per-Gaussian backward kernel writes / adds straight into the lane's bucket (fused accumulate)
                g = ws.backward(dc, dl, dd, bucket=bucket, first=first, bucket_only=True)
                self.pose_grads[v] = g["dL_dtau_sum"].clone()  # [rho | theta], stays on the owning rank
                # overflow bookkeeping stays on the device (no sync inside the loop)
                torch.maximum(self._need[i][0:1], ws.num_rendered[0:1], out=self._need[i][0:1])
                torch.maximum(self._need[i][1:2], ws.bwd_status[0:1], out=self._need[i][1:2])
                torch.maximum(self._ovf[i], torch.maximum(ws.num_rendered[1:2], ws.bwd_status[1:2]), out=self._ovf[i])
        for i in used:
            st = self.lanes[i][2]
            if st is not None and st != main:
                main.wait_stream(st)
        total = self.bucket
        if not used:
            total.zero_()
        else:
            if used[0] != 0:  # (never: lane 0 takes this rank's first view)
                total.sum_storage.copy_(self.lanes[used[0]][1].sum_storage)
                total.rows_unknown()
                total.max_radii.copy_(self.lanes[used[0]][1].max_radii)
            for i in used[1:]:
                total.add_bucket(self.lanes[i][1])
        multi = GradientBucket._multi(self.group)
        if self.exchange == "reduce_scatter":
            self.owned = total.reduce_scatter(self.rank, self.world, self.group)
        elif self.exchange in ("sparse", "auto"):
            self.wire = total.sparse_all_reduce(self.group, auto=(self.exchange == "auto"))
        else:
            total.all_reduce(self.group)
        ovf = self._ovf[0].clone()
        need = self._need[0].clone()
        for i in range(1, L):
            torch.maximum(ovf, self._ovf[i], out=ovf)
            torch.maximum(need, self._need[i], out=need)
        flags = torch.cat([ovf, need])
        if multi:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        ovf, need_R, need_L = (int(x) for x in flags.cpu())  # the step's one host synchronisation
        _raise_on_sync_error(ovf, "frame-sharded step (some rank)")
        if ovf:
            ws = self.ws
            raise OverflowError(f"a view overflowed the workspace on some rank: it needs capacity >= {need_R} instances "
                                f"and row_capacity >= {need_L} gradient rows (has {ws.capacity} / {ws.row_capacity}); "
                                "regrow the RasterWorkspace and repeat the step")
        return total

    def optimizer_step(self, adam: "FusedAdam", params: Dict[str, torch.Tensor], lrs: Dict[str, float]):
        """Apply the step to the raw parameters.  all_reduce / sparse: every rank updates every row (identical
        results everywhere).  reduce_scatter: this rank updates the rows it owns, then the parameter rows are
        all-gathered so that every rank renders the next step from identical Gaussians."""
        import torch.distributed as dist
        if self.exchange != "reduce_scatter" or not GradientBucket._multi(self.group):
            adam.step(self.bucket, params, lrs)
            return
        adam.step(self.bucket, params, lrs, rows=self.owned)
        P = self.bucket.flat.shape[0]
        per = (P + self.world - 1) // self.world
        for name, t in params.items():
            if t is None or t.numel() == 0:
                continue
            rows = t.reshape(P, -1)
            w = rows.shape[1]
            if per * self.world == P:
                dist.all_gather_into_tensor(rows, rows[self.owned[0]:self.owned[1]].clone(), group=self.group)
            else:  # ragged tail: gather into a padded staging buffer
                mine = rows.new_zeros(per, w)
                mine[: self.owned[1] - self.owned[0]] = rows[self.owned[0]:self.owned[1]]
                full = rows.new_empty(per * self.world, w)
                dist.all_gather_into_tensor(full, mine, group=self.group)
                rows.copy_(full[:P])
