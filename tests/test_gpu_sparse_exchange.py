"""The capacity-bound sparse exchange's local kernels (include/olsr.h: olsr_sparse_exchange_mask / _pack / _unpack): two
"ranks" on one GPU, the two collectives replaced by the element-wise max / sum they compute, against the dense result —
and against the torch formulation GradientBucket.sparse_all_reduce_capped keeps for CPU tensors, which is their
specification (tests/test_frame_shard_gloo.py runs that one over real collectives)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bucket(P, width, rows, seed, dev, mask_kind):
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(P, width)
    sel = torch.randperm(P, generator=g)[:rows]
    flat[sel] = torch.randn(rows, width, generator=g)
    # rows whose only non-zero element is the last one, a NaN row and a row of negative zeros (zero for `!= 0`)
    if rows >= 4:
        flat[sel[0]] = 0.0
        flat[sel[0], width - 1] = 3.0
        flat[sel[1], 0] = float("nan")
        flat[sel[2]] = -0.0
    densify = torch.rand(P, 2, generator=g)
    radii = torch.randint(0, 500, (P,), generator=g, dtype=torch.int32)
    nz = (flat != 0).any(dim=1)
    if mask_kind == "none":
        mask = None
    else:
        bits = nz.clone() if mask_kind == "exact" else torch.ones(P, dtype=torch.bool)
        if mask_kind == "superset":
            bits = nz | (torch.rand(P, generator=g) < 0.3)
        pad = torch.zeros((P + 63) // 64 * 64, dtype=torch.int64)
        pad[:P] = bits
        w = pad.view(-1, 64)
        mask = torch.zeros(w.shape[0], dtype=torch.int64)
        for b in range(64):  # (bit 63 through the sign: int64 arithmetic wraps the way the bit pattern needs)
            mask |= w[:, b] << b
        mask = mask.to(dev)
    return flat.to(dev), densify.to(dev), radii.to(dev), mask


def _bits(mask, P):
    return ((mask.view(-1, 1) >> torch.arange(64, device=mask.device)) & 1).reshape(-1)[:P].bool()


@pytest.mark.parametrize("P,width,rows,mask_kind", [
    (5000, 29, 300, "exact"), (5000, 29, 300, "none"), (64 * 16 * 3 + 17, 91, 777, "superset"), (1024, 14, 5, "ones"),
    (100003, 29, 2500, "exact"), (63, 11, 10, "exact"), (2048, 29, 0, "exact")])
def test_two_ranks_on_one_gpu_equal_the_dense_exchange(hip, P, width, rows, mask_kind):
    _two_ranks(P, width, rows, mask_kind)


def test_random_shapes_around_the_block_boundaries(hip):
    """Seeded random sizes: P around the multiples of 64 and 1024 the kernels are cut by, empty to nearly full unions,
    every mask kind, widths from 11 (no SH, no language) to 91."""
    g = torch.Generator().manual_seed(2024)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    for _ in range(40):
        base = (64, 1024, 2048, 4096, 1024 * 7)[ri(0, 4)]
        P = max(1, base * ri(1, 3) + ri(-3, 3))
        width = (11, 14, 29, 46, 91)[ri(0, 4)]
        rows = min(P, (0, 1, 7, P // 50 + 1, P // 3 + 1, P - 1)[ri(0, 5)])
        _two_ranks(P, width, rows, ("exact", "none", "superset", "ones")[ri(0, 3)])


def _two_ranks(P, width, rows, mask_kind):
    from online_lang_splatting_amd._lib import check, lib
    L = lib()
    dev = torch.device(DEV)
    A = _bucket(P, width, rows, 1, dev, mask_kind)
    B = _bucket(P, width, rows // 2, 2, dev, mask_kind)
    dense_flat, dense_den, dense_rad = A[0] + B[0], A[1] + B[1], torch.maximum(A[2], B[2])
    union = ((A[0] != 0).any(1) | (B[0] != 0).any(1))
    n_union = int(union.sum())
    urows = torch.nonzero(union).reshape(-1)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nscr = max(1, int(L.olsr_sparse_exchange_scratch_ints(P)))
    assert nscr == (P + 1023) // 1024

    def p(t):
        return None if t is None else t.data_ptr()
    for cap in sorted({max(n_union, 1), n_union + 37, max(1, n_union // 2)}):
        ranks = []
        for flat, den, rad, mask in (A, B):
            ranks.append(dict(flat=flat.clone(), den=den.clone(), rad=rad.clone(), mask=None if mask is None else mask.clone(),
                              imax=torch.empty(2 * P, dtype=torch.int32, device=dev),
                              idx=torch.empty(cap, dtype=torch.int32, device=dev),
                              fsum=torch.empty(cap * width + 2 * P, dtype=torch.float32, device=dev),
                              scr=torch.empty(nscr, dtype=torch.int32, device=dev),
                              status=torch.zeros(2, dtype=torch.int32, device=dev)))
        for r in ranks:
            check(L.olsr_sparse_exchange_mask(P, width, p(r["flat"]), p(r["mask"]), p(r["rad"]), p(r["imax"]), stream))
            assert torch.equal(r["imax"][:P].bool(), (r["flat"] != 0).any(1)) and torch.equal(r["imax"][P:], r["rad"])
        imax = torch.maximum(ranks[0]["imax"], ranks[1]["imax"])          # collective 1: MAX
        for r in ranks:
            r["imax"].copy_(imax)
            check(L.olsr_sparse_exchange_pack(P, width, cap, p(r["flat"]), p(r["imax"]), p(r["rad"]), p(r["mask"]), p(r["den"]),
                                              p(r["idx"]), p(r["fsum"]), p(r["scr"]), p(r["status"]), stream))
            assert r["status"].tolist() == [n_union, int(n_union > cap)]
            used = min(n_union, cap)
            assert torch.equal(r["idx"][:used].long(), urows[:used]) and bool((r["idx"][used:] == P).all())
            assert torch.equal(r["fsum"][: cap * width].view(cap, width)[:used].nan_to_num(7.0),
                               r["flat"][urows[:used]].nan_to_num(7.0))
            assert bool((r["fsum"][used * width: cap * width] == 0).all())
            assert torch.equal(r["rad"], dense_rad)
            if r["mask"] is not None:
                assert torch.equal(_bits(r["mask"], P), union)
                assert not bool(_bits(r["mask"], r["mask"].numel() * 64)[P:].any())  # no stray bits behind row P
        fsum = ranks[0]["fsum"] + ranks[1]["fsum"]                        # collective 2: SUM
        for r in ranks:
            r["fsum"].copy_(fsum)
            check(L.olsr_sparse_exchange_unpack(P, width, cap, p(r["idx"]), p(r["fsum"]), p(r["flat"]), p(r["den"]), stream))
            used = min(n_union, cap)
            assert torch.equal(r["den"], dense_den)
            assert torch.equal(r["flat"][urows[:used]].nan_to_num(7.0), dense_flat[urows[:used]].nan_to_num(7.0))
            if used == n_union:
                assert torch.equal(r["flat"].nan_to_num(7.0), dense_flat.nan_to_num(7.0))
        torch.cuda.synchronize()


def test_bucket_method_matches_its_cpu_specification(hip):
    """GradientBucket.sparse_all_reduce_capped on a GPU bucket (the kernels) and on a CPU bucket (torch operations) leave the
    same status and, in a group of one, the bucket untouched; the tracked mask comes back exact."""
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket
    dev = torch.device(DEV)
    P, M, F = 9001, 1, 15
    lay = GradLayout(M, F)
    flat, den, rad, _ = _bucket(P, lay.width, 400, 5, torch.device("cpu"), "none")
    out = {}
    for where in ("cpu", "gpu"):
        b = GradientBucket(P, lay, dev if where == "gpu" else "cpu", track_rows=True)
        b.flat.copy_(flat)
        b.densify.copy_(den)
        b.max_radii.copy_(rad)
        for cap in (500, 123):
            st = b.sparse_all_reduce_capped(cap)
            out[(where, cap)] = st.cpu().tolist()
        assert torch.equal(b.flat.cpu().nan_to_num(7.0), flat.nan_to_num(7.0)) and torch.equal(b.densify.cpu(), den)
        if where == "gpu":
            assert torch.equal(_bits(b.row_mask, P).cpu(), (flat != 0).any(1))
    n = int((flat != 0).any(1).sum())
    assert out[("cpu", 500)] == out[("gpu", 500)] == [n, 0] and out[("cpu", 123)] == out[("gpu", 123)] == [n, 1]


def test_count_only_pack_leaves_the_count_and_nothing_else(hip):
    """idx == fsum == NULL: the pack call only counts (what the exact form of the exchange reads back to size its buffer)."""
    from online_lang_splatting_amd._lib import check, lib
    L = lib()
    dev = torch.device(DEV)
    P, width = 70001, 29
    flat, den, rad, mask = _bucket(P, width, 1234, 9, dev, "superset")
    n = int((flat != 0).any(1).sum())
    imax = torch.empty(2 * P, dtype=torch.int32, device=dev)
    scr = torch.empty(int(L.olsr_sparse_exchange_scratch_ints(P)), dtype=torch.int32, device=dev)
    status = torch.full((2,), -7, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    before = flat.clone()
    check(L.olsr_sparse_exchange_mask(P, width, flat.data_ptr(), mask.data_ptr(), rad.data_ptr(), imax.data_ptr(), stream))
    for cap, over in ((n, 0), (n - 1, 1), (1, 1)):
        check(L.olsr_sparse_exchange_pack(P, width, cap, flat.data_ptr(), imax.data_ptr(), rad.data_ptr(), mask.data_ptr(),
                                          den.data_ptr(), None, None, scr.data_ptr(), status.data_ptr(), stream))
        assert status.tolist() == [n, over]
    assert torch.equal(flat.nan_to_num(7.0), before.nan_to_num(7.0)) and torch.equal(_bits(mask, P), (flat != 0).any(1))
    assert L.olsr_sparse_exchange_pack(P, width, 5, flat.data_ptr(), imax.data_ptr(), rad.data_ptr(), None, den.data_ptr(),
                                       None, imax.data_ptr(), scr.data_ptr(), status.data_ptr(), stream) != 0  # one of the pair
