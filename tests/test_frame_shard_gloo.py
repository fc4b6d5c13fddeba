"""The N > 1 path on CPU: two gloo processes shard views, accumulate per-view gradients into the
flat buffer and all-reduce it.  (The rasterization itself needs a GPU; here the per-view gradients
are synthetic, so this covers the sharding rule, the buffer layout and the collectives.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket, views_of_rank


def _fake_view_grads(P, M, F, v):
    g = torch.Generator().manual_seed(100 + v)
    return dict(dL_dmeans3D=torch.randn(P, 3, generator=g), dL_dsh=torch.randn(P, M, 3, generator=g),
                dL_dopacity=torch.randn(P, 1, generator=g), dL_dscales=torch.randn(P, 3, generator=g),
                dL_drotations=torch.randn(P, 4, generator=g), dL_dlanguage=torch.randn(P, F, generator=g),
                dL_dmeans2D=torch.randn(P, 3, generator=g)), torch.randint(0, 9, (P,), generator=g, dtype=torch.int32)


def _worker(rank, world, port, P, M, F, V, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = GradientBucket(P, GradLayout(M, F), "cpu")
    for v in views_of_rank(V, rank, world):
        g, radii = _fake_view_grads(P, M, F, v)
        b.accumulate(g, radii)
    works = b.all_reduce(async_op=(rank % 2 == 0))  # both forms must interoperate
    for w in works:
        w.wait()
    if rank == 0:
        ret["flat"], ret["densify"], ret["max_radii"] = b.flat.clone(), b.densify.clone(), b.max_radii.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_views_of_rank_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(v for r in range(world) for v in views_of_rank(12, r, world))
        assert seen == list(range(12))


def test_two_rank_all_reduce_equals_single_process():
    P, M, F, V, world = 257, 1, 15, 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, P, M, F, V, ret), nprocs=world, join=True)
    ref = GradientBucket(P, GradLayout(M, F), "cpu")
    for v in range(V):
        g, radii = _fake_view_grads(P, M, F, v)
        ref.accumulate(g, radii)
    assert ref.layout.width == 3 + 3 + 1 + 3 + 4 + 15
    torch.testing.assert_close(ret["flat"], ref.flat, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ret["densify"], ref.densify, rtol=1e-6, atol=1e-6)
    assert torch.equal(ret["max_radii"], ref.max_radii)
    # the densification statistic is the sum of per-view norms, not the norm of the summed gradient
    summed = sum(_fake_view_grads(P, M, F, v)[0]["dL_dmeans2D"] for v in range(V))
    assert not torch.allclose(ref.densify[:, 0], summed[:, :2].norm(dim=-1))


def test_sharding_and_layout_properties():
    """Property tests of the host logic (hypothesis): views_of_rank is a partition with balanced shares for every
    (views, world); GradLayout's slices tile [0, width) in the documented order for every (M, F)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 64), st.integers(1, 16))
    def partition(V, world):
        shares = [list(views_of_rank(V, r, world)) for r in range(world)]
        assert sorted(v for s in shares for v in s) == list(range(V))
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
        assert all(v % world == r for r, s in enumerate(shares) for v in s)   # view v -> rank v mod world

    @settings(max_examples=100, deadline=None)
    @given(st.integers(0, 16), st.sampled_from([0, 3, 15, 16, 32]))
    def layout(M, F):
        lay = GradLayout(M, F)
        assert lay.width == 3 + 3 * M + 1 + 3 + 4 + F
        sl = lay.slices()
        names = [n for n, _ in lay.fields]
        assert names[0] == "means3D" and names[-1] == ("language" if F > 0 else names[-1])
        pos = 0
        for n in names:
            assert sl[n].start == pos
            pos = sl[n].stop
        assert pos == lay.width

    partition()
    layout()
