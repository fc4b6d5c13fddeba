"""olsr_grad_bucket.row_mask (include/olsr.h): a bucket that tracks which gradient rows may be non-zero is, after every call,
bit for bit the bucket that rewrites all P rows — over sequences of different views, overwrite / add mixes, frames without any
gradient, and writes to the storage from outside (announced with rows_unknown())."""
import pytest
import torch

from online_lang_splatting_amd.scene import default_camera, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cam(c, dev):
    return dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                tanfovy=c.tanfovy)


def _same(a, b):
    assert torch.equal(a.flat, b.flat)
    assert torch.equal(a.densify, b.densify) and torch.equal(a.max_radii, b.max_radii)


@pytest.mark.parametrize("P,F", [(30000, 15), (777, 0), (64 * 130 + 1, 3)])
def test_tracked_bucket_equals_the_dense_one(hip, P, F):
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket, RasterWorkspace
    dev = torch.device(DEV)
    W, H = 320, 240
    sc = make_scene(P, W, H, F, seed=11, max_sh_degree=1)
    M = sc.shs.shape[1]
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=None if F == 0 else sc.language.to(dev))
    cot = [None if t is None else t.to(dev) for t in sc.cotangents(2)]
    ws = RasterWorkspace(P, W, H, F, M, 1_500_000, dev)
    dense = GradientBucket(P, GradLayout(M, F), dev)
    tracked = GradientBucket(P, GradLayout(M, F), dev, track_rows=True)
    assert tracked.row_mask is not None and tracked.row_mask.numel() == (P + 63) // 64 and dense.row_mask is None
    views = [(0.0, 0.0), (0.0, 0.0), (6.0, 0.2), (-15.0, -0.6), (0.5, 0.0), (0.0, 500.0), (0.0, 0.0)]  # (500 m aside: nothing in view)
    firsts = [True, True, True, False, True, True, True]
    active = []
    for (yaw, tx), first in zip(views, firsts):
        ws.set_scene(sh_degree=sc.sh_degree, **_cam(default_camera(W, H, yaw, tx), dev), **g)
        ws.forward()
        for b in (dense, tracked):
            ws.backward(*cot, bucket=b, first=first, bucket_only=True)
        _same(tracked, dense)
        nz = (dense.flat != 0).any(dim=1)
        active.append(int(nz.sum()))
        # the mask covers every non-zero row
        bits = ((tracked.row_mask.view(-1, 1) >> torch.arange(64, device=dev)) & 1).reshape(-1)[:P].bool()
        assert bool((bits | ~nz).all())
    assert active[0] > 0 and active[5] == 0 and active[2] != active[0]
    # a write from outside, announced: the next overwrite clears it like the dense bucket's does
    for b in (dense, tracked):
        b.flat.fill_(7.0)
    tracked.rows_unknown()
    ws.set_scene(sh_degree=sc.sh_degree, **_cam(default_camera(W, H, 1.0, 0.0), dev), **g)
    ws.forward()
    for b in (dense, tracked):
        ws.backward(*cot, bucket=b, first=True, bucket_only=True)
    _same(tracked, dense)
    assert float(tracked.flat.max()) < 7.0
    # zero_() knows the rows are zero; accumulate() (the stand-alone kernel) does not track
    tracked.zero_()
    assert int(tracked.row_mask.abs().sum()) == 0


def test_frame_lanes_track_rows_and_sum(hip):
    """FrameShardedStep on lanes with tracked buckets (the default of FrameLanes) against untracked ones: same total."""
    from online_lang_splatting_amd.frame_shard import FrameLanes, FrameShardedStep
    dev = torch.device(DEV)
    W, H, F, P = 320, 240, 15, 20000
    sc = make_scene(P, W, H, F, seed=4)
    M = sc.shs.shape[1]
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    cot = [t.to(dev) for t in sc.cotangents(3)]
    totals = {}
    for track in (False, True):
        lanes = FrameLanes(2, P, W, H, F, M, 600_000, dev, track_rows=track)
        step = FrameShardedStep(lanes)
        outs = []
        for views in ([(0.0, 0.0), (3.0, 0.1), (-4.0, -0.1)], [(10.0, 0.3), (11.0, 0.3)], [(0.0, 0.0)]):
            cams = [_cam(default_camera(W, H, y, t), dev) for y, t in views]
            total = step.run(g, cams, lambda v, out: cot, sh_degree=sc.sh_degree)
            torch.cuda.synchronize()
            outs.append((total.flat.clone(), total.densify.clone(), total.max_radii.clone()))
        totals[track] = outs
    for a, b in zip(totals[False], totals[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_a_garbage_tile_order_hint_stays_in_bounds(hip):
    """The launch-order hint is caller memory.  Entries that are no tile ids are ignored (the block keeps its natural tile):
    an all-garbage hint renders the exact frame, and the forward leaves a valid order behind."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    W, H, F, P = 320, 240, 15, 20000
    sc = make_scene(P, W, H, F, seed=9)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 600_000, dev)
    ws.set_scene(sh_degree=sc.sh_degree, **_cam(sc.camera, dev), **g)
    ref = {k: v.clone() for k, v in ws.forward().items()}
    n = ws.tile_order.numel()
    for fill in (0x7FFFFFFF, -1, n, n + 12345):
        ws.tile_order.fill_(fill)
        out = ws.forward()
        for k in ref:
            assert torch.equal(out[k], ref[k]), (fill, k)
        assert sorted(ws.tile_order.tolist()) == list(range(n))   # the order this frame measured: a permutation again


@pytest.mark.parametrize("scene_kind,P,F,lanes_n", [("volume", 30000, 15, 3), ("room", 20000, 15, 4), ("volume", 64 * 9 + 5, 0, 2)])
def test_masked_adam_and_masked_lane_sum_equal_the_dense_ones(hip, scene_kind, P, F, lanes_n):
    """Round 5: FusedAdam does not read gradient rows a bucket's row mask proves zero
    (olsr_adam_step_masked) and the sum of the lane buckets moves only flagged rows (olsr_bucket_add).  The optimiser stays
    DENSE: parameters and both moments after several steps over changing views equal, bit for bit, those of the step that
    reads every row (which tests/test_gpu_api.py holds to torch.optim.Adam); the summed bucket equals torch's add."""
    from online_lang_splatting_amd.frame_shard import FrameLanes, FusedAdam, GradLayout, GradientBucket
    from online_lang_splatting_amd.scene import make_room_scene
    dev = torch.device(DEV)
    W, H = 320, 180
    if scene_kind == "room":
        rs = make_room_scene(P, W, H, F, views=6, seed=3)
        sc, cams = rs.scene, [_cam(c, dev) for c in rs.cameras]
    else:
        sc = make_scene(P, W, H, F, seed=21)
        cams = [_cam(default_camera(W, H, y, t), dev) for y, t in ((0.0, 0.0), (5.0, 0.1), (-7.0, -0.2), (0.0, 300.0), (2.0, 0.0), (9.0, 0.3))]
    M = sc.shs.shape[1]
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=None if F == 0 else sc.language.to(dev))
    cot = [None if t is None else t.to(dev) for t in sc.cotangents(5)]
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    lanes = FrameLanes(lanes_n, sc.P, W, H, F, M, 600_000, dev)   # buckets with row masks
    results = {}
    for masked in (True, False):
        params = dict(means3D=g["means3D"].clone(), shs=g["shs"].clone(), opacities=g["opacities"].clone(),
                      scales=g["scales"].clone(), rotations=g["rotations"].clone(),
                      language=None if F == 0 else g["language"].clone())
        adam = FusedAdam(sc.P, GradLayout(M, F), dev)
        adam.use_row_masks = masked
        for it in range(3):
            used = []
            for v in range(lanes_n + 1):                    # one lane gets two views: an overwrite and an add
                ws, bucket, _ = lanes.lanes[v % lanes_n]
                first = bucket not in used
                if first:
                    used.append(bucket)
                ws.set_scene(sh_degree=sc.sh_degree, **cams[(it + v) % len(cams)], **g)
                ws.forward()
                ws.backward(*cot, bucket=bucket, first=first, bucket_only=True)
            adam.step(used, params, lrs)
            if it == 1:
                adam.step(used[0], params, lrs)             # the single-bucket form takes the masked path too
        torch.cuda.synchronize()
        results[masked] = (params, adam.exp_avg.clone(), adam.exp_avg_sq.clone())
    for k, t in results[True][0].items():
        if t is not None:
            assert torch.equal(t, results[False][0][k]), k
    assert torch.equal(results[True][1], results[False][1]) and torch.equal(results[True][2], results[False][2])
    assert float((results[True][0]["means3D"] - g["means3D"]).abs().max()) > 0
    # the lane sum: olsr_bucket_add against torch's dense add, masks merged
    total, others = lanes.lanes[0][1], [b for _, b, _ in lanes.lanes[1:]]
    ref = total.sum_storage.clone()
    ref_r = total.max_radii.clone()
    for b in others:
        ref.add_(b.sum_storage)
        torch.maximum(ref_r, b.max_radii, out=ref_r)
        total.add_bucket(b)
    torch.cuda.synchronize()
    assert torch.equal(total.sum_storage, ref) and torch.equal(total.max_radii, ref_r)
    nz = (total.flat != 0).any(dim=1)
    bits = ((total.row_mask.view(-1, 1) >> torch.arange(64, device=dev)) & 1).reshape(-1)[:sc.P].bool()
    assert bool((bits | ~nz).all())
    # a source without a mask is added densely and leaves the destination's mask "unknown"
    plain = GradientBucket(sc.P, GradLayout(M, F), dev)
    plain.flat.fill_(0.5)
    before = total.flat.clone()
    total.add_bucket(plain)
    assert torch.equal(total.flat, before + 0.5) and int((total.row_mask != -1).sum()) == 0
