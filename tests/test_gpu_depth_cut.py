"""Per-tile depth cut-offs (include/olsr.h, scene->tile_depth_cut; csrc/k_preprocess.hip, k_binning.hip, k_render_fwd.hip):
the contract is "status 0 => the frame is the one without cut-offs, bit for bit; otherwise OLSR_STATUS_CUT_MISS, zero
gradients, no pose step, and the offending tiles are uncut again".  Every case compares a RasterWorkspace(depth_cut=True)
with a plain one on the same inputs; the plain one is what tests/test_gpu_parity.py holds against the oracle."""
import numpy as np
import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import default_camera, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CUT_MISS = 3


def _cam(c, dev):
    return dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                tanfovy=c.tanfovy)


def _setup(P=40000, W=320, H=240, F=15, seed=5):
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(P, W, H, F, seed=seed)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    cot = [t.to(dev) for t in sc.cotangents(1)]
    plain = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 2_000_000, dev)
    cut = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 2_000_000, dev, depth_cut=True)
    return sc, g, cot, plain, cut, dev


def _frame(ws, sc, cam, g, cot):
    ws.set_scene(sh_degree=sc.sh_degree, **cam, **g)
    out = {k: v.clone() for k, v in ws.forward().items()}
    grads = {k: v.clone() for k, v in ws.backward(*cot).items()}
    return out, grads


def _assert_same_frame(out, grads, out_ref, grads_ref):
    for k in out_ref:
        assert torch.equal(out[k], out_ref[k]), k  # images, radii, n_touched: bit for bit
    for k in grads_ref:
        # the per-Gaussian sums group their rows by instance index, which the cut-offs shift: last-bit differences on a
        # handful of elements, far inside the backward's 1e-4 contract
        scale = float(grads_ref[k].abs().max())
        err = (grads[k] - grads_ref[k]).abs()
        assert bool((err <= 1e-5 * grads_ref[k].abs() + 1e-7 * scale).all()), (k, float(err.max()), scale)


def test_same_view_is_exact_and_its_lists_shrink(hip):
    sc, g, cot, plain, cut, dev = _setup()
    cam = _cam(sc.camera, dev)
    out_ref, grads_ref = _frame(plain, sc, cam, g, cot)
    R_full = plain.rendered()[0]
    assert bool(torch.isinf(cut.depth_cut).all())
    Rs = []
    for it in range(3):
        out, grads = _frame(cut, sc, cam, g, cot)
        assert cut.forward_status() == 0
        _assert_same_frame(out, grads, out_ref, grads_ref)
        Rs.append(cut.rendered()[0])
    assert Rs[0] == R_full            # the first frame had no cut-offs
    assert Rs[1] == Rs[2] < 0.5 * R_full, (Rs, R_full)
    # tiles whose every pixel saturated carry a finite cut-off; it lies behind the depth the tile stopped at
    finite = torch.isfinite(cut.depth_cut)
    assert int(finite.sum()) > 0.5 * cut.depth_cut.numel()
    assert float(cut.depth_cut[finite].min()) > 0.01


def test_a_cut_that_hides_contributions_is_reported_and_heals(hip):
    sc, g, cot, plain, cut, dev = _setup()
    cam = _cam(sc.camera, dev)
    out_ref, grads_ref = _frame(plain, sc, cam, g, cot)
    _frame(cut, sc, cam, g, cot)
    assert cut.forward_status() == 0
    good = cut.depth_cut.clone()
    # shrink every cut-off to a third of the stop depth: lists end before their tiles saturate
    cut.depth_cut.copy_(torch.where(torch.isfinite(good), (good - 0.01) / 1.1 * 0.33, good))
    out, grads = _frame(cut, sc, cam, g, cot)
    assert cut.forward_status() == CUT_MISS
    assert int(cut.bwd_status.cpu()[1]) == CUT_MISS
    for k, v in grads.items():
        assert float(v.abs().max()) == 0.0, k          # a frame that missed hands out no gradient
    assert not torch.equal(out["color"], out_ref["color"])  # (it really was a different image)
    # the offending tiles are uncut again, the others keep a cut-off: the next frame is exact without any host action
    healed = cut.depth_cut.clone()
    assert int(torch.isinf(healed).sum()) > int(torch.isinf(good).sum())
    out, grads = _frame(cut, sc, cam, g, cot)
    assert cut.forward_status() == 0
    _assert_same_frame(out, grads, out_ref, grads_ref)
    # reset_depth_cut: back to the plain path
    cut.reset_depth_cut()
    out, grads = _frame(cut, sc, cam, g, cot)
    assert cut.forward_status() == 0 and cut.rendered()[0] == plain.rendered()[0]
    _assert_same_frame(out, grads, out_ref, grads_ref)


def test_status_zero_means_exact_on_other_views(hip):
    """Cut-offs left by one view, frames of others: whatever the verdict, 0 must mean bit-identical."""
    sc, g, cot, plain, cut, dev = _setup()
    W, H = sc.camera.width, sc.camera.height
    views = [default_camera(W, H, yaw, tx) for yaw, tx in ((0.0, 0.0), (0.5, 0.01), (2.0, 0.05), (8.0, 0.3), (-12.0, -0.45),
                                                            (0.0, 0.0))]
    verdicts = []
    for v in views:
        cam = _cam(v, dev)
        out_ref, grads_ref = _frame(plain, sc, cam, g, cot)
        out, grads = _frame(cut, sc, cam, g, cot)
        st = cut.forward_status()
        verdicts.append(st)
        assert st in (0, CUT_MISS)
        if st == 0:
            _assert_same_frame(out, grads, out_ref, grads_ref)
        else:
            out, grads = _frame(cut, sc, cam, g, cot)     # the retry sees the offending tiles uncut
            if cut.forward_status() == 0:
                _assert_same_frame(out, grads, out_ref, grads_ref)
    assert verdicts[0] == 0


def test_gated_pose_step_leaves_the_state_alone(hip):
    from online_lang_splatting_amd.slam_iterations import PoseState
    dev = torch.device(DEV)
    cam = default_camera(320, 240)
    ps = PoseState(torch.eye(4, device=dev), cam.projection_matrix.to(dev), cam.tanfovx, cam.tanfovy,
                   device_step_count=True)
    grad = torch.tensor([0.3, -0.2, 0.1, 0.05, -0.04, 0.02], device=dev)
    dexp = torch.tensor([0.1, -0.1], device=dev)
    ok = torch.tensor([123, 0], dtype=torch.int32, device=dev)
    ps.step(grad, dexp, frame_status=ok)
    state1, status1 = ps.state.clone(), ps.status.clone()
    assert int(status1[1]) == 1
    for word in (1, 2, CUT_MISS):
        ps.step(grad, dexp, frame_status=torch.tensor([123, word], dtype=torch.int32, device=dev))
        assert torch.equal(ps.state, state1) and int(ps.status[1]) == 1 and int(ps.status[0]) == 0
    ps.step(grad, dexp, frame_status=ok)
    assert int(ps.status[1]) == 2 and not torch.equal(ps.state[:16], state1[:16])
    host_counted = PoseState(torch.eye(4, device=dev), cam.projection_matrix.to(dev), cam.tanfovx, cam.tanfovy)
    with pytest.raises(ValueError, match="device_step_count"):
        host_counted.step(grad, dexp, frame_status=ok)


def test_tracking_loop_with_cut_offs_walks_the_same_poses(hip):
    """TrackingLoop on a depth_cut workspace against the plain loop from the same perturbed pose: the k-th COUNTED step of
    the one is the k-th step of the other (an iteration whose frame missed is a device-side no-op)."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
    from oracle.pose_oracle import se3_exp
    dev = torch.device(DEV)
    W, H, F = 320, 240, 15
    sc = make_scene(20000, W, H, F, seed=21)
    cam = default_camera(W, H)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    proj = cam.projection_matrix.to(dev)
    T_gt = torch.eye(4, device=dev)
    T0 = torch.from_numpy(se3_exp(np.array([0.02, -0.015, 0.01, 0.004, -0.006, 0.003], dtype=np.float32))).to(dev) @ T_gt
    STEPS = 40
    traj, iters, Rsum = {}, {}, {}
    for name in ("plain", "cut"):
        ws = RasterWorkspace(sc.P, W, H, F, sc.shs.shape[1], 2_000_000, dev, depth_cut=(name == "cut"))
        ps = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy, optimise_exposure=False, device_step_count=True)
        ws.set_scene(sh_degree=sc.sh_degree, **ps.camera(), **g)
        out = ws.forward()
        gt_image, gt_depth = out["color"].clone(), out["depth"][0].clone()
        ps.reset(T0)
        loop = TrackingLoop(ws, g, sc.sh_degree, ps, gt_image, gt_depth)
        poses, n, Rs = [], 0, 0
        while len(poses) < STEPS:
            before = loop.steps_done()
            loop.iteration()
            n += 1
            Rs += ws.rendered()[0]
            if loop.steps_done() > before:
                poses.append(ps.T_w2c.clone())
            assert n < 3 * STEPS, "the cut-offs miss more often than they hit"
        traj[name], iters[name], Rsum[name] = torch.stack(poses), n, Rs / n
    assert iters["plain"] == STEPS
    # (the cut loop's gradients differ from the plain one's in the last bit of a few elements; Adam's first steps are sign-like
    #  in the gradient, so the poses agree far below a step of 1e-3)
    assert float((traj["cut"] - traj["plain"]).abs().max()) < 2e-5, float((traj["cut"] - traj["plain"]).abs().max())
    assert iters["cut"] <= STEPS + STEPS // 4, iters     # few iterations are lost to misses
    assert Rsum["cut"] < 0.7 * Rsum["plain"], Rsum       # and the lists really are shorter
    print("tracking with cut-offs:", iters, Rsum)


def test_combinations_without_a_cut_off_kernel_are_refused(hip):
    """The cut-off bookkeeping exists for the default accumulation, with images or the tracking loss; the mapping loss and the
    alternative accumulations answer OLSR_ERR_ARG instead of rendering a frame whose lists were cut and never checked."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    sc, g, cot, plain, cut, dev = _setup(P=3000, W=160, H=120)
    cam = _cam(sc.camera, dev)
    H, W = sc.camera.height, sc.camera.width
    gt_image, gt_depth = torch.rand(3, H, W, device=dev), torch.rand(H, W, device=dev) + 0.5
    gt_lang = torch.rand(sc.F, 24, 32, device=dev)
    cut.set_scene(sh_degree=sc.sh_degree, **cam, **g)
    with pytest.raises(RuntimeError, match="tile_depth_cut"):
        cut.forward_loss(gt_image, gt_depth, gt_lang, None, None, tracking=False)
    lo = cut.forward_loss(gt_image, gt_depth, None, None, None, tracking=True, skip_images=True)   # the tracking loss is fine
    assert cut.forward_status() == 0 and float(lo["loss"][0]) > 0
    weight = RasterWorkspace(sc.P, W, H, sc.F, sc.shs.shape[1], 500_000, dev, flags=_abi.FLAG_FWD_ACCUM_WEIGHT,
                             depth_cut=True)
    weight.set_scene(sh_degree=sc.sh_degree, **cam, **g)
    with pytest.raises(RuntimeError, match="tile_depth_cut"):
        weight.forward()


def test_tracking_loop_run_takes_the_requested_number_of_steps(hip):
    """with depth cut-offs an iteration whose frame missed is a device-side no-op, so a fixed budget of
    iterations takes fewer optimiser steps than the reference's tracking_itr_num.  TrackingLoop.run(steps) iterates until
    that many steps were TAKEN (the count is read back every few iterations); without cut-offs it is `steps` iterations."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.scene import make_scene
    from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
    dev = torch.device("cuda:0")
    W, H, F, P = 320, 240, 15, 30000
    sc = make_scene(P, W, H, F, seed=31)
    cam = sc.camera
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    ws0 = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 1_200_000, dev)
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy)
    ws0.set_scene(sh_degree=sc.sh_degree, **c, **g)
    o = ws0.forward()
    gt_image, gt_depth = o["color"].clone(), o["depth"][0].clone()
    T0 = torch.eye(4)
    T0[0, 3], T0[1, 3] = 0.03, -0.02
    for cut in (True, False):
        ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 1_200_000, dev, depth_cut=cut)
        pose = PoseState(T0.to(dev), cam.projection_matrix.to(dev), cam.tanfovx, cam.tanfovy, device_step_count=True)
        loop = TrackingLoop(ws, g, sc.sh_degree, pose, gt_image, gt_depth)
        issued = loop.run(25, check_every=8)
        assert loop.steps_done() == 25 and issued >= 25, (cut, issued, loop.steps_done())
        if not cut:
            assert issued == 25
        fin = loop.run(3, write_final_images=True)
        assert loop.steps_done() == 28 and fin >= 3
        assert float(ws.out["opacity"].max()) > 0      # the final images were left behind
