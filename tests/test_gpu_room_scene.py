"""A surface-structured SLAM map through the whole path: scene.make_room_scene builds ~500 k
Gaussians the way the reference's back end does (gaussian_splatting/scene/gaussian_model.py:180-281: depth back-projection
per keyframe, pcd_downsample 32 / 64, scales from distCUDA2 x point_size, identity rotations, opacity 0.5) from ray-cast
views of a closed box room, and is rendered from keyframe poses of the mapping window.  Unlike the i.i.d. volume of SURVEY
8(d) nothing saturates: nearly every visible Gaussian receives a gradient, lists are short and read to the end.

Checked here: the library's own distCUDA2 builds the map (bit-identical to the host k-d tree), the HIP path meets the oracle
on it — forward bit-identical, lists bit-identical, every gradient per element, the per-Gaussian chain on identical inputs
EXACT — at full config-3 size for two views and at a small size with both tiles and the exact backward, and the mapping
iteration on it reduces its loss."""
import json
import os

import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import knn_mean_dist2_host, make_room_scene
from test_gpu_parity import _check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(tag, log):
    out = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(out):
        return
    path = os.path.join(out, "parity_room_scene.json")
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[tag] = log
    json.dump(data, open(path, "w"), indent=1)


def test_the_library_knn_builds_the_same_map_as_the_host_kdtree(hip):
    a = make_room_scene(30_000, 400, 230, 15, views=3, seed=5).scene            # olsr_knn_mean_dist2 (a GPU is present)
    b = make_room_scene(30_000, 400, 230, 15, views=3, seed=5, knn=knn_mean_dist2_host).scene
    assert torch.equal(a.scales, b.scales) and torch.equal(a.means3D, b.means3D)


@pytest.mark.parametrize("tile,mode", [(15, _abi.BWD_REFERENCE), (16, _abi.BWD_REFERENCE), (15, _abi.BWD_EXACT)])
def test_small_room_against_the_oracle(hip, oracle, tile, mode):
    rs = make_room_scene(20_000, 320, 180, 15, views=4, seed=2)
    for v in (0, 3):
        _check(hip, oracle, rs.view(v), seed=11 + v, tile=tile, mode=mode, elementwise=True)


def test_rgb_only_room_against_the_oracle(hip, oracle):
    rs = make_room_scene(20_000, 320, 180, 0, views=2, seed=6)
    _check(hip, oracle, rs.view(1), seed=3, elementwise=True)


@pytest.mark.parametrize("view", [0, 7])
def test_full_size_room_against_the_oracle(hip, oracle, view):
    """~500 k Gaussians of 39 keyframes, 1200x680, F = 15: what BASELINE configs[3] looks like to the rasterizer."""
    rs = make_room_scene(500_000, 1200, 680, 15, views=10, seed=3)
    log = []
    _check(hip, oracle, rs.view(view), seed=3, elementwise=True, worst_bound=1e-2, log=log)
    _dump(f"room_view{view}", log)
    torch.cuda.empty_cache()


def test_mapping_iterations_on_the_room_fit_the_raycast_targets(hip):
    """12 views x (render from raw parameters + mapping loss incl. the 192x192 language target + backward into the bucket) +
    Adam, as BackEnd.map does (utils/slam_backend.py:499-670), on the fresh map: the loss falls."""
    from online_lang_splatting_amd.frame_shard import FrameLanes
    from online_lang_splatting_amd.slam_iterations import MappingStep
    W, H, F = 400, 230, 15
    rs = make_room_scene(40_000, W, H, F, views=6, seed=8)
    sc = rs.scene
    dev = torch.device(DEV)
    params = dict(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=torch.logit(sc.opacities).to(dev).contiguous(),
                  scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev), language=sc.language.to(dev))
    camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in rs.cameras]
    lanes = FrameLanes(2, sc.P, W, H, F, 1, 400_000, dev)
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    for fused in (True, False):
        p = {k: v.clone() for k, v in params.items()}
        st = MappingStep(lanes, p, sc.bg.to(dev), 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=fused)
        st.iteration()
        first = float(st.last_loss[0])
        for _ in range(25):
            st.iteration()
        last = float(st.last_loss[0])
        assert last < 0.9 * first, (fused, first, last)
        assert not any(ws.rendered()[1] or ws.backward_status()[1] for ws, _, _ in lanes.lanes)


def test_mapping_step_measures_its_loss_form_and_keeps_per_view_hints_across_a_sliding_window(hip):
    """MappingStep(fused_loss="auto") alternates the fused and the two-kernel loss in its iterations 3-6, keeps the faster and
    says so; the parameters it reaches are those of either fixed form (the cotangents are bit-identical).  view_ids key the
    tile-order hints: the keyframe window may grow, shrink and be reordered between iterations, and with more
    than eight lanes in a single process the Adam step still sums every lane's bucket."""
    from online_lang_splatting_amd.frame_shard import FrameLanes
    from online_lang_splatting_amd.slam_iterations import MappingStep
    W, H, F = 320, 180, 15
    rs = make_room_scene(20_000, W, H, F, views=6, seed=9)
    sc = rs.scene
    dev = torch.device(DEV)
    start = dict(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=torch.logit(sc.opacities).to(dev).contiguous(),
                 scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev), language=sc.language.to(dev))
    camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in rs.cameras]
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    out = {}
    for form in ("auto", True, False):
        lanes = FrameLanes(3, sc.P, W, H, F, 1, 300_000, dev)
        p = {k: v.clone() for k, v in start.items()}
        st = MappingStep(lanes, p, sc.bg.to(dev), 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=form)
        for _ in range(12):
            st.iteration()
            torch.cuda.synchronize()   # (lets the calibration's events complete between iterations)
        out[form] = ({k: v.clone() for k, v in p.items()}, st)
    cal = out["auto"][1].calibration
    assert cal is not None and cal["chosen"] in ("fused", "two_kernel") and out["auto"][1].auto is False
    assert len(cal["fused_ms"]) >= 2 and len(cal["two_kernel_ms"]) >= 2
    for k in start:   # the same parameters whichever form ran in which iteration
        assert torch.equal(out["auto"][0][k], out[True][0][k]) and torch.equal(out[True][0][k], out[False][0][k]), k
    # a sliding window: views keyed by id, reassigned / reordered / grown between iterations
    st = out[True][1]
    st.view_ids = [10, 11, 12, 13, 14, 15]
    st.iteration()
    st.cameras, st.targets, st.view_ids = camd[2:] + camd[:1], rs.targets[2:] + rs.targets[:1], [12, 13, 14, 15, 10]
    st.iteration()
    st.cameras, st.targets, st.view_ids = camd + camd[:2], rs.targets + rs.targets[:2], [10, 11, 12, 13, 14, 15, 16, 17]
    st.iteration()
    assert set(st.view_hints) >= {10, 11, 12, 13, 14, 15, 16, 17}
    total = st.summed_gradients()
    assert total.shape == (sc.P, 11 + 3 + F) and float(total.abs().max()) > 0
    # more lanes than the Adam kernel sums at once: same parameters as with three lanes
    lanes9 = FrameLanes(9, sc.P, W, H, F, 1, 300_000, dev)
    views9 = camd + camd[:3]
    tg9 = rs.targets + rs.targets[:3]
    res = {}
    for nl, lanes_ in ((9, lanes9), (3, FrameLanes(3, sc.P, W, H, F, 1, 300_000, dev))):
        p = {k: v.clone() for k, v in start.items()}
        st = MappingStep(lanes_, p, sc.bg.to(dev), 0, views9, tg9, lrs, exposure=torch.zeros(2, device=dev), fused_loss=True)
        tot = st.iteration()
        torch.cuda.synchronize()
        res[nl] = (p, tot.densify.clone(), tot.max_radii.clone())
    for k in start:   # (the sum over nine views associates differently with nine lanes than with three: to rounding)
        assert torch.allclose(res[9][0][k], res[3][0][k], rtol=1e-4, atol=1e-6), k
    # the densification statistics of the step (xyz_gradient_accum, denom, max_radii2D): every lane counted ONCE, also the ones
    # beyond the eight the Adam kernel sums (the fold of lanes 9+ used to add their statistics a second time)
    assert torch.equal(res[9][1][:, 1], res[3][1][:, 1])   # denom: an integer count per Gaussian
    assert torch.allclose(res[9][1][:, 0], res[3][1][:, 0], rtol=1e-5, atol=1e-9)
    assert torch.equal(res[9][2], res[3][2])


def test_tracking_loop_can_leave_the_final_images_behind(hip, oracle):
    """the fused tracking iteration writes no images, so ws.out is stale afterwards — and the
    reference's front end reads the last iteration's depth / opacity / render (utils/slam_frontend.py:664-665).
    iteration(write_images=True) and render_final() leave the images of the pose that was rendered; they equal a plain
    forward at that pose bit for bit."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
    W, H, F = 320, 180, 15
    rs = make_room_scene(20_000, W, H, F, views=3, seed=12)
    sc, cam = rs.scene, rs.cameras[0]
    dev = torch.device(DEV)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    T = torch.eye(4)
    T[:3, :3], T[:3, 3] = cam.R, cam.T
    T[0, 3] += 0.02
    pose = PoseState(T.to(dev), cam.projection_matrix.to(dev), cam.tanfovx, cam.tanfovy)
    ws = RasterWorkspace(sc.P, W, H, F, 1, 300_000, dev)
    loop = TrackingLoop(ws, g, 0, pose, rs.targets[0][0].to(dev), rs.targets[0][1].to(dev))
    ws.out["depth"].fill_(-7.0)
    for _ in range(4):
        loop.iteration()
    torch.cuda.synchronize()
    assert float(ws.out["depth"].max()) == -7.0                      # stale: the fused iteration wrote nothing
    cam_before = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in pose.camera().items()}
    loop.iteration(write_images=True)                                # renders at cam_before, then steps the pose
    got = {k: ws.out[k].clone() for k in ("color", "depth", "opacity", "language")}
    ws2 = RasterWorkspace(sc.P, W, H, F, 1, 300_000, dev)
    ws2.set_scene(sh_degree=0, **cam_before, **g)
    ref = ws2.forward()
    for k in got:
        assert torch.equal(got[k], ref[k]), k
    fin = loop.render_final()                                        # the images of the pose AFTER the last step
    ws2.set_scene(sh_degree=0, **pose.camera(), **g)
    ref = ws2.forward()
    for k in ("color", "depth", "opacity", "language"):
        assert torch.equal(fin[k], ref[k]), k


def _train_room(rs, dev, iterations, W, H, F, lanes_n=4, capacity=2_000_000):
    """`iterations` mapping iterations of the product on the fresh map (BackEnd.map's 150 per keyframe,
    configs/rgbd/replicav2/base_config.yaml:38): returns the ACTIVATED parameters as a CPU Scene factory."""
    from online_lang_splatting_amd.frame_shard import FrameLanes
    from online_lang_splatting_amd.scene import Scene
    from online_lang_splatting_amd.slam_iterations import MappingStep
    sc = rs.scene
    p = dict(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=torch.logit(sc.opacities).to(dev).contiguous(),
             scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev).clone(),
             language=None if F == 0 else sc.language.to(dev).clone())
    camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in rs.cameras]
    lanes = FrameLanes(lanes_n, sc.P, W, H, F, 1, capacity, dev)
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    st = MappingStep(lanes, p, sc.bg.to(dev), 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=True)
    st.iteration()
    first = float(st.last_loss[0])
    for _ in range(iterations - 1):
        st.iteration()
    last = float(st.last_loss[0])
    assert not any(ws.rendered()[1] or ws.backward_status()[1] for ws, _, _ in lanes.lanes)
    rot = p["rotations"] / p["rotations"].norm(dim=1, keepdim=True).clamp_min(1e-12)

    def view(v):
        return Scene(rs.cameras[v], p["means3D"].cpu(), torch.sigmoid(p["opacities"]).cpu().contiguous(),
                     torch.exp(p["scales"]).cpu().contiguous(), rot.cpu().contiguous(), p["shs"].cpu(),
                     None if F == 0 else p["language"].cpu(), 0, sc.bg, F)
    return view, first, last


def test_trained_room_against_the_oracle(hip, oracle):
    """The fresh map is what the back end builds at a keyframe; what it RENDERS most of the time has been through 150 mapping
    iterations (opacities pushed towards 0 and 1, scales grown over the gaps, anisotropic, rotations off identity, language
    codes off the unit sphere).  The product trains the full-size room map for 150 iterations against its ray-cast targets
    (its own MappingStep: the optimiser is not what is compared), and the result meets the oracle like every other scene:
    forward bit-identical, lists identical, every gradient per element, the chain on identical inputs exact."""
    dev = torch.device(DEV)
    W, H, F = 1200, 680, 15
    rs = make_room_scene(500_000, W, H, F, views=10, random_views=2, seed=3)
    view, first, last = _train_room(rs, dev, 150, W, H, F)
    assert last < 0.75 * first, (first, last)
    sc = view(4)
    iso = (sc.scales.max(dim=1).values / sc.scales.min(dim=1).values)
    assert float(iso.max()) > 1.05 and float((sc.opacities - 0.5).abs().max()) > 0.2     # it did leave the initial state
    log = []
    _check(hip, oracle, sc, seed=5, elementwise=True, worst_bound=1e-2, log=log)
    _dump("room_trained_150_view4", log)
    torch.cuda.empty_cache()


def test_trained_small_room_both_tiles(hip, oracle):
    dev = torch.device(DEV)
    W, H, F = 320, 180, 15
    rs = make_room_scene(20_000, W, H, F, views=6, seed=14)
    view, first, last = _train_room(rs, dev, 80, W, H, F, lanes_n=2, capacity=400_000)
    assert last < first
    for tile, mode in ((15, _abi.BWD_REFERENCE), (16, _abi.BWD_REFERENCE), (15, _abi.BWD_EXACT)):
        _check(hip, oracle, view(2), seed=6, tile=tile, mode=mode, elementwise=True)
