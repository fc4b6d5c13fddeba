"""The oracle against the golden vectors generated from the reference's own Python helpers
(tests/golden/make_golden.py -> conventions.npz), plus internal known-answer checks.
CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from online_lang_splatting_amd import scene as S
from parity_common import fwd_args

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "conventions.npz"))


def test_camera_conventions_match_reference_helpers():
    """scene.Camera must hand the rasterizer exactly what utils/camera_utils.Camera does."""
    for i in range(int(GOLD["num_cams"])):
        W, H, fx, fy, cx, cy = GOLD[f"cam{i}_spec"]
        cam = S.Camera(int(W), int(H), fx, fy, cx, cy, torch.tensor(GOLD[f"cam{i}_R"]), torch.tensor(GOLD[f"cam{i}_T"]))
        np.testing.assert_allclose(cam.world_view_transform.numpy(), GOLD[f"cam{i}_viewmatrix"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(cam.projection_matrix.numpy(), GOLD[f"cam{i}_projmatrix_raw"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), GOLD[f"cam{i}_projmatrix"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), GOLD[f"cam{i}_campos"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose([cam.tanfovx, cam.tanfovy], GOLD[f"cam{i}_tanfov"], rtol=1e-6)
        # the kernel-side reads of CR/backward.cu:596-600
        pr = cam.projection_matrix.reshape(-1)
        assert abs(pr[0].item() - 2 * fx / W) < 1e-6 and abs(pr[5].item() - 2 * fy / H) < 1e-6 and pr[11].item() == 1.0


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_matches_reference_eval_sh(oracle, deg):
    """computeColorFromSH (CR/forward.cu:23-74) as restated by the oracle == eval_sh(+0.5, clamp)."""
    dirs = torch.tensor(GOLD["sh_dirs"])
    sh = torch.tensor(GOLD["sh_coeffs"])
    P = dirs.shape[0]
    # camera at the origin looking down +z; put Gaussian i at 2*dir_i rotated into view:
    # direction = mean - campos = mean, so choose means = dirs * 2 with z forced positive by flipping
    # the camera (we only need `rgb`, which preprocess writes for every Gaussian that survives culling).
    flip = torch.where(dirs[:, 2:3] > 0, 1.0, -1.0)
    means = dirs * 2.0
    sc = S.make_scene(P, 64, 64, 0, seed=0, max_sh_degree=3, sh_degree=deg)
    got = torch.zeros(P, 3)
    seen = torch.zeros(P, dtype=torch.bool)
    for sign in (1.0, -1.0):
        # a camera looking down +z (sign=1) or -z (sign=-1), both centred at the origin
        R = torch.diag(torch.tensor([sign, 1.0, sign]))
        cam = S.Camera(64, 64, 8.0, 8.0, 31.5, 31.5, R, torch.zeros(3))  # very wide field of view
        sc.camera, sc.means3D, sc.shs = cam, means.contiguous(), sh.contiguous()
        sc.scales = torch.full((P, 3), 0.01)
        oracle.TILE = 15
        r = oracle.rasterize_gaussians(*fwd_args(sc))
        radii, geom = r[2], r[3]
        rgb = oracle.get_field(geom, "rgb").view(P, 3)
        vis = radii > 0
        got[vis] = rgb[vis]
        seen |= vis
        oracle.release(geom)
    assert seen.float().mean() > 0.5
    ref = torch.tensor(GOLD[f"sh_rgb_deg{deg}"])
    torch.testing.assert_close(got[seen], ref[seen], rtol=1e-5, atol=1e-6)
    del flip


def test_se3_parametrisation_matches_reference():
    """dense_ref's Exp(tau) (tau = [rho | theta]) agrees with utils/pose_utils.SE3_exp to 2nd order."""
    from dense_ref import se3_exp_first_order
    taus, ref = GOLD["se3_tau"], GOLD["se3_exp"]
    for t, T in zip(taus, ref):
        mine = se3_exp_first_order(torch.tensor(t, dtype=torch.float64)).numpy()
        # the series is cut after the 2nd-order term: error <= |tau|^3 / 6 (+ the reference's fp32 noise)
        np.testing.assert_allclose(mine, T, atol=float(np.linalg.norm(t)) ** 3 / 4 + 2e-6)


def test_pinned_expf_accuracy(oracle):
    worst = 0.0
    for x in np.linspace(-87.0, 0.0, 20001):
        got = oracle.expf(float(np.float32(x)))
        ref = math.exp(float(np.float32(x)))
        worst = max(worst, abs(got - ref) / ref)
    assert worst < 2.5e-7, worst  # ~2 ulp
    assert oracle.expf(0.0) == 1.0


def test_reference_tree_survivors_closed_form():
    """render_cuda_reduce_sum over 225 lanes (CR/backward.cu:691-702): which ranks reach element 0."""
    def survivors(n):
        contrib = [{i} for i in range(n)]
        i = n // 2
        while i > 0:
            for lane in range(i):
                contrib[lane] = contrib[lane] | contrib[lane + i]
            i //= 2
        return contrib[0]
    s225 = survivors(225)
    assert len(s225) == 128
    assert s225 == {r for r in range(224) if r % 7 in (0, 1, 3, 4)}
    assert survivors(256) == set(range(256))


def test_oracle_empty_and_culled(oracle):
    sc = S.make_scene(0, 32, 32, 15, seed=0)
    sc.language = torch.zeros(0, 15)
    r = oracle.rasterize_language_gaussians(*fwd_args(sc))
    assert r[0] == 0 and float(r[1].abs().max()) == 0.0
    oracle.release(r[4])
    # everything behind the near plane
    sc = S.make_scene(50, 32, 32, 15, seed=1)
    sc.means3D[:, 2] = -1.0
    r = oracle.rasterize_language_gaussians(*fwd_args(sc))
    assert r[0] == 0 and int((r[3] > 0).sum()) == 0
    oracle.release(r[4])


def test_oracle_sort_order_and_ranges(oracle):
    sc = S.make_scene(3000, 150, 90, 0, seed=5)
    r = oracle.rasterize_gaussians(*fwd_args(sc))
    geom = r[3]
    keys = oracle.get_field(geom, "keys")
    assert bool((keys[1:] >= keys[:-1]).all())
    pl = oracle.get_field(geom, "point_list").long()
    tiles = keys >> 32
    same = (keys[1:] == keys[:-1])
    assert bool((pl[1:][same] > pl[:-1][same]).all())  # ties broken by Gaussian index (stable sort)
    rg = oracle.get_field(geom, "ranges").view(-1, 2).long()
    lens = rg[:, 1] - rg[:, 0]
    assert int(lens.sum()) == r[0]
    counts = torch.bincount(tiles, minlength=rg.shape[0])
    assert torch.equal(counts, lens)
    oracle.release(geom)


GEO = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry.npz"))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_cov3d_matches_reference_build_scaling_rotation(oracle, k):
    """computeCov3D (CR/forward.cu:121-155) as restated by the oracle == strip_symmetric(L L^T) with
    L = build_scaling_rotation(modifier * scaling, rotation) of the reference's general_utils.py:97-149 (the
    composition of GaussianModel.build_covariance_from_scaling_rotation), for three scale modifiers."""
    scales, rot = torch.tensor(GEO["cov_scales"]), torch.tensor(GEO["cov_rotations"])
    P = scales.shape[0]
    sc = S.make_scene(P, 96, 96, 0, seed=3)
    # every Gaussian in front of the camera, inside the frustum, so preprocess reaches computeCov3D for all of them
    g = torch.Generator().manual_seed(1)
    z = torch.rand(P, generator=g) * 3.0 + 1.0
    sc.means3D = torch.stack([(torch.rand(P, generator=g) - 0.5) * z, (torch.rand(P, generator=g) - 0.5) * z, z], 1).contiguous()
    sc.scales, sc.rotations = scales.contiguous(), rot.contiguous()
    oracle.TILE = 15
    r = oracle.rasterize_gaussians(*fwd_args(sc, scale_modifier=float(GEO[f"cov3D_modifier{k}"])))
    radii, geom = r[2], r[3]
    cov = oracle.get_field(geom, "cov3D").view(P, 6)
    oracle.release(geom)
    vis = radii > 0
    assert int(vis.sum()) > P // 2
    ref = torch.tensor(GEO[f"cov3D_mod{k}"])
    # off-diagonal entries cancel: compare relative to the covariance's own scale (its largest entry)
    scale = ref[vis].abs().max(dim=1, keepdim=True).values
    assert float(((cov[vis] - ref[vis]).abs() / scale).max()) <= 2e-6


def test_oracle_projection_matches_reference_camera_matrices(oracle):
    """means2D / depths of the oracle's preprocess (CR/forward.cu:300-303,343; ndc2Pix CR/auxiliary.h:41-44) == the
    points multiplied, in float64, with the matrices the reference's Camera hands to the rasterizer."""
    for i in range(int(GEO["num_proj"])):
        W, H, fx, fy, cx, cy = GEO[f"proj{i}_spec"]
        cam = S.Camera(int(W), int(H), fx, fy, cx, cy, torch.tensor(GEO[f"proj{i}_R"]), torch.tensor(GEO[f"proj{i}_T"]))
        np.testing.assert_allclose(cam.world_view_transform.numpy(), GEO[f"proj{i}_viewmatrix"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), GEO[f"proj{i}_projmatrix"], rtol=1e-5, atol=1e-6)
        pts = torch.tensor(GEO[f"proj{i}_points"])
        N = pts.shape[0]
        sc = S.make_scene(N, int(W), int(H), 0, seed=4, camera=cam)
        sc.means3D = pts.contiguous()
        sc.scales = torch.full((N, 3), 0.02)
        oracle.TILE = 15
        r = oracle.rasterize_gaussians(*fwd_args(sc))
        radii, geom = r[2], r[3]
        m2d = oracle.get_field(geom, "means2D").view(N, 2).double()
        dep = oracle.get_field(geom, "depths").double()
        oracle.release(geom)
        vis = radii > 0
        assert int(vis.sum()) > N // 2
        # fp32 matrix products against float64: a few ulp of the largest term (pixels up to ~2000, depths up to ~6)
        np.testing.assert_allclose(m2d[vis].numpy(), GEO[f"proj{i}_pix"][vis.numpy()], rtol=0, atol=2e-3)
        np.testing.assert_allclose(dep[vis].numpy(), GEO[f"proj{i}_depth"][vis.numpy()], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_renderer_eval_sh_matches_reference(deg):
    """online_lang_splatting_amd.renderer.eval_sh (the convert_SHs_python branch of render()) == the reference's
    gaussian_splatting/utils/sh_utils.eval_sh on the golden directions / coefficients."""
    from online_lang_splatting_amd.renderer import eval_sh
    dirs, sh = torch.tensor(GOLD["sh_dirs"]), torch.tensor(GOLD["sh_coeffs"])
    got = eval_sh(deg, sh.transpose(1, 2), dirs)
    torch.testing.assert_close(got, torch.tensor(GOLD[f"sh_raw_deg{deg}"]), rtol=1e-5, atol=1e-6)
