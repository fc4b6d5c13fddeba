"""The oracle's analytic backward routines against GRADIENT goldens that autograd produced through the reference's own
Python (tests/golden/make_golden_gradients.py -> gradients.npz; VERDICT round 3, missing #4).  CPU only.

  computeCov3D backward       CR/backward.cu:350-413   <- build_covariance_from_scaling_rotation (gaussian_model.py:119-124)
  computeColorFromSH backward CR/backward.cu:21-145    <- eval_sh + 0.5, clamp_min 0 of the view direction (sh_utils.py:55-126)
  projection Jacobian         CR/backward.cu:571-590, 640-646 <- [m,1] @ full_proj_transform / world_view_transform of Camera
The forward value pins are tests/test_oracle_golden.py and tests/test_oracle_geometry_golden (geometry.npz)."""
import os

import numpy as np
import torch

from online_lang_splatting_amd import scene as S
from parity_common import fwd_args

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "gradients.npz"))
RTOL = 1e-4  # the north-star tolerance, per element, relative to the tensor's largest magnitude for near-zero elements


def _close(got, ref, what):
    got, ref = torch.as_tensor(got).double(), torch.as_tensor(ref).double()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    bad = err > RTOL * ref.abs() + 1e-6 * scale
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements outside 1e-4, worst abs {err.max():.3e}"


def test_cov3d_backward_matches_autograd_through_the_reference(oracle):
    scales, rot, w = (torch.tensor(GOLD[k]) for k in ("cov_scales", "cov_rotations", "cov_cotangent"))
    for i in range(int(GOLD["cov_num"])):
        mod = float(GOLD[f"cov_modifier{i}"])
        ds, dq = oracle.cov3d_backward(scales, mod, rot, w)
        # What the golden makes visible: the CUDA reference returns dL / d(mod * scale) as the scale gradient — s = mod * scale
        # at CR/backward.cu:367, dL_dscale = dot(Rt[k], dL_dMt[k]) at :394-397, no factor `mod` — so with a scale modifier other
        # than 1 (only the GUI passes one, gui/slam_gui.py:588-605) it differs from autograd through the reference's own Python by
        # exactly that factor.  Parity follows the CUDA; the golden pins the routine up to the documented factor.
        _close(ds * mod, GOLD[f"cov_dL_dscales{i}"], f"mod * dL_dscales (modifier {mod})")
        if mod != 1.0:
            assert not torch.allclose(ds.double(), torch.tensor(GOLD[f"cov_dL_dscales{i}"]).double(), rtol=1e-3)
        # the reference's Python normalises the quaternion (general_utils.py:113-117), the kernel does not (CR/forward.cu:130):
        # for |q| = 1 autograd returns the kernel's gradient projected on the tangent space of the unit sphere
        q = rot.double()
        tangent = dq.double() - q * (q * dq.double()).sum(dim=1, keepdim=True)
        _close(tangent, GOLD[f"cov_dL_drotations_tangent{i}"], f"(I - q q^T) dL_drotations (modifier {mod})")
        assert float((q * dq.double()).sum(dim=1).abs().max()) > 1e-3  # (the projection removed something)


def _golden_scene(deg, F=0):
    """The golden's points in front of the golden's camera, as a rasterizer scene every Gaussian of which is visible."""
    W, H, fx, fy, cx, cy = GOLD["cam_spec"]
    cam = S.Camera(int(W), int(H), fx, fy, cx, cy, torch.tensor(GOLD["cam_R"]), torch.tensor(GOLD["cam_T"]))
    np.testing.assert_allclose(cam.world_view_transform.numpy(), GOLD["cam_viewmatrix"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), GOLD["cam_projmatrix"], rtol=1e-5, atol=1e-6)
    pts = torch.tensor(GOLD["points"])
    N = pts.shape[0]
    sc = S.make_scene(N, int(W), int(H), F, seed=3, max_sh_degree=deg, sh_degree=deg, camera=cam)
    sc.means3D = pts.contiguous()
    sc.scales = torch.full((N, 3), 0.02)
    sc.shs = torch.tensor(GOLD["sh_coeffs"])[:, : (deg + 1) ** 2, :].contiguous()
    return sc, N


def _chain(oracle, sc, N, dL_dmeans2D=None, dL_dcolors=None, dL_ddepths=None):
    """The oracle's per-Gaussian backward chain on the given composite-level gradients (the others zero)."""
    a = fwd_args(sc)
    R, color, radii, geom, binb, img, depth, opac, nt = oracle.rasterize_gaussians(*a)
    assert int((radii > 0).sum()) == N  # every golden point is visible
    z = torch.zeros
    comp = dict(dL_dmeans2D=dL_dmeans2D if dL_dmeans2D is not None else z(N, 3), dL_dconic=z(N, 2, 2),
                dL_dcolors=dL_dcolors if dL_dcolors is not None else z(N, 3),
                dL_ddepths=dL_ddepths if dL_ddepths is not None else z(N, 1))
    H, W = sc.camera.height, sc.camera.width
    b = [a[0], a[1], radii, a[2], None, a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], z(3, H, W), None,
         z(1, H, W), a[15], a[16], a[17], geom, R, binb, img, False]
    g = oracle.backward_chain(0, comp, *b)
    rgb = oracle.get_field(geom, "rgb").view(N, 3).clone()
    oracle.release(geom)
    return g, rgb


def test_sh_backward_matches_autograd_through_eval_sh(oracle):
    w = torch.tensor(GOLD["sh_cotangent"])
    for deg in range(4):
        sc, N = _golden_scene(deg)
        g, rgb = _chain(oracle, sc, N, dL_dcolors=w)
        _close(rgb, GOLD[f"sh_colors_deg{deg}"], f"colours, degree {deg}")   # (value pin on the same inputs)
        _close(g["dL_dsh"], GOLD[f"sh_dL_dsh_deg{deg}"], f"dL_dsh, degree {deg}")
        # the colour also depends on the mean through the view direction (nothing else in this chain run does)
        _close(g["dL_dmeans3D"], GOLD[f"sh_dL_dmeans_deg{deg}"], f"dL_dmeans3D via the view direction, degree {deg}")
        if deg > 0:
            assert float(GOLD[f"sh_clamped_fraction_deg{deg}"]) > 0.05  # the clamp mask is exercised
            clamped = torch.tensor(GOLD[f"sh_colors_deg{deg}"]) == 0
            assert float(g["dL_dsh"][:, 0, :][clamped].abs().max()) == 0.0


def test_projection_jacobian_matches_autograd(oracle):
    sc, N = _golden_scene(0)
    w_ndc, w_depth = torch.tensor(GOLD["proj_cotangent_ndc"]).float(), torch.tensor(GOLD["proj_cotangent_depth"]).float()
    d2 = torch.zeros(N, 3)
    d2[:, :2] = w_ndc   # the chain's dL_dmean2D is the cotangent of the NDC coordinates (CR/backward.cu:571-590)
    g, _ = _chain(oracle, sc, N, dL_dmeans2D=d2, dL_ddepths=w_depth.reshape(N, 1))
    _close(g["dL_dmeans3D"], GOLD["proj_dL_dmeans"], "dL_dmeans3D from dL_dmean2D and dL_ddepth")
    # ... and the forward values those gradients belong to
    a = fwd_args(sc)
    r = oracle.rasterize_gaussians(*a)
    W, H = sc.camera.width, sc.camera.height
    m2 = oracle.get_field(r[3], "means2D").view(N, 2).double()
    ndc = torch.tensor(GOLD["proj_ndc"])
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    assert float((m2 - pix).abs().max()) < 2e-3
    _close(oracle.get_field(r[3], "depths"), GOLD["proj_depth"], "depths")
    oracle.release(r[3])
