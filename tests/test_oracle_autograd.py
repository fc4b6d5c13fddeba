"""Pins the oracle's `exact` backward (and its forward) against PyTorch autograd on an
independent dense formulation (tests/dense_ref.py).  CPU only, small scenes."""
import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import default_camera, make_scene, projection_matrix2, world2view2
from dense_ref import render_dense
from parity_common import run_backend


def _dense(sc, tile, tau_grad=True):
    cam, W, H = sc.camera, sc.camera.width, sc.camera.height
    D = torch.float64

    def leaf(t):
        return t.double().clone().requires_grad_(True)
    m, op, s, r, sh = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.scales), leaf(sc.rotations), leaf(sc.shs)
    lg = leaf(sc.language) if sc.F > 0 else None
    tau = torch.zeros(6, dtype=D, requires_grad=True)
    W2C = world2view2(cam.R, cam.T).double()
    Pm = projection_matrix2(cam.znear, cam.zfar, cam.cx, cam.cy, cam.fx, cam.fy, W, H).double()
    out = render_dense(m, op, s, r, sh, None, lg, W2C, Pm, tau, width=W, height=H, tile=tile,
                       sh_degree=sc.sh_degree, bg=sc.bg)
    return out, dict(means3D=m, opacity=op, scales=s, rotations=r, sh=sh, language=lg, tau=tau)


def _close(a, b, rtol=2e-4, what=""):
    a, b = a.double(), b.double()
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("P,W,H,F,deg,tile,seed,bgv,yaw,center", [
    (300, 45, 30, 15, 0, 15, 1, 0.0, 0.0, True),
    (300, 45, 30, 15, 3, 15, 2, 0.7, 0.0, False),
    (400, 48, 32, 3, 1, 16, 3, 0.0, 0.0, False),
    (300, 45, 30, 15, 0, 15, 1, 0.3, 3.0, True),
    (250, 40, 40, 0, 2, 15, 4, 0.2, 7.0, False),
])
def test_exact_backward_matches_autograd(oracle, P, W, H, F, deg, tile, seed, bgv, yaw, center):
    cam = default_camera(W, H, yaw, 0.1 if yaw else 0.0)
    if center:
        cam.cx, cam.cy = W / 2.0, H / 2.0
    sc = make_scene(P, W, H, F, seed=seed, max_sh_degree=deg, bg=torch.tensor([bgv, 0.5 * bgv, 0.2 * bgv]), camera=cam)
    fo, go = run_backend(oracle, sc, None, seed, tile, _abi.BWD_EXACT)
    out, leaves = _dense(sc, tile)
    _close(fo["color"], out["color"], 1e-5, "color")
    _close(fo["depth"], out["depth"], 1e-5, "depth")
    _close(fo["opacity"], out["opacity"], 1e-5, "opacity")
    if F > 0:
        _close(fo["language"], out["language"], 1e-5, "language")
    assert torch.equal(fo["radii"], out["radii"])
    assert torch.equal(fo["n_touched"].long(), out["n_touched"].long())
    dc, dl, dd = sc.cotangents(seed)
    loss = (out["color"] * dc.double()).sum() + (out["depth"] * dd.double()).sum()
    if F > 0:
        loss = loss + (out["language"] * dl.double()).sum()
    loss.backward()
    _close(go["dL_dmeans3D"], leaves["means3D"].grad, what="dL_dmeans3D")
    _close(go["dL_dopacity"], leaves["opacity"].grad, what="dL_dopacity")
    _close(go["dL_dscales"], leaves["scales"].grad, what="dL_dscales")
    _close(go["dL_drotations"], leaves["rotations"].grad, what="dL_drotations")
    _close(go["dL_dsh"], leaves["sh"].grad, what="dL_dsh")
    if F > 0:
        _close(go["dL_dlanguage"], leaves["language"].grad, what="dL_dlanguage")
    # The reference's pose Jacobian drops the principal-point term of the projection and uses the
    # frustum-clamped mean in the rotation part (CR/backward.cu:278,596-611) and its SH term is not
    # a true derivative; it is exact for degree 0, a centred principal point and no clamped Gaussian.
    if deg == 0 and center and abs(yaw) < 4.0:
        _close(go["dL_dtau"].sum(0), leaves["tau"].grad, what="dL_dtau")
    oracle.release(fo["geom"])


def test_reference_mode_differs_only_where_documented(oracle):
    """reference vs exact on the same scene: identical forward; colour/mean gradients shrink (128 of
    225 ranks) and language gradients come from the tile's rank-0 pixel only."""
    sc = make_scene(400, 60, 45, 15, seed=9)
    fr, gr = run_backend(oracle, sc, None, 9, 15, _abi.BWD_REFERENCE)
    fe, ge = run_backend(oracle, sc, None, 9, 15, _abi.BWD_EXACT)
    for k in ("color", "language", "depth", "opacity"):
        assert torch.equal(fr[k], fe[k])
    assert not torch.allclose(gr["dL_dlanguage"], ge["dL_dlanguage"])
    assert gr["dL_dlanguage"].abs().sum() < ge["dL_dlanguage"].abs().sum()
    # with 16x16 tiles the reduction tree is exact: only the language quirks remain
    fr16, gr16 = run_backend(oracle, sc, None, 9, 16, _abi.BWD_REFERENCE)
    fe16, ge16 = run_backend(oracle, sc, None, 9, 16, _abi.BWD_EXACT)
    assert torch.equal(fr16["color"], fe16["color"])
    for f in (fr, fe, fr16, fe16):
        oracle.release(f["geom"])


@pytest.mark.parametrize("F,mode", [(15, _abi.BWD_REFERENCE), (0, _abi.BWD_EXACT)])
def test_chain_replay_reproduces_the_full_backward(oracle, F, mode):
    """oracle_backward_chain (the per-Gaussian half alone, on caller-provided composite-level gradients) fed with the
    oracle's own composite-level gradients gives the full backward's outputs bit for bit; fed with perturbed ones it
    moves — the GPU suite replays it on the product's gradients (tests/test_gpu_parity.py::_check, chain=True)."""
    sc = make_scene(900, 96, 64, F, seed=5)
    fo, go = run_backend(oracle, sc, None, 2, 15, mode)
    comp = {k: go[k] for k in ("dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_ddepths")}
    ch = oracle.backward_chain(F, comp, *fo["bwd_args"])
    assert set(ch) == set(oracle.CHAIN_KEYS)
    for k in oracle.CHAIN_KEYS:
        assert torch.equal(ch[k], go[k]), k
    comp["dL_dconic"] = comp["dL_dconic"] * 1.01
    ch2 = oracle.backward_chain(F, comp, *fo["bwd_args"])
    assert not torch.equal(ch2["dL_dcov3D"], go["dL_dcov3D"])
    assert torch.equal(ch2["dL_dsh"], go["dL_dsh"])  # (colour path: untouched by the conic)
    oracle.release(fo["geom"])
