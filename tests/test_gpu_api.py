"""The drop-in Python API on the GPU: autograd wiring of GaussianRasterizer /
LanguageGaussianRasterizer, the render() harness of the reference caller, the sync-free
workspace, and size-independent properties at BASELINE.json's full config-3 size."""
import math

import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import make_config_scene, make_scene
from parity_common import fwd_args, rel_err, run_backend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _settings(sc, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = sc.camera
    return GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        projmatrix_raw=cam.projection_matrix.to(dev), sh_degree=sc.sh_degree, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=False)


def test_language_rasterizer_autograd_matches_c_level(hip, oracle):
    """What gaussian_renderer._language_render does (GS/gaussian_renderer/__init__.py:195-347)."""
    from diff_gaussian_rasterization import LanguageGaussianRasterizer
    dev = torch.device(DEV)
    sc = make_scene(3000, 160, 120, 15, seed=5)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    means3D, opac, scales, rots, shs, lang = (leaf(t) for t in (sc.means3D, sc.opacities, sc.scales, sc.rotations,
                                                                sc.shs, sc.language))
    means2D = torch.zeros_like(means3D, requires_grad=True)
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    rast = LanguageGaussianRasterizer(raster_settings=_settings(sc, dev))
    image, language, radii, depth, opacity, n_touched = rast(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, language_precomp=lang, opacities=opac,
        scales=scales, rotations=rots, cov3D_precomp=None, theta=theta, rho=rho)
    assert image.shape == (3, 120, 160) and language.shape == (15, 120, 160) and depth.shape == (1, 120, 160)
    assert opacity.shape == (1, 120, 160) and radii.dtype == torch.int32 and n_touched.dtype == torch.int32
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(3))
    # opacity takes part in the loss but, like the reference, contributes no gradient
    loss = (image * dc).sum() + (language * dl).sum() + (depth * dd).sum() + opacity.sum() * 0.123
    loss.backward()
    fo, go = run_backend(oracle, sc, None, 3, 15, _abi.BWD_REFERENCE)
    assert torch.equal(image.detach().cpu(), fo["color"]) and torch.equal(language.detach().cpu(), fo["language"])
    for name, t in (("dL_dmeans3D", means3D), ("dL_dopacity", opac), ("dL_dscales", scales), ("dL_drotations", rots),
                    ("dL_dsh", shs), ("dL_dlanguage", lang), ("dL_dmeans2D", means2D)):
        assert rel_err(t.grad, go[name])[0] <= 1e-4, name
    tau = go["dL_dtau"].sum(0)
    assert rel_err(rho.grad, tau[:3])[0] <= 1e-4 and rel_err(theta.grad, tau[3:])[0] <= 1e-4
    assert theta.grad.shape == (3,) and rho.grad.shape == (3,)
    oracle.release(fo["geom"])
    # pose parameters of the reference's other shape ([1, 3] views are what its Function returns): same numbers, their shape
    theta2 = torch.zeros(1, 3, device=dev, requires_grad=True)
    rho2 = torch.zeros(1, 3, device=dev, requires_grad=True)
    out = rast(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, language_precomp=lang, opacities=opac,
               scales=scales, rotations=rots, cov3D_precomp=None, theta=theta2, rho=rho2)
    ((out[0] * dc).sum() + (out[1] * dl).sum() + (out[3] * dd).sum()).backward()
    assert theta2.grad.shape == (1, 3) and torch.equal(theta2.grad.reshape(3), theta.grad)
    assert torch.equal(rho2.grad.reshape(3), rho.grad)


@pytest.mark.parametrize("F", [0, 15])
def test_compiled_and_ctypes_bindings_agree_bit_for_bit(hip, F, monkeypatch):
    """The five `_C` functions through csrc/olsr_torch.cpp and through ctypes reach the same library calls: every
    output, state-dependent gradient and mark_visible mask must be identical."""
    dev = torch.device(DEV)
    sc = make_scene(5000, 200, 150, F, seed=21)
    out = {}
    for binding in ("torch", "ctypes"):
        monkeypatch.setenv("OLSR_BINDING", binding)
        assert (hip.compiled_binding() is not None) == (binding == "torch")
        fw, gr = run_backend(hip, sc, dev, 9, 15, _abi.BWD_REFERENCE)
        vis = hip.mark_visible(sc.means3D.to(dev), sc.camera.world_view_transform.to(dev),
                               sc.camera.full_proj_transform.to(dev))
        out[binding] = (fw, gr, vis)
    (fa, ga, va), (fb, gb, vb) = out["torch"], out["ctypes"]
    assert fa["R"] == fb["R"] and torch.equal(va, vb) and va.dtype == torch.bool
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert (fa[k] is None and fb[k] is None) or torch.equal(fa[k], fb[k]), k
    assert set(ga) == set(gb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


class _Model:
    """The attributes of GaussianModel that gaussian_renderer.render touches."""

    def __init__(self, sc, dev, language=True, isotropic=False):
        leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
        self.get_xyz, self.get_opacity, self.get_rotation = leaf(sc.means3D), leaf(sc.opacities), leaf(sc.rotations)
        self.get_scaling = leaf(sc.scales[:, :1] if isotropic else sc.scales)
        self.get_features = leaf(sc.shs)
        self.get_language_features = leaf(sc.language) if language else None
        self.active_sh_degree = sc.sh_degree
        self.max_sh_degree = int(math.isqrt(sc.shs.shape[1])) - 1
        self.is_language = language


def _view(sc, dev):
    from types import SimpleNamespace
    cam = sc.camera
    return SimpleNamespace(FoVx=2 * math.atan(cam.tanfovx), FoVy=2 * math.atan(cam.tanfovy), image_height=cam.height,
                           image_width=cam.width, world_view_transform=cam.world_view_transform.to(dev),
                           full_proj_transform=cam.full_proj_transform.to(dev), projection_matrix=cam.projection_matrix.to(dev),
                           camera_center=cam.camera_center.to(dev), cam_rot_delta=torch.zeros(3, device=dev, requires_grad=True),
                           cam_trans_delta=torch.zeros(3, device=dev, requires_grad=True))


def test_render_harness_returns_the_callers_dict(hip, oracle):
    """online_lang_splatting_amd.render — the counterpart of gaussian_renderer.render / _language_render
    (GS/gaussian_renderer/__init__.py:25-58, 195-347): settings from a Camera, the dict keys the SLAM code reads
    (:338-347), gradients into the model's leaves, viewspace_points.grad and the pose deltas; the SH-in-Python
    branch, the isotropic-scaling branch, the empty model and the mask path."""
    from types import SimpleNamespace
    from online_lang_splatting_amd import render
    dev = torch.device(DEV)
    sc = make_scene(3000, 160, 120, 15, seed=5, max_sh_degree=1, sh_degree=1)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False)
    bg = sc.bg.to(dev)
    pc, view = _Model(sc, dev), _view(sc, dev)
    pkg = render(view, pc, pipe, bg)
    assert list(pkg) == ["render", "language", "viewspace_points", "visibility_filter", "radii", "depth", "opacity", "n_touched"]
    fo, go = run_backend(oracle, sc, None, 3, 15, _abi.BWD_REFERENCE)
    assert torch.equal(pkg["render"].detach().cpu(), fo["color"]) and torch.equal(pkg["language"].detach().cpu(), fo["language"])
    assert torch.equal(pkg["depth"].detach().cpu(), fo["depth"]) and torch.equal(pkg["radii"].cpu(), fo["radii"])
    assert torch.equal(pkg["visibility_filter"].cpu(), fo["radii"] > 0) and torch.equal(pkg["n_touched"].cpu(), fo["n_touched"])
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(3))
    ((pkg["render"] * dc).sum() + (pkg["language"] * dl).sum() + (pkg["depth"] * dd).sum()).backward()
    assert rel_err(pc.get_xyz.grad, go["dL_dmeans3D"])[0] <= 1e-4
    assert rel_err(pkg["viewspace_points"].grad, go["dL_dmeans2D"])[0] <= 1e-4
    tau = go["dL_dtau"].double().sum(0).float()
    assert rel_err(view.cam_trans_delta.grad.reshape(-1), tau[:3])[0] <= 1e-4   # rho
    assert rel_err(view.cam_rot_delta.grad.reshape(-1), tau[3:])[0] <= 1e-4     # theta
    oracle.release(fo["geom"])
    # SH evaluated in Python (convert_SHs_python): same image up to the fp32 order of the SH sum
    pkg2 = render(view, _Model(sc, dev), SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False), bg)
    assert rel_err(pkg2["render"], pkg["render"].detach())[0] <= 1e-5
    # RGB-only model: no "language" key; isotropic scaling is repeated to three axes
    sc0 = make_scene(1500, 96, 64, 0, seed=6)
    pkg3 = render(_view(sc0, dev), _Model(sc0, dev, language=False, isotropic=True), pipe, sc0.bg.to(dev))
    assert list(pkg3) == ["render", "viewspace_points", "visibility_filter", "radii", "depth", "opacity", "n_touched"]
    sc_iso = make_scene(1500, 96, 64, 0, seed=6)
    sc_iso.scales = sc_iso.scales[:, :1].repeat(1, 3).contiguous()
    fo3, _ = run_backend(oracle, sc_iso, None, 0, 15, 0)
    assert torch.equal(pkg3["render"].detach().cpu(), fo3["color"])
    oracle.release(fo3["geom"])
    # mask: a subset is rendered, per-Gaussian outputs come back for all P
    mask = torch.zeros(sc.P, dtype=torch.bool, device=dev)
    mask[::2] = True
    pkg4 = render(view, _Model(sc, dev), pipe, bg, mask=mask)
    assert pkg4["radii"].shape[0] == sc.P and int(pkg4["radii"][~mask].abs().sum()) == 0 and int(pkg4["radii"][mask].sum()) > 0
    # empty model
    empty = _Model(make_scene(0, 96, 64, 0, seed=1), dev, language=False)
    assert render(_view(sc0, dev), empty, pipe, sc0.bg.to(dev)) is None


def test_rgb_rasterizer_autograd_and_mark_visible(hip, oracle):
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = torch.device(DEV)
    sc = make_scene(2000, 128, 96, 0, seed=6, max_sh_degree=1, sh_degree=1)
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    means3D, opac, scales, rots, shs = (leaf(t) for t in (sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.shs))
    means2D = torch.zeros_like(means3D, requires_grad=True)
    rast = GaussianRasterizer(raster_settings=_settings(sc, dev))
    image, radii, depth, opacity, n_touched = rast(means3D=means3D, means2D=means2D, shs=shs, opacities=opac,
                                                   scales=scales, rotations=rots)
    dc, _, dd = sc.cotangents(4)
    ((image * dc.to(dev)).sum() + (depth * dd.to(dev)).sum()).backward()
    fo, go = run_backend(oracle, sc, None, 4, 15, _abi.BWD_REFERENCE)
    assert torch.equal(image.detach().cpu(), fo["color"])
    assert rel_err(means3D.grad, go["dL_dmeans3D"])[0] <= 1e-4 and rel_err(shs.grad, go["dL_dsh"])[0] <= 1e-4
    vis = rast.markVisible(means3D.detach())
    assert torch.equal(vis.cpu(), sc.means3D[:, 2] > 0.2)
    oracle.release(fo["geom"])


def test_workspace_async_equals_dropin_path(hip):
    """olsr_forward_async / olsr_backward with pre-allocated buffers == the allocating drop-in path."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(20000, 320, 240, 15, seed=7)
    fg, gg = run_backend(hip, sc, dev, 5, 15, 0)
    cam = sc.camera
    ws = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], int(fg["R"] * 1.2) + 1000, dev)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(5))
    kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
              viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
              projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    for _ in range(2):  # buffers are reused: the second pass must not see stale state
        ws.set_scene(**kw)
        out = ws.forward()
        g = ws.backward(dc, dl, dd)
    assert ws.rendered() == (fg["R"], False)
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert torch.equal(out[k], fg[k]), k
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dlanguage", "dL_dtau"):
        assert torch.equal(g[k].reshape(gg[k].shape), gg[k]), k
    # the launch-order hint never changes a result: any permutation of the tiles gives the same bits, and the
    # workspace leaves this frame's heaviest-first order behind for the next one
    ntiles = ws.tile_order.numel()
    ws.tile_order.copy_(torch.randperm(ntiles, generator=torch.Generator().manual_seed(3)).to(torch.int32))
    ws.set_scene(**kw)
    out_p = ws.forward()
    g_p = ws.backward(dc, dl, dd)
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert torch.equal(out_p[k], fg[k]), k
    assert torch.equal(g_p["dL_dmeans3D"].reshape(gg["dL_dmeans3D"].shape), gg["dL_dmeans3D"])
    assert torch.equal(torch.sort(ws.tile_order.long()).values.cpu(), torch.arange(ntiles))
    # capacity overflow is reported, nothing is rendered, nothing crashes
    small = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], 1000, dev)
    small.set_scene(**kw)
    o2 = small.forward()
    R, overflow = small.rendered()
    assert overflow and R == fg["R"] and float(o2["opacity"].abs().max()) == 0.0
    # ... and a backward after the overflowed forward reports it and writes ZERO gradients: the frame's instance
    # tables were never written, so nothing of them may be read (first on the fresh workspace, then again after a
    # valid frame has left its tables behind in the same buffers)
    for attempt in range(2):
        if attempt == 1:
            big = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], int(fg["R"] * 1.2) + 1000, dev)
            big.set_scene(**kw)
            big.forward()
            big.backward(dc, dl, dd)
            small.geom.copy_(big.geom)          # stale but plausible inst_start / offsets / counters of another frame
            small.set_scene(**kw)
            small.forward()
            assert small.rendered()[1]
        gz = small.backward(dc, dl, dd)
        assert small.backward_status()[1], "an overflowed forward must surface in the backward's status"
        for k, v in gz.items():
            assert float(v.abs().max()) == 0.0, (attempt, k)
    # backward scratch too small for the live (instance, slot) rows: reported, not a crash
    L, ov = ws.backward_status()
    assert not ov and 0 < L <= 4 * fg["R"]
    tight = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], int(fg["R"] * 1.2) + 1000, dev, row_capacity=L - 1)
    tight.set_scene(**kw)
    tight.forward()
    tight.backward(dc, dl, dd)
    assert tight.backward_status() == (L, True)
    exact = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], int(fg["R"] * 1.2) + 1000, dev, row_capacity=L)
    exact.set_scene(**kw)
    exact.forward()
    g3 = exact.backward(dc, dl, dd)
    assert exact.backward_status() == (L, False)
    assert torch.equal(g3["dL_dmeans3D"], gg["dL_dmeans3D"])


def test_fused_bucket_equals_separate_accumulate(hip):
    """olsr_backward with an olsr_grad_bucket == olsr_backward + olsr_accumulate_gradients, for the first
    view (assign) and a second view (add), with and without the separate per-Gaussian arrays; SH degree 3
    so the 48-float SH slice of the bucket row is exercised."""
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket, RasterWorkspace
    from online_lang_splatting_amd.scene import arc_cameras
    dev = torch.device(DEV)
    sc = make_scene(6000, 200, 150, 15, seed=13, max_sh_degree=3, sh_degree=3)
    M = sc.shs.shape[1]
    cams = arc_cameras(200, 150, 2)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(4))
    gk = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev), sh_degree=3)
    ws = RasterWorkspace(sc.P, 200, 150, 15, M, 300000, dev)
    ref = GradientBucket(sc.P, GradLayout(M, 15), dev)
    fused = GradientBucket(sc.P, GradLayout(M, 15), dev)
    only = GradientBucket(sc.P, GradLayout(M, 15), dev)
    for b in (ref, fused, only):  # stale contents must be overwritten by the first view
        b.flat.fill_(7.0)
        b.densify.fill_(7.0)
        b.max_radii.fill_(7)
    for i, c in enumerate(cams):
        ws.set_scene(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                     projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                     tanfovy=c.tanfovy, **gk)
        out = ws.forward()
        g = ws.backward(dc, dl, dd)
        sep = {k: v.clone() for k, v in g.items()}
        ref.accumulate(sep, out["radii"], first=(i == 0))
        g2 = ws.backward(dc, dl, dd, bucket=fused, first=(i == 0))
        for k in sep:  # the separate arrays are still written, identically
            assert torch.equal(g2[k], sep[k]), k
        g3 = ws.backward(dc, dl, dd, bucket=only, first=(i == 0), bucket_only=True)
        assert torch.equal(g3["dL_dtau_sum"], sep["dL_dtau_sum"])
        g4 = ws.backward(dc, dl, dd, pose_only=True)  # tracking: the pose gradient alone
        assert torch.equal(g4["dL_dtau_sum"], sep["dL_dtau_sum"]) and g4["dL_dmeans3D"] is None
        assert not ws.rendered()[1]
    for b in (fused, only):
        assert torch.equal(b.flat, ref.flat)
        assert torch.equal(b.densify, ref.densify)
        assert torch.equal(b.max_radii, ref.max_radii)
    assert float(ref.flat.abs().max()) > 0 and int((ref.densify[:, 1] == 2).sum()) > 0


def test_fused_activations_match_the_pytorch_chain(hip):
    """OLSR_ACT_*: raw opacity / scale / rotation parameters in, gradients with respect to them out — against
    PyTorch's sigmoid / exp / normalize around the un-fused path (GaussianModel.get_opacity / get_scaling /
    get_rotation, gaussian_model.py:95-105, and the autograd chain through them)."""
    import torch.nn.functional as Fn
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(8000, 200, 150, 15, seed=23)
    cam = sc.camera
    g = torch.Generator().manual_seed(23)
    raw_op = torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)).to(dev)
    raw_sc = torch.log(sc.scales).to(dev)
    raw_rot = (sc.rotations * (0.3 + 2 * torch.rand(sc.P, 1, generator=g))).to(dev)   # un-normalised quaternions
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(6))
    base = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
                viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
                projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
                tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    ws = RasterWorkspace(sc.P, 200, 150, 15, sc.shs.shape[1], 600000, dev)
    # un-fused: PyTorch activations, gradients chained by autograd
    ro, rs, rr = (t.clone().requires_grad_(True) for t in (raw_op, raw_sc, raw_rot))
    act_o, act_s, act_r = torch.sigmoid(ro), torch.exp(rs), Fn.normalize(rr)
    ws.set_scene(opacities=act_o.detach().contiguous(), scales=act_s.detach().contiguous(),
                 rotations=act_r.detach().contiguous(), **base)
    out_ref = {k: v.clone() for k, v in ws.forward().items()}
    gref = {k: v.clone() for k, v in ws.backward(dc, dl, dd).items()}
    torch.autograd.backward([act_o, act_s, act_r],
                            [gref["dL_dopacity"].reshape(act_o.shape), gref["dL_dscales"], gref["dL_drotations"]])
    # fused
    ws.set_scene(opacities=raw_op, scales=raw_sc, rotations=raw_rot, activations=_abi.ACT_ALL, **base)
    out = ws.forward()
    for k in ("color", "language", "depth", "opacity"):
        r, e = rel_err(out[k], out_ref[k])
        assert r <= 2e-5, (k, r)           # expf / sigmoid differ from PyTorch's by an ulp or two
    assert int((out["radii"] != out_ref["radii"]).sum()) <= 2
    gf = ws.backward(dc, dl, dd)
    for name, got, exp in (("opacity", gf["dL_dopacity"].reshape(-1), ro.grad.reshape(-1)), ("scales", gf["dL_dscales"], rs.grad),
                           ("rotations", gf["dL_drotations"], rr.grad), ("means3D", gf["dL_dmeans3D"], gref["dL_dmeans3D"]),
                           ("language", gf["dL_dlanguage"], gref["dL_dlanguage"])):
        r, e = rel_err(got, exp)
        assert r <= 2e-4, (name, r, e)
    # a raw opacity must be handed to the backward too, and activations exclude a precomputed covariance
    assert float(gf["dL_drotations"].abs().max()) > 0 and float(rr.grad.abs().max()) > 0


def test_fused_adam_equals_torch_optim_adam(hip):
    """olsr_adam_step over the gradient bucket == torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) over the seven
    parameter tensors of the reference's GaussianModel (gaussian_model.py:393-440), three steps, dense semantics
    (a third of the rows get zero gradient in step 2 and must still move)."""
    from online_lang_splatting_amd.frame_shard import FusedAdam, GradLayout, GradientBucket
    dev = torch.device(DEV)
    P, M, F = 5000, 4, 15
    g = torch.Generator().manual_seed(31)
    init = dict(means3D=torch.randn(P, 3, generator=g), shs=torch.randn(P, M, 3, generator=g) * 0.3,
                opacities=torch.randn(P, 1, generator=g), scales=torch.randn(P, 3, generator=g) - 3,
                rotations=torch.randn(P, 4, generator=g), language=torch.zeros(P, F))
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=2.5e-3 / 20, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    # the reference's optimiser (features split into dc / rest as in GaussianModel)
    ref = {k: v.clone().requires_grad_(True) for k, v in init.items() if k != "shs"}
    f_dc = init["shs"][:, :1].clone().requires_grad_(True)
    f_rest = init["shs"][:, 1:].clone().requires_grad_(True)
    opt = torch.optim.Adam([dict(params=[ref["means3D"]], lr=lrs["xyz"]), dict(params=[f_dc], lr=lrs["sh_dc"]),
                            dict(params=[f_rest], lr=lrs["sh_rest"]), dict(params=[ref["opacities"]], lr=lrs["opacity"]),
                            dict(params=[ref["scales"]], lr=lrs["scale"]), dict(params=[ref["rotations"]], lr=lrs["rotation"]),
                            dict(params=[ref["language"]], lr=lrs["language"])], lr=0.0, eps=1e-15)
    mine = {k: v.clone().to(dev).contiguous() for k, v in init.items()}
    bucket = GradientBucket(P, GradLayout(M, F), dev)
    adam = FusedAdam(P, GradLayout(M, F), dev)
    for step in range(3):
        grads = dict(dL_dmeans3D=torch.randn(P, 3, generator=g) * 1e-3, dL_dsh=torch.randn(P, M, 3, generator=g) * 1e-3,
                     dL_dopacity=torch.randn(P, 1, generator=g) * 1e-2, dL_dscales=torch.randn(P, 3, generator=g) * 1e-3,
                     dL_drotations=torch.randn(P, 4, generator=g) * 1e-3, dL_dlanguage=torch.randn(P, F, generator=g) * 1e-3,
                     dL_dmeans2D=torch.randn(P, 3, generator=g))
        if step == 1:
            for v in grads.values():
                v[::3] = 0.0
        radii = torch.ones(P, dtype=torch.int32)
        bucket.accumulate({k: v.to(dev) for k, v in grads.items()}, radii.to(dev), first=True)
        adam.step(bucket, mine, lrs)
        ref["means3D"].grad, ref["opacities"].grad = grads["dL_dmeans3D"].clone(), grads["dL_dopacity"].clone()
        ref["scales"].grad, ref["rotations"].grad = grads["dL_dscales"].clone(), grads["dL_drotations"].clone()
        ref["language"].grad = grads["dL_dlanguage"].clone()
        f_dc.grad, f_rest.grad = grads["dL_dsh"][:, :1].clone(), grads["dL_dsh"][:, 1:].clone()
        opt.step()
    expect = dict(ref, shs=torch.cat([f_dc, f_rest], dim=1))
    for k in init:
        torch.testing.assert_close(mine[k].cpu(), expect[k].detach(), rtol=2e-6, atol=2e-7, msg=lambda m, k=k: f"{k}: {m}")  # one ulp of an O(1) parameter
    m_ref = torch.cat([opt.state[p_]["exp_avg"].reshape(P, -1) for p_ in
                       (ref["means3D"], f_dc, f_rest, ref["opacities"], ref["scales"], ref["rotations"], ref["language"])], dim=1)
    v_ref = torch.cat([opt.state[p_]["exp_avg_sq"].reshape(P, -1) for p_ in
                       (ref["means3D"], f_dc, f_rest, ref["opacities"], ref["scales"], ref["rotations"], ref["language"])], dim=1)
    # (torch's CPU lerp / addcmul fuse some multiply-adds; a few ulp of the largest entries after three steps)
    torch.testing.assert_close(adam.exp_avg.cpu(), m_ref, rtol=1e-5, atol=2e-9)
    torch.testing.assert_close(adam.exp_avg_sq.cpu(), v_ref, rtol=1e-5, atol=1e-11)


def test_frames_in_flight_are_independent(hip):
    """FrameLanes: three views rendered concurrently on three HIP streams == the same views rendered
    one after the other (bit for bit: no shared scratch, no cross-stream race)."""
    from online_lang_splatting_amd.frame_shard import FrameLanes
    from online_lang_splatting_amd.scene import arc_cameras
    dev = torch.device(DEV)
    sc = make_scene(20000, 320, 240, 15, seed=11)
    cams = arc_cameras(320, 240, 3)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(3))
    gk = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
              sh_degree=sc.sh_degree)

    def cam_kw(c):
        return dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                    projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev),
                    tanfovx=c.tanfovx, tanfovy=c.tanfovy)

    def render(lanes, repeat):
        res = []
        for _ in range(repeat):
            res = []
            for c in cams:
                ws, bucket, st = lanes.next_lane()
                st.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(st):
                    ws.set_scene(**cam_kw(c), **gk)
                    out = ws.forward()
                    g = ws.backward(dc, dl, dd)
                    bucket.accumulate(g, out["radii"], first=True)
                res.append((ws, bucket))
        lanes.synchronize()
        torch.cuda.synchronize(dev)
        return [(ws.rendered(), {k: v.clone() for k, v in ws.out.items()}, b.flat.clone()) for ws, b in res]

    serial = []
    one = FrameLanes(1, sc.P, 320, 240, 15, sc.shs.shape[1], 400000, dev)
    for c in cams:
        ws, bucket, st = one.next_lane()
        ws.set_scene(**cam_kw(c), **gk)
        out = ws.forward()
        g = ws.backward(dc, dl, dd)
        bucket.accumulate(g, out["radii"], first=True)
        torch.cuda.synchronize(dev)
        serial.append((ws.rendered(), {k: v.clone() for k, v in ws.out.items()}, bucket.flat.clone()))
    par = render(FrameLanes(3, sc.P, 320, 240, 15, sc.shs.shape[1], 400000, dev), repeat=3)
    assert len({r[0][0] for r in serial}) > 1  # the three views really differ
    for (ra, oa, fa), (rb, ob, fb) in zip(serial, par):
        assert ra == rb and not ra[1]
        for k in oa:
            assert torch.equal(oa[k], ob[k]), k
        assert torch.equal(fa, fb)


def test_full_size_config3_properties(hip):
    """BASELINE.json configs[2] (500 k Gaussians, 1200x680, F=15) on the GPU: properties that do not
    need the oracle — sort order, range partition, transmittance/opacity identities, linearity of
    the backward in the cotangent, determinism."""
    dev = torch.device(DEV)
    sc = make_config_scene(3)
    P, W, H, F = sc.P, 1200, 680, 15
    fg, g1 = run_backend(hip, sc, dev, 3, 15, 0)
    R = fg["R"]
    cnt = hip.state_field("geometry", fg["geom"], "counters", P=P, F=F, dtype=torch.int32, count=8)
    assert int(cnt[3]) == 5189186  # the reference's num_rendered for this scene (oracle)
    assert 2_000_000 < R < 3_500_000  # exact tile lists keep about half of the bounding-square instances
    gx, gy = math.ceil(W / 15), math.ceil(H / 15)
    pl = hip.state_field("binning", fg["binning"], "point_list", R=R, F=F, dtype=torch.int32, count=R).long()
    rg = hip.state_field("image", fg["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2 * gx * gy).view(-1, 2).long()
    lens = rg[:, 1] - rg[:, 0]
    assert int(lens.sum()) == R
    nz = lens > 0
    starts = rg[nz, 0]
    assert bool((starts[1:] == rg[nz, 1][:-1]).all()) and int(starts[0]) == 0  # contiguous partition
    depths = hip.state_field("geometry", fg["geom"], "depths", P=P, F=F, dtype=torch.float32, count=P)
    d = depths[pl]
    tile_of = torch.repeat_interleave(torch.arange(gx * gy, device=dev), lens)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((d[1:][same_tile] >= d[:-1][same_tile]).all())  # front-to-back inside every tile
    tie = same_tile & (d[1:] == d[:-1])
    assert bool((pl[1:][tie] > pl[:-1][tie]).all())  # ties by Gaussian index
    tt = hip.state_field("geometry", fg["geom"], "tiles_touched", P=P, F=F, dtype=torch.int32, count=P).long()
    assert torch.equal(torch.bincount(pl, minlength=P), tt * (fg["radii"] > 0))
    final_T = hip.state_field("image", fg["img"], "final_T", W=W, H=H, dtype=torch.float32, count=W * H)
    assert torch.equal(fg["opacity"].reshape(-1), 1 - final_T)
    assert float(fg["opacity"].min()) >= 0 and float(fg["opacity"].max()) <= 1
    assert bool(torch.isfinite(fg["color"]).all()) and bool(torch.isfinite(fg["language"]).all())
    assert bool((fg["n_touched"] <= 0).logical_or(fg["radii"] > 0).all())
    # backward is linear in the cotangent and deterministic
    _, g2 = run_backend(hip, sc, dev, 3, 15, 0)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        assert bool(torch.isfinite(g1[k]).all()), k
    # the reference's bounding-square binning gives the same images bit for bit and the same gradients
    fr, gr = run_backend(hip, sc, dev, 3, 15, 0, binning=0)
    assert fr["R"] == 5189186
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert torch.equal(fr[k], fg[k]), k
    for k in g1:
        r, _ = rel_err(g1[k], gr[k])
        assert r <= 1e-5, (k, r)
    hip.TILE, hip.BWD_MODE, hip.BINNING = 15, 0, 1


def test_full_size_config5_properties(hip):
    """BASELINE.json configs[4], one of its eight views: 2 M Gaussians, 1920x1080, F = 32 — the large-sort paths
    (> 1000 radix blocks, device-wide scans), 9 216 tiles, 192-byte gradient rows.  Oracle-free properties:
    range partition, front-to-back order, opacity identity, finite outputs, run-to-run determinism, and
    agreement of the two binning modes."""
    dev = torch.device(DEV)
    sc = make_config_scene(5)
    P, W, H, F = sc.P, 1920, 1080, 32
    fg, g1 = run_backend(hip, sc, dev, 5, 15, 0)
    R = fg["R"]
    cnt = hip.state_field("geometry", fg["geom"], "counters", P=P, F=F, dtype=torch.int32, count=8)
    assert int(cnt[3]) == 14948564 and 6_000_000 < R < 11_000_000
    gx, gy = math.ceil(W / 15), math.ceil(H / 15)
    pl = hip.state_field("binning", fg["binning"], "point_list", R=R, F=F, dtype=torch.int32, count=R).long()
    rg = hip.state_field("image", fg["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2 * gx * gy).view(-1, 2).long()
    lens = rg[:, 1] - rg[:, 0]
    assert int(lens.sum()) == R
    depths = hip.state_field("geometry", fg["geom"], "depths", P=P, F=F, dtype=torch.float32, count=P)
    d = depths[pl]
    tile_of = torch.repeat_interleave(torch.arange(gx * gy, device=dev), lens)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((d[1:][same_tile] >= d[:-1][same_tile]).all())
    final_T = hip.state_field("image", fg["img"], "final_T", W=W, H=H, dtype=torch.float32, count=W * H)
    assert torch.equal(fg["opacity"].reshape(-1), 1 - final_T)
    assert bool(torch.isfinite(fg["color"]).all()) and bool(torch.isfinite(fg["language"]).all())
    _, g2 = run_backend(hip, sc, dev, 5, 15, 0)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        assert bool(torch.isfinite(g1[k]).all()), k
    del g2
    fr, gr = run_backend(hip, sc, dev, 5, 15, 0, binning=0)
    assert fr["R"] == 14948564
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert torch.equal(fr[k], fg[k]), k
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dlanguage", "dL_dmeans3D"):
        r, _ = rel_err(g1[k], gr[k])
        assert r <= 1e-5, (k, r)
    hip.TILE, hip.BWD_MODE, hip.BINNING = 15, 0, 1


@pytest.mark.parametrize("W,H,tile", [(7, 5, 15), (16, 16, 16), (4111, 37, 15), (33, 4200, 16)])
def test_extreme_image_shapes(hip, oracle, W, H, tile):
    """Images smaller than one tile, exactly one tile, and very wide / very tall strips (hundreds of tiles in one
    row or column): forward bit-exact and gradients within tolerance against the oracle."""
    from test_gpu_parity import _check
    sc = make_scene(1500, W, H, 15, seed=W + H)
    _check(hip, oracle, sc, seed=3, tile=tile)


def test_fused_accumulate_matches_torch_formulation(hip):
    """olsr_accumulate_gradients == GradientBucket's torch specification (the gloo tests' path)."""
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket
    dev = torch.device(DEV)
    P, M, F = 1000, 4, 15
    g = torch.Generator().manual_seed(3)
    grads = dict(dL_dmeans3D=torch.randn(P, 3, generator=g), dL_dsh=torch.randn(P, M, 3, generator=g),
                 dL_dopacity=torch.randn(P, 1, generator=g), dL_dscales=torch.randn(P, 3, generator=g),
                 dL_drotations=torch.randn(P, 4, generator=g), dL_dlanguage=torch.randn(P, F, generator=g),
                 dL_dmeans2D=torch.randn(P, 3, generator=g))
    radii = torch.randint(0, 5, (P,), generator=g, dtype=torch.int32)
    ref = GradientBucket(P, GradLayout(M, F), "cpu")
    got = GradientBucket(P, GradLayout(M, F), dev)
    for _ in range(3):
        ref.accumulate(grads, radii)
        got.accumulate({k: v.to(dev) for k, v in grads.items()}, radii.to(dev))
    torch.testing.assert_close(got.flat.cpu(), ref.flat, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got.densify.cpu(), ref.densify, rtol=1e-6, atol=1e-6)
    assert torch.equal(got.max_radii.cpu(), ref.max_radii)


@pytest.mark.parametrize("mode", [_abi.BWD_REFERENCE, _abi.BWD_EXACT])
@pytest.mark.parametrize("tile", [15, 16])
def test_null_language_and_depth_cotangents_equal_zero_cotangents(hip, oracle, mode, tile):
    """olsr_backward without a language / depth cotangent (the tracking loss has no language term,
    utils/slam_utils.py:92-121; autograd hands None, DGR/diff_gaussian_rasterization/__init__.py:296-345): the RGB
    instantiation of the composite backward runs on the language forward's state.  Every gradient must equal what
    zero-filled cotangents give (value for value: D - A == 0 and the rank-0 language row == 0), dL_dlanguage must be
    zeros, and both must sit within the parity tolerance of the oracle fed zeros."""
    from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(12000, 300, 210, 15, seed=11)
    cam = sc.camera
    hip.TILE, hip.BWD_MODE = tile, mode
    fg, _ = run_backend(hip, sc, dev, 4, tile, mode)
    M = sc.shs.shape[1]
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(4))
    kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
              viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
              projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    ws = RasterWorkspace(sc.P, 300, 210, 15, M, int(fg["R"] * 1.2) + 1000, dev, tile=tile, bwd_mode=mode)
    ws.set_scene(**kw)
    ws.forward()
    names = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dlanguage", "dL_ddepths", "dL_dmeans3D",
             "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtau", "dL_dtau_sum")
    zl, zd = torch.zeros_like(dl), torch.zeros_like(dd)
    for lang_cot, depth_cot, zlang, zdepth in ((None, dd, zl, dd), (None, None, zl, zd), (dl, None, dl, zd)):
        ref = {k: v.clone() for k, v in ws.backward(dc, zlang, zdepth).items()}
        got = ws.backward(dc, lang_cot, depth_cot)
        for k in names:
            if mode == _abi.BWD_REFERENCE or lang_cot is not None:
                assert torch.equal(got[k], ref[k]), (k, lang_cot is None, depth_cot is None)
            else:
                # exact mode reduces 10 + F values per visit: without the language channels the colour / depth sums go
                # through a differently shaped permlane tree (same values, another rounding order)
                assert rel_err(got[k], ref[k])[0] <= 2e-6, (k, rel_err(got[k], ref[k]))
        if lang_cot is None:
            assert float(got["dL_dlanguage"].abs().max()) == 0.0
    # pose-only (what tracking runs) and the bucket: same values through the NULL path
    ref_tau = ws.backward(dc, zl, dd, pose_only=True)["dL_dtau_sum"].clone()
    got_tau = ws.backward(dc, None, dd, pose_only=True)["dL_dtau_sum"]
    assert torch.equal(got_tau, ref_tau) if mode == _abi.BWD_REFERENCE else rel_err(got_tau, ref_tau)[0] <= 2e-6
    lay = GradLayout(M, 15)
    b0, b1 = GradientBucket(sc.P, lay, dev), GradientBucket(sc.P, lay, dev)
    ws.backward(dc, zl, dd, bucket=b0, first=True, bucket_only=True)
    ws.backward(dc, None, dd, bucket=b1, first=True, bucket_only=True)
    if mode == _abi.BWD_REFERENCE:
        assert torch.equal(b0.flat, b1.flat) and torch.equal(b0.densify, b1.densify)
    else:
        assert rel_err(b1.flat, b0.flat)[0] <= 2e-6 and rel_err(b1.densify, b0.densify)[0] <= 2e-6
    assert float(b1.view("language").abs().max()) == 0.0
    # against the oracle with zero language cotangent
    import types
    sc0 = types.SimpleNamespace(**{k: getattr(sc, k) for k in ("camera", "means3D", "opacities", "scales", "rotations",
                                                               "shs", "language", "bg", "sh_degree", "F", "P")})
    sc0.cotangents = lambda seed: (lambda c: (c[0], torch.zeros_like(c[1]), c[2]))(sc.cotangents(seed))
    fo, go = run_backend(oracle, sc0, None, 4, tile, mode)
    got = ws.backward(dc, None, dd)
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D", "dL_dtau"):
        assert rel_err(got[k].reshape(go[k].shape), go[k])[0] <= 1e-4, k
    assert float(go["dL_dlanguage"].abs().max()) == 0.0
    oracle.release(fo["geom"])


def test_autograd_without_language_loss_uses_the_null_path(hip, oracle):
    """A loss without a language term through the drop-in autograd API (what front-end tracking does with
    gaussian_renderer.render): language_precomp.grad comes back as zeros, everything else as with a zero cotangent."""
    from diff_gaussian_rasterization import LanguageGaussianRasterizer
    dev = torch.device(DEV)
    sc = make_scene(3000, 160, 120, 15, seed=5)

    def grads(with_zero_lang_term, with_depth):
        leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
        means3D, opac, scales, rots, shs, lang = (leaf(t) for t in (sc.means3D, sc.opacities, sc.scales, sc.rotations,
                                                                    sc.shs, sc.language))
        means2D = torch.zeros_like(means3D, requires_grad=True)
        theta = torch.zeros(3, device=dev, requires_grad=True)
        rho = torch.zeros(3, device=dev, requires_grad=True)
        rast = LanguageGaussianRasterizer(raster_settings=_settings(sc, dev))
        image, language, radii, depth, opacity, n_touched = rast(
            means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, language_precomp=lang, opacities=opac,
            scales=scales, rotations=rots, cov3D_precomp=None, theta=theta, rho=rho)
        dc, dl, dd = (t.to(dev) for t in sc.cotangents(3))
        loss = (image * dc).sum()
        if with_depth:
            loss = loss + (depth * dd).sum()
        if with_zero_lang_term:
            loss = loss + (language * torch.zeros_like(dl)).sum() + (depth * torch.zeros_like(dd)).sum()
        loss.backward()
        return [t.grad for t in (means3D, opac, scales, rots, shs, lang, means2D, theta, rho)]
    for with_depth in (True, False):
        a, b = grads(False, with_depth), grads(True, with_depth)
        for x, y in zip(a, b):
            assert x is not None and torch.equal(x, y)
        assert float(a[5].abs().max()) == 0.0 and a[5].shape == (sc.P, 15)


def test_live_row_counts_reach_the_host_without_a_sync(hip):
    """The drop-in forward posts the frame's gradient-row counts into mapped host memory; the matching backward — also from
    PyTorch's autograd thread — sizes its scratch from them (olsr_live_rows) instead of the bound of 2 / 4 rows per
    instance.  Counts equal what the backward's own row compaction finds; a token whose slot was reused says so."""
    from online_lang_splatting_amd._lib import lib
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(20000, 320, 240, 15, seed=7)
    cam = sc.camera
    kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
              viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
              projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(5))
    want = {}
    for mode in (_abi.BWD_REFERENCE, _abi.BWD_EXACT):
        ws = RasterWorkspace(sc.P, 320, 240, 15, sc.shs.shape[1], 600000, dev, bwd_mode=mode)
        ws.set_scene(**kw)
        ws.forward()
        ws.backward(dc, dl, dd)
        want[mode] = ws.backward_status()[0]
    fg, gg = run_backend(hip, sc, dev, 5, 15, 0)
    tok = hip.last_forward_token()
    torch.cuda.synchronize()
    L = lib()
    assert tok > 0
    assert L.olsr_live_rows(tok, 1) == want[_abi.BWD_REFERENCE] and L.olsr_live_rows(tok, 0) == want[_abi.BWD_EXACT]
    assert 0 < want[_abi.BWD_REFERENCE] < want[_abi.BWD_EXACT] <= 4 * fg["R"]
    assert L.olsr_live_rows(tok + 1, 1) == -1 and L.olsr_live_rows(0, 1) == -1
    # the exact count suffices (a backward with it equals the backward with the bound)
    a = hip.backward_all(15, *_bwd_args(sc, fg, dev, 5), rows_token=tok)
    b = hip.backward_all(15, *_bwd_args(sc, fg, dev, 5))
    for k in b:
        assert torch.equal(a[k], b[k]), k
    # 256 forwards later the slot belongs to another frame
    small = make_scene(50, 32, 32, 15, seed=1)
    for _ in range(257):
        run_fwd_only(hip, small, dev)
    torch.cuda.synchronize()
    assert L.olsr_live_rows(tok, 1) == -1


@pytest.mark.parametrize("wh", [(10, 10), (20, 20), (40, 31), (130, 100)])
def test_live_row_counts_with_few_tiles(hip, wh):
    """The counts are summed by the first block of each of the eight tile chunks; images with 1, 4, 9 and 63 tiles leave
    chunks empty or one tile long."""
    from online_lang_splatting_amd._lib import lib
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    W, H = wh
    sc = make_scene(300, W, H, 15, seed=11)
    cam = sc.camera
    kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
              rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
              viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
              projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(5))
    want = {}
    for mode in (_abi.BWD_REFERENCE, _abi.BWD_EXACT):
        ws = RasterWorkspace(sc.P, W, H, 15, sc.shs.shape[1], 100000, dev, bwd_mode=mode)
        ws.set_scene(**kw)
        ws.forward()
        ws.backward(dc, dl, dd)
        want[mode] = ws.backward_status()[0]
    run_fwd_only(hip, sc, dev)
    tok = hip.last_forward_token()
    torch.cuda.synchronize()
    assert lib().olsr_live_rows(tok, 1) == want[_abi.BWD_REFERENCE] > 0
    assert lib().olsr_live_rows(tok, 0) == want[_abi.BWD_EXACT] >= want[_abi.BWD_REFERENCE]


def test_dropin_launch_order_hint_is_invisible(hip):
    """olsr_forward keeps, per stream and tile count, the heaviest-first tile order of the previous frame issued on that
    stream (the forward composite finishes sooner with it).  It must never show in a result: the same scene after a
    different scene of the same image size, on the default stream and on two side streams interleaved, gives the same bits
    as a first frame."""
    dev = torch.device(DEV)
    a = make_scene(9000, 250, 190, 15, seed=31)
    b = make_scene(7000, 250, 190, 15, seed=32, scale_mult=3.0)
    first, g_first = run_backend(hip, a, dev, 1, 15, 0)   # (whatever earlier tests left behind for 17 x 13 tiles)
    run_backend(hip, b, dev, 2, 15, 0)                    # another scene leaves its order behind
    again, g_again = run_backend(hip, a, dev, 1, 15, 0)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    side = []
    for st, sc_ in ((s1, b), (s2, a), (s1, a), (s2, b), (s2, a)):
        with torch.cuda.stream(st):
            side.append((sc_ is a, run_backend(hip, sc_, dev, 1, 15, 0)))
    torch.cuda.synchronize()
    for is_a, (f, g) in [(True, (again, g_again))] + side:
        if not is_a:
            continue
        assert f["R"] == first["R"]
        for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
            assert torch.equal(f[k], first[k]), k
        for k in g_first:
            assert torch.equal(g[k], g_first[k]), k


def test_backward_scratch_is_exact_when_the_host_runs_ahead(hip):
    """A training loop reaches its backward while the forward is still executing: the binding then waits for the forward's
    posted row count (olsr_backward_rows; the GPU is busy meanwhile) instead of allocating the bound of two rows per
    instance — seen here as the peak of the allocator over un-synchronised forward + backward pairs."""
    from online_lang_splatting_amd._lib import lib
    from parity_common import fwd_args
    dev = torch.device(DEV)
    sc = make_scene(150000, 800, 600, 15, seed=21)
    a = fwd_args(sc, dev)
    cots = [t.to(dev) for t in sc.cotangents(3)]

    def pair():
        R, color, lang, radii, geom, binb, img, depth, opac, nt = hip.rasterize_language_gaussians(*a)
        tok = hip.last_forward_token()
        fwd = dict(R=R, radii=radii, geom=geom, binning=binb, img=img)
        args = [a[0], a[1], radii, a[2], a[3], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], *cots,
                a[16], a[17], a[18], geom, R, binb, img, False]
        g = hip.backward_all(15, *args, rows_token=tok)
        return R, tok, g

    R, tok, g = pair()
    torch.cuda.synchronize()
    bound_bytes = lib().olsr_backward_scratch_bytes(2 * R, 15)
    exact = lib().olsr_live_rows(tok, 1)
    exact_bytes = lib().olsr_backward_scratch_bytes(exact, 15)
    assert bound_bytes > (64 << 20) and exact_bytes < bound_bytes // 3
    del g
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    for _ in range(3):
        R, tok, g = pair()  # no synchronisation between the forward and its backward
        del g
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev) - base
    assert peak < bound_bytes, (peak, bound_bytes, exact_bytes)  # the bound alone would exceed this
    # the policy call itself: exact once posted, the bound for an unknown token
    assert lib().olsr_backward_rows(tok, 1, R, 15) == lib().olsr_live_rows(tok, 1)
    assert lib().olsr_backward_rows(0, 1, R, 15) == 2 * R
    assert lib().olsr_live_rows_wait(tok + 7, 1, 1000) == -1  # never issued: times out


@pytest.mark.parametrize("binding", ["torch", "ctypes"])
def test_a_guessed_backward_scratch_is_verified_and_redone_exactly(hip, binding, monkeypatch):
    """Round 5: when the host reaches the backward before the forward has posted its row count, the bindings launch the
    backward at once with a GUESSED scratch size (1.5 x the rows per instance of the last verified frame) and verify the
    guess while the GPU works.  A guess that was too small — here forced by a ratio 100 x too low — makes that first backward
    write zeros; the binding follows it with an exact second backward on the same stream.  The gradients are those of the
    exactly sized call either way, bit for bit, and the redo is counted."""
    from parity_common import fwd_args
    monkeypatch.setenv("OLSR_BINDING", binding)
    dev = torch.device(DEV)
    sc = make_scene(150000, 800, 600, 15, seed=21)
    a = fwd_args(sc, dev)
    cots = [t.to(dev) for t in sc.cotangents(3)]

    def pair():
        R, color, lang, radii, geom, binb, img, depth, opac, nt = hip.rasterize_language_gaussians(*a)
        tok = hip.last_forward_token()
        args = [a[0], a[1], radii, a[2], a[3], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], *cots,
                a[16], a[17], a[18], geom, R, binb, img, False]
        return hip.backward_all(15, *args, rows_token=tok)      # no synchronisation in between

    hip.debug_rows_ratio(True, 0.0)                             # (forget what earlier tests of this process left behind)
    ref = pair()                                                # (establishes the ratio: exact count, waited for or posted)
    torch.cuda.synchronize()
    ratio, redone0 = hip.debug_rows_ratio(True)
    assert 0.0 < ratio < 2.0
    good = pair()                                               # a guess of 1.5 x that: large enough, no redo
    torch.cuda.synchronize()
    assert hip.debug_rows_ratio(True)[1] == redone0
    hip.debug_rows_ratio(True, ratio / 100.0)
    redo = pair()                                               # far too small: zeros first, then the exact backward
    torch.cuda.synchronize()
    ratio2, redone1 = hip.debug_rows_ratio(True)
    assert redone1 == redone0 + 1 and abs(ratio2 - ratio) < 1e-4 * ratio   # (and the ratio is the verified one again)
    for k in ref:
        assert torch.equal(ref[k], good[k]) and torch.equal(ref[k], redo[k]), k
    assert float(ref["dL_dmeans3D"].abs().max()) > 0


def test_more_than_64_hint_keys_retire_buffers_instead_of_freeing_them(hip):
    """the synchronising entry keeps one hint buffer per (device, stream, tile count), at most 64; a 65th key
    used to hipFree() the least recently used one although another thread might hold its pointer and its stream might still
    be writing it.  Evicted buffers are now retired to a spare list and handed to later keys of the same size.  Seventy
    streams and three resolutions through the drop-in forward, twice over: every result equals the first stream's."""
    from parity_common import fwd_args
    dev = torch.device(DEV)
    scenes = [make_scene(1500, w, h, 15, seed=40 + i) for i, (w, h) in enumerate(((160, 120), (128, 96), (200, 90)))]
    args = [fwd_args(sc, dev) for sc in scenes]
    ref = [hip.rasterize_language_gaussians(*a)[1].clone() for a in args]
    streams = [torch.cuda.Stream(dev) for _ in range(70)]
    torch.cuda.synchronize()
    for rnd in range(2):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append((i % 3, hip.rasterize_language_gaussians(*args[i % 3])[1]))
        torch.cuda.synchronize()
        for k, o in outs:
            assert torch.equal(o, ref[k])


def run_fwd_only(hip, sc, dev):
    from parity_common import fwd_args
    return hip.rasterize_language_gaussians(*fwd_args(sc, dev))


def _bwd_args(sc, fwd, dev, seed):
    from parity_common import fwd_args
    a = fwd_args(sc, dev)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(seed))
    return [a[0], a[1], fwd["radii"], a[2], a[3], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], dc, dl, dd,
            a[16], a[17], a[18], fwd["geom"], fwd["R"], fwd["binning"], fwd["img"], False]


# ---- synchronisation errors are reported, never silent -----------------------------------------
def _sync_error_workspace(dev, P=20000, W=320, H=240, F=15, **ws_kw):
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.scene import make_scene
    sc = make_scene(P, W, H, F, seed=123)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    cam = sc.camera
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
             tanfovy=cam.tanfovy)
    ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 600000, dev, **ws_kw)
    ws.set_scene(sh_degree=sc.sh_degree, **c, **g)
    cot = [t.to(dev) for t in sc.cotangents(1)]
    return sc, ws, cot


def test_corrupted_row_compaction_ticket_is_reported_and_gradients_are_zero(hip):
    """Between a forward and its backward the test overwrites the row compaction's ticket word in the binning buffer: the
    blocks then hold tickets whose predecessors do not exist, their look-back runs into its spin bound.  The library must not
    hang, must not return gradients computed from garbage row offsets, and must say so: status_dev[1] ==
    OLSR_STATUS_SYNC_ERROR on the sync-free path (zero gradients), OLSR_ERR_DEVICE from the synchronising olsr_backward."""
    from online_lang_splatting_amd._lib import lib
    dev = torch.device(DEV)
    # (rows_in_forward=False: the compaction is the backward's own first launch, so a word corrupted between the forward and
    #  the backward reaches it; the default workspace compacts in the forward's last launch, covered below)
    sc, ws, cot = _sync_error_workspace(dev, rows_in_forward=False)
    L = lib()
    L.olsr_debug_sync_fault(-1, 2000)   # give up after 2000 polls instead of 2^22 (keeps the test short)
    try:
        ws.forward()
        good = {k: v.clone() for k, v in ws.backward(*cot).items()}
        assert ws.backward_status()[0] > 0 and float(good["dL_dmeans3D"].abs().max()) > 0
        ws.forward()
        sync = hip.state_field("binning", ws.binning, "row_sync", R=ws.capacity, F=sc.F, dtype=torch.int32, count=2)
        sync[0] = 3   # tickets now start at 3: blocks 0..2 never publish
        g = ws.backward(*cot)
        torch.cuda.synchronize()
        assert int(ws.bwd_status.cpu()[1]) == 2
        with pytest.raises(RuntimeError, match="synchronisation error"):
            ws.backward_status()
        for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dlanguage", "dL_dmeans2D"):
            assert float(g[k].abs().max()) == 0.0, k
        # the next frame on the same workspace is healthy again (the forward re-zeroes its synchronisation words)
        ws.forward()
        again = ws.backward(*cot)
        assert ws.backward_status()[1] is False
        for k in good:
            assert torch.equal(again[k], good[k]), k
    finally:
        L.olsr_debug_sync_fault(0, 0)


@pytest.mark.parametrize("tile,mode", [(15, 0), (15, 1), (16, 0)])
def test_rows_compacted_by_the_forwards_last_launch(hip, tile, mode):
    """olsr_scene.backward_row_capacity (RasterWorkspace(rows_in_forward=True), the default): the forward's last launch is the
    tile order AND the row compaction; the backward skips its own.  Gradients, the status words and a repeated backward on the
    same forward are bit-identical to the workspace whose backward compacts; a too-small row capacity is reported the same
    way; a backward whose scratch does not match the announced capacity is refused."""
    from online_lang_splatting_amd._lib import lib
    dev = torch.device(DEV)
    res = {}
    for rif in (True, False):
        sc, ws, cot = _sync_error_workspace(dev, P=30000, tile=tile, bwd_mode=mode, rows_in_forward=rif)
        ws.forward()
        g1 = {k: v.clone() for k, v in ws.backward(*cot).items()}
        st1 = ws.backward_status()
        g2 = {k: v.clone() for k, v in ws.backward(*cot).items()}   # the same forward, once more
        assert ws.backward_status() == st1
        for k in g1:
            assert torch.equal(g1[k], g2[k]), (rif, k)
        ws.forward()                                                  # and a second frame on the same buffers
        g3 = ws.backward(*cot)
        for k in g1:
            assert torch.equal(g1[k], g3[k]), (rif, k)
        res[rif] = (g1, st1, ws)
    assert res[True][1] == res[False][1] and res[True][1][0] > 0 and not res[True][1][1]
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], res[False][0][k]), k
    assert float(res[True][0]["dL_dmeans3D"].abs().max()) > 0
    L_rows = res[True][1][0]
    # too few rows: reported as an overflow, zero gradients
    sc, tight, cot = _sync_error_workspace(dev, P=30000, tile=tile, bwd_mode=mode, row_capacity=L_rows - 1)
    tight.forward()
    gz = tight.backward(*cot)
    assert tight.backward_status() == (L_rows, True)
    assert all(float(v.abs().max()) == 0.0 for v in gz.values())
    # the backward must be given the scratch the forward was told about
    ws = res[True][2]
    ws.forward()
    ws.row_capacity -= 1
    try:
        with pytest.raises(RuntimeError, match="backward_row_capacity"):
            ws.backward(*cot)
    finally:
        ws.row_capacity += 1
    assert lib().olsr_last_error() is not None
    # ... and the FORWARD must really have compacted them for that scratch: a forward issued without the
    # announcement, then a backward that claims it — stale rowbase / counters — hands out zero gradients and says so, and so does
    # a backward that announces another capacity than the forward compacted for
    def backward_with_scene_capacity(w, cap):
        w._scene.backward_row_capacity = cap
        return w.backward(*cot)
    plain = res[False][2]                      # (rows_in_forward=False: its forward announces nothing)
    plain.forward()
    gz = backward_with_scene_capacity(plain, plain.row_capacity)
    torch.cuda.synchronize()
    assert int(plain.bwd_status.cpu()[1]) == 1 and all(float(v.abs().max()) == 0.0 for v in gz.values())
    plain._scene.backward_row_capacity = 0
    g_ok = plain.backward(*cot)                # the honest backward on the same forward is healthy
    assert plain.backward_status() == res[False][1] and float(g_ok["dL_dmeans3D"].abs().max()) > 0
    ws.forward()                               # compacted for ws.row_capacity ...
    keep_cap, keep_scratch = ws.row_capacity, ws.scratch
    ws.row_capacity += 4096                    # ... but the backward announces (and brings) a larger scratch
    ws.scratch = torch.empty(lib().olsr_backward_scratch_bytes(ws.row_capacity, ws.F), dtype=torch.uint8, device=dev)
    try:
        gz = backward_with_scene_capacity(ws, ws.row_capacity)
        torch.cuda.synchronize()
        assert int(ws.bwd_status.cpu()[1]) == 1 and all(float(v.abs().max()) == 0.0 for v in gz.values())
    finally:
        ws.row_capacity, ws.scratch = keep_cap, keep_scratch
        ws._scene.backward_row_capacity = keep_cap
    ws.forward()
    g_ok = ws.backward(*cot)
    for k in res[True][0]:
        assert torch.equal(res[True][0][k], g_ok[k]), k


@pytest.mark.parametrize("which,bit", [("depth sort", 1), ("tile sort", 2)])
def test_lost_digit_counts_in_a_radix_pass_are_reported(hip, which, bit):
    """Fault injection (olsr_debug_sync_fault): the block holding ticket 0 of the first depth / tile pass never publishes its
    digit counts — what its successors see when a status word is lost mid-frame.  The forward reports
    OLSR_STATUS_SYNC_ERROR in num_rendered_dev[1], the backward zero gradients and the same status; through the reference's
    API the gradients are zeros and the library's next call raises (an error detected on the device surfaces asynchronously)."""
    from online_lang_splatting_amd._lib import lib
    dev = torch.device(DEV)
    # (more than one block per pass: 20 000 keys would fit one block of the depth sort)
    sc, ws, cot = _sync_error_workspace(dev, P=60000)
    L = lib()
    L.olsr_debug_sort_knobs(2, -1, -1)   # 2 keys per thread: 2048 keys per block, so every pass has several blocks
    L.olsr_debug_sync_fault(bit, 2000)
    try:
        ws.forward()
        torch.cuda.synchronize()
        assert int(ws.num_rendered.cpu()[1]) == 2, which
        with pytest.raises(RuntimeError, match="synchronisation error"):
            ws.rendered()
        g = ws.backward(*cot)
        torch.cuda.synchronize()
        assert int(ws.bwd_status.cpu()[1]) == 2 and float(g["dL_dmeans3D"].abs().max()) == 0.0
        # the drop-in API: forward (synchronising) + backward -> RuntimeError carrying the library's message
        a = fwd_args(sc, dev)
        r = hip.rasterize_language_gaussians(*a)
        b = [a[0], a[1], r[3], a[2], a[3], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], cot[0], cot[1], cot[2],
             a[16], a[17], a[18], r[4], r[0], r[5], r[6], False]
        g2 = hip.rasterize_language_gaussians_backward(*b)   # (may be issued before the GPU has reached the error)
        torch.cuda.synchronize()
        assert float(g2[4].abs().max()) == 0.0                 # dL_dmeans3D: zeros, not garbage
        # the pending error belongs to THIS device and stream (round 6): a healthy call on another stream is not failed by it
        L.olsr_debug_sync_fault(0, 0)
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            r_side = hip.rasterize_language_gaussians(*a)
        side.synchronize()
        assert r_side[0] > 0
        # ... and the first library call on the stream of the broken frame fails, the way an asynchronous HIP error surfaces
        with pytest.raises(RuntimeError, match="synchronisation error"):
            hip.rasterize_language_gaussians(*a)
        hip.rasterize_language_gaussians(*a)                     # (reported once)
    finally:
        L.olsr_debug_sync_fault(0, 0)
        L.olsr_debug_sort_knobs(0, -1, -1)
    ws.forward()
    ws.backward(*cot)
    assert ws.rendered()[1] is False and ws.backward_status()[1] is False


def test_per_view_tile_orders_of_the_dropin_entry_never_change_a_result(hip):
    """olsr_forward keeps, per stream, one launch-order hint per VIEW it has seen (16 slots, nearest view matrix, least
    recently used recycled).  Twenty views in turn — more than there are slots — then the first ones again: every call's
    images, radii and n_touched equal what a fresh workspace renders for that view without any hint."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.scene import arc_cameras
    dev = torch.device(DEV)
    W, H, F = 200, 150, 15
    sc = make_scene(8000, W, H, F, seed=77)
    cams = arc_cameras(W, H, n=20)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    ws = RasterWorkspace(sc.P, W, H, F, sc.shs.shape[1], 600000, dev)
    ws.tile_order = None  # (no hint at all on the reference side of the comparison)
    ref = []
    for c in cams:
        cd = dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                  projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                  tanfovy=c.tanfovy)
        ws.set_scene(sh_degree=sc.sh_degree, **cd, **g)
        check_ = __import__("online_lang_splatting_amd._lib", fromlist=["check"]).check
        lib_ = __import__("online_lang_splatting_amd._lib", fromlist=["lib"]).lib()
        import ctypes as C
        o = ws.out
        check_(lib_.olsr_forward_async(C.byref(ws._scene), ws.geom.data_ptr(), ws.binning.data_ptr(), ws.capacity,
                                       ws.img.data_ptr(), o["color"].data_ptr(), o["language"].data_ptr(),
                                       o["depth"].data_ptr(), o["opacity"].data_ptr(), o["radii"].data_ptr(),
                                       o["n_touched"].data_ptr(), ws.num_rendered.data_ptr(), None, ws._stream()))
        ref.append({k: v.clone() for k, v in o.items()})
    order = list(range(20)) + [0, 1, 2, 19, 0, 10, 0]
    for v in order:
        c = cams[v]
        sc.camera = c
        a = fwd_args(sc, dev)
        r = hip.rasterize_language_gaussians(*a)
        assert torch.equal(r[1], ref[v]["color"]) and torch.equal(r[2], ref[v]["language"]), v
        assert torch.equal(r[3], ref[v]["radii"]) and torch.equal(r[7].reshape(-1), ref[v]["depth"].reshape(-1)), v
        assert torch.equal(r[9], ref[v]["n_touched"]), v


def test_adam_step_over_several_buckets_equals_the_summed_bucket(hip):
    """olsr_adam_step_sum: the lane buckets of a step summed inside the Adam kernel, in list order — parameters and moments
    bit-identical to adding the buckets into the first one and stepping on it (what MappingStep used to do)."""
    from online_lang_splatting_amd.frame_shard import FusedAdam, GradientBucket, GradLayout
    dev = torch.device(DEV)
    P, M, F = 5003, 1, 15
    lay = GradLayout(M, F)
    gen = torch.Generator().manual_seed(5)
    buckets = [GradientBucket(P, lay, dev) for _ in range(4)]
    for i, b in enumerate(buckets):
        x = torch.randn(P, lay.width, generator=gen)
        x[torch.rand(P, generator=gen) < 0.9] = 0.0   # mostly empty rows, as in a real step
        b.flat.copy_(x.to(dev))
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)

    def params():
        g = torch.Generator().manual_seed(9)
        return dict(means3D=torch.randn(P, 3, generator=g).to(dev), shs=torch.randn(P, M, 3, generator=g).to(dev),
                    opacities=torch.randn(P, 1, generator=g).to(dev), scales=torch.randn(P, 3, generator=g).to(dev),
                    rotations=torch.randn(P, 4, generator=g).to(dev), language=torch.randn(P, F, generator=g).to(dev))
    pa, pb = params(), params()
    a, b = FusedAdam(P, lay, dev), FusedAdam(P, lay, dev)
    total = GradientBucket(P, lay, dev)
    for step in range(3):
        total.flat.copy_(buckets[0].flat)
        for k in buckets[1:]:
            total.flat.add_(k.flat)
        a.step(total, pa, lrs)
        b.step(buckets, pb, lrs)
    torch.cuda.synchronize()
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
