"""olsr_mapping_loss (HIP, through the C-ABI) against the CPU oracle and the reference-generated golden
vectors.  Tolerance: the per-pixel gradients are products of exact signs / masks and constants (a few ulp);
the scalar sums are accumulated in a different order than torch's (block trees, final add in double):
1e-5 relative is asserted, ~1e-7 is observed."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss_oracle  # noqa: E402
from test_loss_oracle_golden import golden_cases, golden_tracking_cases, run_oracle  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-5


def _close(a, b, name, atol_scale=1.0):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= RTOL * scale * atol_scale, f"{name}: abs {err:.3e} vs scale {scale:.3e}"


def _run_hip(c, hip_losses, **kw):
    dev = torch.device(DEV)
    mv = lambda t: None if t is None else t.to(dev)
    exposure = None if c.get("a") is None else torch.cat([c["a"].reshape(1), c["b"].reshape(1)]).float().to(dev)
    return hip_losses.mapping_loss(mv(c["image"]), mv(c["depth"]), mv(c.get("lang")), mv(c["gt_image"]), mv(c["gt_depth"]),
                                   mv(c.get("gt_lang")), exposure, **kw)


def test_golden_vectors_from_the_reference(hip):
    from online_lang_splatting_amd import losses
    for c in golden_cases():
        o = _run_hip(c, losses, alpha=float(c["alpha"]), rgb_boundary_threshold=float(c["thr"]), lamda_lang=1.0,
                     initialization=bool(int(c["init"])))
        _close(o["loss"][0], c["loss"], "loss")
        _close(o["loss"][1] + o["loss"][2], c["loss_map"], "loss_map")
        _close(o["loss"][3], c["loss_lang"], "loss_lang")
        _close(o["dL_dimage"], c["d_image"], "dL_dimage")
        _close(o["dL_ddepth"], c["d_depth"], "dL_ddepth")
        _close(o["dL_dlanguage"], c["d_lang"], "dL_dlanguage")
        # exposure gradients are sums of +-1 over ~1e3 pixels with heavy cancellation: compare on the scale of
        # the sum of magnitudes (alpha / 3 here), not of the cancelled result
        assert abs(float(o["dL_dexposure"][0]) - float(c["d_a"])) <= 1e-6
        assert abs(float(o["dL_dexposure"][1]) - float(c["d_b"])) <= 1e-6
        # decisions are exact: the zero pattern (masks, ties) is identical
        assert torch.equal(o["dL_dimage"].cpu() == 0, c["d_image"] == 0)
        assert torch.equal(o["dL_ddepth"].cpu() == 0, c["d_depth"] == 0)
        assert torch.equal(o["dL_dlanguage"].cpu() == 0, c["d_lang"] == 0)


@pytest.mark.parametrize("F,H,W,lh,lw", [(15, 680, 1200, 192, 192), (32, 135, 241, 192, 192), (0, 97, 33, 0, 0),
                                         (16, 64, 64, 64, 64), (3, 50, 70, 200, 300)])
def test_against_the_oracle(hip, F, H, W, lh, lw):
    """Full-size config 3 frame (1200x680, 15 channels, 192x192 target as in the reference), ragged sizes,
    every supported F, identity-size and down-sampled targets."""
    from online_lang_splatting_amd import losses
    g = torch.Generator().manual_seed(F + H)
    c = dict(image=torch.rand(3, H, W, generator=g), depth=torch.rand(1, H, W, generator=g) * 5,
             lang=torch.randn(F, H, W, generator=g) * 0.3 if F else None, gt_image=torch.rand(3, H, W, generator=g),
             gt_depth=torch.rand(H, W, generator=g) * 5, gt_lang=torch.randn(F, lh, lw, generator=g) * 0.3 if F else None,
             a=torch.tensor([0.11]), b=torch.tensor([-0.03]))
    c["gt_image"][:, : H // 3] *= 0.001
    c["gt_depth"][:, : W // 4] = 0.0
    kw = dict(alpha=0.95, rgb_boundary_threshold=0.01, lamda_lang=1.0)
    ref = loss_oracle.mapping_loss_and_grads(c["image"], c["depth"], c["lang"], c["gt_image"], c["gt_depth"], c["gt_lang"],
                                             c["a"], c["b"], **kw)
    o = _run_hip(c, losses, **kw)
    _close(o["loss"][0], ref["loss"], "loss")
    _close(o["loss"][1], ref["rgb"], "rgb")
    _close(o["loss"][2], ref["depth"], "depth")
    _close(o["dL_dimage"], ref["dL_dimage"], "dL_dimage")
    _close(o["dL_ddepth"], ref["dL_ddepth"], "dL_ddepth")
    if F:
        _close(o["loss"][3], ref["lang"], "lang")
        # a bilinear sample that lands within rounding of the rendered value flips a sign: allow a handful
        d = (o["dL_dlanguage"].cpu() - ref["dL_dlanguage"]).abs()
        bad = int((d > RTOL * float(ref["dL_dlanguage"].abs().max())).sum())
        assert bad <= 3, bad
    assert abs(float(o["dL_dexposure"][0]) - float(ref["dL_da"])) <= 2e-6
    assert abs(float(o["dL_dexposure"][1]) - float(ref["dL_db"])) <= 2e-6
    # without a language target the language cotangent is zero-filled; initialization skips the exposure
    c2 = dict(c, gt_lang=None)
    o2 = _run_hip(c2, losses, initialization=True, **kw)
    ref2 = loss_oracle.mapping_loss_and_grads(c["image"], c["depth"], c["lang"], c["gt_image"], c["gt_depth"], None,
                                              c["a"], c["b"], initialization=True, **kw)
    _close(o2["loss"][0], ref2["loss"], "loss (init)")
    _close(o2["dL_dimage"], ref2["dL_dimage"], "dL_dimage (init)")
    assert float(o2["dL_dlanguage"].abs().max()) == 0 if F else True
    assert float(o2["dL_dexposure"].abs().max()) == 0


def test_loss_feeds_the_rasterizer_backward(hip):
    """End to end on the GPU: render -> olsr_mapping_loss -> olsr_backward; the cotangents are consumed as they
    are, and a finite-difference step along the gradient of the opacities decreases the loss."""
    from online_lang_splatting_amd import losses
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.scene import make_scene
    dev = torch.device(DEV)
    sc = make_scene(5000, 160, 120, 15, seed=17)
    cam = sc.camera
    ws = RasterWorkspace(sc.P, 160, 120, 15, sc.shs.shape[1], 300000, dev)
    g = torch.Generator().manual_seed(3)
    gt_image, gt_depth = torch.rand(3, 120, 160, generator=g).to(dev), (torch.rand(120, 160, generator=g) * 4).to(dev)
    gt_lang = (torch.randn(15, 48, 48, generator=g) * 0.3).to(dev)
    kw = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), scales=sc.scales.to(dev), rotations=sc.rotations.to(dev),
              shs=sc.shs.to(dev), language=sc.language.to(dev), viewmatrix=cam.world_view_transform.to(dev),
              projmatrix=cam.full_proj_transform.to(dev), projmatrix_raw=cam.projection_matrix.to(dev),
              campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)

    def loss_at(op):
        ws.set_scene(opacities=op, **kw)
        out = ws.forward()
        lo = losses.mapping_loss(out["color"], out["depth"], out["language"], gt_image, gt_depth, gt_lang)
        return out, lo

    op0 = sc.opacities.to(dev)
    out, lo = loss_at(op0)
    grads = ws.backward(lo["dL_dimage"], lo["dL_dlanguage"], lo["dL_ddepth"])
    gop = grads["dL_dopacity"].reshape(op0.shape).clone()
    assert bool(torch.isfinite(gop).all()) and float(gop.abs().max()) > 0
    l0 = float(lo["loss"][0])
    step = 0.02 / float(gop.abs().max())
    _, lo1 = loss_at((op0 - step * gop).clamp(1e-4, 1 - 1e-4))
    assert float(lo1["loss"][0]) < l0


def test_tracking_loss_golden_and_oracle(hip):
    """olsr_tracking_loss against the vectors from the reference's get_loss_tracking and, at full size, the oracle."""
    from online_lang_splatting_amd import losses
    dev = torch.device(DEV)
    for c in golden_tracking_cases():
        expo = torch.cat([c["a"], c["b"]]).float().to(dev)
        o = losses.tracking_loss(c["image"].to(dev), c["depth"].to(dev), c["opacity"].to(dev), c["gt_image"].to(dev),
                                 c["gt_depth"].to(dev), c["grad_mask"].to(dev), expo, alpha=float(c["alpha"]),
                                 rgb_boundary_threshold=float(c["thr"]))
        _close(o["loss"][0], c["loss"], "loss")
        _close(o["dL_dimage"], c["d_image"], "dL_dimage")
        _close(o["dL_ddepth"], c["d_depth"], "dL_ddepth")
        assert abs(float(o["dL_dexposure"][0]) - float(c["d_a"])) <= 1e-6
        assert abs(float(o["dL_dexposure"][1]) - float(c["d_b"])) <= 1e-6
        assert torch.equal(o["dL_dimage"].cpu() == 0, c["d_image"] == 0)
        assert torch.equal(o["dL_ddepth"].cpu() == 0, c["d_depth"] == 0)
    H, W = 680, 1200
    g = torch.Generator().manual_seed(9)
    image, depth = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g) * 5
    opacity = torch.rand(1, H, W, generator=g) * 0.2 + 0.85
    gt_image, gt_depth = torch.rand(3, H, W, generator=g), torch.rand(H, W, generator=g) * 5
    gm = torch.rand(1, H, W, generator=g) > 0.5
    a, b = torch.tensor([0.05]), torch.tensor([0.01])
    ref = loss_oracle.tracking_loss_and_grads(image, depth, opacity, gt_image, gt_depth, gm, a, b)
    o = losses.tracking_loss(image.to(dev), depth.to(dev), opacity.to(dev), gt_image.to(dev), gt_depth.to(dev), gm.to(dev),
                             torch.cat([a, b]).to(dev))
    _close(o["loss"][0], ref["loss"], "loss")
    _close(o["dL_dimage"], ref["dL_dimage"], "dL_dimage")
    _close(o["dL_ddepth"], ref["dL_ddepth"], "dL_ddepth")
    assert abs(float(o["dL_dexposure"][0]) - float(ref["dL_da"])) <= 2e-6


def test_mapping_iterations_reduce_the_loss(hip):
    """The whole loop of utils/slam_backend.py:510-760 on the library alone — render from raw parameters
    (OLSR_ACT_*), olsr_mapping_loss, backward into the bucket, olsr_adam_step — fits a perturbed scene back to
    renders of the original: the summed loss of the views must fall by a clear margin in 25 iterations."""
    from online_lang_splatting_amd import _abi, losses
    from online_lang_splatting_amd.frame_shard import FusedAdam, GradLayout, GradientBucket, RasterWorkspace
    from online_lang_splatting_amd.scene import arc_cameras, make_scene
    dev = torch.device(DEV)
    W, H, F = 160, 120, 15
    sc = make_scene(4000, W, H, F, seed=41)
    M = sc.shs.shape[1]
    cams = arc_cameras(W, H, 3)
    camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in cams]
    truth = dict(means3D=sc.means3D.to(dev).contiguous(), shs=sc.shs.to(dev).contiguous(),
                 opacities=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)).to(dev).contiguous(),
                 scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev).contiguous(),
                 language=sc.language.to(dev).contiguous())
    ws = RasterWorkspace(sc.P, W, H, F, M, 400000, dev)
    bg = sc.bg.to(dev)

    def render(p, cam):
        ws.set_scene(bg=bg, sh_degree=0, activations=_abi.ACT_ALL, **cam, **p)
        return ws.forward()

    gts = []
    for cam in camd:
        o = render(truth, cam)
        gts.append((o["color"].clone(), o["depth"][0].clone(), o["language"].clone()))   # full-resolution language target
    g = torch.Generator().manual_seed(1)
    params = {k: v.clone() for k, v in truth.items()}
    params["shs"] += 0.3 * torch.randn(params["shs"].shape, generator=g).to(dev)
    params["language"] += 0.3 * torch.randn(params["language"].shape, generator=g).to(dev)
    params["opacities"] -= 0.5
    bucket = GradientBucket(sc.P, GradLayout(M, F), dev)
    adam = FusedAdam(sc.P, GradLayout(M, F), dev)
    lrs = dict(xyz=1e-4, sh_dc=1e-2, sh_rest=5e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=1e-2)
    hist = []
    for it in range(25):
        total = 0.0
        for v, cam in enumerate(camd):
            out = render(params, cam)
            lo = losses.mapping_loss(out["color"], out["depth"], out["language"], *gts[v])
            ws.backward(lo["dL_dimage"], lo["dL_dlanguage"], lo["dL_ddepth"], bucket=bucket, first=(v == 0), bucket_only=True)
            total += float(lo["loss"][0])
        adam.step(bucket, params, lrs)
        hist.append(total)
    assert all(torch.isfinite(t).all() for t in params.values())
    assert hist[-1] < 0.6 * hist[0], (hist[0], hist[-1])


# ---- the loss in the forward composite's epilogue (olsr_forward_async_loss) ---------------------------------------------
def _fused_case(F, W, H, tile, seed, bg=None, lang_hw=(37, 53)):
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.scene import make_scene
    dev = torch.device(DEV)
    sc = make_scene(5000, W, H, F, seed=seed, bg=bg)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=None if F == 0 else sc.language.to(dev))
    cam = sc.camera
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
             tanfovy=cam.tanfovy)
    ws = RasterWorkspace(sc.P, W, H, F, sc.shs.shape[1], 400000, dev, tile=tile)
    ws.set_scene(sh_degree=sc.sh_degree, **c, **g)
    gen = torch.Generator().manual_seed(seed + 1)
    gt_image = torch.rand(3, H, W, generator=gen)
    gt_image[:, : H // 3] *= 0.001                       # below the rgb boundary threshold
    gt_depth = torch.rand(H, W, generator=gen) * 5
    gt_depth[:, : W // 4] = 0.0                          # invalid depth
    gt_lang = torch.nn.functional.normalize(torch.randn(max(F, 1), lang_hw[0], lang_hw[1], generator=gen), dim=0)[:F] if F else None
    grad_mask = (torch.rand(H, W, generator=gen) > 0.3).float()
    exposure = torch.tensor([0.11, -0.03])
    mv = lambda t: None if t is None else t.to(dev).contiguous()
    return ws, mv(gt_image), mv(gt_depth), mv(gt_lang), mv(grad_mask), mv(exposure)


@pytest.mark.parametrize("F,W,H,tile,lang_hw", [(15, 200, 150, 15, (37, 53)), (0, 157, 101, 15, (37, 53)),
                                                (32, 128, 96, 16, (37, 53)), (3, 64, 64, 15, (200, 300)),
                                                (16, 171, 93, 16, (93, 171)), (15, 1200, 680, 15, (192, 192))])
@pytest.mark.parametrize("use_exposure", [True, False])
def test_fused_mapping_loss_equals_the_two_kernel_path(hip, F, W, H, tile, lang_hw, use_exposure):
    """olsr_forward_async_loss against olsr_forward_async + olsr_mapping_loss on the same render: the cotangents are
    bit-identical (same per-pixel source, csrc/olsr_loss_device.h), the loss and the exposure gradient agree to the
    summation order (per-tile instead of per-256-pixel partials), the state the backward reads is the same, and so the
    gradients are bit-identical.  The stand-alone kernel is what the reference-generated goldens pin."""
    from online_lang_splatting_amd import losses
    # (the language target's window of a tile is staged in LDS when it fits — a small target enlarged, the reference's 192 x 192
    #  at 1200 x 680, an identity-size one — and gathered from global memory when it does not: a 200 x 300 target for 64 x 64)
    ws, gt_image, gt_depth, gt_lang, _, exposure = _fused_case(F, W, H, tile, seed=300 + F, bg=torch.tensor([0.2, 0.1, 0.3]),
                                                               lang_hw=lang_hw)
    ex = exposure if use_exposure else None
    out = ws.forward()
    two = losses.mapping_loss(out["color"], out["depth"], out["language"] if F else None, gt_image, gt_depth, gt_lang, ex)
    g2 = {k: v.clone() for k, v in ws.backward(two["dL_dimage"], two["dL_dlanguage"] if F else None, two["dL_ddepth"]).items()}
    images = {k: out[k].clone() for k in ("color", "depth", "opacity", "language")}
    for skip in (False, True):
        for k in images:
            ws.out[k].fill_(-7.0)
        fu = ws.forward_loss(gt_image, gt_depth, gt_lang, ex, skip_images=skip)
        torch.cuda.synchronize()
        for k in images:  # written exactly as by the plain forward, or not at all
            assert torch.equal(ws.out[k], images[k] if not skip else torch.full_like(images[k], -7.0)), (k, skip)
        assert torch.equal(fu["dL_dimage"], two["dL_dimage"]) and torch.equal(fu["dL_ddepth"], two["dL_ddepth"])
        if F:
            assert torch.equal(fu["dL_dlanguage"], two["dL_dlanguage"])
        else:
            assert fu["dL_dlanguage"] is None
        _close(fu["loss"], two["loss"], "loss")
        assert float(two["loss"][0]) > 0 and (F == 0 or float(two["loss"][3]) > 0)
        assert float((fu["dL_dexposure"] - two["dL_dexposure"]).abs().max()) <= 1e-6
        gf = ws.backward(fu["dL_dimage"], fu["dL_dlanguage"], fu["dL_ddepth"])
        for k in g2:
            assert torch.equal(gf[k], g2[k]), k


@pytest.mark.parametrize("F,W,H,tile", [(15, 200, 150, 15), (0, 157, 101, 16)])
@pytest.mark.parametrize("masked", [True, False])
def test_fused_tracking_loss_equals_the_two_kernel_path(hip, F, W, H, tile, masked):
    from online_lang_splatting_amd import losses
    ws, gt_image, gt_depth, _, grad_mask, exposure = _fused_case(F, W, H, tile, seed=400 + F)
    gm = grad_mask if masked else None
    out = ws.forward()
    two = losses.tracking_loss(out["color"], out["depth"], out["opacity"], gt_image, gt_depth, gm, exposure)
    g2 = ws.backward(two["dL_dimage"], None, two["dL_ddepth"], pose_only=True)["dL_dtau_sum"].clone()
    fu = ws.forward_loss(gt_image, gt_depth, None, exposure, gm, tracking=True)
    assert fu["dL_dlanguage"] is None
    assert torch.equal(fu["dL_dimage"], two["dL_dimage"]) and torch.equal(fu["dL_ddepth"], two["dL_ddepth"])
    _close(fu["loss"], two["loss"], "loss")
    assert float((fu["dL_dexposure"] - two["dL_dexposure"]).abs().max()) <= 1e-6
    gf = ws.backward(fu["dL_dimage"], None, fu["dL_ddepth"], pose_only=True)["dL_dtau_sum"]
    assert torch.equal(gf, g2) and float(g2.abs().max()) > 0


def test_fused_loss_argument_checks(hip):
    from online_lang_splatting_amd import _abi
    from online_lang_splatting_amd._lib import OlsrError
    ws, gt_image, gt_depth, gt_lang, _, _ = _fused_case(15, 64, 48, 15, seed=9)
    with pytest.raises(RuntimeError):
        ws.forward_loss(gt_image.cpu(), gt_depth)
    with pytest.raises(RuntimeError):
        ws.forward_loss(gt_image[:, :10].contiguous(), gt_depth)
    ws.flags = _abi.FLAG_FWD_ACCUM_WEIGHT
    ws._scene.flags = _abi.FLAG_FWD_ACCUM_WEIGHT
    with pytest.raises(OlsrError, match="default forward accumulation"):
        ws.forward_loss(gt_image, gt_depth, gt_lang)
