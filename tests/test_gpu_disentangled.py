"""Row f4: the disentangled-language rasterizer (online_lang_splatting_amd/disentangled.py, shim package
diff_gaussian_rasterization_disentangle) against tests/dgrd_oracle.py."""
import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import make_scene
from parity_common import elementwise_report, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene2(P, W, H, F=3, seed=0, border=False):
    """A make_scene scene plus a second opacity / scale / rotation set.  border: a handful of Gaussians sit just outside
    the image with a first set too small to reach it and a second set large enough (and the reverse), the case in
    which DGR-D reports the radius of a set that covers no tile."""
    sc = make_scene(P, W, H, F, seed=seed, max_sh_degree=1)
    g = torch.Generator().manual_seed(4242 + seed)
    s = dict(camera=sc.camera, bg=sc.bg, means3D=sc.means3D.clone(), opacities=sc.opacities, scales=sc.scales.clone(),
             rotations=sc.rotations, shs=sc.shs, sh_degree=sc.sh_degree, language=sc.language)
    s["opacities_lang"] = torch.rand(P, 1, generator=g) * 0.9 + 0.05
    s["scales_lang"] = sc.scales * torch.exp(torch.randn(P, 3, generator=g) * 0.5)
    q = torch.randn(P, 4, generator=g)
    s["rotations_lang"] = q / q.norm(dim=1, keepdim=True)
    if border:
        k = min(64, P // 4)
        z = 3.0
        # make_scene's default camera sits at the origin and looks down +z: pixel x = fx * X / Z + (W - 1) / 2.
        # 4.5 pixels right of the tile grid: the small set (radius 3) reaches no tile, the large one (radius 11) does.
        fx = W / 2.0
        gx = (W + 15) // 16
        x_out = ((gx * 16 + 4.5) - (W - 1) / 2.0) / fx * z
        ys = (torch.rand(k, generator=g) - 0.5) * (H / fx) * z * 0.8
        s["means3D"][:k] = torch.stack([torch.full((k,), x_out), ys, torch.full((k,), z)], 1)
        small, large = 0.0005, 0.08
        s["scales"][: k // 2] = small
        s["scales_lang"][: k // 2] = large
        s["scales"][k // 2: k] = large
        s["scales_lang"][k // 2: k] = small
    return s


def _settings(s, dev):
    from diff_gaussian_rasterization_disentangle import GaussianRasterizationSettings
    cam = s["camera"]
    return GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=s["bg"].to(dev),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        projmatrix_raw=cam.projection_matrix.to(dev), sh_degree=s["sh_degree"], campos=cam.camera_center.to(dev),
        prefiltered=False, debug=False)


def _run_product(s, dev, seed, cot):
    from diff_gaussian_rasterization_disentangle import LanguageGaussianRasterizer
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)  # noqa: E731
    names = ("means3D", "opacities", "opacities_lang", "scales", "scales_lang", "rotations", "rotations_lang", "shs",
             "language")
    L = {n: leaf(s[n]) for n in names}
    means2D = torch.zeros_like(L["means3D"], requires_grad=True)
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    rast = LanguageGaussianRasterizer(raster_settings=_settings(s, dev))
    out = rast(means3D=L["means3D"], means2D=means2D, opacities=L["opacities"], opacities_lang=L["opacities_lang"],
               shs=L["shs"], colors_precomp=None, language_precomp=L["language"], scales=L["scales"],
               scales_lang=L["scales_lang"], rotations=L["rotations"], rotations_lang=L["rotations_lang"],
               cov3D_precomp=None, cov3D_precomp_lang=None, theta=theta, rho=rho)
    color, language, radii, radii_lang, depth, opacity, opacity_lang, n_touched, n_touched_lang = out
    dc, dl, dd = (t.to(dev) for t in cot)
    # the opacity images take part in the loss but, like in the reference, carry no gradient
    loss = (color * dc).sum() + (language * dl).sum() + (depth * dd).sum() + 0.3 * opacity.sum() + 0.2 * opacity_lang.sum()
    loss.backward()
    grads = {n: L[n].grad for n in names}
    grads.update(means2D=means2D.grad, theta=theta.grad, rho=rho.grad)
    fwd = dict(color=color, language=language, radii=radii, radii_lang=radii_lang, depth=depth, opacity=opacity,
               opacity_lang=opacity_lang, n_touched=n_touched, n_touched_lang=n_touched_lang)
    return {k: v.detach().cpu() for k, v in fwd.items()}, {k: v.detach().cpu() for k, v in grads.items()}


def _cot(s, F, seed):
    g = torch.Generator().manual_seed(99 + seed)
    H, W = s["camera"].height, s["camera"].width
    n = float(H * W)
    return (torch.randn(3, H, W, generator=g) / n, torch.randn(F, H, W, generator=g) / n,
            torch.randn(1, H, W, generator=g) / n)


@pytest.mark.parametrize("binning", [_abi.BINNING_RECT, _abi.BINNING_ELLIPSE])
@pytest.mark.parametrize("P,W,H,F,seed,border", [(4000, 200, 150, 3, 11, False), (3000, 173, 131, 3, 12, True),
                                                  (2500, 160, 120, 15, 13, True)])
def test_disentangled_rasterizer_matches_the_oracle(hip, oracle, P, W, H, F, seed, border, binning):
    import dgrd_oracle as D
    dev = torch.device(DEV)
    s = _scene2(P, W, H, F, seed, border)
    cot = _cot(s, F, seed)
    fo, saved = D.forward(s)
    go = D.backward(s, saved, *cot)
    hip.BINNING = binning
    fg, gg = _run_product(s, dev, seed, cot)
    # forward: every image and counter bit for bit
    for k in ("color", "language", "depth", "opacity", "opacity_lang", "radii", "radii_lang", "n_touched",
              "n_touched_lang"):
        assert torch.equal(fg[k], fo[k]), k
    if border:
        r1, r2 = fo["raw_radii"]
        assert int(((r1 < 0) & (r2 > 0)).sum()) > 0 and int(((r2 < 0) & (r1 > 0)).sum()) > 0, "the border case must occur"
        # ... and there the reported radius is the magnitude of the signed one
        m = (r1 < 0) & (r2 > 0)
        assert torch.equal(fg["radii"][m], -r1[m]) and bool((fg["radii"][m] > 0).all())
    # backward: DGR-D's 16 gradients
    pairs = dict(means3D="means3D", means2D="means2D", shs="sh", language="language", opacities="opacities",
                 opacities_lang="opacities_lang", scales="scales", scales_lang="scales_lang", rotations="rotations",
                 rotations_lang="rotations_lang", theta="theta", rho="rho")
    for pk, ok in pairs.items():
        a, b = gg[pk], go[ok]
        assert a is not None, pk
        assert rel_err(a.reshape(-1), b.reshape(-1))[0] <= 1e-4, pk
    # per element too (the tensors are small: a 1e-4 fraction is less than one element, so allow two outliers)
    for pk, ok in (("opacities_lang", "opacities_lang"), ("scales_lang", "scales_lang"), ("language", "language"),
                   ("means3D", "means3D"), ("scales", "scales")):
        r = elementwise_report(gg[pk].reshape(-1), go[ok].reshape(-1))
        assert r["worst"] <= 2e-2 and r["frac_within"] >= 1.0 - max(1e-4, 2.0 / r["n"]), (pk, r)
    D.release(saved)


def test_language_loss_gives_no_mean_or_pose_gradient(hip):
    dev = torch.device(DEV)
    s = _scene2(2000, 160, 120, 3, 21)
    H, W = 120, 160
    zero3, zero1 = torch.zeros(3, H, W), torch.zeros(1, H, W)
    dl = torch.randn(3, H, W, generator=torch.Generator().manual_seed(5)) / (H * W)
    _f, g = _run_product(s, dev, 21, (zero3, dl, zero1))
    for k in ("means3D", "means2D", "theta", "rho", "scales", "rotations", "opacities", "shs"):
        assert float(g[k].abs().max()) == 0.0, k
    for k in ("language", "opacities_lang", "scales_lang", "rotations_lang"):
        assert float(g[k].abs().max()) > 0.0, k


def test_rgb_rasterizer_of_the_disentangled_package_uses_16_pixel_tiles(hip, oracle):
    from diff_gaussian_rasterization_disentangle import GaussianRasterizer
    from parity_common import run_backend
    dev = torch.device(DEV)
    sc = make_scene(3000, 160, 120, 0, seed=31, max_sh_degree=1)
    s = dict(camera=sc.camera, bg=sc.bg, sh_degree=sc.sh_degree)
    fo, _go = run_backend(oracle, sc, None, 3, 16, _abi.BWD_REFERENCE)
    rast = GaussianRasterizer(raster_settings=_settings(s, dev))
    m = sc.means3D.to(dev)
    color, radii, depth, opacity, n_touched = rast(
        means3D=m, means2D=torch.zeros_like(m), opacities=sc.opacities.to(dev), shs=sc.shs.to(dev),
        scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    assert torch.equal(color.cpu(), fo["color"]) and torch.equal(n_touched.cpu(), fo["n_touched"])
    assert torch.equal(depth.cpu(), fo["depth"]) and torch.equal(radii.cpu(), fo["radii"])
    oracle.release(fo["geom"])
