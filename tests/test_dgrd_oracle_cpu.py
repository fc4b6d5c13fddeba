"""CPU checks around row f4 (no GPU): the DGR-D oracle composition (tests/dgrd_oracle.py) and the Python surface of
the disentangled shim package."""
import inspect

import pytest
import torch

import dgrd_oracle as D
from test_gpu_disentangled import _cot, _scene2


def test_radii_rule_and_language_only_gradients():
    s = _scene2(1500, 173, 131, 3, 12, border=True)
    fo, saved = D.forward(s)
    r1, r2 = fo["raw_radii"]
    only2, only1 = (r1 < 0) & (r2 > 0), (r2 < 0) & (r1 > 0)
    assert int(only2.sum()) > 0 and int(only1.sum()) > 0
    # DGR-D forward.cu:391-431: both radii are written as soon as one square covers a tile
    assert bool((fo["radii"][only2] > 0).all()) and bool((fo["radii_lang"][only1] > 0).all())
    assert bool(((fo["radii"] > 0) == (fo["radii_lang"] > 0)).all())
    # a set that covers no tile emits nothing and is touched by no pixel
    assert int(fo["n_touched"][only2].sum()) == 0 and int(fo["n_touched_lang"][only1].sum()) == 0
    dc, dl, dd = _cot(s, 3, 12)
    g = D.backward(s, saved, torch.zeros_like(dc), dl, torch.zeros_like(dd))
    for k in ("means2D", "means3D", "rho", "theta", "scales", "rotations", "opacities", "sh", "colors"):
        assert float(g[k].abs().max()) == 0.0, k  # computeCov2DCUDA_no_tau: the language set moves no mean, no pose
    for k in ("language", "opacities_lang", "scales_lang", "rotations_lang"):
        assert float(g[k].abs().max()) > 0.0, k
    D.release(saved)


def test_disentangled_shim_surface():
    import diff_gaussian_rasterization_disentangle as pkg
    from online_lang_splatting_amd import disentangled
    assert pkg.LanguageGaussianRasterizer is disentangled.LanguageGaussianRasterizer
    # argument order of DGR-D/diff_gaussian_rasterization/__init__.py:528-545
    names = list(inspect.signature(pkg.LanguageGaussianRasterizer.forward).parameters)
    assert names == ["self", "means3D", "means2D", "opacities", "opacities_lang", "shs", "colors_precomp",
                     "language_precomp", "scales", "scales_lang", "rotations", "rotations_lang", "cov3D_precomp",
                     "cov3D_precomp_lang", "theta", "rho"]
    assert pkg.GaussianRasterizationSettings._fields[:5] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg")
    rast = pkg.LanguageGaussianRasterizer(raster_settings=None)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(m, m, m[:, :1], m[:, :1], scales=m, rotations=torch.zeros(4, 4), scales_lang=m,
             rotations_lang=torch.zeros(4, 4), language_precomp=m)
    with pytest.raises(Exception, match="for language"):
        rast(m, m, m[:, :1], m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), language_precomp=m)
    assert disentangled.TILE == 16
