"""MI355X-native language-Gaussian rasterizer (hot path of rpng/online_lang_splatting).

`from online_lang_splatting_amd import GaussianRasterizationSettings, GaussianRasterizer,
LanguageGaussianRasterizer` — or, as a drop-in, `import diff_gaussian_rasterization` (the
top-level shim package re-exports the same names).

Importing this package loads no native code; the first rasterizer call loads libolsr.so and
raises if it is missing (there is no CPU fallback).
"""
from ._abi import BINNING_ELLIPSE, BINNING_RECT, BWD_EXACT, BWD_REFERENCE  # noqa: F401
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                         LanguageGaussianRasterizer, rasterize_gaussians, rasterize_language_gaussians)

from .renderer import render  # noqa: F401,E402  (the caller-side façade: gaussian_renderer.render)

__all__ = ["render", "GaussianRasterizationSettings", "GaussianRasterizer", "LanguageGaussianRasterizer",
           "rasterize_gaussians", "rasterize_language_gaussians", "BWD_REFERENCE", "BWD_EXACT", "set_backward_mode",
           "set_tile", "BINNING_RECT", "BINNING_ELLIPSE", "set_binning"]


def set_backward_mode(mode):
    """BWD_REFERENCE (default, what the shipped reference computes) or BWD_EXACT (true gradient)."""
    from . import _C
    if mode not in (BWD_REFERENCE, BWD_EXACT):
        raise ValueError("mode must be BWD_REFERENCE or BWD_EXACT")
    _C.BWD_MODE = mode


def set_binning(binning):
    """BINNING_ELLIPSE (default): a Gaussian is listed only in the tiles its alpha >= 1/255 ellipse reaches —
    identical images and gradients, about half the instances.  BINNING_RECT: the reference's bounding-square
    lists (getRect, CR/auxiliary.h:46-56), for bit-identical num_rendered / point lists."""
    from . import _C
    if binning not in (BINNING_RECT, BINNING_ELLIPSE):
        raise ValueError("binning must be BINNING_RECT or BINNING_ELLIPSE")
    _C.BINNING = binning


def set_tile(tile):
    """Logical tile edge: 15 (reference, CR/config.h:17-18) or 16 (upstream 3DGS / MonoGS)."""
    from . import _C
    if tile not in (15, 16):
        raise ValueError("tile must be 15 or 16")
    _C.TILE = tile
