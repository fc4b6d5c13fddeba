"""Drop-in name for the reference's `diff_gaussian_rasterization` package
(submodules/diff-gaussian-rasterization): the callers' import line

    from diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                             LanguageGaussianRasterizer)

(gaussian_splatting/gaussian_renderer/__init__.py:15-19) resolves to the MI355X-native
implementation in online_lang_splatting_amd.
"""
from online_lang_splatting_amd import _C  # noqa: F401  (same attribute name as the reference's extension)
from online_lang_splatting_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                                  LanguageGaussianRasterizer, rasterize_gaussians,
                                                  rasterize_language_gaussians)
