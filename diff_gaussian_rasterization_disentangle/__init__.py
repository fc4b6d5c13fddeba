"""Drop-in for the reference's disentangled rasterizer package (submodules/diff-gaussian-rasterization-disentangle-optim,
which installs under the same import name `diff_gaussian_rasterization` as the plain one — hence a second shim name):

    from diff_gaussian_rasterization_disentangle import (GaussianRasterizationSettings, GaussianRasterizer,
                                                         LanguageGaussianRasterizer)

resolves to online_lang_splatting_amd.disentangled (two opacity / scale / rotation sets per Gaussian, 16x16 tiles).
"""
from online_lang_splatting_amd import _C  # noqa: F401
from online_lang_splatting_amd.disentangled import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                                    LanguageGaussianRasterizer, rasterize_gaussians,
                                                    rasterize_language_gaussians)
