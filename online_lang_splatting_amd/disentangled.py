"""The disentangled-language rasterizer on the MI355X-native library.

Reference: DGR-D = submodules/diff-gaussian-rasterization-disentangle-optim (its package is also called
`diff_gaussian_rasterization`; the shim for it here is `diff_gaussian_rasterization_disentangle`).  A Gaussian carries
TWO opacity / scale / rotation (or 3D covariance) sets over one mean: the first composites colour and depth, the
second ("_lang") composites the language features.  Public surface and argument order follow
DGR-D/diff_gaussian_rasterization/__init__.py (settings :444-457, modules :459-666, autograd bridge :213-441).

What DGR-D computes, and how it maps onto the two rasterizers of this library (both run with DGR-D's 16x16 tiles,
DGR-D/cuda_rasterizer/config.h:17-18):

  forward   languagePreprocessCUDA (DGR-D/cuda_rasterizer/forward.cu:262-433) evaluates both covariance sets of a
            Gaussian; binning, sorting and tile ranges run once per set (rasterizer_impl.cu:489-565) and
            language_renderCUDA (forward.cu:437-655) holds two independent compositing loops — colour + depth over
            the first set's lists with the first set's conics, language over the second's.  That is
              pass 1 = the RGB rasterizer (F = 0) with (opacities, scales, rotations | cov3D_precomp)
              pass 2 = the language rasterizer with (opacities_lang, scales_lang, rotations_lang | cov3D_precomp_lang),
                       of which only the language image, the opacity image, n_touched and radii are kept.
            One coupling: preprocess returns early only if BOTH bounding squares cover no tile (forward.cu:391-397)
            and then writes both radii (:421-431), so a set whose square covers no tile still reports its radius when
            the other set is visible.  The passes run with OLSR_FLAG_SIGNED_EMPTY_RADII and `_merge_radii` applies
            the rule.  (DGR-D also keeps going when exactly one determinant is zero, :375-378; a 2D covariance with
            0.3 added to its diagonal has det >= 0.09, so that branch is not reachable with finite inputs.)
  backward  language_render_cuda (DGR-D/cuda_rasterizer/backward.cu:1052-1428): the colour loop is the RGB
            rasterizer's backward (skip-guarded recursions, 256-lane block sum); the language loop is the language
            rasterizer's loop with the colour / depth terms removed — the language recursion is not skip-guarded
            (:1385-1393) and the feature gradient is taken from thread 0 of the tile (:1423-1425), the two quirks
            OLSR_BWD_REFERENCE reproduces — and it produces NO mean gradient.  BACKWARD::language_preprocess
            (:1505-1622) runs the full chain for the first set and, for the second, computeCov2DCUDA_no_tau
            (:354-436): dL_dconic_lang -> dL_dcov3D_lang -> scale_lang / rotation_lang only, no dL_dmean3D, no
            dL_dtau.  Hence
              pass 1 backward: every gradient of the RGB rasterizer (means2D, means3D, sh / colours, opacities,
                               scales, rotations, cov3D, tau)
              pass 2 backward: the language rasterizer's backward with zero colour / depth cotangents, of which
                               dL_dlanguage, dL_dopacity, dL_dcov3D, dL_dscales, dL_drotations are kept and the mean
                               and pose gradients dropped.
"""
import torch

from . import _C, _abi
from .rasterizer import (GaussianRasterizationSettings, _check_exclusive, _cotangent, _or_empty,  # noqa: F401
                         _RasterizerBase, _settings_args, _split_tau)

TILE = 16  # BLOCK_X = BLOCK_Y = 16, DGR-D/cuda_rasterizer/config.h:17-18
LANGUAGE_CHANNELS = 3  # NUM_LANGUAGE_CHANNELS as shipped, DGR-D/cuda_rasterizer/config.h:16 (any supported F works here)


def _cfg():
    return (TILE, _C.BWD_MODE, _C.BINNING)


def _merge_radii(r1, r2):
    """radii, radii_lang of DGR-D from the two passes' signed radii (> 0 visible, < 0 bounding square without a tile,
    0 outside the frustum): both are reported as soon as one set is visible (DGR-D forward.cu:391-397, :421-431)."""
    either = (r1 > 0) | (r2 > 0)
    zero = torch.zeros_like(r1)
    return torch.where(either, r1.abs(), zero), torch.where(either, r2.abs(), zero)


class _RasterizeGaussians16(torch.autograd.Function):
    """DGR-D's RGB-only rasterizer (rasterize_gaussians, DGR-D __init__.py:86-211): the one of rasterizer.py with
    16x16 tiles."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, rs):
        cfg = _cfg()
        R, color, _l, radii, geom, binning, img, depth, opacity, n_touched = _C._forward(
            0, rs.bg, means3D, colors_precomp, None, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            *_settings_args(rs), rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug,
            cfg=cfg)
        ctx.rs, ctx.R, ctx.cfg = rs, R, cfg
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii, n_touched)
        ctx.set_materialize_grads(False)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_opacity, g_n_touched):
        rs = ctx.rs
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        g_color = _cotangent(g_color, (3, H, W), means3D)  # (a missing depth cotangent stays None: NULL for the library)
        (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot, _g_tau,
         tau_sum) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            *_settings_args(rs), g_color, g_depth, sh, rs.sh_degree, rs.campos, geom, ctx.R, binning, img, rs.debug,
            cfg=ctx.cfg, with_tau_sum=True)
        g_theta, g_rho = _split_tau(tau_sum)
        return g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rot, g_cov3D, g_theta, g_rho, None


class _RasterizeLanguageGaussiansDisentangled(torch.autograd.Function):
    """DGR-D __init__.py:213-441; inputs and gradients in its order."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, language_precomp, opacities, opacities_lang, scales,
                scales_lang, rotations, rotations_lang, cov3Ds_precomp, cov3Ds_precomp_lang, theta, rho, rs):
        cfg = _cfg()
        flags = _abi.FLAG_SIGNED_EMPTY_RADII
        common = (*_settings_args(rs), rs.image_height, rs.image_width)
        # pass 1: colour + depth over the first set
        R1, color, _l, r1, geom1, bin1, img1, depth, opacity, n_touched = _C._forward(
            0, rs.bg, means3D, colors_precomp, None, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            *common, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, cfg=cfg, flags=flags)
        # pass 2: language over the second set (its colour / depth images are by-products: constant colours, no SH)
        dummy_rgb = torch.zeros(means3D.shape[0], 3, dtype=torch.float32, device=means3D.device)
        R2, _c, language, r2, geom2, bin2, img2, _d, opacity_lang, n_touched_lang = _C._forward(
            language_precomp.shape[1], rs.bg, means3D, dummy_rgb, language_precomp, opacities_lang, scales_lang,
            rotations_lang, rs.scale_modifier, cov3Ds_precomp_lang, *common, None, 0, rs.campos, rs.prefiltered,
            rs.debug, cfg=cfg, flags=flags)
        radii, radii_lang = _merge_radii(r1, r2)
        ctx.rs, ctx.R1, ctx.R2, ctx.cfg = rs, R1, R2, cfg
        ctx.save_for_backward(colors_precomp, language_precomp, means3D, scales, scales_lang, rotations, rotations_lang,
                              cov3Ds_precomp, cov3Ds_precomp_lang, r1.clamp(min=0), r2.clamp(min=0), sh, dummy_rgb,
                              geom1, bin1, img1, geom2, bin2, img2)
        ctx.mark_non_differentiable(radii, radii_lang, n_touched, n_touched_lang)
        ctx.set_materialize_grads(False)
        return color, language, radii, radii_lang, depth, opacity, opacity_lang, n_touched, n_touched_lang

    @staticmethod
    def backward(ctx, g_color, g_language, g_radii, g_radii_lang, g_depth, g_opacity, g_opacity_lang, g_nt, g_ntl):
        rs = ctx.rs
        (colors_precomp, language_precomp, means3D, scales, scales_lang, rotations, rotations_lang, cov3Ds_precomp,
         cov3Ds_precomp_lang, radii1, radii2, sh, dummy_rgb, geom1, bin1, img1, geom2, bin2, img2) = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        g_color = _cotangent(g_color, (3, H, W), means3D)  # (a missing depth cotangent stays None: NULL for the library)
        g_language = _cotangent(g_language, (language_precomp.shape[1], H, W), means3D)
        (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot, _g_tau,
         tau_sum) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii1, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            *_settings_args(rs), g_color, g_depth, sh, rs.sh_degree, rs.campos, geom1, ctx.R1, bin1, img1, rs.debug,
            cfg=ctx.cfg, with_tau_sum=True)
        # the language loop sees no colour / depth cotangent (DGR-D backward.cu:1337-1428); its mean and pose
        # gradients do not exist in DGR-D (computeCov2DCUDA_no_tau, :354-436) and are dropped here
        (_m2, _c, g_language_precomp, g_opac_lang, _m3, g_cov3D_lang, _sh, g_scales_lang, g_rot_lang,
         _tau) = _C.rasterize_language_gaussians_backward(
            rs.bg, means3D, radii2, dummy_rgb, language_precomp, scales_lang, rotations_lang, rs.scale_modifier,
            cov3Ds_precomp_lang, *_settings_args(rs), torch.zeros_like(g_color), g_language, None,
            None, 0, rs.campos, geom2, ctx.R2, bin2, img2, rs.debug, cfg=ctx.cfg)
        g_theta, g_rho = _split_tau(tau_sum)
        return (g_means3D, g_means2D, g_sh, g_colors, g_language_precomp, g_opac, g_opac_lang, g_scales, g_scales_lang,
                g_rot, g_rot_lang, g_cov3D, g_cov3D_lang, g_theta, g_rho, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    return _RasterizeGaussians16.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                       cov3Ds_precomp, theta, rho, raster_settings)


def rasterize_language_gaussians(means3D, means2D, sh, colors_precomp, language_precomp, opacities, opacities_lang,
                                 scales, scales_lang, rotations, rotations_lang, cov3Ds_precomp, cov3Ds_precomp_lang,
                                 theta, rho, raster_settings):
    return _RasterizeLanguageGaussiansDisentangled.apply(
        means3D, means2D, sh, colors_precomp, language_precomp, opacities, opacities_lang, scales, scales_lang,
        rotations, rotations_lang, cov3Ds_precomp, cov3Ds_precomp_lang, theta, rho, raster_settings)


class GaussianRasterizer(_RasterizerBase):
    """DGR-D __init__.py:459-502."""

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), opacities,
                                   _or_empty(scales), _or_empty(rotations), _or_empty(cov3D_precomp), _or_empty(theta),
                                   _or_empty(rho), self.raster_settings)


class LanguageGaussianRasterizer(_RasterizerBase):
    """DGR-D __init__.py:504-666.  Returns (colors, language, radii, radii_lang, depth, opacity, opacity_lang,
    n_touched, n_touched_lang)."""

    def forward(self, means3D, means2D, opacities, opacities_lang, shs=None, colors_precomp=None, language_precomp=None,
                scales=None, scales_lang=None, rotations=None, rotations_lang=None, cov3D_precomp=None,
                cov3D_precomp_lang=None, theta=None, rho=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        if ((scales_lang is None or rotations_lang is None) and cov3D_precomp_lang is None) or (
                (scales_lang is not None or rotations_lang is not None) and cov3D_precomp_lang is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance "
                            "for language!")  # DGR-D __init__.py:561-567
        if language_precomp is None or language_precomp.dim() != 2:
            raise Exception("language_precomp must be [P, F]")
        return rasterize_language_gaussians(
            means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), language_precomp, opacities, opacities_lang,
            _or_empty(scales), _or_empty(scales_lang), _or_empty(rotations), _or_empty(rotations_lang),
            _or_empty(cov3D_precomp), _or_empty(cov3D_precomp_lang), _or_empty(theta), _or_empty(rho),
            self.raster_settings)
