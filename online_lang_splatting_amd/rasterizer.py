"""Drop-in Python API of `diff_gaussian_rasterization` on the MI355X-native library.

Public surface, argument meaning, return arity/order and error behaviour follow
DGR/diff_gaussian_rasterization/__init__.py (settings tuple :405-419, modules :421-576,
autograd bridges :79-202 and :205-403):

    GaussianRasterizationSettings, GaussianRasterizer, LanguageGaussianRasterizer

so gaussian_splatting/gaussian_renderer/__init__.py runs against it unmodified.  Backward
ignores the cotangents of radii / opacity / n_touched exactly like the reference (:296,
:317-345), and returns gradients in the reference's input order.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _settings_args(rs):
    return (rs.viewmatrix, rs.projmatrix, rs.projmatrix_raw, rs.tanfovx, rs.tanfovy)


def _split_tau(tau_sum, theta_shape=None, rho_shape=None):
    """[6] -> (grad_theta, grad_rho).  dL_dtau = [rho | theta] per Gaussian; the reference sums it over P with
    torch.sum (:383-385), here the library's backward already left the sum behind (fixed order, one tiny kernel).
    The reference hands autograd [1, 3] views for parameters of shape [3] (cam_rot_delta / cam_trans_delta), which makes the
    engine launch a sum-to-size reduction per parameter and per backward; views of the INPUT's shape carry the same three
    numbers into .grad without those two kernels (round 5: 10 us of GPU time and their launch gaps per frame)."""
    def shaped(t, shape):
        if shape is not None:
            n = 1
            for d in shape:
                n *= int(d)
            if n == 3:
                return t.view(shape)
        return t.view(1, -1)
    return shaped(tau_sum[3:], theta_shape), shaped(tau_sum[:3], rho_shape)


def _cotangent(g, shape, like):
    """Autograd hands None for an output the loss never touched (the Functions do not materialise zero gradients: three
    memsets per backward for radii / opacity / n_touched, which are ignored anyway).  The library takes "no language
    cotangent" and "no depth cotangent" as NULL pointers (olsr_backward: the tracking loss has no language term,
    utils/slam_utils.py:92-121, and then the RGB instantiation of the composite backward runs); only the colour
    cotangent, which also carries the image size, is materialised."""
    return g if g is not None else torch.zeros(shape, dtype=torch.float32, device=like.device)


def _cpu_deep_copy(args):
    """cpu_deep_copy_tuple, DGR/diff_gaussian_rasterization/__init__.py:17-19"""
    return tuple(a.cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _debug_call(debug, fn, args, kwargs, dump, what):
    """raster_settings.debug of the RGB rasterizer (DGR/diff_gaussian_rasterization/__init__.py:121-130, 173-183): the arguments
    are copied to the CPU before the call "before they can be corrupted", and an exception leaves them behind as
    snapshot_fw.dump / snapshot_bw.dump in the working directory before it is re-raised.  (The reference's LANGUAGE rasterizer
    has the same lines commented out, :270-281, 357-368: it writes no dump, and neither does this one.)"""
    if not debug:
        return fn(*args, **kwargs)
    cpu_args = _cpu_deep_copy(args)
    try:
        return fn(*args, **kwargs)
    except Exception:
        torch.save(cpu_args, dump)
        print(f"\nAn error occured in {what}. " + ("Please forward snapshot_fw.dump for debugging." if what == "forward"
                                                   else "Writing snapshot_bw.dump for debugging.\n"))
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """RGB + depth + opacity rasterization (reference :79-202)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                raster_settings):
        rs = raster_settings
        cfg = _C.current_config()  # tile / backward mode / binning of THIS forward, reused by its backward
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                *_settings_args(rs), rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        (num_rendered, color, radii, geom, binning, img, depth, opacity, n_touched) = _debug_call(
            rs.debug, _C.rasterize_gaussians, args, {}, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.olsr_cfg = cfg
        ctx.olsr_pose_shapes = (tuple(theta.shape), tuple(rho.shape))
        ctx.olsr_rows_token = _C.last_forward_token()  # the backward sizes its row scratch from this frame's exact count
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii, n_touched)
        ctx.set_materialize_grads(False)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_radii, grad_out_depth, grad_out_opacity, grad_n_touched):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        grad_out_color = _cotangent(grad_out_color, (3, H, W), means3D)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                *_settings_args(rs), grad_out_color, grad_out_depth, sh, rs.sh_degree, rs.campos, geom, ctx.num_rendered,
                binning, img, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, _grad_tau, tau_sum) = _debug_call(
            rs.debug, _C.rasterize_gaussians_backward, args,
            dict(cfg=ctx.olsr_cfg, with_tau_sum=True, rows_token=ctx.olsr_rows_token), "snapshot_bw.dump", "backward")
        grad_theta, grad_rho = _split_tau(tau_sum, *ctx.olsr_pose_shapes)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, grad_theta, grad_rho, None)


class _RasterizeLanguageGaussians(torch.autograd.Function):
    """RGB + language + depth + opacity rasterization (reference :205-403)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, language_precomp, opacities, scales, rotations,
                cov3Ds_precomp, theta, rho, raster_settings):
        rs = raster_settings
        cfg = _C.current_config()
        (num_rendered, color, language, radii, geom, binning, img, depth, opacity,
         n_touched) = _C.rasterize_language_gaussians(
            rs.bg, means3D, colors_precomp, language_precomp, opacities, scales, rotations, rs.scale_modifier,
            cov3Ds_precomp, *_settings_args(rs), rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
            rs.prefiltered, rs.debug)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.olsr_cfg = cfg
        ctx.olsr_pose_shapes = (tuple(theta.shape), tuple(rho.shape))
        ctx.olsr_rows_token = _C.last_forward_token()  # the backward sizes its row scratch from this frame's exact count
        ctx.save_for_backward(colors_precomp, language_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geom, binning, img)
        ctx.mark_non_differentiable(radii, n_touched)
        ctx.set_materialize_grads(False)
        return color, language, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_language, grad_out_radii, grad_out_depth, grad_out_opacity,
                 grad_n_touched):
        rs = ctx.raster_settings
        (colors_precomp, language_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning,
         img) = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        grad_out_color = _cotangent(grad_out_color, (3, H, W), means3D)
        (grad_means2D, grad_colors_precomp, grad_language_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp,
         grad_sh, grad_scales, grad_rotations, _grad_tau, tau_sum) = _C.rasterize_language_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, language_precomp, scales, rotations, rs.scale_modifier,
            cov3Ds_precomp, *_settings_args(rs), grad_out_color, grad_out_language, grad_out_depth, sh, rs.sh_degree,
            rs.campos, geom, ctx.num_rendered, binning, img, rs.debug, cfg=ctx.olsr_cfg, with_tau_sum=True, rows_token=ctx.olsr_rows_token)
        grad_theta, grad_rho = _split_tau(tau_sum, *ctx.olsr_pose_shapes)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_language_precomp, grad_opacities,
                grad_scales, grad_rotations, grad_cov3Ds_precomp, grad_theta, grad_rho, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, theta, rho, raster_settings)


def rasterize_language_gaussians(means3D, means2D, sh, colors_precomp, language_precomp, opacities, scales, rotations,
                                 cov3Ds_precomp, theta, rho, raster_settings):
    return _RasterizeLanguageGaussians.apply(means3D, means2D, sh, colors_precomp, language_precomp, opacities,
                                             scales, rotations, cov3Ds_precomp, theta, rho, raster_settings)


def _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp):
    """The two argument-exclusivity rules of the reference, with its messages (:441-445, :515-528)."""
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")


def _or_empty(t):
    return torch.Tensor([]) if t is None else t


class _RasterizerBase(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points in front of the near plane (reference :426-435, :487-496)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)


class GaussianRasterizer(_RasterizerBase):
    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), opacities,
                                   _or_empty(scales), _or_empty(rotations), _or_empty(cov3D_precomp), _or_empty(theta),
                                   _or_empty(rho), self.raster_settings)


class LanguageGaussianRasterizer(_RasterizerBase):
    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, language_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, theta=None, rho=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_language_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp),
                                            _or_empty(language_precomp), opacities, _or_empty(scales),
                                            _or_empty(rotations), _or_empty(cov3D_precomp), _or_empty(theta),
                                            _or_empty(rho), self.raster_settings)
