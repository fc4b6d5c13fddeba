"""The render façade the reference's callers use, on the MI355X-native rasterizer.

Counterpart of gaussian_splatting/gaussian_renderer/__init__.py: `render(viewpoint_camera, pc, pipe, bg_color,
scaling_modifier, override_color, mask)` (:25-58) dispatching on `pc.is_language` to the RGB (:60-193) or language
(:195-347) variant, returning the dict tracking / mapping / evaluation / GUI code reads:

    render, [language,] viewspace_points, visibility_filter (= radii > 0), radii, depth, opacity, n_touched

Duck-typed on what those functions touch:
  viewpoint_camera  FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
                    projection_matrix, camera_center, cam_rot_delta, cam_trans_delta
  pc                get_xyz, get_opacity, get_scaling ([P,3] or isotropic [P,1]), get_rotation, get_features [P,M,3],
                    get_language_features [P,F], active_sh_degree, max_sh_degree, is_language,
                    get_covariance(scaling_modifier) (only with pipe.compute_cov3D_python)
  pipe              convert_SHs_python, compute_cov3D_python
Differences from the reference, on purpose: `override_color` is honoured (there the branch is unreachable, :271-288)
and the `mask` path of the language variant returns all six outputs (there it unpacks five of six and raises,
:294-314); with a mask, `radii`, `n_touched` and `visibility_filter` are scattered back to all P Gaussians.
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, LanguageGaussianRasterizer

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh(deg, sh, dirs):
    """Real spherical harmonics up to degree 3 (the `convert_SHs_python` branch; same basis and constants as
    CR/auxiliary.h:22-39).  sh: [..., 3, (deg_max + 1)^2], dirs: [..., 3] unit vectors -> [..., 3]."""
    res = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _C2[3] * xz * sh[..., 7] + _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10]
                       + _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14]
                       + _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def settings_from_camera(viewpoint_camera, pc, bg_color, scaling_modifier=1.0):
    """GaussianRasterizationSettings exactly as the reference builds them (:229-247)."""
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, projmatrix_raw=viewpoint_camera.projection_matrix,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mask=None):
    """Render the scene; returns None for an empty model (:210-211).  `bg_color` must be on the GPU."""
    xyz = pc.get_xyz
    P = xyz.shape[0]
    if P == 0:
        return None
    language = bool(getattr(pc, "is_language", False))
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device)
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = settings_from_camera(viewpoint_camera, pc, bg_color, scaling_modifier)
    rasterizer = (LanguageGaussianRasterizer if language else GaussianRasterizer)(raster_settings=rs)

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        if scales.shape[-1] == 1:  # isotropic model
            scales = scales.repeat(1, 3)
        rotations = pc.get_rotation
    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        shs = pc.get_features

    def sel(t):
        return t if (t is None or mask is None) else t[mask]
    kw = dict(means3D=sel(xyz), means2D=sel(screenspace_points), shs=sel(shs), colors_precomp=sel(colors_precomp),
              opacities=sel(pc.get_opacity), scales=sel(scales), rotations=sel(rotations), cov3D_precomp=sel(cov3D_precomp),
              theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta)
    if language:
        image, lang, radii, depth, opacity, n_touched = rasterizer(language_precomp=sel(pc.get_language_features), **kw)
    else:
        image, radii, depth, opacity, n_touched = rasterizer(**kw)
        lang = None
    if mask is not None:  # back to all P Gaussians
        full_r = torch.zeros(P, dtype=radii.dtype, device=radii.device)
        full_n = torch.zeros(P, dtype=n_touched.dtype, device=n_touched.device)
        full_r[mask], full_n[mask] = radii, n_touched
        radii, n_touched = full_r, full_n
    out = {"render": image}
    if language:
        out["language"] = lang
    out.update({"viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii, "depth": depth,
                "opacity": opacity, "n_touched": n_touched})
    return out
