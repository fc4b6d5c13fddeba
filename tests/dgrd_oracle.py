"""CPU oracle of the disentangled rasterizer DGR-D (submodules/diff-gaussian-rasterization-disentangle-optim) — TEST
INFRASTRUCTURE, written independently of online_lang_splatting_amd/disentangled.py.

It restates DGR-D by what its kernels do, on top of the oracle of the plain rasterizer (oracle/oracle_C.py, which
restates DGR = submodules/diff-gaussian-rasterization).  DGR-D's shared device functions are DGR's
(`diff DGR/cuda_rasterizer/{auxiliary.h,math.h} DGR-D/...` shows only a commented-out pair of lines; computeCov3D and
computeCov2DCUDA are identical), its tiles are 16x16 (DGR-D/cuda_rasterizer/config.h:17-18).  Parity unpinned by
execution, like oracle.cpp's compositing core: no CUDA here; every step below cites the DGR-D lines it follows.
"""
import torch

from oracle import oracle_C as O
from online_lang_splatting_amd import _abi

TILE = 16


def _with(tile, mode, flags):
    O.TILE, O.BWD_MODE, O.FLAGS = tile, mode, flags


def forward(sc2, mode=_abi.BWD_REFERENCE):
    """sc2: dict with the DGR-D inputs (CPU tensors).  Returns (outputs dict, saved state)."""
    a = sc2
    cam = a["camera"]
    common = (cam.world_view_transform, cam.full_proj_transform, cam.projection_matrix, cam.tanfovx, cam.tanfovy,
              cam.height, cam.width)
    tail = (a["sh_degree"], cam.camera_center, False, False)
    _with(TILE, mode, _abi.FLAG_SIGNED_EMPTY_RADII)
    try:
        # colour + depth loop of language_renderCUDA (DGR-D forward.cu:506-560): lists, conics and opacities of set 1
        R1, color, r1, geom1, bin1, img1, depth, opacity, n_touched = O.rasterize_gaussians(
            a["bg"], a["means3D"], torch.empty(0), a["opacities"], a["scales"], a["rotations"], 1.0, torch.empty(0),
            *common, a["shs"], *tail)
        # language loop (DGR-D forward.cu:562-633): lists, conics and opacities of set 2; colours play no role in it
        zeros = torch.zeros(a["means3D"].shape[0], 3)
        R2, _c, language, r2, geom2, bin2, img2, _d, opacity_lang, n_touched_lang = O.rasterize_language_gaussians(
            a["bg"], a["means3D"], zeros, a["language"], a["opacities_lang"], a["scales_lang"], a["rotations_lang"], 1.0,
            torch.empty(0), *common, torch.empty(0), 0, cam.camera_center, False, False)
    finally:
        _with(15, _abi.BWD_REFERENCE, 0)
    # languagePreprocessCUDA returns early only when BOTH squares cover no tile (DGR-D forward.cu:391-397) and then
    # writes radii[idx] = my_radius, radii_lang[idx] = my_radius_lang unconditionally (:421-431)
    vis = (r1 > 0) | (r2 > 0)
    radii = torch.where(vis, r1.abs(), torch.zeros_like(r1))
    radii_lang = torch.where(vis, r2.abs(), torch.zeros_like(r2))
    out = dict(color=color, language=language, radii=radii, radii_lang=radii_lang, depth=depth, opacity=opacity,
               opacity_lang=opacity_lang, n_touched=n_touched, n_touched_lang=n_touched_lang, R1=R1, R2=R2,
               raw_radii=(r1, r2))
    saved = dict(common=common[:5], geom1=geom1, bin1=bin1, img1=img1, geom2=geom2, bin2=bin2, img2=img2, R1=R1, R2=R2,
                 r1=r1.clamp(min=0), r2=r2.clamp(min=0), zeros=zeros)
    return out, saved


def backward(sc2, saved, dc, dl, dd, mode=_abi.BWD_REFERENCE):
    """DGR-D's gradients (names of DGR-D __init__.py:389-404)."""
    a, s = sc2, saved
    cam = a["camera"]
    _with(TILE, mode, 0)
    try:
        # colour loop of language_render_cuda (DGR-D backward.cu:1218-1335) + computeCov2DCUDA + the first-set half of
        # language_preprocessCUDA (:1545-1564, :676-808): the RGB rasterizer's backward
        (m2, dcol, dop, m3, dcov, dsh, dsc, drot, dtau) = O.rasterize_gaussians_backward(
            a["bg"], a["means3D"], s["r1"], torch.empty(0), a["scales"], a["rotations"], 1.0, torch.empty(0),
            *s["common"], dc, dd, a["shs"], a["sh_degree"], cam.camera_center, s["geom1"], s["R1"], s["bin1"], s["img1"],
            False)
        # language loop (:1337-1428): dL_dalpha_lang has only the feature term (no colour, depth or background term),
        # the recursion is not skip-guarded (:1385-1393), thread 0's feature gradient is the one added (:1423-1425), no
        # mean gradient is formed; computeCov2DCUDA_no_tau (:354-436) then yields dL_dcov3D_lang only, from which
        # language_preprocessCUDA derives scale_lang / rotation_lang.  With zero colour / depth cotangents the language
        # rasterizer's backward computes exactly these terms (its colour terms multiply the zero cotangents).
        (_m2, _dcol, dlang, dop_l, _m3, dcov_l, _dsh, dsc_l, drot_l, _dtau) = O.rasterize_language_gaussians_backward(
            a["bg"], a["means3D"], s["r2"], s["zeros"], a["language"], a["scales_lang"], a["rotations_lang"], 1.0,
            torch.empty(0), *s["common"], torch.zeros_like(dc), dl, torch.zeros_like(dd), torch.empty(0), 0,
            cam.camera_center, s["geom2"], s["R2"], s["bin2"], s["img2"], False)
    finally:
        _with(15, _abi.BWD_REFERENCE, 0)
    tau = dtau.view(-1, 6).sum(0)  # DGR-D __init__.py:405-407
    return dict(means2D=m2, colors=dcol, language=dlang, opacities=dop, opacities_lang=dop_l, means3D=m3, cov3D=dcov,
                cov3D_lang=dcov_l, sh=dsh, scales=dsc, scales_lang=dsc_l, rotations=drot, rotations_lang=drot_l,
                rho=tau[:3], theta=tau[3:])


def release(saved):
    O.release(saved["geom1"])
    O.release(saved["geom2"])
