"""Host side of olsr_mapping_loss (include/olsr.h): the mapping loss of one view and the cotangents the
rasterizer backward consumes, in one HIP pass.

Mirrors what utils/slam_backend.py:579-597 + utils/slam_utils.py:124-165 of the reference compute with
PyTorch ops + autograd:  loss, d loss / d (image, depth, language), d loss / d (exposure_a, exposure_b).
GPU only; there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _abi
from ._lib import check, lib


def mapping_loss(image, depth, language, gt_image, gt_depth, gt_language=None, exposure=None, *, alpha=0.95,
                 rgb_boundary_threshold=0.01, lamda_lang=1.0, initialization=False):
    """image [3,H,W], depth [1,H,W], language [F,H,W] or None, gt_image [3,H,W], gt_depth [H,W],
    gt_language [F,h,w] or None, exposure = device tensor [2] {exposure_a, exposure_b} or None.
    Returns dict(loss[4] = {total, rgb, depth, language terms}, dL_dimage, dL_ddepth, dL_dlanguage, dL_dexposure[2])."""
    for name, t in (("image", image), ("depth", depth), ("gt_image", gt_image), ("gt_depth", gt_depth)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"mapping_loss: {name} must be a float32 tensor on the GPU")
    dev = image.device
    H, W = image.shape[1], image.shape[2]
    F = 0 if language is None else language.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    c = lambda t: None if t is None else t.contiguous()
    image, depth, language, gt_image, gt_depth, gt_language, exposure = map(
        c, (image, depth, language, gt_image, gt_depth, gt_language, exposure))
    p = _abi.OlsrLossParams(width=W, height=H, F=F, lang_width=0 if gt_language is None else gt_language.shape[2],
                            lang_height=0 if gt_language is None else gt_language.shape[1],
                            initialization=int(bool(initialization)), alpha=float(alpha),
                            rgb_boundary_threshold=float(rgb_boundary_threshold), lamda_lang=float(lamda_lang))
    out = dict(loss=torch.empty(4, **f32), dL_dimage=torch.empty(3, H, W, **f32), dL_ddepth=torch.empty(1, H, W, **f32),
               dL_dlanguage=torch.empty(F, H, W, **f32), dL_dexposure=torch.empty(2, **f32))
    L = lib()
    scratch = torch.empty(L.olsr_mapping_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None and t.numel() > 0 else None
    with torch.cuda.device(dev):
        check(L.olsr_mapping_loss(C.byref(p), ptr(image), ptr(depth), ptr(language), ptr(gt_image), ptr(gt_depth),
                                  ptr(gt_language), ptr(exposure), ptr(out["dL_dimage"]), ptr(out["dL_ddepth"]),
                                  ptr(out["dL_dlanguage"]), ptr(out["loss"]), ptr(out["dL_dexposure"]),
                                  scratch.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out


def tracking_loss(image, depth, opacity, gt_image, gt_depth, grad_mask=None, exposure=None, *, alpha=0.95,
                  rgb_boundary_threshold=0.01):
    """get_loss_tracking (utils/slam_utils.py:92-121) and its image cotangents in one pass.
    image [3,H,W], depth [1,H,W], opacity [1,H,W], gt_image [3,H,W], gt_depth [H,W], grad_mask [1,H,W] or [H,W]
    (bool or float) or None, exposure = device tensor [2] or None.
    Returns dict(loss[4] = {total, rgb term, depth term, 0}, dL_dimage, dL_ddepth, dL_dexposure[2])."""
    for name, t in (("image", image), ("depth", depth), ("opacity", opacity), ("gt_image", gt_image),
                    ("gt_depth", gt_depth)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"tracking_loss: {name} must be a float32 tensor on the GPU")
    dev = image.device
    H, W = image.shape[1], image.shape[2]
    f32 = dict(dtype=torch.float32, device=dev)
    c = lambda t: None if t is None else t.contiguous()
    gm = None if grad_mask is None else grad_mask.to(torch.float32)
    image, depth, opacity, gt_image, gt_depth, gm, exposure = map(c, (image, depth, opacity, gt_image, gt_depth, gm, exposure))
    p = _abi.OlsrLossParams(width=W, height=H, F=0, lang_width=0, lang_height=0, initialization=0, alpha=float(alpha),
                            rgb_boundary_threshold=float(rgb_boundary_threshold), lamda_lang=0.0)
    out = dict(loss=torch.empty(4, **f32), dL_dimage=torch.empty(3, H, W, **f32), dL_ddepth=torch.empty(1, H, W, **f32),
               dL_dexposure=torch.empty(2, **f32))
    L = lib()
    scratch = torch.empty(L.olsr_mapping_loss_scratch_bytes(W, H), dtype=torch.uint8, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None and t.numel() > 0 else None
    with torch.cuda.device(dev):
        check(L.olsr_tracking_loss(C.byref(p), ptr(image), ptr(depth), ptr(opacity), ptr(gt_image), ptr(gt_depth), ptr(gm),
                                   ptr(exposure), ptr(out["dL_dimage"]), ptr(out["dL_ddepth"]), ptr(out["loss"]),
                                   ptr(out["dL_dexposure"]), scratch.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
