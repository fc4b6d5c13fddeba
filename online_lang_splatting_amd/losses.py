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


def _rendered(name, t, shape, dev=None):
    """A rendered image handed over by the rasterizer: float32, on the GPU, of the expected shape (no conversion:
    these are the arrays whose cotangents are returned)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor on the GPU")
    if dev is not None and t.device != dev:
        raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


def _image3(name, image):
    if not isinstance(image, torch.Tensor) or image.dim() != 3 or image.shape[0] != 3:
        raise RuntimeError(f"{name} must be a [3, H, W] tensor")
    return _rendered(name, image, image.shape)


def _target(name, t, shape, dev):
    """A target / side input (ground truth, exposure, mask): the reference keeps some of these on the CPU or in other
    dtypes (viewpoint.gt_lang_feat, utils/slam_backend.py:576; bool masks), so they are converted to float32 on
    the device of the rendered image; the shape must match (`None` entries of `shape` are free)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if len(t.shape) != len(shape) or any(e is not None and int(s_) != int(e) for s_, e in zip(t.shape, shape)):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple('*' if e is None else e for e in shape)}")
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _exposure(exposure, dev):
    """{exposure_a, exposure_b} as one device float32[2]; accepts a [2] tensor or the reference's pair of
    1-element parameters (viewpoint.exposure_a, viewpoint.exposure_b)."""
    if exposure is None:
        return None
    if isinstance(exposure, (tuple, list)):
        if len(exposure) != 2:
            raise RuntimeError("exposure must be a [2] tensor or a pair (exposure_a, exposure_b)")
        exposure = torch.cat([e.detach().reshape(1) for e in exposure])
    return _target("exposure", exposure.reshape(-1), (2,), dev)


def mapping_loss(image, depth, language, gt_image, gt_depth, gt_language=None, exposure=None, *, alpha=0.95,
                 rgb_boundary_threshold=0.01, lamda_lang=1.0, initialization=False):
    """image [3,H,W], depth [1,H,W], language [F,H,W] or None, gt_image [3,H,W], gt_depth [H,W],
    gt_language [F,h,w] or None, exposure = device tensor [2] {exposure_a, exposure_b} or None.
    Returns dict(loss[4] = {total, rgb, depth, language terms}, dL_dimage, dL_ddepth, dL_dlanguage, dL_dexposure[2])."""
    image = _image3("mapping_loss: image", image)
    dev = image.device
    H, W = image.shape[1], image.shape[2]
    depth = _rendered("mapping_loss: depth", depth, (1, H, W), dev)
    F = 0 if language is None else int(language.shape[0])
    if language is not None:
        language = _rendered("mapping_loss: language", language, (F, H, W), dev)
    gt_image = _target("mapping_loss: gt_image", gt_image, (3, H, W), dev)
    if gt_depth is not None and gt_depth.dim() == 3:
        gt_depth = gt_depth.reshape(gt_depth.shape[-2], gt_depth.shape[-1])
    gt_depth = _target("mapping_loss: gt_depth", gt_depth, (H, W), dev)
    if gt_image is None or gt_depth is None:
        raise RuntimeError("mapping_loss: gt_image and gt_depth are required")
    gt_language = _target("mapping_loss: gt_language", gt_language, (F, None, None), dev) if F > 0 else None
    exposure = _exposure(exposure, dev)
    f32 = dict(dtype=torch.float32, device=dev)
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
    image = _image3("tracking_loss: image", image)
    dev = image.device
    H, W = image.shape[1], image.shape[2]
    depth = _rendered("tracking_loss: depth", depth, (1, H, W), dev)
    opacity = _rendered("tracking_loss: opacity", opacity, (1, H, W), dev)
    gt_image = _target("tracking_loss: gt_image", gt_image, (3, H, W), dev)
    if gt_depth is not None and gt_depth.dim() == 3:
        gt_depth = gt_depth.reshape(gt_depth.shape[-2], gt_depth.shape[-1])
    gt_depth = _target("tracking_loss: gt_depth", gt_depth, (H, W), dev)
    if gt_image is None or gt_depth is None:
        raise RuntimeError("tracking_loss: gt_image and gt_depth are required")
    if grad_mask is not None and grad_mask.dim() == 3:
        grad_mask = grad_mask.reshape(grad_mask.shape[-2], grad_mask.shape[-1])
    gm = _target("tracking_loss: grad_mask", grad_mask, (H, W), dev)
    exposure = _exposure(exposure, dev)
    f32 = dict(dtype=torch.float32, device=dev)
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
