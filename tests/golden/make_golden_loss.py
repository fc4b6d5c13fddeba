"""Generates tests/golden/mapping_loss.npz by calling the REFERENCE's own loss code.

Run in the authoring container only (needs /root/reference):  python tests/golden/make_golden_loss.py

  utils.slam_utils.get_loss_mapping (imported from /root/reference) is called unmodified.  It moves the
  ground-truth image with `.cuda()` (slam_utils.py:141); there is no GPU here, so the stub viewpoint hands it
  a tensor subclass whose .cuda() is the identity — the arithmetic that follows is the reference's.
  The language term is the three reference lines of utils/slam_backend.py:579-590 (F.interpolate bilinear,
  align_corners=False; l1_loss = mean|a-b|, gaussian_splatting/utils/loss_utils.py:21-22 — that module imports
  cv2, which is absent, so its one-line l1_loss is restated) with lamda_lang = 1.0 (slam_backend.py:80).
Gradients are taken by autograd through exactly these calls.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, "/root/reference")
from utils.slam_utils import get_loss_mapping, get_loss_tracking  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class _StaysHere(torch.Tensor):
    def cuda(self, *a, **k):
        return torch.Tensor(self)


class _Viewpoint:
    pass


def case(seed, H, W, Fch, lh, lw, a, b, alpha, thr, init):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(3, H, W, generator=g, requires_grad=True)
    depth = (torch.rand(1, H, W, generator=g) * 4).requires_grad_(True)
    lang = torch.randn(Fch, H, W, generator=g).mul(0.3).requires_grad_(True)
    gt_image = torch.rand(3, H, W, generator=g)
    gt_image[:, : H // 4, : W // 3] = 0.0          # below the rgb boundary threshold
    with torch.no_grad():                           # an exact tie after the exposure transform: |0| has gradient 0
        ab = image if init else torch.exp(torch.tensor([a])) * image + torch.tensor([b])
        gt_image[:, H - 1, W - 1] = ab[:, H - 1, W - 1]
    gt_depth = torch.rand(H, W, generator=g) * 4
    gt_depth[H // 2:, : W // 5] = 0.0               # invalid depth
    gt_lang = torch.randn(Fch, lh, lw, generator=g).mul(0.3)
    vp = _Viewpoint()
    vp.original_image = gt_image.as_subclass(_StaysHere)
    vp.depth = gt_depth.numpy()
    vp.exposure_a = torch.tensor([a], requires_grad=True)
    vp.exposure_b = torch.tensor([b], requires_grad=True)
    cfg = {"Training": {"alpha": alpha, "rgb_boundary_threshold": thr}}
    loss_map = get_loss_mapping(cfg, image, depth, vp, None, initialization=init)        # the reference
    resized = F.interpolate(gt_lang.unsqueeze(0), size=(H, W), mode="bilinear", align_corners=False).squeeze(0)
    l_lang = torch.abs(lang - resized).mean()                                              # l1_loss
    loss = loss_map + 1.0 * l_lang                                                         # lamda_lang = 1.0
    loss.backward()
    z = torch.zeros(1)
    return dict(image=image.detach(), depth=depth.detach(), lang=lang.detach(), gt_image=gt_image, gt_depth=gt_depth,
                gt_lang=gt_lang, a=torch.tensor([a]), b=torch.tensor([b]), alpha=torch.tensor(alpha, dtype=torch.float64),
                thr=torch.tensor(thr, dtype=torch.float64), init=torch.tensor(int(init)), loss=loss.detach(), loss_map=loss_map.detach(),
                loss_lang=l_lang.detach(), d_image=image.grad, d_depth=depth.grad, d_lang=lang.grad,
                d_a=vp.exposure_a.grad if vp.exposure_a.grad is not None else z,
                d_b=vp.exposure_b.grad if vp.exposure_b.grad is not None else z)


def tracking_case(seed, H, W, a, b, alpha, thr):
    """get_loss_tracking (utils/slam_utils.py:92-121) on a stub viewpoint; opacity enters as a constant (the
    rasterizer's backward discards its cotangent)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(3, H, W, generator=g, requires_grad=True)
    depth = (torch.rand(1, H, W, generator=g) * 4).requires_grad_(True)
    opacity = torch.rand(1, H, W, generator=g)
    opacity[:, :, W // 2:] = 0.96 + 0.04 * opacity[:, :, W // 2:]   # half of the pixels pass opacity > 0.95
    gt_image = torch.rand(3, H, W, generator=g)
    gt_image[:, : H // 4, : W // 3] = 0.0
    gt_depth = torch.rand(H, W, generator=g) * 4
    gt_depth[H // 2:, : W // 5] = 0.0
    grad_mask = torch.rand(1, H, W, generator=g) > 0.4               # Camera.compute_grad_mask yields a bool mask
    vp = _Viewpoint()
    vp.original_image = gt_image.as_subclass(_StaysHere)
    vp.depth = gt_depth.numpy()
    vp.grad_mask = grad_mask
    vp.exposure_a = torch.tensor([a], requires_grad=True)
    vp.exposure_b = torch.tensor([b], requires_grad=True)
    cfg = {"Training": {"alpha": alpha, "rgb_boundary_threshold": thr}}
    loss = get_loss_tracking(cfg, image, depth, opacity, vp)                                # the reference
    loss.backward()
    return dict(image=image.detach(), depth=depth.detach(), opacity=opacity, gt_image=gt_image, gt_depth=gt_depth,
                grad_mask=grad_mask.to(torch.float32), a=torch.tensor([a]), b=torch.tensor([b]),
                alpha=torch.tensor(alpha, dtype=torch.float64), thr=torch.tensor(thr, dtype=torch.float64),
                loss=loss.detach(), d_image=image.grad, d_depth=depth.grad, d_a=vp.exposure_a.grad,
                d_b=vp.exposure_b.grad)


def main():
    out = {}
    tcases = [(11, 24, 40, 0.07, -0.02, 0.9, 0.01), (12, 33, 21, -0.2, 0.04, 0.95, 0.5)]
    for i, c in enumerate(tcases):
        for k, v in tracking_case(*c).items():
            out[f"t{i}_{k}"] = v.numpy()
    out["n_tracking_cases"] = np.array(len(tcases))
    cases = [(1, 24, 40, 15, 12, 12, 0.07, -0.02, 0.95, 0.01, False),
             (2, 33, 21, 15, 48, 40, -0.3, 0.05, 0.9, 0.6, False),       # downsampling target, odd sizes
             (3, 16, 16, 3, 5, 7, 0.0, 0.0, 0.95, 0.01, True)]           # initialization: no exposure transform
    for i, c in enumerate(cases):
        for k, v in case(*c).items():
            out[f"c{i}_{k}"] = v.numpy()
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "mapping_loss.npz"), **out)
    print("wrote mapping_loss.npz", {k: v.shape for k, v in out.items() if k.startswith("c0_")})


if __name__ == "__main__":
    main()
