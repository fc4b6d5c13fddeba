"""Times olsr_mapping_loss on a config-3 sized frame (1200x680, F=15, 192x192 language target) against the
PyTorch formulation of the reference (autograd) on the same GPU; prints one JSON line.
HBM roofline: every image plane read once, every cotangent plane written once:
(3+3+1+1+F) reads + (3+1+F) writes = (12+2F) floats per pixel."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd import losses
from oracle import loss_oracle

dev = torch.device("cuda:0")
H, W, F = 680, 1200, 15
g = torch.Generator().manual_seed(0)
image, depth = torch.rand(3, H, W, generator=g).to(dev), (torch.rand(1, H, W, generator=g) * 5).to(dev)
lang = (torch.randn(F, H, W, generator=g) * 0.3).to(dev)
gt_image, gt_depth = torch.rand(3, H, W, generator=g).to(dev), (torch.rand(H, W, generator=g) * 5).to(dev)
gt_lang = (torch.randn(F, 192, 192, generator=g) * 0.3).to(dev)
expo = torch.tensor([0.1, -0.02], device=dev)


def timed(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_hip = timed(lambda: losses.mapping_loss(image, depth, lang, gt_image, gt_depth, gt_lang, expo))


def torch_way():
    a = expo[0:1].clone().requires_grad_(True); b = expo[1:2].clone().requires_grad_(True)
    im = image.clone().requires_grad_(True); d = depth.clone().requires_grad_(True); l = lang.clone().requires_grad_(True)
    loss, *_ = loss_oracle.mapping_loss(im, d, l, gt_image, gt_depth, gt_lang, a, b)
    loss.backward()


t_torch = timed(torch_way, n=20, warm=5)
bytes_ = (12 + 2 * F) * 4 * H * W
print(json.dumps({"kernel": "olsr_mapping_loss", "H": H, "W": W, "F": F, "ms": round(t_hip, 4),
                  "algorithmic_bytes": bytes_, "achieved_GBs": round(bytes_ / t_hip / 1e6, 1),
                  "frac_of_8TBs": round(bytes_ / t_hip / 1e6 / 8000, 4),
                  "pytorch_autograd_ms_same_gpu": round(t_torch, 4), "speedup": round(t_torch / t_hip, 1)}))
