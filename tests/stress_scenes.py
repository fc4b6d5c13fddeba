"""The random scenes of the oracle-vs-GPU campaign (scripts/oracle_stress.py, tests/test_gpu_stress.py): Gaussian count,
image size, tile edge, language width, footprint scale over two decades, camera yaw / offset, SH degree, backward mode,
precomputed colours / 3D covariances / scale modifiers; generation "vary" also draws intrinsics, a camera pitch, the
background and the opacity range, generation "big" 10 k - 80 k Gaussians on up to 964 x 748 pixels."""
import math

import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import default_camera, make_scene


def random_scene(k, seed0=0, generation="base"):
    """-> (scene, tile, backward mode, keyword arguments of _check, description)."""
    VARY, BIG = generation in ("vary", "big"), generation == "big"
    g = torch.Generator().manual_seed(77_000 + seed0 + k)
    r = lambda: float(torch.rand(1, generator=g))  # noqa: E731
    P = int(300 + r() * 9000)
    W, H = int(64 + r() * 400), int(48 + r() * 300)
    if BIG:  # (third generation: several staging batches per tile)
        P = int(10_000 + r() * 70_000)
        W, H = int(200 + r() * 764), int(150 + r() * 598)
    tile = 16 if r() < 0.4 else 15
    F = (0, 3, 15, 16, 32)[int(r() * 5) % 5]
    deg = int(r() * 4) % 4
    cam = default_camera(W, H, yaw_deg=r() * 50 - 25, tx=r() - 0.5)
    sc = make_scene(P, W, H, F, seed=900_000 + seed0 + k, camera=cam, scale_mult=10 ** (r() * 2.2 - 1.2), max_sh_degree=deg)
    if VARY:  # (second-generation scenes: intrinsics, background, opacity range, a pitch on top of the yaw)
        cam.fx, cam.fy = W * (0.3 + 0.9 * r()), W * (0.3 + 0.9 * r())
        cam.cx, cam.cy = (W - 1) / 2 + (r() - 0.5) * 0.3 * W, (H - 1) / 2 + (r() - 0.5) * 0.3 * H
        a_ = (r() - 0.5) * 0.5
        Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(a_), -math.sin(a_)], [0.0, math.sin(a_), math.cos(a_)]])
        cam.R = (Rx @ cam.R).contiguous()
        cam.T = cam.T + torch.tensor([0.0, (r() - 0.5) * 0.6, (r() - 0.5) * 0.6])
        sc.bg = torch.rand(3, generator=g) if r() < 0.6 else sc.bg
        if r() < 0.5:
            sc.opacities[:] = torch.sigmoid(torch.randn(sc.opacities.shape, generator=g) * (1 + 4 * r()) + (r() - 0.5) * 4)
    mode = _abi.BWD_EXACT if r() < 0.3 else _abi.BWD_REFERENCE
    kw = {}
    if r() < 0.25:
        kw["colors_precomp"] = torch.rand(P, 3, generator=g)
    if r() < 0.2:  # precomputed 3D covariance: a random positive semi-definite matrix of the scene's scale
        L = torch.randn(P, 3, 3, generator=g) * sc.scales.mean()
        Sg = L @ L.transpose(1, 2)
        kw["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).contiguous()
    elif r() < 0.3:
        kw["scale_modifier"] = 0.5 + r()
    desc = f"scene {k}: P={P} {W}x{H} tile={tile} F={F} deg={deg} mode={mode} {sorted(kw)}"
    return sc, tile, mode, kw, desc


def random_room_scene(k, seed0=0):
    """A random SURFACE scene (round 5): scene.make_room_scene at a random size / resolution / language width, seen from a
    random keyframe of its window, with the perturbations a SLAM map goes through between keyframes — scales grown and made
    anisotropic, rotations off identity, opacities spread, a camera nudged off the keyframe pose.
    -> (scene, tile, backward mode, keyword arguments of _check, description)."""
    from online_lang_splatting_amd.scene import Camera, knn_mean_dist2_host, make_room_scene
    g = torch.Generator().manual_seed(88_000 + seed0 + k)
    r = lambda: float(torch.rand(1, generator=g))  # noqa: E731
    W, H = int(96 + r() * 380), int(64 + r() * 260)
    P = int(1500 + r() * 30000)
    F = (0, 3, 15, 16, 32)[int(r() * 5) % 5]
    views = 3 + int(r() * 4)
    rs = make_room_scene(P, W, H, F, views=views, seed=500 + seed0 + k, knn=knn_mean_dist2_host)
    v = int(r() * views) % views
    sc = rs.view(v)
    n = sc.P
    if r() < 0.7:   # a map that has been optimised for a while
        sc.scales = (sc.scales * torch.exp(0.5 * torch.randn(n, 3, generator=g) + 0.6 * r())).contiguous()
        q = sc.rotations + 0.3 * torch.randn(n, 4, generator=g)
        sc.rotations = (q / q.norm(dim=1, keepdim=True)).contiguous()
        sc.opacities = torch.sigmoid(torch.randn(n, 1, generator=g) * 2.0 + 1.0).contiguous()
    if r() < 0.5:   # the tracked pose is never exactly a keyframe's
        c = sc.camera
        a_ = (r() - 0.5) * 0.2
        Ry = torch.tensor([[math.cos(a_), 0.0, math.sin(a_)], [0.0, 1.0, 0.0], [-math.sin(a_), 0.0, math.cos(a_)]])
        sc.camera = Camera(c.width, c.height, c.fx, c.fy, c.cx, c.cy, (Ry @ c.R).contiguous(),
                           (Ry @ c.T + torch.tensor([(r() - 0.5) * 0.2, (r() - 0.5) * 0.1, (r() - 0.5) * 0.2])).contiguous())
    tile = 16 if r() < 0.4 else 15
    mode = _abi.BWD_EXACT if r() < 0.3 else _abi.BWD_REFERENCE
    desc = f"room scene {k}: P={n} {W}x{H} tile={tile} F={F} view {v}/{views} mode={mode}"
    return sc, tile, mode, {}, desc
