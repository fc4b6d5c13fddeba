"""Randomised oracle-vs-GPU campaign (one-off confidence run, not part of the test suite): N random scenes — Gaussian
count, image size, tile edge, language width, footprint scale over two decades, camera yaw / offset, SH degree — each put
through tests/test_gpu_parity.py::_check (forward bit-exact in both binning modes, instance lists, every gradient).

    python scripts/oracle_stress.py [N=200] [seed0=0] [vary|big]"""
import math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from online_lang_splatting_amd import _C as hip, _abi
from online_lang_splatting_amd.scene import default_camera, make_scene
from oracle import oracle_C as oracle
import test_gpu_parity as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
VARY = len(sys.argv) > 3 and sys.argv[3] in ("vary", "big")
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # also randomise intrinsics, background, opacity range, pitch
t0 = time.time()
fails = 0
notes = 0
for k in range(N):
    g = torch.Generator().manual_seed(77_000 + seed0 + k)
    r = lambda: float(torch.rand(1, generator=g))
    P = int(300 + r() * 9000)
    W, H = int(64 + r() * 400), int(48 + r() * 300)
    if BIG:  # (third generation: up to 80 k Gaussians on up to 964 x 748 pixels — several staging batches per tile)
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
    if r() < 0.2:  # precomputed 3D covariance: Sigma = R S S^T R^T of the scene's own scales / rotations, perturbed
        L = torch.randn(P, 3, 3, generator=g) * sc.scales.mean()
        Sg = L @ L.transpose(1, 2)
        kw["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).contiguous()
    elif r() < 0.3:
        kw["scale_modifier"] = 0.5 + r()
    try:
        T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, **kw)
    except AssertionError as e:
        # the max-norm bound (1e-4 of the tensor's largest magnitude) is what the test suite asserts on its fixed scenes;
        # a breach here is re-judged by the north-star criterion per element (>= 99.99 % within 1e-4, worst <= 2e-2)
        note = f"scene {k}: P={P} {W}x{H} tile={tile} F={F} deg={deg} mode={mode} {sorted(kw)}: {str(e)[:200]}"
        try:
            T._check(hip, oracle, sc, seed=k, tile=tile, mode=mode, elementwise=True, worst_bound=2e-2, **kw)
            notes += 1
            print("NOTE (max-norm only) " + note, flush=True)
        except AssertionError as e2:
            fails += 1
            print("FAIL " + note + " | per element: " + str(e2)[:200], flush=True)
    finally:
        hip.TILE, hip.BWD_MODE, hip.BINNING = 15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE
        oracle.TILE, oracle.BWD_MODE = 15, 0
    if (k + 1) % 25 == 0:
        print(f"{k + 1}/{N} scenes, {fails} failures, {notes} notes, {time.time() - t0:.0f} s", flush=True)
print(f"done: {N} scenes, {fails} failures, {notes} max-norm notes")
sys.exit(1 if fails else 0)
