"""Randomised campaign for the carried depth order (csrc/k_order_carry.hip): runs ON THE GPU BOX.
N sequences; each: a random scene (size, image, channels, tile, frames-in-flight block shape), a random walk of the pose with
steps from optimiser-sized to large, parameter noise on the means between frames (what an Adam step does to the order), and —
half of the frames — damage to the carried array itself between two frames (swaps near and far, duplicated entries, a shifted
block, out-of-range words).  Every frame of the carrying workspace must equal the plain workspace's bit for bit: instance
lists, images, radii, n_touched, gradients.  Prints hits / misses and the first failure.

    python scripts/carry_fuzz.py [N=100] [seed0=0]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_order_carry as T  # noqa: E402
from online_lang_splatting_amd import _abi  # noqa: E402
from online_lang_splatting_amd.scene import default_camera  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
hits = misses = frames = fails = 0
for k in range(N):
    gen = torch.Generator().manual_seed(seed0 + k)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))  # noqa: E731
    u = lambda: float(torch.rand(1, generator=gen))  # noqa: E731
    P = [r(1, 3000), r(3000, 20000), r(20000, 120000), r(120000, 400000)][r(0, 3)]
    W, H = r(64, 480), r(64, 360)
    F = [0, 3, 15][r(0, 2)]
    tile = [15, 16][r(0, 1)]
    flags = _abi.FLAG_FRAMES_IN_FLIGHT if (hasattr(_abi, "FLAG_FRAMES_IN_FLIGHT") and r(0, 1)) else 0
    desc = f"seq {k}: P={P} {W}x{H} F={F} tile={tile} flags={flags}"
    try:
        sc, g, cot, plain, carry, dev = T._setup(P=P, W=W, H=H, F=F, seed=seed0 + k, tile=tile, flags=flags,
                                                 capacity=6_000_000)
        if u() < 0.3 and P > 8:   # ties in quantity
            sc.means3D[::3, 2] = sc.means3D[0, 2]
            g["means3D"] = sc.means3D.to(dev)
        yaw, tx = 0.0, 0.0
        for f in range(6):
            step = [0.0, 0.01, 0.05, 0.3, 2.0, 8.0][r(0, 5)]
            yaw += (u() - 0.5) * 2 * step
            tx += (u() - 0.5) * 0.05 * step
            if u() < 0.5:   # an optimiser step on the means
                g["means3D"] = (g["means3D"] + (10 ** -r(2, 5)) * torch.randn(g["means3D"].shape, generator=gen).to(dev)).contiguous()
            if f > 0 and u() < 0.5:   # damage the carried array
                a = carry.depth_order_carry
                kind = r(0, 4)
                if kind == 0 and P > 4:      # swaps: neighbours and far apart
                    for _ in range(r(1, 50)):
                        i = r(0, P - 1)
                        j = min(P - 1, i + [1, 7, 500, 5000, P][r(0, 4)])
                        ai, aj = int(a[i]), int(a[j])
                        a[i], a[j] = aj, ai
                elif kind == 1:              # duplicated entries
                    idx = torch.randint(0, P, (r(1, 20),), generator=gen).to(dev)
                    a[idx] = int(a[0])
                elif kind == 2 and P > 10:   # a block shifted by one
                    i = r(0, P - 3)
                    j = min(P, i + r(2, 4000))
                    a[i:j] = torch.roll(a[i:j].clone(), 1)
                elif kind == 3:              # out-of-range words
                    idx = torch.randint(0, P, (r(1, 20),), generator=gen).to(dev)
                    a[idx] = torch.randint(-2**31, 2**31 - 1, (idx.numel(),), generator=gen).to(dev).to(a.dtype)
                else:                        # reversed
                    carry.depth_order_carry.copy_(torch.flip(a.clone(), [0]))
            cam = T._cam(default_camera(W, H, yaw, tx), dev)
            ref = T._frame(plain, sc, cam, g, cot)
            got = T._frame(carry, sc, cam, g, cot)
            T._assert_identical(got, ref)
            T._is_order_of(carry, dev)
            frames += 1
            if carry.carry_missed():
                misses += 1
            else:
                hits += 1
        del plain, carry
    except AssertionError as e:
        fails += 1
        print("FAIL " + desc + ": " + str(e)[:300], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR " + desc + ": " + repr(e)[:300], flush=True)
    if (k + 1) % 20 == 0:
        print(f"[{k + 1}/{N}] frames {frames} hits {hits} misses {misses} failures {fails} ({time.time() - t0:.0f} s)", flush=True)
print(f"done: {N} sequences, {frames} frames ({hits} repaired, {misses} fell back to the passes), {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
