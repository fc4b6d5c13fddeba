"""A/B of the carried depth order (include/olsr.h; csrc/k_order_carry.hip): one frame in flight, the same view repeated, with
and without RasterWorkspace(carry_order=True) — frames/s and the library's stage times — on the room map and on the volume
configs.  Also a perturbed-pose sequence (the hit rate of optimiser-sized steps).

    python scripts/probe/carry_timing.py [room] [1] [2] [3]   ->  one JSON object on stdout
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from online_lang_splatting_amd import _abi, _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import default_camera, make_room_scene, make_scene  # noqa: E402


def measure(sc, cams, dev, W, H, F, carry, flags=0, n=60):
    M = sc.shs.shape[1]
    g_dev, _ = bench.device_inputs(sc, cams[0], dev)
    camd = [bench.device_inputs(sc, c, dev)[1] for c in cams]
    dc, dl, dd = [None if t is None else t.to(dev) for t in sc.cotangents(3)]
    R0 = bench._sized_capacity(F, g_dev, camd[0], H, W, sc.sh_degree, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE))
    cap = int(1.5 * R0) + (1 << 16)
    ws = RasterWorkspace(sc.P, W, H, F, M, cap, dev, carry_order=carry, flags=flags)
    bk = GradientBucket(sc.P, GradLayout(M, F), dev, track_rows=True)

    def step(cam):
        ws.set_scene(sh_degree=sc.sh_degree, **cam, **g_dev)
        ws.forward()
        ws.backward(dc, dl, dd, bucket=bk, first=True, bucket_only=True)
    for _ in range(10):
        step(camd[0])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        step(camd[0])
    torch.cuda.synchronize(dev)
    fps = n / (time.perf_counter() - t0)
    _lib.set_profiling(True)
    for _ in range(10):
        step(camd[0])
    per = {}
    for name, ms in _lib.stage_times():
        per.setdefault(name, []).append(ms)
    _lib.set_profiling(False)
    res = {"fps": round(fps, 1), "stage_ms": {k: round(sum(v) / len(v), 4) for k, v in per.items()}}
    if carry:
        res["missed_last"] = ws.carry_missed()
        # a sequence of optimiser-sized pose steps: which frames the repair was enough for
        hits = []
        for cam in camd[1:]:
            step(cam)
            hits.append(not ws.carry_missed())
        res["sequence_hits"] = hits
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            step(camd[1 + i % (len(camd) - 1)] if len(camd) > 1 else camd[0])
        torch.cuda.synchronize(dev)
        res["fps_over_the_sequence"] = round(n / (time.perf_counter() - t0), 1)
    return res


def main():
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["room", "1", "3"]
    out = {}
    for w in which:
        if w == "room":
            P, W, H, F = 500_000, 1200, 680, 15
            rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=3)
            sc = rs.scene
            c0 = rs.cameras[0]
            # small rotations / translations about view 0 (1e-3 rad = 0.057 deg)
            import math
            cams = [c0]
            from online_lang_splatting_amd.scene import Camera
            for k, (yaw, tx) in enumerate([(0.02, 0.001), (0.05, 0.002), (0.1, 0.004), (0.2, 0.008), (0.4, 0.016), (0.8, 0.03),
                                           (0.8, 0.03), (0.4, 0.016)]):
                a = math.radians(yaw)
                Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
                cams.append(Camera(W, H, c0.fx, c0.fy, c0.cx, c0.cy, Ry @ c0.R, Ry @ c0.T + torch.tensor([tx, 0.0, 0.0])))
        else:
            cfg = bench.CONFIGS[int(w)]
            P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
            sc = make_scene(P, W, H, F, seed=3, max_sh_degree=cfg["max_sh_degree"])
            cams = [default_camera(W, H, yaw, tx) for yaw, tx in [(0.0, 0.0), (0.02, 0.001), (0.05, 0.002), (0.1, 0.004),
                                                                  (0.2, 0.008), (0.4, 0.016), (0.8, 0.03), (0.8, 0.03),
                                                                  (0.4, 0.016)]]
        r = {}
        for carry in (False, True):
            r["carry" if carry else "plain"] = measure(sc, cams, dev, W, H, F, carry)
        r["carry_frames_in_flight_shape"] = measure(sc, cams, dev, W, H, F, True, flags=_abi.FLAG_FRAMES_IN_FLIGHT)
        out[w] = r
        print(w, json.dumps(r), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
