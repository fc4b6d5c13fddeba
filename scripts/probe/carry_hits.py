"""How far does the depth order move between two iterations of the reference's dependent loops?  Room map (and the volume):
the tracking loop from a perturbed pose and the mapping loop over the 12-view window, with RasterWorkspace(carry_order=True) —
per iteration whether the repair was enough (a synchronising read of the device flag: a probe, not a benchmark) and the rank
displacement between consecutive orders (max, 99.9 %, median).

    python scripts/probe/carry_hits.py [room|volume]  ->  JSON on stdout
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from online_lang_splatting_amd import _abi  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes, RasterWorkspace  # noqa: E402
from online_lang_splatting_amd.scene import make_room_scene, make_scene, default_camera  # noqa: E402
from online_lang_splatting_amd.slam_iterations import MappingStep, PoseState, TrackingLoop  # noqa: E402


def displacement(prev, cur):
    P = prev.numel()
    pos_prev = torch.empty(P, dtype=torch.int64, device=prev.device)
    pos_prev[prev.long()] = torch.arange(P, device=prev.device)
    pos_cur = torch.empty(P, dtype=torch.int64, device=prev.device)
    pos_cur[cur.long()] = torch.arange(P, device=prev.device)
    d = (pos_cur - pos_prev).abs().float()
    return {"max": int(d.max()), "p999": int(d.quantile(0.999)) if P <= 16_000_000 else None, "median": int(d.median()),
            "moved_beyond_1024": int((d > 1024).sum())}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "room"
    dev = torch.device("cuda:0")
    P, W, H, F = 500_000, 1200, 680, 15
    if which == "room":
        rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=3)
        sc, cams, targets = rs.scene, rs.cameras, rs.targets
    else:
        sc = make_scene(P, W, H, F, seed=3)
        cams = [default_camera(W, H, (k - 5.5) * 2.0, (k - 5.5) * 0.05) for k in range(12)]
        gen = torch.Generator().manual_seed(1)
        targets = [(torch.rand(3, H, W, generator=gen), torch.rand(H, W, generator=gen) + 1.5, torch.rand(F, 192, 192, generator=gen))
                   for _ in cams]
    M = sc.shs.shape[1]
    g_dev, _ = bench.device_inputs(sc, cams[0], dev)
    camd = [bench.device_inputs(sc, c, dev)[1] for c in cams]
    R0 = max(bench._sized_capacity(F, g_dev, c, H, W, 0, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE)) for c in camd)
    cap = int(1.5 * R0) + (1 << 16)
    out = {"scene": which, "P": sc.P}
    # ---- tracking: 60 iterations from a pose a few cm / tenths of a degree off
    ws = RasterWorkspace(sc.P, W, H, F, M, cap, dev, carry_order=True)
    cam0 = cams[0]
    T_gt = torch.eye(4)
    T_gt[:3, :3], T_gt[:3, 3] = cam0.R, cam0.T
    T_gt = T_gt.to(dev)
    pose = PoseState(T_gt, cam0.projection_matrix.to(dev), cam0.tanfovx, cam0.tanfovy)
    tau0 = torch.tensor([0.02, -0.015, 0.01, 0.004, -0.006, 0.003])
    th = tau0[3:]
    Wm = torch.tensor([[0.0, -th[2], th[1]], [th[2], 0.0, -th[0]], [-th[1], th[0], 0.0]])
    dT = torch.eye(4)
    dT[:3, :3] = torch.eye(3) + Wm + 0.5 * Wm @ Wm
    dT[:3, 3] = tau0[:3]
    T0 = (dT.to(dev) @ T_gt).contiguous()
    pose.reset(T0)
    loop = TrackingLoop(ws, g_dev, 0, pose, targets[0][0].to(dev), targets[0][1].to(dev), language_cotangent="null")
    rows = []
    prev = None
    for it in range(60):
        loop.iteration()
        torch.cuda.synchronize(dev)
        cur = ws.depth_order_carry.clone()
        r = {"it": it, "missed": ws.carry_missed(), "pose_step": float(pose.last_tau.abs().max())}
        if prev is not None:
            r.update(displacement(prev, cur))
        prev = cur
        rows.append(r)
    out["tracking"] = {"hits": sum(not r["missed"] for r in rows), "iterations": len(rows), "rows": rows}
    del ws, loop
    # ---- mapping: 12 views, 8 iterations, one order per view
    params = dict(means3D=g_dev["means3D"].clone(), shs=g_dev["shs"].clone(),
                  opacities=torch.logit(g_dev["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
                  scales=torch.log(g_dev["scales"]).contiguous(), rotations=g_dev["rotations"].clone(),
                  language=g_dev["language"].clone())
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    lanes = FrameLanes(1, sc.P, W, H, F, M, cap, dev)
    stp = MappingStep(lanes, params, g_dev["bg"], 0, camd, targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=True)
    mrows = []
    prev = {}
    for it in range(8):
        # (one lane: the views run in turn; read each view's flag through a hook on the workspace)
        stp.iteration()
        torch.cuda.synchronize(dev)
        per_view = []
        for v in range(len(camd)):
            cur = stp.view_orders[v].clone()
            d = displacement(prev[v], cur) if v in prev else None
            prev[v] = cur
            per_view.append(d)
        mrows.append({"it": it, "per_view": per_view})
    out["mapping"] = {"rows": mrows,
                      "views_beyond_1024": [sum(1 for d in r["per_view"] if d and d["max"] > 1024) for r in mrows]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
