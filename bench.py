#!/usr/bin/env python3
"""Headline benchmark: language-Gaussian rasterizer forward+backward frames/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3]

A "step" is one pass of the hot path over one frame: forward (preprocess, depth sort, binning,
tile sort, composite) + backward (composite backward, per-Gaussian backward) of BASELINE.json's
config 3 — 500 k Gaussians, 1200x680, RGB + depth + 15 language channels — through the
sync-free C-ABI entry points (olsr_forward_async / olsr_backward) with every input already
resident in HBM; the backward writes the view's gradients into the flat per-Gaussian gradient
buffer (olsr_grad_bucket) that a mapping step all-reduces.  With N > 1 (one process per GPU, torch.distributed over RCCL) every rank
renders its own viewpoint of the same Gaussians per step and the step ends with the one
all-reduce of the shared-Gaussian gradient buffer (frame sharding, weak scaling).

Frames are independent (the views of a mapping iteration), so `--streams S` (default 4) keeps S
frames in flight on S HIP streams with S workspaces (frame k on stream k mod S): the small binning
kernels of one frame overlap the compositing kernels of another, and with N > 1 the all-reduce of
one frame overlaps the compute of the next.  The timed region still issues exactly K steps.

`python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run on 127.0.0.1, one per GPU over
RCCL; fewer GPUs than ranks is an error, OLSR_BENCH_BACKEND=gloo shares devices for functional checks).  `--exchange
{sparse,all_reduce,reduce_scatter}` picks how the shared-Gaussian gradients travel; sparse (default) exchanges only the rows
that are non-zero on some rank, capacity-bound and without a host synchronisation.  The contract's K-step region is run
`--repeats` times back to back (default: until the timed regions total `--min-timed-s` = 2 s): `value` is the median run,
`config.value_runs` lists all of them.  `--scene room` runs the same legs on the surface-structured map of scene.make_room_scene.

Rank 0 prints ONE JSON line.  `non_coherent`: the camera changes every step.  `isolated` repeats the measurement with ONE frame in flight: there the
intervals between the library's HIP events (recorded on the launch stream) ARE the kernel durations, whereas
with S > 1 an interval also contains the time a kernel queues behind other frames' kernels.  `roofline`
therefore takes the dominant kernel — the longest stage of the isolated leg — and prices it three ways:
  * achieved / frac: SURVEY section 8(d)'s per-unit bytes x the units the launch PROCESSES (instances kept by the
    exact tile binning, `R_binned`) / its duration, against the 8 TB/s HBM peak (the tier's convention);
  * model_reference_R: the same per-unit bytes x the reference algorithm's units (its num_rendered);
  * traffic: HBM bytes per launch by the PMC counters (profiles/*_pmc_traffic.json);
  * valu: the compositing kernels are bound by the VALU pipe (bound = "valu"): wave-instructions per launch by the
    PMC counters (profiles/*_pmc_valu.json) / duration against the issue peak of the hardware guide — one full-rate
    wave64 instruction per 2 cycles per SIMD-32, 1024 SIMDs x 2.4 GHz / 2 — with the figure the micro-benchmark
    measured beside it (profiles/r3_valu_ubench.json: 2.3-2.7 cycles; packed fp32, selects, DPP at half rate,
    transcendentals and permlane swaps at quarter rate) and the pipe's busy fraction by SQ_ACTIVE_INST_VALU.
`latency_ms` holds median / p10 / p90 of the per-step completion intervals of the timed region and of the
isolated leg's per-frame GPU time.  `bracket`: one frame in flight under the settings the headline does not use
(exact backward, 16x16 tiles, the reference's tile lists, the one-fma accumulation).  `config4_substitute`: the
tracking and the 12-view mapping iteration with the loss in the forward composite's epilogue, beside the two-kernel
formulation, with a stage breakdown (BASELINE configs[3] itself is blocked in this image).  `cpu_baseline` is
the CPU oracle (a port — the reference has no CPU path) on whole frames of the same workload with the CPUs the
container may use and, once, with one thread; the frame it times is checked against the GPU's result of the same
view.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from online_lang_splatting_amd import _abi, _lib  # noqa: E402
from online_lang_splatting_amd.frame_shard import FrameLanes  # noqa: E402
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_scene, shard_cameras  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


JSON_OUT = sys.stdout  # main() re-points it at a private copy of the original stdout


def algorithmic_bytes(P, R, N, C=3, F=15, M=1):
    """SURVEY.md §8(d) / DESIGN.md §6 byte model, per frame and per dominant stage."""
    per = {
        "frame": (466 + 8 * C + 4 * F + 48 * M) * P + (136 + 12 * C + 12 * F) * R + (28 + 8 * C + 8 * F) * N,
        # forward composite: per-instance gather (id 4 + xy 8 + conic/opacity 16 + depth 4 + colour 4C + lang 4F)
        # + per-pixel outputs (colour 4C, lang 4F, depth 4, opacity 4, T 4, n_contrib 4)
        "render_forward": (32 + 4 * C + 4 * F) * R + (16 + 4 * C + 4 * F) * N,
        # backward composite: the same gather + the partial-gradient row 4(7+C+F) per instance
        # + per-pixel dL_dpix (4C+4F+4), T 4, n_contrib 4
        "render_backward": (60 + 8 * C + 8 * F) * R + (12 + 4 * C + 4 * F) * N,
    }
    return per


# the committed counter summaries of the current kernels, per workload: profiles/<tag>_pmc_{traffic,valu}.json.  A counter is
# attached to a line only when it was collected on THAT line's scene and config (the room map's
# lines carried the volume's 264 MB); anything else reports null.
PROFILE_TAGS = {("volume", 3): "r6", ("room", 3): "r6_room"}


def _pmc_summary(kind, scene="volume", config=3):
    """profiles/<tag of (scene, config)>_pmc_<kind>.json, or None: no counters were collected on this workload."""
    tag = PROFILE_TAGS.get((scene, config))
    if tag is None:
        return None
    tagged = os.path.join(ROOT, "profiles", f"{tag}_pmc_{kind}.json")
    return tagged if os.path.exists(tagged) else None


def measured_traffic(kernel_stage, F, scene="volume", config=3):
    """HBM bytes per launch of the stage's kernel from the workload's profiles/<tag>_pmc_traffic.json (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, corrected as MI355X_MICROARCH.md prescribes);
    None when no summary of THIS scene and config has been committed."""
    files = [_pmc_summary("traffic", scene, config)]
    if not files[-1]:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("scene", scene) != scene or d.get("config", config) != config:   # (the summary says what it was collected on)
        return None, None
    prefix = {"render_forward": "render_fwd_kernel", "render_backward": "render_bwd_kernel"}[kernel_stage]
    for k, v in d["kernels"].items():
        if k.startswith(prefix) and f", {F}" in k:
            return v["traffic_bytes"], os.path.basename(files[-1])
    return None, None


def measured_valu(kernel_stage, F, scene="volume", config=3):
    """VALU wave-instructions per launch of the stage's kernel from the workload's profiles/<tag>_pmc_valu.json
    (rocprofv3 --pmc SQ_INSTS_VALU ... of this same command); None when none of THIS scene and config has been committed."""
    files = [_pmc_summary("valu", scene, config)]
    if not files[-1]:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("scene", scene) != scene or d.get("config", config) != config:
        return None, None
    prefix = {"render_forward": "render_fwd_kernel", "render_backward": "render_bwd_kernel"}[kernel_stage]
    for k, v in d["kernels"].items():
        if k.startswith(prefix) and f", {F}" in k:
            return v, os.path.basename(files[-1])
    return None, None


# wave64 VALU instructions / s the chip can issue = 256 CUs x 4 SIMDs x clock / (cycles one instruction holds a SIMD).
# MI355X_MICROARCH.md: `v_fma_f32 (wave64): 2 cyc (SIMD-32)`.  Measured here (scripts/probe/valu_ubench.hip ->
# profiles/r3_valu_ubench.json): full-rate instructions (v_fma / v_mul / v_add / v_mov / v_and)
# reach 2.3-2.6 cycles per instruction per SIMD from 4 resident waves on — the guide's figure, not the one instruction per
# quad-cycle rounds 1-2 assumed; v_pk_*_f32, v_max / v_min, v_cvt, v_rndne, v_cndmask, DPP: ~4.2-5 cycles (half rate);
# v_exp / v_rcp / v_sqrt, v_permlane*_swap: ~8.2; v_readlane ~10.  A single wave issues at most one instruction per
# ~5 cycles (dependent: ~8.5), so fewer than ~3 waves per SIMD cannot saturate the pipe.  `peak` below is the guide's
# 2 cycles; `peak_measured` the best full-rate row of the committed micro-benchmark.
VALU_CYCLES_GUIDE = 2.0
VALU_ISSUE_PEAK = 1024 * 2.4e9 / VALU_CYCLES_GUIDE


def measured_valu_cycles():
    """Best cycles-per-instruction-per-SIMD of a full-rate VALU instruction in profiles/r3_valu_ubench.json."""
    path = os.path.join(ROOT, "profiles", "r3_valu_ubench.json")
    if not os.path.exists(path):
        return None, None
    rows = json.load(open(path))["rows"]
    full = [r["cycles_per_inst_per_simd"] for r in rows if r["op"] in ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_fmac_f32")]
    return (min(full) if full else None), os.path.basename(path)


def percentiles(xs):
    if not xs:
        return None
    v = sorted(xs)
    q = lambda f: v[min(len(v) - 1, max(0, int(round(f * (len(v) - 1)))))]  # noqa: E731
    return {"median": round(q(0.5), 4), "p10": round(q(0.1), 4), "p90": round(q(0.9), 4), "n": len(v)}


def device_inputs(sc, cam, dev):
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev),
             language=None if sc.language is None else sc.language.to(dev))
    c = dict(viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
             tanfovy=cam.tanfovy)
    return g, c


def cpu_baseline(sc, seed, gpu=None, budget_s=12.0, max_frames=16, single_thread=True):
    """The oracle (port) on full frames of the same workload, forward+backward, all host cores: whole
    frames until about `budget_s` seconds of CPU work have been timed (at least one, at most max_frames);
    then ONE frame on one thread.  `gpu` = (forward outputs, gradient bucket rows) of the same view from the
    HIP library: the frame the CPU timed must be that frame (images bit-identical, gradients within 1e-4)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_common import rel_err, run_backend
    from oracle import oracle_C as O
    # the CPUs the process may really use (affinity capped by the container's quota), not every visible one: the GPU box
    # shows 256 hardware threads and grants 16, and 256 OpenMP threads on 16 CPUs ran the frame 7x slower than 16 do
    threads = O.usable_cpus()
    torch.set_num_threads(threads)
    O.set_threads(threads)
    frames, R, dt = 0, 0, 0.0
    checked = None
    while True:
        t0 = time.perf_counter()
        fo, go = run_backend(O, sc, None, seed, 15, _abi.BWD_REFERENCE)
        dt += time.perf_counter() - t0
        R = fo["R"]
        frames += 1
        if frames == 1 and gpu is not None:  # (outside the timed sample)
            out, flat, sl = gpu
            same = all(torch.equal(out[k].cpu().reshape(fo[k].shape), fo[k]) for k in ("color", "language", "depth", "opacity")
                       if fo.get(k) is not None and fo[k].numel())
            worst = 0.0
            for name, key in (("means3D", "dL_dmeans3D"), ("opacity", "dL_dopacity"), ("scales", "dL_dscales"),
                              ("rotations", "dL_drotations"), ("language", "dL_dlanguage")):
                if go[key].numel():
                    worst = max(worst, rel_err(flat[:, sl[name]].reshape(go[key].shape), go[key])[0])
            checked = {"forward_bit_identical": bool(same), "gradient_max_rel_err": float(f"{worst:.3e}"),
                       "ok": bool(same and worst <= 1e-4)}
        O.release(fo["geom"])
        if dt >= budget_s or frames >= max_frames:
            break
    res = {"value": round(frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{frames} full frame(s) of the same workload (P={sc.P}, R={R}), forward+backward, "
                     f"OpenMP over tiles/Gaussians on {threads} threads ({os.cpu_count()} visible CPUs, quota {threads}), "
                     f"{dt:.2f} s wall",
           "checked_against_gpu": checked}
    if single_thread:
        O.set_threads(1)
        t1 = time.perf_counter()
        fo, _ = run_backend(O, sc, None, seed, 15, _abi.BWD_REFERENCE)
        d1 = time.perf_counter() - t1
        O.release(fo["geom"])
        O.set_threads(threads)
        res["single_thread"] = {"value": round(1.0 / d1, 5), "unit": "frames/s", "cores": 1,
                                "sample": f"1 full frame of the same workload, {d1:.2f} s wall"}
    return res


def workload_stats(ws, sc_P, W, H, F, tile=15):
    """What the frame on workspace `ws` (already rendered, synchronised here) looked like to the rasterizer: the reference's
    instance count R (bounding squares), the instances the exact binning kept, visible Gaussians, Gaussians some pixel blended,
    and how far the tile lists were READ: sum over tiles of the deepest list position any pixel of the tile blended
    (max n_contrib) / sum of the list lengths — saturation ends a list early, a surface map reads it to the end."""
    from online_lang_splatting_amd import _C
    torch.cuda.synchronize(ws.device)
    Rb, _ = ws.rendered()
    cnt = _C.state_field("geometry", ws.geom, "counters", P=sc_P, F=F, dtype=torch.int32, count=8).cpu()
    nc = _C.state_field("image", ws.img, "n_contrib", W=W, H=H, dtype=torch.int32, count=W * H).view(H, W)
    tx, ty = (W + tile - 1) // tile, (H + tile - 1) // tile
    pad = torch.zeros(ty * tile, tx * tile, dtype=torch.int32, device=nc.device)
    pad[:H, :W] = nc
    deepest = pad.view(ty, tile, tx, tile).amax(dim=(1, 3)).to(torch.int64)
    rg = _C.state_field("image", ws.img, "ranges", W=W, H=H, dtype=torch.int32, count=2 * tx * ty).view(-1, 2).to(torch.int64)
    lens = rg[:, 1] - rg[:, 0]
    bl = _C.state_field("geometry", ws.geom, "blended", P=sc_P, F=F, dtype=torch.uint8, count=sc_P)
    vis = int((ws.out["radii"] > 0).sum())
    return {"R": int(cnt[3]), "R_over_P": round(int(cnt[3]) / max(sc_P, 1), 3), "R_binned": int(Rb),
            "R_binned_over_R": round(int(Rb) / max(int(cnt[3]), 1), 4), "visible_gaussians": vis,
            "visible_fraction": round(vis / max(sc_P, 1), 4), "blended_gaussians": int((bl != 0).sum()),
            "blended_of_visible": round(int((bl != 0).sum()) / max(vis, 1), 4),
            "mean_list_length": round(float(lens.double().mean()), 1),
            "list_fraction_read_before_saturation": round(float(deepest.sum()) / max(float(lens.sum()), 1.0), 4),
            "pixels_saturated_fraction": round(float((ws.out["opacity"] > 1.0 - 1e-4).float().mean()), 4)
            if ws.out["opacity"].numel() else None}


def choose_exchange(union_rows, P, width):
    """The step's exchange, from the data: the capacity-bound sparse exchange moves 8 P bytes of
    flags and radii, 8 P bytes of densification statistics and 1.25 x the union's rows; the dense two-phase exchange moves
    the bucket (P x width floats + the statistics + the radii).  Sparse only when it is the smaller payload."""
    from online_lang_splatting_amd.frame_shard import GradientBucket
    cap = min(P, int(1.25 * union_rows) + 4096)
    sparse_bytes = 8 * P + (cap * width + 2 * P) * 4
    dense_bytes = (P * width + 2 * P) * 4 + 4 * P
    pays = sparse_bytes < GradientBucket.SPARSE_MARGIN * dense_bytes   # (GradientBucket.sparse_pays: the same rule)
    return ("sparse" if pays else "reduce_scatter"), cap, sparse_bytes, dense_bytes


def room_scene_leg(dev, dims, steps, seed=3):
    """config4_substitute.room_scene: BASELINE configs[3] is blocked in this image, and every other
    number of this line is measured on SURVEY 8(d)'s i.i.d. VOLUME of Gaussians, where saturation ends 87 % of every tile list
    and 98 % of the Gaussians receive no gradient.  A SLAM map is one surface layer deep.  scene.make_room_scene builds one the
    way the reference's back end does (gaussian_splatting/scene/gaussian_model.py:180-281) from ray-cast keyframes of a closed
    box room, ~P Gaussians; this leg renders the 10-keyframe window + 2 random keyframes of BackEnd.map against it and reports
    the same quantities as the headline: workload shape, stage times, frames/s (one and four frames in flight, coherent and
    cycling through the views), the tracking and the mapping iteration, and what the round-3/4 optimisations are worth here."""
    from online_lang_splatting_amd.frame_shard import FrameLanes, GradientBucket, GradLayout, RasterWorkspace
    from online_lang_splatting_amd.scene import make_room_scene
    from online_lang_splatting_amd.slam_iterations import MappingStep, PoseState, TrackingLoop
    P, W, H, F, M = dims
    t_build = time.perf_counter()
    rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=seed)
    t_build = time.perf_counter() - t_build
    sc = rs.scene
    g_dev, _ = device_inputs(sc, rs.cameras[0], dev)
    camd = [device_inputs(sc, c_, dev)[1] for c_ in rs.cameras]
    dc, dl, dd = [None if t is None else t.to(dev) for t in sc.cotangents(seed)]
    cfg0 = (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE)
    R0 = max(_sized_capacity(F, g_dev, c_, H, W, 0, dev, cfg0) for c_ in camd)
    cap = int(1.5 * R0) + (1 << 16)
    out = {"what": f"{sc.P} Gaussians of {rs.keyframes} ray-cast keyframes of a 7.0 x 2.8 x 5.0 m box room with furniture, built by "
                   "the reference's recipe (depth back-projection, pcd_downsample 32 / 64, scale = sqrt(distCUDA2 x point_size) "
                   "through olsr_knn_mean_dist2, identity rotations, opacity 0.5, unit-norm language codes), rendered from the "
                   "10-keyframe window + 2 random keyframes",
           "P": sc.P, "keyframes": rs.keyframes, "width": W, "height": H, "F": F, "views": len(camd),
           "scene_build_s": round(t_build, 2)}
    # (the lanes carry their view's depth order: the legs that repeat one view use it, the cycling legs — a lane's previous
    #  frame is another view — and the tracking loop — a pose step moves the order by tens of thousands of ranks on a surface
    #  map, scripts/probe/carry_hits.py — run without; MappingStep keeps one order per view by itself)
    lanes = FrameLanes(4, sc.P, W, H, F, M, cap, dev, carry_order=True)

    class no_carry:
        def __init__(self, *lane_sets):
            self.ws = [l_[0] for ls in lane_sets for l_ in ls]

        def __enter__(self):
            self.keep = [w_.depth_order_carry for w_ in self.ws]
            for w_ in self.ws:
                w_.depth_order_carry = None

        def __exit__(self, *exc):
            for w_, k_ in zip(self.ws, self.keep):
                w_.depth_order_carry = k_

    def step(lane, cam_):
        ws_, bucket_, stream_ = lane
        with torch.cuda.stream(stream_):
            ws_.set_scene(sh_degree=0, **cam_, **g_dev)
            ws_.forward()
            ws_.backward(dc, dl, dd, bucket=bucket_, first=True, bucket_only=True)

    def rate(n, pick_lane, pick_cam, warm=8, repeats=3):
        # (the median of `repeats` timed regions of n frames each: a region of 20 frames of 0.33 ms carries the closing
        #  synchronisation's ~25 us as half a per cent, and one slow region of a shared box moved the driver's line by 2 %)
        for i in range(warm):
            step(pick_lane(), pick_cam(i))
        runs = []
        for _ in range(repeats):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n):
                step(pick_lane(), pick_cam(i))
            torch.cuda.synchronize(dev)
            runs.append(n / (time.perf_counter() - t0))
        return sorted(runs)[len(runs) // 2]
    lane0 = FrameLanes(1, sc.P, W, H, F, M, cap, dev, carry_order=True).lanes[0]   # (one frame in flight: without OLSR_FLAG_FRAMES_IN_FLIGHT)
    n = max(steps, 60)
    out["isolated"] = {"value": round(rate(n, lambda: lane0, lambda i: camd[0]), 1), "unit": "frames/s", "frames_in_flight": 1,
                       "carried_depth_order": True}
    out["four_in_flight"] = {"value": round(rate(2 * n, lanes.next_lane, lambda i: camd[0], warm=40), 1), "unit": "frames/s",
                             "same_view_every_step": True, "carried_depth_order": True}
    with no_carry(lanes.lanes, [lane0]):
        out["isolated"]["without_carried_order"] = round(rate(n, lambda: lane0, lambda i: camd[0]), 1)
        out["four_in_flight"]["without_carried_order"] = round(rate(2 * n, lanes.next_lane, lambda i: camd[0], warm=40), 1)
        out["four_in_flight_cycling_12_views"] = {"value": round(rate(2 * n, lanes.next_lane, lambda i: camd[i % len(camd)], warm=48), 1),
                                                  "unit": "frames/s", "same_view_every_step": False, "carried_depth_order": False}
        out["isolated_cycling_12_views"] = {"value": round(rate(n, lambda: lane0, lambda i: camd[i % len(camd)], warm=24), 1),
                                            "unit": "frames/s", "carried_depth_order": False}

    # stage times, one frame in flight (events between the stages)
    def stages():
        for _ in range(3):
            step(lane0, camd[0])
        torch.cuda.synchronize(dev)
        _lib.set_profiling(True)
        for _ in range(10):
            step(lane0, camd[0])
        per = {}
        for name, ms in _lib.stage_times():
            per.setdefault(name, []).append(ms)
        _lib.set_profiling(False)
        return {k: round(sum(v) / len(v), 4) for k, v in per.items()}
    with no_carry([lane0]):
        out["stage_ms_without_carried_order"] = stages()
    out["stage_ms"] = stages()
    out["carried_order_missed_last_frame"] = bool(lane0[0].carry_missed())
    out["stage_ms_sum"] = round(sum(out["stage_ms"].values()), 4)
    ws0, b0 = lane0[0], lane0[1]
    st = workload_stats(ws0, sc.P, W, H, F)
    live = int((b0.flat != 0).any(dim=1).sum())
    L_rows, row_ovf = ws0.backward_status()
    st.update({"live_gradient_rows_gaussians": live, "live_rows_of_P": round(live / sc.P, 4),
               "live_rows_of_visible": round(live / max(st["visible_gaussians"], 1), 4), "gradient_rows_L": L_rows,
               "capacity_overflow": bool(ws0.rendered()[1] or row_ovf)})
    out["workload"] = st
    model = algorithmic_bytes(sc.P, st["R_binned"], W * H, 3, F, M)
    out["frame_model"] = {"algorithmic_bytes": int(model["frame"]),
                          "frac_of_8TBs_four_in_flight": round(model["frame"] * out["four_in_flight"]["value"] / 1e9 / HBM_PEAK_GBS, 4),
                          "frac_of_8TBs_isolated": round(model["frame"] * out["isolated"]["value"] / 1e9 / HBM_PEAK_GBS, 4)}
    # the union of the gradient rows over the 12 views of a mapping iteration, and what the run-time rule makes of it
    union = torch.zeros(sc.P, dtype=torch.bool, device=dev)
    for c_ in camd:
        step(lane0, c_)
        union |= (b0.flat != 0).any(dim=1)
    torch.cuda.synchronize(dev)
    width = 11 + 3 * M + F
    choice, cap_rows, sp_b, de_b = choose_exchange(int(union.sum()), sc.P, width)
    out["exchange"] = {"union_rows_over_12_views": int(union.sum()), "union_of_P": round(int(union.sum()) / sc.P, 4),
                       "rows_one_view": live, "sparse_payload_bytes": sp_b, "dense_payload_bytes": de_b, "chosen": choice,
                       "rule": "sparse iff 8 P + (1.25 x union rows x width + 2 P) x 4 bytes < 0.75 x (the dense bucket + statistics + radii)"}
    # what the round-3/4 optimisations are worth on a surface: A/B, one frame in flight
    ab = {}
    for name, kw, track in (("default", {}, True), ("no_row_mask", {}, False),
                            ("rect_binning", {"binning": _abi.BINNING_RECT}, True)):
        capn = cap if not kw else int(1.5 * max(_sized_capacity(F, g_dev, camd[0], H, W, 0, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_RECT)), R0)) + (1 << 16)
        ws_ = RasterWorkspace(sc.P, W, H, F, M, capn, dev, **kw)
        bk_ = GradientBucket(sc.P, GradLayout(M, F), dev, track_rows=track)
        lane_ = (ws_, bk_, torch.cuda.current_stream(dev))
        ab[name] = round(rate(n, lambda: lane_, lambda i: camd[0]), 1)
        del ws_, bk_, lane_
    out["ab_isolated_fps"] = dict(ab, note="no_row_mask: the bucket overwrite stores all P rows (round-4 row mask off); rect_binning: "
                                           "the reference's bounding-square lists instead of the exact ellipse lists; the `blended` "
                                           "byte cannot be switched off at run time - its effect is bounded by blended_of_visible")
    # tracking iteration against the ray-cast frame of view 0, from a pose a few cm off (as in config4_substitute.tracking)
    cam0 = rs.cameras[0]
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
    gt_image, gt_depth = rs.targets[0][0].to(dev), rs.targets[0][1].to(dev)
    trk = {}
    keep_carry, ws0.depth_order_carry = ws0.depth_order_carry, None   # (tracking: no carried order, see above)
    for fused in (True, False):
        pose.reset(T0)
        loop = TrackingLoop(ws0, g_dev, 0, pose, gt_image, gt_depth, language_cotangent="null", fused_loss=fused)
        for _ in range(5):
            loop.iteration()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(60):
            loop.iteration()
        torch.cuda.synchronize(dev)
        trk["fused_loss" if fused else "two_kernel_loss"] = round(1e3 * (time.perf_counter() - t0) / 60, 4)
    ws0.depth_order_carry = keep_carry
    out["tracking"] = {"ms_per_iteration": trk, "iterations": 60,
                       "pose_error_start": round(float((T0 - T_gt).abs().max()), 6),
                       "pose_error_after": round(float((pose.T_w2c - T_gt).abs().max()), 6)}
    # mapping iteration: the 12 views against their ray-cast targets, raw parameters, four views in flight
    params = dict(means3D=g_dev["means3D"].clone(), shs=g_dev["shs"].clone(),
                  opacities=torch.logit(g_dev["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
                  scales=torch.log(g_dev["scales"]).contiguous(), rotations=g_dev["rotations"].clone(),
                  language=None if F == 0 else g_dev["language"].clone())
    start = {k: (None if v is None else v.clone()) for k, v in params.items()}
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    mp = {}
    for fused in (True, False):
        for k, v in params.items():
            if v is not None:
                v.copy_(start[k])
        stp = MappingStep(lanes, params, g_dev["bg"], 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev),
                          fused_loss=fused)
        stp.iteration()
        first = float(stp.last_loss[0])
        stp.iteration()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(6):
            stp.iteration()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        mp["fused_loss" if fused else "two_kernel_loss"] = {
            "ms_per_iteration": round(1e3 * el / 6, 3), "views_per_s": round(len(camd) * 6 / el, 1),
            "loss_last_view_first_iteration": round(first, 6), "loss_last_view_final_iteration": round(float(stp.last_loss[0]), 6),
            "capacity_overflow": bool(any(w_.rendered()[1] or w_.backward_status()[1] for w_, _, _ in lanes.lanes))}
        del stp
    for k, v in params.items():
        if v is not None:
            v.copy_(start[k])
    stp = MappingStep(lanes, params, g_dev["bg"], 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev))
    for _ in range(10):
        stp.iteration()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(6):
        stp.iteration()
    torch.cuda.synchronize(dev)
    mp["auto_loss"] = {"ms_per_iteration": round(1e3 * (time.perf_counter() - t0) / 6, 3), "calibration": stp.calibration}
    del stp
    out["mapping"] = dict(mp, views=len(camd), views_in_flight=len(lanes))
    # ... and the map the loop renders MOST of the time: after 150 mapping iterations on those 12 views (BackEnd.map runs
    # mapping_itr_num = 150 per keyframe, configs/rgbd/replicav2/base_config.yaml:38) the opacities have left 0.5, the scales
    # have grown over the gaps and gone anisotropic.  The same workload statistics and rates on the trained parameters.
    for k, v in params.items():
        if v is not None:
            v.copy_(start[k])
    stp = MappingStep(lanes, params, g_dev["bg"], 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=True)
    stp.iteration()
    loss0 = float(stp.last_loss[0])
    for _ in range(149):
        stp.iteration()
    loss1 = float(stp.last_loss[0])
    del stp
    rot_n = params["rotations"] / params["rotations"].norm(dim=1, keepdim=True).clamp_min(1e-12)
    g_tr = dict(g_dev, means3D=params["means3D"], shs=params["shs"], opacities=torch.sigmoid(params["opacities"]).contiguous(),
                scales=torch.exp(params["scales"]).contiguous(), rotations=rot_n.contiguous(), language=params["language"])
    g_fresh, g_dev = g_dev, g_tr          # (step() renders from g_dev)
    R1 = max(_sized_capacity(F, g_dev, c_, H, W, 0, dev, cfg0) for c_ in camd[:3])
    if int(1.3 * R1) + (1 << 16) > cap:
        lanes = FrameLanes(4, sc.P, W, H, F, M, int(1.5 * R1) + (1 << 16), dev, carry_order=True)
        lane0 = FrameLanes(1, sc.P, W, H, F, M, int(1.5 * R1) + (1 << 16), dev, carry_order=True).lanes[0]
    tr = {"mapping_iterations": 150, "loss_first": round(loss0, 6), "loss_last": round(loss1, 6),
          "isolated": {"value": round(rate(n, lambda: lane0, lambda i: camd[0]), 1), "unit": "frames/s"},
          "four_in_flight": {"value": round(rate(2 * n, lanes.next_lane, lambda i: camd[0], warm=40), 1), "unit": "frames/s"}}
    step(lane0, camd[0])
    st_tr = workload_stats(lane0[0], sc.P, W, H, F)
    st_tr["live_gradient_rows_gaussians"] = int((lane0[1].flat != 0).any(dim=1).sum())
    st_tr["live_rows_of_visible"] = round(st_tr["live_gradient_rows_gaussians"] / max(st_tr["visible_gaussians"], 1), 4)
    st_tr["mean_opacity_image"] = round(float(lane0[0].out["opacity"].mean()), 4)
    st_tr["opacity_parameter_quantiles_5_50_95"] = [round(float(x), 3) for x in
                                                    torch.quantile(g_tr["opacities"].flatten().float()[::7], torch.tensor([0.05, 0.5, 0.95], device=dev))]
    st_tr["scale_anisotropy_median"] = round(float((g_tr["scales"].max(1).values / g_tr["scales"].min(1).values).median()), 3)
    tr["workload"] = st_tr
    out["after_150_mapping_iterations"] = tr
    g_dev = g_fresh
    del lanes
    torch.cuda.empty_cache()
    return out


def dropin_leg(sc, dev, steps, warmup):
    """The reference's own Python API on the same workload: diff_gaussian_rasterization.LanguageGaussianRasterizer
    (GaussianRasterizer for F == 0), autograd forward + backward with the image cotangents fed straight to
    torch.autograd.backward (no loss kernels), fresh output / gradient / state tensors per call and the one host
    synchronisation the API implies (num_rendered is a Python int).  One frame in flight."""
    from diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                             LanguageGaussianRasterizer)
    cam = sc.camera
    rs = GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        projmatrix_raw=cam.projection_matrix.to(dev), sh_degree=sc.sh_degree, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=False)
    lang = sc.F > 0
    rast = (LanguageGaussianRasterizer if lang else GaussianRasterizer)(raster_settings=rs)
    names = ("means3D", "opacities", "scales", "rotations", "shs") + (("language",) if lang else ())
    p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
    means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    dc, dl, dd = [None if t is None else t.to(dev) for t in sc.cotangents(3)]
    leaves = list(p.values()) + [means2D, theta, rho]

    def step():
        kw = dict(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
                  rotations=p["rotations"], theta=theta, rho=rho)
        if lang:
            color, language, radii, depth, opacity, n_touched = rast(language_precomp=p["language"], **kw)
            outs, cots = [color, language, depth], [dc, dl, dd]
        else:
            color, radii, depth, opacity, n_touched = rast(**kw)
            outs, cots = [color, depth], [dc, dd]
        for t in leaves:
            t.grad = None
        torch.autograd.backward(outs, cots)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    for t in leaves:
        t.grad = None
    torch.cuda.reset_peak_memory_stats(dev)
    mem0 = torch.cuda.memory_allocated(dev)
    # (no events inside the timed loop: an event is a barrier packet on the stream, ~6 us per frame; the intervals come from a
    #  second pass)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    peak_mb = round((torch.cuda.max_memory_allocated(dev) - mem0) / 2**20, 1)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    gaps = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    res = {"path": "diff_gaussian_rasterization autograd API (forward + backward, torch allocations, 1 host sync)",
           "value": round(steps / dt, 3), "unit": "frames/s", "ms_per_frame": round(1e3 * dt / steps, 4), "steps": steps,
           "frame_interval_ms": percentiles(gaps),
           "peak_allocated_MB_per_frame": peak_mb}
    # the reference's mapping loop renders its window of keyframes one after the other on one stream: four arc views in turn
    # (the library keeps one tile order per view it has seen on the stream, so each view still starts its heavy tiles first)
    from online_lang_splatting_amd.scene import arc_cameras
    rasts = []
    for c in arc_cameras(cam.width, cam.height, n=8)[2:6]:
        rs_v = rs._replace(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                           campos=c.camera_center.to(dev))
        rasts.append((LanguageGaussianRasterizer if lang else GaussianRasterizer)(raster_settings=rs_v))
    k = [0]

    def step_alt():
        nonlocal rast
        rast = rasts[k[0] % len(rasts)]
        k[0] += 1
        step()
    for _ in range(3 * len(rasts)):
        step_alt()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_alt()
    torch.cuda.synchronize(dev)
    res["alternating_4_views"] = {"value": round(steps / (time.perf_counter() - t0), 3), "unit": "frames/s"}
    return res



def _sized_capacity(F, g_dev, c0, H, W, sh_degree, dev, cfg):
    """Instances of one synchronous forward under the knobs cfg = (tile, backward mode, binning)."""
    from online_lang_splatting_amd import _C
    e = torch.empty(0, device=dev)
    r = _C._forward(F, g_dev["bg"], g_dev["means3D"], e, g_dev["language"] if F > 0 else None, g_dev["opacities"],
                    g_dev["scales"], g_dev["rotations"], 1.0, e, c0["viewmatrix"], c0["projmatrix"], c0["projmatrix_raw"],
                    c0["tanfovx"], c0["tanfovy"], H, W, g_dev["shs"], sh_degree, c0["campos"], False, False, cfg=cfg)
    return int(r[0])


def bracket_legs(sc, g_dev, c0, cots, dev, steps, dims):
    """One frame in flight, un-profiled, through the same sync-free entry points as the headline, for the settings the
    headline does NOT use (missing #3): the exact backward (all 225 pixels of a tile, four waves, no
    survivor shortcut), 16x16 tiles (DGR-D/cuda_rasterizer/config.h:17-18; the reduction tree keeps every rank) and the
    reference's bounding-square tile lists (OLSR_BINNING_RECT).  They bracket the reference-mode headline from below."""
    from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace
    P, W, H, F, M = dims
    dc, dl, dd = cots
    legs = {}
    for name, cfg, flags in (("exact_mode", (15, _abi.BWD_EXACT, _abi.BINNING_ELLIPSE), 0),
                             ("tile16", (16, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE), 0),
                             ("rect_binning", (15, _abi.BWD_REFERENCE, _abi.BINNING_RECT), 0),
                             # ... and one that bounds it from above: the headline's settings with the forward's
                             # one-fma-per-channel accumulation (images to 1e-7 of the reference's rounding)
                             ("fwd_accum_weight", (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE), _abi.FLAG_FWD_ACCUM_WEIGHT)):
        tile, mode, binning = cfg
        R = _sized_capacity(F, g_dev, c0, H, W, sc.sh_degree, dev, cfg)
        # (rows: up to four per instance in the exact backward; a surface map blends nearly every listed pair — the room
        #  scene's exact-mode leg overflowed the one-row-per-instance default and reported the rate of a frame that rendered nothing)
        cap_ = int(R * 1.1) + (1 << 16)
        ws = RasterWorkspace(P, W, H, F, M, cap_, dev, tile=tile, bwd_mode=mode, binning=binning, flags=flags,
                             row_capacity=4 * cap_ if mode == _abi.BWD_EXACT or tile == 16 else 2 * cap_)
        bucket = GradientBucket(P, GradLayout(M, F), dev)

        def one():
            ws.set_scene(sh_degree=sc.sh_degree, **c0, **g_dev)
            ws.forward()
            ws.backward(dc, dl, dd, bucket=bucket, first=True, bucket_only=True)
        for _ in range(3):
            one()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        L_rows, row_ovf = ws.backward_status()
        legs[name] = {"tile": tile, "backward_mode": "exact" if mode == _abi.BWD_EXACT else "reference",
                      "binning": "rect" if binning == _abi.BINNING_RECT else "ellipse",
                      "forward_accumulation": "weight" if flags & _abi.FLAG_FWD_ACCUM_WEIGHT else "valu",
                      "value": round(steps / el, 3),
                      "unit": "frames/s", "ms_per_frame": round(1e3 * el / steps, 4), "steps": steps, "R_binned": R,
                      "live_gradient_rows": L_rows, "capacity_overflow": bool(ws.rendered()[1] or row_ovf)}
        del ws, bucket
    return legs


def config4_substitute(sc, g_dev, dev, dims, tracking_iters=60, mapping_iters=4, views=12):
    """BASELINE.json configs[3] (Replica room0 `slam.py` loop) cannot run in this image (no data, no SED / auto-encoder
    checkpoints, front-end dependencies absent: SURVEY.md section 8(d)); what it spends its rasterizer time in can:
      tracking iteration  utils/slam_frontend.py:218-243 — render, get_loss_tracking, backward to the pose only, Adam on the
                          pose increments, update_pose; iterations DEPEND on each other (latency, not throughput)
      mapping iteration   utils/slam_backend.py:510-760 — 12 views of the same Gaussians, mapping + language loss, summed
                          gradients, one Adam step
    on synthetic data of config 3's shape, entirely on this library (online_lang_splatting_amd/slam_iterations.py)."""
    from online_lang_splatting_amd.frame_shard import FrameLanes, RasterWorkspace
    from online_lang_splatting_amd.scene import arc_cameras, default_camera
    from online_lang_splatting_amd.slam_iterations import MappingStep, PoseState, TrackingLoop
    P, W, H, F, M = dims
    out = {"note": "substitute for BASELINE configs[3] (blocked: no Replica data / checkpoints / front-end dependencies); "
                   f"synthetic scene of config 3: {P} Gaussians, {W}x{H}, F={F}"}
    cam = default_camera(W, H)
    proj = cam.projection_matrix.to(dev)
    T_gt = torch.eye(4, device=dev)
    pose = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy)
    c_gt = pose.camera()
    R0 = _sized_capacity(F, g_dev, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in c_gt.items()}, H, W,
                         sc.sh_degree, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE))
    ws = RasterWorkspace(P, W, H, F, M, int(1.4 * R0) + (1 << 16), dev)
    ws.set_scene(sh_degree=sc.sh_degree, **c_gt, **g_dev)
    o = ws.forward()
    gt_image, gt_depth = o["color"].clone(), o["depth"][0].clone()
    # a few cm / a fraction of a degree off, like the constant-velocity prior of the front end
    tau0 = torch.tensor([0.02, -0.015, 0.01, 0.004, -0.006, 0.003])
    th = tau0[3:]
    Wm = torch.tensor([[0.0, -th[2], th[1]], [th[2], 0.0, -th[0]], [-th[1], th[0], 0.0]])
    T0 = torch.eye(4)
    T0[:3, :3] = torch.eye(3) + Wm + 0.5 * Wm @ Wm
    T0[:3, 3] = tau0[:3]
    T0 = T0.to(dev)
    # library stages of the iteration (HIP events between the stages, ~6 us each: a separate, profiled pass)
    pose.reset(T0)
    loop = TrackingLoop(ws, g_dev, sc.sh_degree, pose, gt_image, gt_depth, language_cotangent="null")
    for _ in range(3):
        loop.iteration()
    torch.cuda.synchronize(dev)
    _lib.set_profiling(True)
    for _ in range(10):
        loop.iteration()
    per = {}
    for name, ms in _lib.stage_times():
        per.setdefault(name, []).append(ms)
    _lib.set_profiling(False)
    stage = {k: round(sum(v) / len(v), 4) for k, v in per.items()}
    trk = {}
    for variant in ("null", "zeros"):
        for readback in (False, True):
            pose.reset(T0)
            loop = TrackingLoop(ws, g_dev, sc.sh_degree, pose, gt_image, gt_depth, language_cotangent=variant)
            for _ in range(5):
                loop.iteration(readback)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(tracking_iters):
                loop.iteration(readback)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            key = ("no_language_cotangent" if variant == "null" else "zero_language_cotangent") + \
                  ("_with_convergence_readback" if readback else "")
            trk[key] = round(1e3 * el / tracking_iters, 4)
    # the tracking loss reads colour, depth and opacity only: a front end may render the SAME images with the RGB rasterizer
    # (identical arithmetic for those channels) and skip the language accumulation altogether
    if F > 0:
        ws_rgb = RasterWorkspace(P, W, H, 0, M, int(1.4 * R0) + (1 << 16), dev)
        g_rgb = {k: v for k, v in g_dev.items() if k != "language"}
        g_rgb["language"] = None
        pose.reset(T0)
        loop = TrackingLoop(ws_rgb, g_rgb, sc.sh_degree, pose, gt_image, gt_depth)
        for _ in range(5):
            loop.iteration()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(tracking_iters):
            loop.iteration()
        torch.cuda.synchronize(dev)
        trk["rgb_rasterizer_render"] = round(1e3 * (time.perf_counter() - t0) / tracking_iters, 4)
        del ws_rgb
    # ... and the two-kernel formulation of round 3 (olsr_forward_async + olsr_tracking_loss) instead of the loss in the
    # composite's epilogue, for comparison
    pose.reset(T0)
    loop = TrackingLoop(ws, g_dev, sc.sh_degree, pose, gt_image, gt_depth, language_cotangent="null", fused_loss=False)
    for _ in range(5):
        loop.iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(tracking_iters):
        loop.iteration()
    torch.cuda.synchronize(dev)
    trk["no_language_cotangent_two_kernel_loss"] = round(1e3 * (time.perf_counter() - t0) / tracking_iters, 4)
    # Per-tile depth cut-offs (include/olsr.h, RasterWorkspace(depth_cut=True)): every iteration bins only what the
    # previous one needed; an iteration whose cut-offs hid something is a device-side no-op and is counted as lost.
    # Reported beside the plain loop, never as the headline: same poses (tests/test_gpu_depth_cut.py), shorter lists.
    ws_cut = RasterWorkspace(P, W, H, F, M, int(1.4 * R0) + (1 << 16), dev, depth_cut=True)
    pose_c = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy, device_step_count=True)
    pose_c.reset(T0)
    loop = TrackingLoop(ws_cut, g_dev, sc.sh_degree, pose_c, gt_image, gt_depth, language_cotangent="null")
    for _ in range(5):
        loop.iteration()
    torch.cuda.synchronize(dev)
    steps0 = loop.steps_done()
    t0 = time.perf_counter()
    for _ in range(tracking_iters):
        loop.iteration()
    torch.cuda.synchronize(dev)
    el_cut = time.perf_counter() - t0
    counted = loop.steps_done() - steps0
    _lib.set_profiling(True)
    for _ in range(10):
        loop.iteration()
    per = {}
    for name, ms in _lib.stage_times():
        per.setdefault(name, []).append(ms)
    _lib.set_profiling(False)
    stage_cut = {k: round(sum(v) / len(v), 4) for k, v in per.items()}
    trk["depth_cut_offs"] = round(1e3 * el_cut / tracking_iters, 4)
    out["tracking_depth_cut"] = {
        "ms_per_iteration": trk["depth_cut_offs"], "iterations": tracking_iters, "iterations_that_counted": counted,
        "ms_per_counted_iteration": round(1e3 * el_cut / max(counted, 1), 4),
        "instances_last_frame": ws_cut.rendered()[0], "instances_without_cut": ws.rendered()[0],
        "tiles_with_a_cut": int(torch.isfinite(ws_cut.depth_cut).sum()), "tiles": ws_cut.depth_cut.numel(),
        "library_stage_ms": stage_cut, "library_ms": round(sum(stage_cut.values()), 4),
        "pose_error_after": round(float((pose_c.T_w2c - T_gt).abs().max()), 6),
        "what": "the no_language_cotangent loop on a workspace with per-tile depth cut-offs: emission, tile sort and row "
                "compaction see only the instances in front of the depth at which each tile saturated one iteration earlier "
                "(x1.1 + 0.01); exact or flagged, a flagged iteration takes no pose step"}
    del ws_cut
    out["tracking_iteration_ms"] = trk["no_language_cotangent"]
    out["tracking"] = {"ms_per_iteration": trk, "iterations": tracking_iters,
                       "what": "render (language rasterizer, as gaussian_renderer.render does for a language map) with the "
                               "tracking loss in the composite's epilogue (olsr_forward_async_loss) + pose-only olsr_backward + "
                               "olsr_pose_step; dependent iterations",
                       "library_stage_ms": stage, "library_ms": round(sum(stage.values()), 4),
                       "pose_error_start": round(float((T0 - T_gt).abs().max()), 6),
                       "pose_error_after": round(float((pose.T_w2c - T_gt).abs().max()), 6)}
    # mapping iteration: raw parameters (what GaussianModel stores), activations folded into the kernels
    params = dict(means3D=g_dev["means3D"].clone(), shs=g_dev["shs"].clone(),
                  opacities=torch.logit(g_dev["opacities"].clamp(1e-4, 1 - 1e-4)).contiguous(),
                  scales=torch.log(g_dev["scales"]).contiguous(), rotations=g_dev["rotations"].clone(),
                  language=None if F == 0 else g_dev["language"].clone())
    cams = arc_cameras(W, H, n=views)
    camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in cams]
    lanes = FrameLanes(4, P, W, H, F, M, int(1.5 * R0) + (1 << 16), dev)
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    gen = torch.Generator().manual_seed(0)
    targets = []
    ws0 = lanes.lanes[0][0]
    step = MappingStep(lanes, params, g_dev["bg"], sc.sh_degree, camd, None, lrs, exposure=torch.zeros(2, device=dev))
    for c in camd:  # targets: renders of the scene itself, perturbed so that every loss term has a gradient
        o = step.render(ws0, c)
        targets.append((torch.clamp(o["color"] + 0.05 * torch.randn(3, H, W, generator=gen).to(dev), 0, 1).contiguous(),
                        (o["depth"][0] * (1 + 0.02 * torch.randn(H, W, generator=gen).to(dev))).contiguous(),
                        None if F == 0 else torch.nn.functional.normalize(torch.randn(F, 192, 192, generator=gen), dim=0).to(dev)))
    del step
    start = {k: (None if v is None else v.clone()) for k, v in params.items()}

    def mapping_leg(fused):
        """mapping_iters timed iterations from the same start parameters (four views in flight), then one PROFILED iteration
        whose HIP events give the breakdown: library stages per view (events on the lanes' streams), the stand-alone loss kernel
        (two-kernel formulation only), the sum of the lane buckets and the Adam step."""
        for k, v in params.items():
            if v is not None:
                v.copy_(start[k])
        st = MappingStep(lanes, params, g_dev["bg"], sc.sh_degree, camd, targets, lrs, exposure=torch.zeros(2, device=dev),
                         fused_loss=fused)
        st.iteration()
        first = float(st.last_loss[0])
        st.iteration()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(mapping_iters):
            st.iteration()
        issue = time.perf_counter() - t0  # host time to enqueue the iterations (the GPU is still working)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        last = float(st.last_loss[0])
        ovf = any(ws_.rendered()[1] or ws_.backward_status()[1] for ws_, _, _ in lanes.lanes)
        _lib.set_profiling(True)
        st.profile = True
        st.iteration()
        per = {}
        for name, ms in list(_lib.stage_times()) + list(st.stage_ms):
            per.setdefault(name, []).append(ms)
        _lib.set_profiling(False)
        st.profile = False
        per_view = {k: round(sum(v) / views, 4) for k, v in per.items() if k not in ("lane_sum", "adam")}
        return {"ms_per_iteration": round(1e3 * el / mapping_iters, 3), "views_per_s": round(views * mapping_iters / el, 1),
                "host_enqueue_ms_per_iteration": round(1e3 * issue / mapping_iters, 3),
                "loss_last_view_first_iteration": round(first, 6), "loss_last_view_final_iteration": round(last, 6),
                "capacity_overflow": bool(ovf),
                "profiled_iteration": {"stage_ms_per_view": per_view,
                                       "stage_ms_per_view_sum": round(sum(per_view.values()), 4),
                                       "lane_sum_ms": round(sum(per.get("lane_sum", [0.0])), 4),
                                       "adam_ms": round(sum(per.get("adam", [0.0])), 4),
                                       "note": "events between the stages (~6 us each) on four lane streams: an interval also "
                                               "holds the time a kernel queues behind other lanes' kernels"}}

    fused_leg = mapping_leg(True)
    two_leg = mapping_leg(False)
    # MappingStep's default: the form is MEASURED by the object itself (iterations 3-6 alternate the two between HIP events)
    for k, v in params.items():
        if v is not None:
            v.copy_(start[k])
    st_auto = MappingStep(lanes, params, g_dev["bg"], sc.sh_degree, camd, targets, lrs, exposure=torch.zeros(2, device=dev))
    for _ in range(9):   # (synchronised like a caller that reads the loss every iteration: the calibration's events complete)
        st_auto.iteration()
        torch.cuda.synchronize(dev)
    st_auto.iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(mapping_iters):
        st_auto.iteration()
    torch.cuda.synchronize(dev)
    auto_leg = {"ms_per_iteration": round(1e3 * (time.perf_counter() - t0) / mapping_iters, 3), "calibration": st_auto.calibration}
    del st_auto
    out["mapping_iteration_ms"] = auto_leg["ms_per_iteration"]
    out["mapping"] = {"views": views, "views_in_flight": len(lanes), "iterations": mapping_iters,
                      "what": f"{views} arc views x (render from raw parameters with the mapping loss incl. the 192x192 "
                              "language target in the composite's epilogue + backward into the gradient bucket) + sum of the "
                              "lane buckets + fused Adam step",
                      **fused_leg,
                      "two_kernel_loss": dict(what="olsr_forward_async + olsr_mapping_loss instead (round 3)", **two_leg),
                      "auto_loss": dict(what="MappingStep's default (fused_loss='auto'): the object times both forms in its "
                                             "iterations 3-6 and keeps the faster one", **auto_leg)}
    del lanes
    return out


def views_mode(a, sc, dev, rank, world, dist):
    """--views V: one step = one mapping iteration (utils/slam_backend.py:510-760 without the language front end):
    V viewpoints of the same Gaussians, view v rendered forward + backward by rank v mod N straight into the flat
    gradient bucket, ONE exchange of the bucket per step (--exchange), the fused Adam step, and — owner-applies —
    the all-gather of the updated parameter rows.  Total work is fixed as N grows: strong scaling."""
    from online_lang_splatting_amd import _C
    from online_lang_splatting_amd.frame_shard import (FrameShardedStep, FusedAdam, GradLayout, RasterWorkspace,
                                                       views_of_rank)
    cfg = CONFIGS[a.config]
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    M = sc.shs.shape[1]
    cams = arc_cameras(W, H, n=a.views)
    g_dev, _ = device_inputs(sc, cams[0], dev)
    cam_dev = [device_inputs(sc, c, dev)[1] for c in cams]
    cot = [None if t is None else t.to(dev) for t in sc.cotangents(a.config)]
    mine = views_of_rank(a.views, rank, world)
    need = 0
    for v in mine:  # instance capacity: the largest of this rank's views, measured once
        c = cam_dev[v]
        args = [g_dev["bg"], g_dev["means3D"], torch.empty(0, device=dev)] + ([g_dev["language"]] if F > 0 else []) + [
            g_dev["opacities"], g_dev["scales"], g_dev["rotations"], 1.0, torch.empty(0, device=dev), c["viewmatrix"],
            c["projmatrix"], c["projmatrix_raw"], c["tanfovx"], c["tanfovy"], H, W, g_dev["shs"], sc.sh_degree,
            c["campos"], False, False]
        need = max(need, int((_C.rasterize_language_gaussians if F > 0 else _C.rasterize_gaussians)(*args)[0]))
    # this rank's views are rendered a.streams at a time (FrameLanes: one workspace, bucket and HIP stream per lane)
    lanes = FrameLanes(max(1, min(a.streams, len(mine))), P, W, H, F, M, int(need * 1.3) + (1 << 16), dev)
    step = FrameShardedStep(lanes, rank, world, exchange=a.exchange)
    adam = FusedAdam(P, GradLayout(M, F), dev)
    params = dict(means3D=g_dev["means3D"], shs=g_dev["shs"], opacities=g_dev["opacities"], scales=g_dev["scales"],
                  rotations=g_dev["rotations"], language=g_dev["language"])
    lrs = dict(xyz=1.6e-5, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.01, scale=1e-4, rotation=1e-4, language=2.5e-3)

    def one():
        step.run(g_dev, cam_dev, lambda v, out: cot, sh_degree=sc.sh_degree)
        step.optimizer_step(adam, params, lrs)
    for _ in range(a.warmup):
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        width = 11 + 3 * M + F
        print(json.dumps({
            "metric": f"mapping iteration: {a.views} views rasterizer fwd+bwd + gradient exchange + Adam, frames/sec",
            "value": round(a.views * a.steps / el, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{a.config - 1}] Gaussians, {a.views} arc viewpoints per step, "
                                   f"view v on rank v mod {world}, exchange {a.exchange}, fused Adam",
                       "P": P, "width": W, "height": H, "F": F, "views_per_step": a.views,
                       "views_in_flight_per_gpu": len(lanes),
                       "views_of_rank0": len(views_of_rank(a.views, 0, world)), "parallelism": f"frame-shard x{world}",
                       "exchange": a.exchange, "bucket_bytes": P * width * 4,
                       "exchange_chosen": (step.wire or {}).get("chosen", a.exchange) if a.exchange == "auto" else a.exchange,
                       "wire": step.wire}}), file=JSON_OUT, flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command under torch.distributed.run, one rank per
    GPU on this node (rendezvous on 127.0.0.1, a free port), and pass its output through.  Over RCCL this needs N GPUs:
    checked here, before anything is started, so that a SCALE run on too small a node fails instead of reporting one GPU."""
    import socket
    import subprocess
    backend = os.environ.get("OLSR_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} over RCCL needs {n} GPUs, this node shows {have} (OLSR_BENCH_BACKEND=gloo lets ranks "
              "share a device - a functional check only)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
          [x for x in sys.argv[1:] if x != "--self-launch"]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the bracket legs and the config-4 substitute")
    ap.add_argument("--mode", default="reference", choices=["reference", "exact"])
    ap.add_argument("--binning", default="ellipse", choices=["ellipse", "rect"],
                    help="ellipse: exact tile lists (default); rect: the reference's bounding-square lists")
    ap.add_argument("--fwd-accum", default="valu", choices=["valu", "mfma", "weight"],
                    help="forward feature accumulation: vector ALU (bit-identical to the oracle) or matrix cores "
                         "(OLSR_FLAG_FWD_ACCUM_MFMA: images to ~1e-7 relative)")
    ap.add_argument("--streams", type=int, default=4, help="frames in flight per GPU (workspaces on separate HIP streams)")
    ap.add_argument("--views", type=int, default=0,
                    help="mapping-iteration mode: every step renders this many viewpoints in total, view v on rank v mod N "
                         "(BackEnd.map renders 12, utils/slam_backend.py:510-670), then ONE exchange of the bucket; "
                         "0 (default): one view per rank per step, weak scaling")
    ap.add_argument("--exchange", default="auto", choices=["auto", "all_reduce", "reduce_scatter", "sparse"],
                    help="how the shared-Gaussian gradients travel when N > 1.  auto (default): chosen from the data - the "
                         "union of the ranks' non-zero gradient rows is measured (at set-up in the weak-scaling mode, every "
                         "step with --views) and the rows travel packed (sparse) only when that is the smaller payload by a "
                         "quarter, else the bucket goes densely in two direct phases (reduce_scatter); sparse: only the rows "
                         "that are non-zero on some rank (2 %% of the rows of a config-3 view, 20 %% of a room-map view) - "
                         "capacity-bound and sync-free in the default weak-scaling mode, exact in --views mode; all_reduce: "
                         "the whole bucket; reduce_scatter: two direct phases over all xGMI links (with --views: "
                         "owner-applies Adam in between)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (N > 1 does so by itself when "
                         "WORLD_SIZE is not set)")
    ap.add_argument("--rank-view", default="", help="K,N[,arc]: with one rank, render what rank K of an N-GPU run renders instead of "
                                                    "the identity pose (the per-rank cost of a weak-scaling point); ',arc': "
                                                    "pose K of the N-pose arc of the mapping mode")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the contract's K-step region is run; the value is the median run (all in "
                         "config.value_runs).  0 (default): as many as it takes for the timed regions to total --min-timed-s "
                         "seconds (at least 5)")
    ap.add_argument("--min-timed-s", type=float, default=2.0,
                    help="with --repeats 0: repeat the K-step region until the timed regions total this many seconds, so that "
                         "the GPU is visibly busy to an outside sampler (the K = 20 region alone lasts 9 ms)")
    ap.add_argument("--scene", default="volume", choices=["volume", "room"],
                    help="volume (default): SURVEY 8(d)'s i.i.d. Gaussians, the BASELINE workload; room: the surface-structured "
                         "map of scene.make_room_scene (a SLAM map by the reference's recipe) at the config's size, rank r "
                         "rendering keyframe r of the mapping window")
    ap.add_argument("--exchange-via", default="torch", choices=["rccl", "torch"],
                    help="N > 1 over RCCL: who issues the step's collectives - torch (default): torch.distributed, on its own stream "
                         "beside the lane's next forward; rccl: RCCL's C API directly on the lane's stream (rccl_direct.py) - no hop, "
                         "but the collective then serialises with the lane: measured SLOWER, profiles/r6_exchange_via.json")
    ap.add_argument("--carry-order", type=int, default=1, choices=[0, 1],
                    help="1 (default): the lanes carry their view's depth order from frame to frame (olsr_scene.depth_order_carry: "
                         "repaired in two launches, exact by a device-side fall-back to the radix passes); the legs whose camera "
                         "changes every step run without, and the line reports the same-view rates without it as well")
    ap.add_argument("--setup-steps", type=int, default=40, help="untimed frames before the W warm-up steps (steady state)")
    ap.add_argument("--isolated-steps", type=int, default=30, help="steps of the single-stream re-measurement (0 = skip)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.self_launch):
        sys.exit(self_launch(a.gpus))
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to the process's stdout when its first
    # communicator is created ("RCCL version : ...", measured in round 4 on a one-rank group), and native code may print
    # more.  File descriptor 1 is pointed at stderr for the life of the process; the line goes to the saved descriptor.
    global JSON_OUT
    sys.stdout.flush()
    JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    # OLSR_BENCH_FORCE_EXCHANGE=1 with one rank: a group of ONE over RCCL, every collective of the chosen exchange issued
    # (identities there) - what a one-GPU box can measure of the multi-GPU step: the exchange's local kernels and launches,
    # everything but the wire (reported as n_gpus 1 with config.exchange set)
    forced = world == 1 and os.environ.get("OLSR_BENCH_FORCE_EXCHANGE") == "1"
    if forced and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one process per GPU over RCCL ("nccl" on ROCm).  OLSR_BENCH_BACKEND=gloo exists only to exercise the N > 1
        # code path on a box with fewer GPUs than ranks (ranks then share devices; the numbers mean nothing).
        backend = os.environ.get("OLSR_BENCH_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows "
                             f"{torch.cuda.device_count()} (OLSR_BENCH_BACKEND=gloo lets ranks share a device - a functional "
                             "check only)")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
        if forced:
            from online_lang_splatting_amd.frame_shard import GradientBucket
            GradientBucket.exchange_single_rank = True
            GradientBucket.capped_torch_formulation = os.environ.get("OLSR_BENCH_EXCHANGE_TORCH") == "1"
        # --exchange-via rccl: the step's collectives through RCCL's C API on the lane's own stream (rccl_direct.py) instead of
        # torch.distributed's hop to the process group's stream and back (built, measured, slower)
        if backend == "nccl" and a.exchange_via == "rccl":
            from online_lang_splatting_amd.frame_shard import GradientBucket
            from online_lang_splatting_amd.rccl_direct import DirectComm
            GradientBucket.direct_comm = DirectComm.from_process_group(device=torch.device("cuda", local_rank))
            exchange_via = "rccl C API on the lane's stream"
        else:
            exchange_via = "torch.distributed"
    else:
        dist = None
        exchange_via = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    cfg = CONFIGS[a.config]
    P, W, H, F = cfg["P"], cfg["W"], cfg["H"], cfg["F"]
    room = None
    if a.scene == "room":
        from online_lang_splatting_amd.scene import make_room_scene
        room = make_room_scene(P, W, H, F, views=max(world, 10), random_views=2, seed=a.config,
                               max_sh_degree=cfg["max_sh_degree"])
        sc = room.scene
    else:
        sc = make_scene(P, W, H, F, seed=a.config, max_sh_degree=cfg["max_sh_degree"])
    P = sc.P
    M = sc.shs.shape[1]
    if a.views > 0:
        views_mode(a, sc, dev, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # weak scaling = the same work per GPU whatever N: rank r renders pose r of N poses 3 cm apart without rotation, whose
    # cost equals the identity pose's (n == 1 -> the identity pose of config 3); the arc's rotated poses cost up to 12 % more
    cams = shard_cameras(W, H, n=max(world, 1)) if room is None else room.cameras[:max(world, 1)]
    if a.rank_view and world == 1:
        k_, n_ = (int(x) for x in a.rank_view.split(",")[:2])
        cams = [(arc_cameras if a.rank_view.endswith("arc") else shard_cameras)(W, H, n=n_)[k_]]
    g_dev, _ = device_inputs(sc, cams[0], dev)
    cam_dev = [device_inputs(sc, c, dev)[1] for c in cams]
    dc, dl, dd = [None if t is None else t.to(dev) for t in sc.cotangents(a.config)]
    mode = _abi.BWD_REFERENCE if a.mode == "reference" else _abi.BWD_EXACT

    # size the instance capacity from one synchronous forward of this rank's view
    from online_lang_splatting_amd import _C
    binning = _abi.BINNING_ELLIPSE if a.binning == "ellipse" else _abi.BINNING_RECT
    _C.BINNING = binning
    my_view = rank % len(cams)
    c0 = cam_dev[my_view]
    r = _C.rasterize_language_gaussians(g_dev["bg"], g_dev["means3D"], torch.empty(0, device=dev), g_dev["language"],
                                        g_dev["opacities"], g_dev["scales"], g_dev["rotations"], 1.0,
                                        torch.empty(0, device=dev), c0["viewmatrix"], c0["projmatrix"],
                                        c0["projmatrix_raw"], c0["tanfovx"], c0["tanfovy"], H, W, g_dev["shs"],
                                        sc.sh_degree, c0["campos"], False, False) if F > 0 else \
        _C.rasterize_gaussians(g_dev["bg"], g_dev["means3D"], torch.empty(0, device=dev), g_dev["opacities"],
                               g_dev["scales"], g_dev["rotations"], 1.0, torch.empty(0, device=dev), c0["viewmatrix"],
                               c0["projmatrix"], c0["projmatrix_raw"], c0["tanfovx"], c0["tanfovy"], H, W,
                               g_dev["shs"], sc.sh_degree, c0["campos"], False, False)
    R = int(r[0])
    del r
    _C.BINNING = _abi.BINNING_ELLIPSE
    capacity = int(R * 1.25) + (1 << 16)
    fwd_flags = {"mfma": _abi.FLAG_FWD_ACCUM_MFMA, "weight": _abi.FLAG_FWD_ACCUM_WEIGHT}.get(a.fwd_accum, 0)
    carry = bool(a.carry_order)
    lanes = FrameLanes(a.streams, P, W, H, F, M, capacity, dev, tile=15, bwd_mode=mode, binning=binning, flags=fwd_flags,
                       carry_order=carry)

    class carried_order_off:
        """the legs in which a lane's previous frame is another view (or that measure the plain sort): no carried order"""
        def __init__(self, *lane_sets):
            self.ws = [l_[0] for ls in lane_sets for l_ in ls]

        def __enter__(self):
            self.keep = [w_.depth_order_carry for w_ in self.ws]
            for w_ in self.ws:
                w_.depth_order_carry = None

        def __exit__(self, *exc):
            for w_, k_ in zip(self.ws, self.keep):
                w_.depth_order_carry = k_

    pending = {}  # bucket id -> outstanding all-reduce handles of that lane's previous frame
    step_done = []  # one event per step of the current timed region, recorded on the step's stream
    sparse_cap = [0]  # rows of the capacity-bound sparse exchange (sized below, from the measured row sparsity)

    chosen = [a.exchange]  # "auto" is resolved below, from the measured union of the ranks' gradient rows

    # OLSR_BENCH_EXCHANGE_STREAM=1 (an experiment of round 5, OFF by default): the exchange on ONE side stream shared by the
    # lanes, so that a lane's stream goes on to its next forward while the exchange's launches and the hops to RCCL's own
    # stream and back happen beside it (the lane's next backward waits for the completion event).  Measured on one rank over
    # RCCL it is WORSE than the exchange in the lane's own stream — room map 3 274 against 3 764 fps (sparse), volume 1 999
    # against 2 217: a sixth stream takes a hardware queue from the lanes.
    class _Waitable:
        def __init__(self, ev, stream):
            self.ev, self.stream = ev, stream

        def wait(self):
            torch.cuda.current_stream(dev).wait_event(self.ev)
    side = [None]
    use_side = os.environ.get("OLSR_BENCH_EXCHANGE_STREAM", "0") == "1"

    def exchange(bucket):
        """The step's one exchange of the shared-Gaussian gradients (no host sync)."""
        if dist is None:
            return
        if use_side and chosen[0] != "all_reduce":
            if side[0] is None:
                side[0] = torch.cuda.Stream(dev)
            lane_stream = torch.cuda.current_stream(dev)
            side[0].wait_stream(lane_stream)
            with torch.cuda.stream(side[0]):
                if chosen[0] == "sparse":
                    bucket.sparse_all_reduce_capped(sparse_cap[0])
                else:
                    bucket.reduce_scatter_all_gather(rank, world)
                ev = torch.cuda.Event()
                ev.record(side[0])
            pending[id(bucket)] = [_Waitable(ev, side[0])]
            return
        if chosen[0] == "all_reduce":
            pending[id(bucket)] = bucket.all_reduce(async_op=True)
        elif chosen[0] == "sparse":
            bucket.sparse_all_reduce_capped(sparse_cap[0])
        else:
            bucket.reduce_scatter_all_gather(rank, world)

    view_cycle = [None]  # non-coherent leg: the cameras one_step cycles through instead of the rank's own view
    step_no = [0]

    def one_step(lane, record=False):
        # every rank renders exactly one view per step (weak scaling): its own
        ws, bucket, stream = lane
        cam_ = c0
        if view_cycle[0] is not None:
            cam_ = view_cycle[0][step_no[0] % len(view_cycle[0])]
            step_no[0] += 1
        with torch.cuda.stream(stream):
            ws.set_scene(sh_degree=sc.sh_degree, **cam_, **g_dev)
            out = ws.forward()
            # the previous frame's all-reduce of this lane ran while the forward above was enqueued and executed;
            # the bucket is only rewritten by the backward below
            for w_ in pending.pop(id(bucket), ()):
                w_.wait()
            # gradients go straight into the flat bucket (what a mapping step consumes): dL_dmeans3D, dL_dsh,
            # dL_dopacity, dL_dscales, dL_drotations, dL_dlanguage + densification statistics + dL_dtau_sum
            ws.backward(dc, dl, dd, bucket=bucket, first=True, bucket_only=True)
            exchange(bucket)
            if record:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream)
                step_done.append(ev)

    def timed(nsteps, warmup, pick, profile=False, events=True):
        for _ in range(warmup):
            one_step(pick())
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        # per-stage HIP events cost ~6 us each (a barrier packet between two kernels): only the separate `profiled`
        # leg records them, the timed region and the isolated leg run without
        _lib.set_profiling(profile and rank == 0)
        step_done.clear()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        t0 = time.perf_counter()
        for _ in range(nsteps):
            one_step(pick(), record=events)
        for lane_ in lanes.lanes:  # the last frames' exchanges belong to the timed region
            with torch.cuda.stream(lane_[2]):
                for w_ in pending.pop(id(lane_[1]), ()):
                    w_.wait()
        issue = time.perf_counter() - t0  # host time to enqueue everything (diagnostic: << el when GPU-bound)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if os.environ.get("OLSR_BENCH_DEBUG"):
            print(f"[bench] enqueue {1e3 * issue / nsteps:.3f} ms/step, total {1e3 * el / nsteps:.3f} ms/step", file=sys.stderr)
        st = _lib.stage_times() if (profile and rank == 0) else []
        _lib.set_profiling(False)
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per, frames_ms = {}, []
        for name, ms in st:
            per.setdefault(name, []).append(ms)
            if name == "preprocess":
                frames_ms.append(0.0)
            if frames_ms:
                frames_ms[-1] += ms
        # completion time of every step since the start of the timed region -> intervals between completions
        done = sorted(ev0.elapsed_time(e) for e in step_done)
        gaps = [b_ - a_ for a_, b_ in zip([0.0] + done[:-1], done)]
        lat = {"step_completion_interval_ms": percentiles(gaps), "frame_gpu_ms": percentiles(frames_ms)}
        return float(t.item()), {k: sum(v) / len(v) for k, v in per.items()}, lat

    # Setup, untimed: every lane runs a few frames so that allocations, code objects, the tile-order hints and the clocks
    # are in their steady state before the contract's W warm-up and K timed steps (with K = 20 the timed region is ~10 ms,
    # and it used to swing by +-8 % with what happened to precede it)
    active_rows = None
    if dist is not None:
        # row sparsity of this workload: saturation ends most tile lists early, so only the front layer of Gaussians
        # receives any gradient.  One frame per rank, rows counted, MAX over the ranks -> capacity of the packed buffer
        # (the union over the ranks holds at most world x that many rows; 25 % head-room)
        ws_, bucket_, stream_ = lanes.lanes[0]
        ws_.set_scene(sh_degree=sc.sh_degree, **c0, **g_dev)
        ws_.forward()
        ws_.backward(dc, dl, dd, bucket=bucket_, first=True, bucket_only=True)
        nz = (bucket_.flat != 0).any(dim=1).sum().to(torch.int64).reshape(1)
        dist.all_reduce(nz, op=dist.ReduceOp.MAX)
        active_rows = int(nz.item())
        sparse_cap[0] = min(P, int(1.25 * world * active_rows) + 4096)
        union_rows = None
        if a.exchange in ("sparse", "auto"):
            # ... and the union itself: neighbouring views share most of their front layer (the overlapping keyframes of a
            # mapping window do too), so the union is far smaller than world x rows.  One exchange at the safe capacity,
            # its row count read back (set-up: the one host synchronisation of this path), 25 % head-room on that.
            union_rows = int(bucket_.sparse_all_reduce_capped(sparse_cap[0])[0].item())  # identical on every rank
            sparse_cap[0] = min(P, int(1.25 * union_rows) + 4096)
        if a.exchange == "auto":
            # the exchange is chosen from the data (identical on every rank: the union is): packed rows only when the
            # capacity-bound payload is the smaller one by a quarter (GradientBucket.sparse_pays), else the dense bucket in
            # two direct phases
            pays_, sp_b_, de_b_ = bucket_.sparse_pays(union_rows, capacity=sparse_cap[0])
            chosen[0] = "sparse" if pays_ else "reduce_scatter"
    for _ in range(a.setup_steps):
        one_step(lanes.next_lane())
    for lane_ in lanes.lanes:
        with torch.cuda.stream(lane_[2]):
            for w_ in pending.pop(id(lane_[1]), ()):
                w_.wait()
    torch.cuda.synchronize(dev)
    # the headline region: no per-stage events, no per-step events (an event is a barrier packet on its stream)
    # The contract's region (W warm-up steps, then exactly K timed steps between barriers) is run `--repeats` times back to
    # back; the line's value is the MEDIAN run, all runs are listed (with K = 20 the region is
    # 10 ms long and a single run swings by several per cent).
    runs = []
    while True:
        runs.append(timed(a.steps, a.warmup if not runs else 0, lanes.next_lane, events=False))
        if a.repeats > 0:
            if len(runs) >= a.repeats:
                break
        elif len(runs) >= 5 and (sum(r_[0] for r_ in runs) >= a.min_timed_s or len(runs) >= 2000):
            # (identical on every rank: the elapsed times are the MAX over the ranks)
            break
    run_fps = [world * a.steps / r_[0] for r_ in runs]
    order_ = sorted(range(len(runs)), key=lambda i: run_fps[i])
    elapsed, avg, lat = runs[order_[len(runs) // 2]]   # the median run (the upper one of an even count), as value_runs.median
    iso = prof = nonco = None
    # ONE frame in flight is its own workspace: the lanes' workspaces carry OLSR_FLAG_FRAMES_IN_FLIGHT (four-wave radix blocks,
    # which get onto the CUs beside another lane's composite: + 2 % with four frames in flight, - 11 % with one)
    iso_lane = FrameLanes(1, P, W, H, F, M, capacity, dev, tile=15, bwd_mode=mode, binning=binning, flags=fwd_flags,
                          carry_order=carry).lanes[0] \
        if (a.isolated_steps > 0 and len(lanes) > 1) else lanes.lanes[0]
    plain_sort = None   # the same-view rates WITHOUT the carried depth order (the five-launch sort every frame)
    if carry and world == 1:
        with carried_order_off(lanes.lanes):
            for _ in range(2 * len(lanes)):
                one_step(lanes.next_lane())
            p4 = timed(a.steps, a.warmup, lanes.next_lane, events=False)
        plain_sort = {"value": round(a.steps / p4[0], 3), "unit": "frames/s", "frames_in_flight_per_gpu": len(lanes)}
        for _ in range(2 * len(lanes)):
            one_step(lanes.next_lane())
    if a.isolated_steps > 0:
        for _ in range(8):
            one_step(iso_lane)
        iso = timed(a.isolated_steps, 3, lambda: iso_lane)
        prof = timed(a.isolated_steps, 3, lambda: iso_lane, profile=True)
        if plain_sort is not None:
            with carried_order_off([iso_lane]):
                for _ in range(3):
                    one_step(iso_lane)
                p1 = timed(a.isolated_steps, 3, lambda: iso_lane)
                p1p = timed(a.isolated_steps, 3, lambda: iso_lane, profile=True)
            plain_sort["isolated_value"] = round(a.isolated_steps / p1[0], 3)
            plain_sort["isolated_depth_sort_ms"] = round(p1p[1].get("depth_sort", 0.0), 4)
            for _ in range(3):
                one_step(iso_lane)
        if world == 1 and not a.no_extra_legs:
            # non-coherent frames: the camera changes EVERY step (eight arc views, yaw -14 .. +14 degrees), so a lane's
            # tile-order hint comes from another view and nothing of the previous frame can be reused
            view_cycle[0] = [device_inputs(sc, c_, dev)[1] for c_ in (arc_cameras(W, H, n=8) if room is None else room.cameras)]
            step_no[0] = 0
            with carried_order_off(lanes.lanes, [iso_lane]):
                nc4 = timed(a.steps, a.warmup, lanes.next_lane, events=False)
                nc1 = timed(a.isolated_steps, 3, lambda: iso_lane)
            view_cycle[0] = None
            nonco = {"what": "the camera changes every step (8 arc views, yaw -14..+14 deg, 0.15 m apart): the tile-order hint "
                             "a lane carries belongs to another view",
                     "value": round(a.steps / nc4[0], 3), "unit": "frames/s", "frames_in_flight_per_gpu": len(lanes),
                     "isolated_value": round(a.isolated_steps / nc1[0], 3), "carried_depth_order": False,
                     "overflow": bool(any(l_[0].rendered()[1] for l_ in lanes.lanes))}
            # (back to the coherent steady state for the legs below)
            for _ in range(2 * len(lanes)):
                one_step(lanes.next_lane())
            one_step(iso_lane)
            torch.cuda.synchronize(dev)
    ws0 = iso_lane[0] if a.isolated_steps > 0 else lanes.lanes[0][0]
    exch_detail = None
    if dist is not None:
        b0 = lanes.lanes[0][1]
        exch_detail = {"bucket_bytes": b0.sum_storage.numel() * 4 + P * 4, "gradient_rows": P,
                       "rows_nonzero_per_view_max_over_ranks": active_rows,
                       "live_row_fraction_per_view": round(active_rows / max(P, 1), 4),
                       "requested": a.exchange, "chosen": chosen[0]}
        if union_rows is not None:
            pays_, sp_b_, de_b_ = b0.sparse_pays(union_rows, capacity=sparse_cap[0])
            exch_detail.update({"rows_in_union_at_setup": union_rows, "union_fraction": round(union_rows / max(P, 1), 4),
                                "sparse_payload_bytes": sp_b_, "dense_payload_bytes": de_b_, "sparse_pays": bool(pays_),
                                "rule": "sparse iff 8 P + (capacity x width + 2 P) x 4 < 0.75 x dense payload"})
        if chosen[0] == "sparse":
            stt = torch.stack([lane_[1]._capped["status"] for lane_ in lanes.lanes if getattr(lane_[1], "_capped", None)])
            exch_detail.update({"packed_capacity_rows": sparse_cap[0], "rows_in_union_last_step": int(stt[:, 0].max()),
                                "overflow": bool(int(stt[:, 1].max()))})
        # The exchange, checked where it ran (outside every timed region): one more frame of this rank's view into lane 0's
        # bucket, the chosen exchange on it, against the dense all-reduce (SUM of gradients and statistics, MAX of the radii)
        # of a copy of the same partial sums.  Two statements come back in the line: every rank holds the SAME bucket
        # afterwards (checksums of the bits, up to the sign of zeros), and it is the dense result (bit for bit with two ranks; beyond two the
        # collectives may associate the ranks' terms differently for buffers of different size, so the largest difference
        # relative to the tensor's largest element is reported and held to 1e-5).
        ws_, bucket_, stream_ = lanes.lanes[0]
        with torch.cuda.stream(stream_):
            for w_ in pending.pop(id(bucket_), ()):
                w_.wait()
            ws_.set_scene(sh_degree=sc.sh_degree, **c0, **g_dev)
            ws_.forward()
            ws_.backward(dc, dl, dd, bucket=bucket_, first=True, bucket_only=True)
            ref_sum, ref_rad = bucket_.sum_storage.clone(), bucket_.max_radii.clone()
            if world > 1:
                dist.all_reduce(ref_sum, op=dist.ReduceOp.SUM)
                dist.all_reduce(ref_rad, op=dist.ReduceOp.MAX)
            exchange(bucket_)
            for w_ in pending.pop(id(bucket_), ()):
                w_.wait()
            got = bucket_.sum_storage
            scale_ = ref_sum.abs().max().clamp_min(1e-30)
            diff = ((got - ref_sum).abs().max() / scale_).to(torch.float64).reshape(1)
            same_rad = (bucket_.max_radii == ref_rad).all().to(torch.float64).reshape(1)
            # identical on every rank: compare the bits through an order-independent integer checksum
            # (-0.0 + 0.0 = +0.0: a row outside the union keeps this rank's own zeros, whose sign the dense sum would erase)
            csum = (got + 0.0).view(torch.int32).to(torch.int64).sum().reshape(1)
            cmax, cmin = csum.clone(), csum.clone()
            if world > 1:
                dist.all_reduce(diff, op=dist.ReduceOp.MAX)
                dist.all_reduce(same_rad, op=dist.ReduceOp.MIN)
                dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize(dev)
        exch_detail["check"] = {"what": "one frame per rank after the timed regions: the chosen exchange against the dense "
                                        "all-reduce of the same partial sums",
                                "max_abs_diff_over_max_abs": float(diff.item()),
                                "equals_dense": bool(float(diff.item()) <= (0.0 if world <= 2 else 1e-5)),
                                "radii_equal": bool(same_rad.item() == 1.0),
                                "identical_on_every_rank": bool(int(cmax.item()) == int(cmin.item())),
                                "nonzero_gradient_rows_after": int((bucket_.flat != 0).any(dim=1).sum().item())}
    Rr, overflow = ws0.rendered()
    # the reference's num_rendered (bounding-square instances) of this view: the R of the byte model
    R_ref = int(_C.state_field("geometry", ws0.geom, "counters", P=P, F=F, dtype=torch.int32, count=8)[3])
    L_rows, row_overflow = ws0.backward_status()
    wstats = workload_stats(ws0, P, W, H, F) if rank == 0 else None

    if rank == 0:
        frames = world * a.steps
        fps = frames / elapsed
        N = W * H
        model = algorithmic_bytes(P, Rr, N, 3, F, M)          # units the launches process (exact tile lists)
        model_ref = algorithmic_bytes(P, R_ref, N, 3, F, M)   # units of the reference algorithm (its num_rendered)

        def roofline_of(av, where):
            comp = {k: av[k] for k in ("render_forward", "render_backward") if k in av}
            if not comp:
                return None
            dom = max(comp, key=comp.get)
            t_s = av[dom] * 1e-3
            achieved = model[dom] / t_s / 1e9
            traffic, traffic_src = measured_traffic(dom, F, a.scene, a.config)
            valu, valu_src = measured_valu(dom, F, a.scene, a.config)
            r = {"bound": "valu", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                 "traffic_source": traffic_src, "units": {"instances_processed": Rr, "pixels": N},
                 "algorithmic_bytes_per_launch": int(model[dom]), "avg_launch_ms": round(av[dom], 4),
                 "measured_in": where,
                 "model_reference_R": {"instances": R_ref, "algorithmic_bytes_per_launch": int(model_ref[dom]),
                                       "achieved": round(model_ref[dom] / t_s / 1e9, 2),
                                       "frac": round(model_ref[dom] / t_s / 1e9 / HBM_PEAK_GBS, 5)}}
            if traffic:
                r["traffic_frac"] = round(traffic / t_s / 1e9 / HBM_PEAK_GBS, 5)
            if valu:
                ips = valu["SQ_INSTS_VALU"] / t_s
                cyc_meas, cyc_src = measured_valu_cycles()
                r["valu"] = {"insts_per_launch": int(valu["SQ_INSTS_VALU"]), "achieved": round(ips / 1e9, 1),
                             "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "G wave-instructions/s",
                             "frac": round(ips / VALU_ISSUE_PEAK, 4),
                             "peak_basis": "MI355X_MICROARCH.md: v_fma_f32 (wave64) 2 cycles per SIMD-32",
                             "cycles_per_inst_per_simd": round(1024 * 2.4e9 / ips, 2),
                             "busy_by_counters": valu.get("valu_busy"), "source": valu_src}
                if cyc_meas:
                    r["valu"]["peak_measured"] = {"cycles_per_full_rate_inst": round(cyc_meas, 2),
                                                  "G_wave_instructions_per_s": round(1024 * 2.4e9 / cyc_meas / 1e9, 1),
                                                  "frac": round(ips / (1024 * 2.4e9 / cyc_meas), 4), "source": cyc_src}
            return r
        roof = roofline_of(prof[1], "profiled leg, 1 frame in flight: event intervals == kernel durations") \
            if prof is not None else None
        gpu_ms = sum(prof[1].values()) if prof is not None else None
        frame = {"algorithmic_bytes": int(model["frame"]),
                 "achieved_GBs_wall": round(model["frame"] * fps / world / 1e9, 2),
                 "frac_of_8TBs_wall": round(model["frame"] * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                 "algorithmic_bytes_reference_R": int(model_ref["frame"]),
                 "frac_of_8TBs_wall_reference_R": round(model_ref["frame"] * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                 "gpu_stage_ms_sum_1_in_flight": None if gpu_ms is None else round(gpu_ms, 4)}
        out = {
            "metric": "rasterizer fwd+bwd frames/sec @500k Gaussians, 1200x680, 15-dim lang"
                      if a.config == 3 else f"rasterizer fwd+bwd frames/sec, config {a.config}",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # what `value` is a rate OF: this many frames in flight per GPU, the same camera every step (the headline's
            # tile-order hints therefore belong to the view rendered; `non_coherent` and `isolated` are the other cases)
            "frames_in_flight": len(lanes), "same_view_every_step": True,
            "config": {"workload": (f"BASELINE.json configs[{a.config - 1}]: {P} Gaussians, {W}x{H}, RGB+depth+{F} "
                                    if room is None else
                                    f"room map (scene.make_room_scene) at the size of BASELINE.json configs[{a.config - 1}]: "
                                    f"{sc.P} Gaussians of {room.keyframes} keyframes, {W}x{H}, RGB+depth+{F} ") +
                                   f"language channels, forward+backward, tile 15, backward mode {a.mode}; "
                                   f"{a.setup_steps} untimed set-up frames precede the warm-up (allocations, tile-order "
                                   "hints, clocks)" +
                                   ("; every lane carries its view's depth order from its previous frame (repaired in two launches, "
                                    "the radix passes as the device-side fall-back: lists bit-identical; `carried_depth_order`)"
                                    if carry else ""),
                       "scene": a.scene,
                       "carried_depth_order": {"on": carry, "same_view_rates_without": plain_sort,
                                               "missed_last_frame": bool(ws0.carry_missed()) if carry else None},
                       "value_runs": {"runs": len(run_fps), "steps_per_run": a.steps,
                                      "timed_seconds_total": round(sum(r_[0] for r_ in runs), 3),
                                      "median": round(sorted(run_fps)[len(run_fps) // 2], 3),
                                      "min": round(min(run_fps), 3), "max": round(max(run_fps), 3),
                                      "p10": round(sorted(run_fps)[int(0.1 * (len(run_fps) - 1))], 3),
                                      "p90": round(sorted(run_fps)[int(0.9 * (len(run_fps) - 1) + 0.5)], 3),
                                      "fps": [int(round(x)) for x in run_fps],
                                      "note": "the contract's K-step region (barrier, K steps, barrier) repeated back to back "
                                              "until the timed regions total --min-timed-s; value = the median run"},
                       "frames_in_flight": len(lanes), "same_view_every_step": True,
                       "P": sc.P, "width": W, "height": H, "F": F, "R": R_ref, "R_over_P": round(R_ref / max(sc.P, 1), 3),
                       "binning": a.binning, "R_binned": Rr, "forward_accumulation": a.fwd_accum,
                       "views_per_step": world, "parallelism": f"frame-shard x{world}",
                       "rccl_ranks": world if backend == "nccl" else 0, "backend": backend,
                       "exchange": chosen[0] if dist is not None else None,
                       "exchange_requested": a.exchange if dist is not None else None,
                       "exchange_forced_single_rank": forced, "exchange_via": exchange_via,
                       "exchange_bytes_per_step": None if dist is None else
                       lanes.lanes[0][1].exchange_bytes(chosen[0], sparse_cap[0]),
                       "exchange_detail": exch_detail,
                       "frames_in_flight_per_gpu": len(lanes), "untimed_setup_steps": a.setup_steps,
                       "live_gradient_rows": L_rows,
                       "workload_stats": wstats,
                       "capacity_overflow": bool(overflow or row_overflow)},
            "roofline": roof,
            "frame_model": frame,
            "target": {"fps": 40.0, "met": fps / world >= 40.0},
        }
        if nonco is not None:
            out["non_coherent"] = nonco
        if iso is not None:
            iso_el, iso_avg, iso_lat = iso
            out["isolated"] = {"frames_in_flight_per_gpu": 1, "steps": a.isolated_steps,
                               "value": round(world * a.isolated_steps / iso_el, 3), "unit": "frames/s",
                               "ms_per_frame": round(1e3 * iso_el / a.isolated_steps, 4),
                               "latency_ms": iso_lat,
                               "stage_ms": {k: round(v, 4) for k, v in prof[1].items()},
                               "stage_ms_note": "from a separate profiled leg (HIP events between the stages, ~6 us each)"}
        if world == 1 and a.isolated_steps > 0:
            out["dropin"] = dropin_leg(sc, dev, max(3 * a.isolated_steps, 30), 10)
        if world == 1 and a.isolated_steps > 0 and not a.no_extra_legs:
            dims = (P, W, H, F, M)
            out["bracket"] = bracket_legs(sc, g_dev, c0, (dc, dl, dd), dev, max(10, a.isolated_steps), dims)
            if a.config == 3 and room is None:
                out["config4_substitute"] = config4_substitute(sc, g_dev, dev, dims)
                out["config4_substitute"]["room_scene"] = room_scene_leg(dev, dims, a.isolated_steps)
        # the driver keeps `config` and `roofline` of this line: the rates a caller sees, beside the headline's
        out["config"]["isolated_fps"] = out["isolated"]["value"] if iso is not None else None
        out["config"]["dropin_fps"] = out["dropin"]["value"] if "dropin" in out else None
        out["config"]["non_coherent_fps"] = nonco["value"] if nonco is not None else None
        out["config"]["frame_model_frac"] = frame["frac_of_8TBs_wall"]
        if roof is not None:
            roof["frame_model"] = {"frac": frame["frac_of_8TBs_wall"], "algorithmic_bytes": frame["algorithmic_bytes"]}
        rs_ = out.get("config4_substitute", {}).get("room_scene")
        if rs_:
            out["config"]["room_scene"] = {"isolated_fps": rs_["isolated"]["value"], "four_in_flight_fps": rs_["four_in_flight"]["value"],
                                           "four_in_flight_cycling_12_views_fps": rs_["four_in_flight_cycling_12_views"]["value"],
                                           "tracking_ms": rs_["tracking"]["ms_per_iteration"],
                                           "mapping_ms": {k: v["ms_per_iteration"] for k, v in rs_.get("mapping", {}).items()
                                                          if isinstance(v, dict) and "ms_per_iteration" in v},
                                           "stage_ms": rs_.get("stage_ms")}
        if world == 1 and not a.no_cpu_baseline:
            lane0 = iso_lane
            one_step(lane0)
            torch.cuda.synchronize(dev)
            sl = lane0[1].layout.slices()
            out["cpu_baseline"] = cpu_baseline(sc, a.config, gpu=(lane0[0].out, lane0[1].flat.cpu(), sl),
                                               single_thread=(a.config in (1, 3)))
        print(json.dumps(out), file=JSON_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
