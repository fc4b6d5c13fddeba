"""bench.py's contract line, single process and — to exercise the N > 1 code path on a one-GPU box — two ranks
sharing the GPU over gloo (OLSR_BENCH_BACKEND=gloo; on a multi-GPU node the driver launches it over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(out):
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(out[-2000:])


def test_single_process_line():
    p = subprocess.run([sys.executable, "bench.py", "--config", "1", "--steps", "6", "--warmup", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "weak"
    roof = d["roofline"]
    assert roof["bound"] == "valu" and roof["unit"] == "GB/s" and 0 < roof["frac"] < 1
    assert roof["units"]["instances_processed"] <= roof["model_reference_R"]["instances"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert cpu["checked_against_gpu"]["ok"], cpu["checked_against_gpu"]   # the CPU timed the frame the GPU rendered
    assert cpu["single_thread"]["cores"] == 1 and cpu["single_thread"]["value"] > 0
    assert d["isolated"]["frames_in_flight_per_gpu"] == 1
    q = d["isolated"]["latency_ms"]["step_completion_interval_ms"]
    assert q["p10"] <= q["median"] <= q["p90"] and q["n"] >= 1
    assert d["isolated"]["stage_ms"]["render_backward"] > 0
    # the legs that bracket the reference-mode headline: exact backward, 16x16 tiles, the reference's tile lists
    for name, tile, mode, binning in (("exact_mode", 15, "exact", "ellipse"), ("tile16", 16, "reference", "ellipse"),
                                      ("rect_binning", 15, "reference", "rect")):
        leg = d["bracket"][name]
        assert (leg["tile"], leg["backward_mode"], leg["binning"]) == (tile, mode, binning)
        assert leg["value"] > 0 and not leg["capacity_overflow"] and leg["R_binned"] > 0
    assert d["bracket"]["rect_binning"]["R_binned"] == d["config"]["R"] >= d["config"]["R_binned"]
    assert "config4_substitute" not in d  # (config 3 only)


def test_config3_line_carries_the_config4_substitute():
    """The driver's line (config 3) also reports the measurable substitute of BASELINE configs[3]: the dependent tracking
    iteration and the 12-view mapping iteration, plus the valu roofline priced against the guide's issue rate."""
    p = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "3", "--isolated-steps", "10",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    c4 = d["config4_substitute"]
    assert 0 < c4["tracking_iteration_ms"] < 5 and 0 < c4["mapping_iteration_ms"] < 60
    trk = c4["tracking"]["ms_per_iteration"]
    assert set(trk) == {"no_language_cotangent", "no_language_cotangent_with_convergence_readback",
                        "zero_language_cotangent", "zero_language_cotangent_with_convergence_readback",
                        "rgb_rasterizer_render", "no_language_cotangent_two_kernel_loss", "depth_cut_offs"}
    cut = c4["tracking_depth_cut"]   # the same loop on a workspace with per-tile depth cut-offs: its own leg, not the headline
    assert cut["instances_last_frame"] < 0.5 * cut["instances_without_cut"]
    assert cut["iterations_that_counted"] >= 0.8 * cut["iterations"]
    chain = lambda st: st["depth_sort"] + st["emit"] + st["tile_sort"]
    assert chain(cut["library_stage_ms"]) < chain(c4["tracking"]["library_stage_ms"])
    assert trk["no_language_cotangent"] <= 1.02 * trk["no_language_cotangent_two_kernel_loss"]  # the fused epilogue pays
    assert c4["tracking"]["pose_error_after"] < c4["tracking"]["pose_error_start"]  # it moved towards the target pose
    assert not c4["mapping"]["capacity_overflow"] and c4["mapping"]["views"] == 12
    assert c4["mapping"]["loss_last_view_final_iteration"] < c4["mapping"]["loss_last_view_first_iteration"]
    two = c4["mapping"]["two_kernel_loss"]   # the round-3 formulation beside the fused one: same losses, its own breakdown
    assert abs(two["loss_last_view_final_iteration"] - c4["mapping"]["loss_last_view_final_iteration"]) < 1e-5
    prof, prof2 = c4["mapping"]["profiled_iteration"], two["profiled_iteration"]
    assert prof["adam_ms"] > 0 and prof["lane_sum_ms"] > 0 and "loss" not in prof["stage_ms_per_view"]
    assert prof2["stage_ms_per_view"]["loss"] > 0 and prof["stage_ms_per_view"]["render_backward"] > 0
    assert set(d["bracket"]) == {"exact_mode", "tile16", "rect_binning", "fwd_accum_weight"}
    assert d["bracket"]["fwd_accum_weight"]["forward_accumulation"] == "weight"
    runs = d["config"]["value_runs"]   # the K-step region repeated until the timed regions total 2 s; the value is the median run
    assert len(runs["fps"]) == runs["runs"] >= 5 and runs["timed_seconds_total"] >= 2.0
    assert runs["min"] <= runs["p10"] <= runs["median"] <= runs["p90"] <= runs["max"] and abs(runs["median"] - d["value"]) < 0.01
    assert d["frames_in_flight"] == 4 and d["same_view_every_step"] is True and d["config"]["scene"] == "volume"
    ws_ = d["config"]["workload_stats"]   # the i.i.d. volume: saturation ends the lists early, few Gaussians blend
    assert ws_["list_fraction_read_before_saturation"] < 0.5 and ws_["blended_of_visible"] < 0.2
    room = c4["room_scene"]               # the surface-structured map: nothing saturates, nearly every visible Gaussian is live
    rw = room["workload"]
    assert rw["list_fraction_read_before_saturation"] > 0.8 and rw["live_rows_of_visible"] > 0.9 and not rw["capacity_overflow"]
    assert room["isolated"]["value"] > 0 and room["four_in_flight"]["value"] > room["isolated"]["value"]
    assert room["exchange"]["chosen"] in ("sparse", "reduce_scatter") and room["exchange"]["union_rows_over_12_views"] >= rw["live_gradient_rows_gaussians"]
    assert room["tracking"]["pose_error_after"] < room["tracking"]["pose_error_start"]
    for leg_ in (room["mapping"]["auto_loss"], c4["mapping"]["auto_loss"]):   # MappingStep's default measured both forms itself
        cal_ = leg_["calibration"]
        assert cal_["chosen"] in ("fused", "two_kernel") and len(cal_["fused_ms"]) >= 2 and len(cal_["two_kernel_ms"]) >= 2
        assert (min(cal_["fused_ms"]) <= min(cal_["two_kernel_ms"])) == (cal_["chosen"] == "fused")
    for k_ in ("fused_loss", "two_kernel_loss"):
        m_ = room["mapping"][k_]
        assert not m_["capacity_overflow"] and m_["loss_last_view_final_iteration"] < m_["loss_last_view_first_iteration"]
    assert "untimed set-up frames" in d["config"]["workload"]
    assert d["dropin"]["alternating_4_views"]["value"] > 0.8 * d["dropin"]["value"]   # per-view orders inside the library
    nc = d["non_coherent"]   # the camera changes every step: no reusable tile-order hint
    assert 0 < nc["value"] and 0 < nc["isolated_value"] and not nc["overflow"]
    valu = d["roofline"].get("valu")
    if valu is not None:  # (present when the committed PMC summary covers the kernel)
        assert abs(valu["peak"] - 1228.8) < 0.1 and 0 < valu["frac"] < 1


def test_two_ranks_frame_sharded_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OLSR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--config", "1",
                        "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["views_per_step"] == 2 and d["value"] > 0
    assert "cpu_baseline" not in d


def test_mapping_iteration_mode_two_ranks_over_gloo():
    """bench.py --views: V views sharded over the ranks, one exchange + fused Adam per step, in all three exchange modes."""
    for exchange in ("all_reduce", "reduce_scatter", "sparse"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, OLSR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--config",
                            "1", "--views", "5", "--exchange", exchange, "--steps", "3", "--warmup", "1"], cwd=ROOT,
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (exchange, p.stdout[-1500:], p.stderr[-1500:])
        d = _last_json(p.stdout)
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["views_per_step"] == 5 and d["value"] > 0
        assert d["config"]["views_of_rank0"] == 3 and d["config"]["exchange"] == exchange


def test_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (the flag used to be
    ignored and the line said n_gpus 1).  Over RCCL that needs two GPUs — on a smaller node it must FAIL, loudly, not
    report one GPU; OLSR_BENCH_BACKEND=gloo is the functional fallback in which the ranks share the device."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OLSR_BENCH_BACKEND")}
    args = [sys.executable, "bench.py", "--gpus", "2", "--config", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run(args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
        d = _last_json(p.stdout)
        assert d["n_gpus"] == 2 and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 2
    else:
        assert p.returncode != 0 and "needs 2 GPUs" in p.stderr and "{" not in p.stdout
    for exchange in ("sparse", "all_reduce", "reduce_scatter"):
        p = subprocess.run(args + ["--exchange", exchange], cwd=ROOT, env=dict(env, OLSR_BENCH_BACKEND="gloo"),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (exchange, p.stdout[-1500:], p.stderr[-1500:])
        d = _last_json(p.stdout)
        c = d["config"]
        assert d["n_gpus"] == 2 and c["views_per_step"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
        assert c["backend"] == "gloo" and c["rccl_ranks"] == 0 and c["exchange"] == exchange
        assert c["exchange_bytes_per_step"] > 0 and c["exchange_detail"]["rows_nonzero_per_view_max_over_ranks"] > 0
        chk = c["exchange_detail"]["check"]  # the exchange against the dense all-reduce of the same partial sums, where it ran
        assert chk["equals_dense"] and chk["max_abs_diff_over_max_abs"] == 0.0 and chk["radii_equal"]
        assert chk["identical_on_every_rank"] and chk["nonzero_gradient_rows_after"] > 0
        if exchange == "sparse":
            det = c["exchange_detail"]
            assert not det["overflow"] and 0 < det["rows_in_union_last_step"] <= det["packed_capacity_rows"]
            sparse_bytes = c["exchange_bytes_per_step"]
        else:
            assert c["exchange_bytes_per_step"] == c["exchange_detail"]["bucket_bytes"]
    assert sparse_bytes < c["exchange_detail"]["bucket_bytes"]


def test_self_launched_single_rank_matches_the_plain_line():
    """--self-launch: the same command under torch.distributed.run with one rank (what the driver's N = 1 SCALE leg looks
    like): same workload, same keys, n_gpus 1, no exchange."""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--self-launch", "--config", "1", "--steps", "6", "--warmup",
                        "2", "--no-cpu-baseline", "--no-extra-legs"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    d = _last_json(p.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["exchange"] is None and d["config"]["rccl_ranks"] == 0 and d["config"]["backend"] is None


def test_forced_single_rank_exchange_over_rccl_and_a_clean_stdout():
    """OLSR_BENCH_FORCE_EXCHANGE=1: a group of ONE rank over RCCL with every collective of the exchange issued - what a one-GPU
    box can run of the multi-GPU step.  RCCL prints a version banner to the process's stdout when its first communicator is
    created; the line must stay the ONLY thing on stdout (the driver parses it), whatever native code prints."""
    for exchange in ("auto", "sparse", "all_reduce", "reduce_scatter"):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OLSR_BENCH_BACKEND")}
        p = subprocess.run([sys.executable, "bench.py", "--config", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                            "--no-extra-legs", "--isolated-steps", "0", "--repeats", "2", "--exchange", exchange], cwd=ROOT,
                           env=dict(env, OLSR_BENCH_FORCE_EXCHANGE="1"), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (exchange, p.stdout[-1500:], p.stderr[-1500:])
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        c = d["config"]
        assert d["n_gpus"] == 1 and c["exchange_requested"] == exchange and c["exchange_forced_single_rank"] and c["backend"] == "nccl"
        if exchange == "auto":   # chosen from the measured union of the gradient rows, and the line says how
            det = c["exchange_detail"]
            assert c["exchange"] == det["chosen"] == ("sparse" if det["sparse_pays"] else "reduce_scatter")
            assert 0 < det["rows_in_union_at_setup"] <= c["P"] and 0 < det["live_row_fraction_per_view"] <= 1
        else:
            assert c["exchange"] == exchange
        chk = c["exchange_detail"]["check"]
        assert chk["equals_dense"] and chk["radii_equal"] and chk["identical_on_every_rank"]
        assert chk["nonzero_gradient_rows_after"] > 0
