"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol that
include/olsr.h declares, buffer-size functions behave, the Python API validates arguments with
the reference's exceptions, and the product never silently falls back to the CPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from online_lang_splatting_amd import build
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    from online_lang_splatting_amd import _lib
    return _lib.lib()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "olsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(olsr_[a-z0-9_]+)\s*\(", src)) - {"olsr_alloc_fn"})


def test_library_exports_every_declared_symbol(L):
    from online_lang_splatting_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 14
    assert sorted(_lib.EXPORTS) == declared
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_buffer_sizes(L):
    g1, g2 = L.olsr_geometry_bytes(1000, 15), L.olsr_geometry_bytes(500000, 15)
    assert 0 < g1 < g2
    assert g2 >= 500000 * (4 + 8 + 16 + 24 + 12 + 3 + 4 * 6)
    assert L.olsr_image_bytes(1200, 680, 15) >= 1200 * 680 * 8 + 80 * 46 * 8
    assert L.olsr_binning_bytes(1 << 20, 15) >= (1 << 20) * (5 * 4 + 1 + 4)  # keys x2, val, src, inst_gid, flags, rowbase
    # backward scratch: one row per live (instance, slot) pair, row stride = 10+F floats padded to 16
    s0, s15, s32 = (L.olsr_backward_scratch_bytes(1 << 20, F) for F in (0, 15, 32))
    assert s15 - s0 == (1 << 20) * 4 * (32 - 16) and s32 - s15 == (1 << 20) * 4 * (48 - 32)
    assert L.olsr_geometry_bytes(0, 0) > 0 and L.olsr_binning_bytes(0, 15) > 0 and L.olsr_backward_scratch_bytes(0, 15) > 0
    assert L.olsr_version().startswith(b"olsr")


def test_struct_layout_matches_header():
    from online_lang_splatting_amd._abi import OlsrScene
    # 10 int32 + 4 float + 13 pointers + 2 int32 + 1 pointer (tile_depth_cut, round 4) + 1 int64 (backward_row_capacity,
    # round 5), no implicit padding
    assert ctypes.sizeof(OlsrScene) == 10 * 4 + 4 * 4 + 13 * 8 + 2 * 4 + 8 + 8 + 8
    assert OlsrScene.background.offset == 56 and OlsrScene.cam_pos.offset == 56 + 12 * 8
    assert OlsrScene.tile_depth_cut.offset == 56 + 13 * 8 + 8
    assert OlsrScene.depth_order_carry.offset == 56 + 13 * 8 + 8 + 8 + 8
    from online_lang_splatting_amd._abi import OlsrGradBucket
    assert ctypes.sizeof(OlsrGradBucket) == 3 * 8 + 2 * 4 + 8 and OlsrGradBucket.row_mask.offset == 32


def test_c_abi_argument_errors(L):
    from online_lang_splatting_amd import _abi
    s = _abi.OlsrScene()
    s.P, s.width, s.height, s.tile, s.F = 10, 64, 64, 17, 15
    cb = _abi.ALLOC_FN(lambda u, n: None)
    R = ctypes.c_int32(0)
    rc = L.olsr_forward(ctypes.byref(s), cb, None, cb, None, cb, None, None, None, None, None, None, None,
                        ctypes.byref(R), None)
    assert rc == _abi.OLSR_ERR_ARG and b"tile" in L.olsr_last_error()
    s.tile, s.F = 15, 7
    rc = L.olsr_forward(ctypes.byref(s), cb, None, cb, None, cb, None, None, None, None, None, None, None,
                        ctypes.byref(R), None)
    assert rc == _abi.OLSR_ERR_ARG and b"language channels" in L.olsr_last_error()


def test_python_api_surface_and_exceptions():
    import diff_gaussian_rasterization as D
    from online_lang_splatting_amd import GaussianRasterizationSettings, LanguageGaussianRasterizer
    assert D.GaussianRasterizationSettings is GaussianRasterizationSettings
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "projmatrix_raw", "sh_degree", "campos", "prefiltered", "debug")
    rs = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    P = 4
    m, m2, o = torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1)
    sh, col = torch.zeros(P, 1, 3), torch.zeros(P, 3)
    sc, rot, cov = torch.ones(P, 3), torch.zeros(P, 4), torch.zeros(P, 6)
    lang = torch.zeros(P, 15)
    for cls in (D.GaussianRasterizer, LanguageGaussianRasterizer):
        r = cls(rs)
        extra = dict(language_precomp=lang) if cls is LanguageGaussianRasterizer else {}
        with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
            r(m, m2, o, shs=None, colors_precomp=None, scales=sc, rotations=rot, **extra)
        with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
            r(m, m2, o, shs=sh, colors_precomp=col, scales=sc, rotations=rot, **extra)
        with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
            r(m, m2, o, shs=sh, scales=sc, rotations=None, **extra)
        with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
            r(m, m2, o, shs=sh, scales=sc, rotations=rot, cov3D_precomp=cov, **extra)
        # CPU tensors are refused loudly: there is no CPU fallback in the product
        with pytest.raises(RuntimeError, match="GPU"):
            r(m, m2, o, shs=sh, scales=sc, rotations=rot, **extra)
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        from online_lang_splatting_amd import _C
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(P, 4), col, o, sc, rot, 1.0, torch.empty(0), torch.eye(4),
                               torch.eye(4), torch.eye(4), 1.0, 1.0, 8, 8, torch.empty(0), 0, torch.zeros(3), False,
                               False)


def test_debug_settings_leave_a_snapshot_behind_on_failure(tmp_path, monkeypatch):
    """raster_settings.debug = True: a failing forward of the RGB rasterizer writes its CPU-copied arguments to
    snapshot_fw.dump before re-raising (DGR/diff_gaussian_rasterization/__init__.py:121-130); the language rasterizer does not
    (the reference has those lines commented out, :270-281)."""
    import diff_gaussian_rasterization as D
    from online_lang_splatting_amd import GaussianRasterizationSettings, LanguageGaussianRasterizer
    monkeypatch.chdir(tmp_path)
    rs = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, True)
    P = 4
    m, m2, o = torch.full((P, 3), 2.0), torch.zeros(P, 3), torch.ones(P, 1)
    sh, sc, rot = torch.zeros(P, 1, 3), torch.ones(P, 3), torch.zeros(P, 4)
    with pytest.raises(RuntimeError, match="GPU"):   # (CPU tensors: the product has no CPU path, so the call fails)
        LanguageGaussianRasterizer(rs)(m, m2, o, shs=sh, scales=sc, rotations=rot, language_precomp=torch.zeros(P, 15))
    assert not (tmp_path / "snapshot_fw.dump").exists()
    with pytest.raises(RuntimeError, match="GPU"):
        D.GaussianRasterizer(rs)(m, m2, o, shs=sh, scales=sc, rotations=rot)
    dump = torch.load(tmp_path / "snapshot_fw.dump")
    assert isinstance(dump, tuple) and torch.equal(dump[1], m) and dump[-1] is True   # (bg, means3D, ..., prefiltered, debug)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "online_lang_splatting_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_C" not in txt and "liboracle" not in txt and "import oracle" not in txt, f
    shim = open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")).read()
    assert "oracle" not in shim


def test_scene_generator_is_deterministic():
    from online_lang_splatting_amd.scene import make_config_scene, make_scene
    a, b = make_scene(1000, 64, 48, 15, seed=3), make_scene(1000, 64, 48, 15, seed=3)
    for k in ("means3D", "opacities", "scales", "rotations", "shs", "language"):
        assert torch.equal(getattr(a, k), getattr(b, k))
    c = make_config_scene(1, P=500)
    assert c.shs.shape == (500, 16, 3) and c.F == 0 and c.sh_degree == 3
    assert torch.allclose(a.language.norm(dim=1), torch.ones(1000), atol=1e-5)
    assert torch.allclose(a.rotations.norm(dim=1), torch.ones(1000), atol=1e-5)


def test_compiled_torch_binding_loads_and_rejects_cpu_tensors():
    """The `_C` functions of DGR/ext.cpp:15-21 as a compiled torch extension (csrc/olsr_torch.cpp): it imports, holds
    the three entry points, reports the library's version, and — without a GPU — refuses a CPU tensor like the ctypes
    binding does."""
    from online_lang_splatting_amd import _C
    ext = _C.compiled_binding()
    assert ext is not None, getattr(_C.compiled_binding, "error", None)
    for name in ("forward", "backward", "mark_visible", "version"):
        assert callable(getattr(ext, name))
    assert ext.version() == _C.lib().olsr_version().decode()
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.forward(15, e, torch.zeros(4, 3), e, e, e, e, e, 1.0, e, e, e, e, 1.0, 1.0, 10, 10, e, 0, e, False, False,
                    15, 0, 1, 0)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        ext.forward(15, e, torch.zeros(4, 2), e, e, e, e, e, 1.0, e, e, e, e, 1.0, 1.0, 10, 10, e, 0, e, False, False,
                    15, 0, 1, 0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.mark_visible(torch.zeros(4, 3), e, e)


def test_binding_selector(monkeypatch):
    from online_lang_splatting_amd import _C
    monkeypatch.setenv("OLSR_BINDING", "ctypes")
    assert _C.compiled_binding() is None
    monkeypatch.setenv("OLSR_BINDING", "torch")
    assert _C.compiled_binding() is not None


def test_sort_plan_host_logic(L, monkeypatch):
    """Keys per thread / block count of the one-kernel radix passes (csrc/olsr_state.h: sort_plan): a round of at most
    256 resident blocks costs about (kpt + 11) us; status rows are reserved for the smallest chunk; 4096 blocks at
    most, beyond that the multi-kernel fallback."""
    L.olsr_debug_sort_knobs(0, 0, 0)
    kpt, nblk = ctypes.c_int32(), ctypes.c_int32()

    def plan(n, cap=0):
        ok = L.olsr_debug_sort_plan(n, cap, ctypes.byref(kpt), ctypes.byref(nblk))
        return ok, kpt.value, nblk.value
    # round 5: a scene with OLSR_FLAG_FRAMES_IN_FLIGHT sorts with four-wave (256-thread) blocks of sixteen keys per thread — they
    # get onto the CUs beside another frame's composite workgroups; olsr_debug_sort_threads forces a shape for every call
    assert L.olsr_debug_sort_threads(-1) == 0      # (default: the call decides; 1024 without the flag)
    assert L.olsr_debug_sort_threads(256) == 256
    for n in (1, 1023, 1024, 1025, 500_000, 2_700_146, 10_000_000, 60_000_000):
        ok, k, b = plan(n)
        assert ok == 1 and k == 16 and b == -(-n // (256 * 16)) and b <= 16384
    assert plan(16384 * 4096 + 1)[0] == 0          # more than 16 384 blocks even at kpt 16: the multi-kernel passes
    L.olsr_debug_sort_knobs(2, -1, -1)             # (kpt 2 does not exist for four-wave blocks: 4 runs)
    assert plan(500_000)[1] == 4
    L.olsr_debug_sort_knobs(0, -1, -1)
    assert L.olsr_debug_sort_threads(0) == 0       # the 1024-thread shape of rounds 2-4, below
    assert plan(500_000) == (1, 2, 245)            # the depth sort of the headline frame: one round at kpt 2
    assert plan(2_700_146) == (1, 12, 220)         # its tile sort: one round of 220 fat blocks
    ok, k, b = plan(3_440_000, 1)                  # the same sort launched against a capacity: planned for 85 % fill
    assert (ok, k) == (1, 12) and b == -(-3_440_000 // (1024 * 12))
    assert plan(3_440_000, 0)[1] == 16             # ... while an exact count of that size takes one round at kpt 16
    assert plan(0)[2] == 0
    for n in (1, 2047, 2048, 2049, 10_000_000, 60_000_000):
        ok, k, b = plan(n)
        assert ok == 1 and k in (2, 4, 8, 12, 16) and b == -(-n // (1024 * k)) and b <= 16384
    assert plan(4096 * 16384 + 1)[0] == 0          # beyond what the status rows are reserved for: the multi-kernel passes
    # the tuning knobs: set through the library (the environment is read once, at load), a forced value that would
    # overrun the status rows is ignored, a negative argument leaves a knob alone
    try:
        L.olsr_debug_sort_knobs(4, -1, -1)
        assert plan(500_000)[1:] == (4, 123)
        assert plan(40_000_000)[2] <= 16384
        L.olsr_debug_sort_knobs(-1, 64, -1)
        assert plan(500_000)[1:] == (4, 123)
        L.olsr_debug_sort_knobs(0, -1, -1)
        assert plan(500_000)[1] != 2  # 64 resident blocks per round: fatter blocks win
    finally:
        L.olsr_debug_sort_knobs(0, 0, 0)
    assert plan(500_000) == (1, 2, 245)
    assert L.olsr_debug_sort_threads(-1) == 0


def test_backward_row_policy_without_a_posted_count(L):
    """Host logic of the drop-in backward's scratch sizing (no GPU involved): with nothing posted for a token the bound of
    two rows per instance (packed survivor waves) or four (64-pixel slots) stands in, and waiting for a token that no
    forward ever issued times out."""
    import time
    assert L.olsr_backward_rows(0, 1, 1000, 15) == 2000
    assert L.olsr_backward_rows(0, 0, 1000, 15) == 4000
    assert L.olsr_backward_rows(0, 1, 0, 15) == 0 and L.olsr_backward_rows(0, 1, -5, 15) == 0
    assert L.olsr_live_rows(123, 1) == -1
    t0 = time.perf_counter()
    assert L.olsr_live_rows_wait(123, 1, 2000) == -1
    assert time.perf_counter() - t0 < 0.5
    assert L.olsr_last_forward_token() == 0  # (this thread issued no forward)


def test_usable_cpus_respects_affinity_and_quota(monkeypatch, tmp_path):
    """The oracle's thread pool (and bench.py's cpu_baseline `cores`) is sized by the CPUs the container may use — the
    GPU box shows 256 hardware threads and grants 16 — not by os.cpu_count()."""
    import builtins
    import os
    from oracle import oracle_C
    n = oracle_C.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("300000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert oracle_C.usable_cpus() == 3


def test_sparse_exchange_entries_validate_their_arguments(L):
    """The three local steps of the capacity-bound sparse exchange (include/olsr.h): argument errors come back before any
    launch, P == 0 is a no-op, and the scratch size is one int per 1024 rows."""
    from online_lang_splatting_amd import _abi
    assert L.olsr_sparse_exchange_scratch_ints(0) == 0 and L.olsr_sparse_exchange_scratch_ints(1) == 1
    assert L.olsr_sparse_exchange_scratch_ints(1024) == 1 and L.olsr_sparse_exchange_scratch_ints(500000) == 489
    assert L.olsr_sparse_exchange_mask(0, 29, None, None, None, None, None) == _abi.OLSR_OK
    assert L.olsr_sparse_exchange_mask(10, 0, None, None, None, None, None) == _abi.OLSR_ERR_ARG
    assert L.olsr_sparse_exchange_mask(10, 29, None, None, None, None, None) == _abi.OLSR_ERR_ARG
    assert b"must not be NULL" in L.olsr_last_error()
    assert L.olsr_sparse_exchange_pack(10, 29, 0, *([None] * 10)) == _abi.OLSR_ERR_ARG
    assert L.olsr_sparse_exchange_pack(10, 29, 4, *([None] * 10)) == _abi.OLSR_ERR_ARG
    assert L.olsr_sparse_exchange_pack(0, 29, 4, *([None] * 10)) == _abi.OLSR_OK
    assert L.olsr_sparse_exchange_unpack(10, 29, 4, None, None, None, None, None) == _abi.OLSR_ERR_ARG
    assert L.olsr_sparse_exchange_unpack(-1, 29, 4, None, None, None, None, None) == _abi.OLSR_ERR_ARG


def test_weak_scaling_poses_are_side_by_side_and_centred():
    """shard_cameras: what rank r of N renders in bench.py's weak-scaling mode - N poses 1.5 cm apart, no rotation, centred on
    the identity pose (N = 1: the identity pose itself), so that every rank's view costs what the N = 1 view costs."""
    import torch
    from online_lang_splatting_amd.scene import default_camera, shard_cameras
    ident = default_camera(1200, 680)
    one = shard_cameras(1200, 680, n=1)
    assert len(one) == 1 and torch.equal(one[0].world_view_transform, ident.world_view_transform)
    for n in (2, 4, 8):
        cams = shard_cameras(1200, 680, n=n)
        tx = torch.tensor([float(c.T[0]) for c in cams])
        assert len(cams) == n and all(torch.equal(c.R, ident.R) for c in cams)
        assert abs(float(tx.sum())) < 1e-6 and torch.allclose(tx[1:] - tx[:-1], torch.full((n - 1,), 0.015), atol=1e-6)
        assert float(tx.abs().max()) <= 0.015 * 3.5 + 1e-6
