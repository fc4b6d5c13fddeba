"""Parity proper: the HIP library (through the C-ABI, via the reference-shaped `_C` surface)
against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): fp32 outputs within 1e-4 relative.  Measured: the forward path
(preprocess, sort order, images, counters) is BIT-IDENTICAL to the oracle because every decision
expression shares its fp32 operation order and the pinned exp; gradients differ only in the
order of the cross-pixel summation.  The tests assert exact equality for the forward and
integer state, and RTOL = 1e-4 (relative to the tensor's largest magnitude) for gradients.
"""
import os

import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import default_camera, make_scene
from parity_common import (ELEM_MIN_FRACTION, assert_elementwise, assert_ordered_equals_oracle, assert_rounding_only,
                           fwd_args, ordered_backward, rel_err, run_backend)

pytestmark = pytest.mark.gpu
RTOL = 1e-4
DEV = "cuda:0"


def _tile_pairs(hip, f, sc, tile):
    """(tile * P + Gaussian) per sorted list position, and the per-position blend flags."""
    W, H, F, R = sc.camera.width, sc.camera.height, sc.F, f["R"]
    hip.TILE = tile
    pl = hip.state_field("binning", f["binning"], "point_list", R=R, F=F, dtype=torch.int32, count=R).long()
    src = hip.state_field("binning", f["binning"], "src", R=R, F=F, dtype=torch.int32, count=R).long()
    fl = hip.state_field("binning", f["binning"], "flags", R=R, F=F, dtype=torch.uint8, count=R)
    nt = ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
    rg = hip.state_field("image", f["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2 * nt).view(-1, 2).long()
    lens = rg[:, 1] - rg[:, 0]
    assert int(lens.sum()) == R
    tile_of = torch.repeat_interleave(torch.arange(nt, device=pl.device), lens)
    # list positions of tile t are ranges[t]; walk them tile by tile
    starts = torch.repeat_interleave(rg[:, 0], lens)
    pos = starts + (torch.arange(R, device=pl.device) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens))
    return tile_of * max(sc.P, 1) + pl[pos], fl[src[pos]], pos


COMPOSITE_GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dlanguage", "dL_ddepths")


COMPOSITE_KEYS = COMPOSITE_GRADS


def _check(hip, oracle, sc, seed=0, tile=15, mode=0, grad_keys=None, elementwise=False, worst_bound=1e-2, log=None,
           chain=True, chain_worst_bound=1e-3, chain_min_fraction=ELEM_MIN_FRACTION, composite_worst_bound=2e-4,
           chain_exact=True, ordered=True, rounding_k_bound=64.0, **kw):
    """Oracle vs the HIP library in both binning modes.
    RECT: images, counters AND the instance lists equal the reference's bit for bit.
    ELLIPSE (the product's default): identical images / radii / n_touched / final_T, gradients to RTOL,
    and every tile list is the RECT list minus instances that blend nothing, in the same order.
    elementwise: additionally assert, per gradient tensor, the north-star criterion per ELEMENT (>= 99.99 % of the
    elements within 1e-4 relative + 1e-6 of the tensor's largest magnitude, and the worst element within
    `worst_bound`, both printed) instead of only the max-norm.
    chain (on by default, every scene of the suite): additionally replay the reference's per-Gaussian chain (computeCov2DCUDA + preprocessCUDA backward) in the
    oracle on the PRODUCT's composite-level gradients and hold the product's per-Gaussian outputs to it per element:
    same inputs on both sides, so what is compared is the chain's arithmetic and not its sensitivity to summation-order
    noise in dL_dconic / dL_dmean2D (which the reference's own float atomics have from run to run).  Criterion: the
    north-star one per element (two elements may leave the band in tensors too small for 99.99 % to allow any), worst
    element within chain_worst_bound = 1e-3 (the 2 800 random scenes of round 3's campaigns: every element of every
    tensor within 1e-4 except ONE element — 6.5e-5 in a tensor whose largest is 1.3e-2 — at 8.5e-4; the suite's own
    scenes: one element of 9 308 at 1.5e-4).
    chain_exact (late round 4, on by default): the product's chain is written in the association of the reference's source,
    like the oracle, and neither is built with contraction - on identical inputs every element of dL_dmeans3D, dL_dcov3D,
    dL_dsh, dL_dscales, dL_drotations and dL_dtau must EQUAL the oracle's replay (the sign of a zero aside).
    ordered (round 6, on by default): the composite backward in the reference's own association on the GPU
    (olsr_debug_backward_ordered) must EQUAL the oracle's composite-level gradients bit for bit on both binning modes' state,
    and the product's fast kernel must differ from it BY ROUNDING ONLY: every element within rounding_k_bound x 2^-24 x the
    element's condition (the same sums over magnitudes; measured K <= 13 on the suite's scenes, <= 30 at the full configs).
    With the chain exact on identical inputs, this closes the backward: what is not an equality is bounded per element by
    the arithmetic's own rounding."""
    fo, go = run_backend(oracle, sc, None, seed, tile, mode, **kw)
    assert grad_keys is None or all(k in go for k in grad_keys), [k for k in grad_keys if k not in go]
    fr, gr = run_backend(hip, sc, torch.device(DEV), seed, tile, mode, binning=_abi.BINNING_RECT, **kw)
    fg, gg = run_backend(hip, sc, torch.device(DEV), seed, tile, mode, binning=_abi.BINNING_ELLIPSE, **kw)
    torch.cuda.synchronize()
    P, F = sc.P, sc.F
    W, H = sc.camera.width, sc.camera.height
    assert fr["R"] == fo["R"]
    assert fg["R"] <= fo["R"]
    for f_, g_, name in ((fr, gr, "rect"), (fg, gg, "ellipse")):
        assert torch.equal(f_["radii"].cpu(), fo["radii"]), name
        assert torch.equal(f_["n_touched"].cpu(), fo["n_touched"]), name
        for k in ("color", "language", "depth", "opacity"):
            if fo[k] is not None and fo[k].numel():
                assert torch.equal(f_[k].cpu(), fo[k]), \
                    f"{name}: forward {k} not bit-identical (rel {rel_err(f_[k], fo[k])[0]:.2e})"
        if P:
            hip.TILE = tile
            ft = hip.state_field("image", f_["img"], "final_T", W=W, H=H, dtype=torch.float32, count=W * H)
            assert torch.equal(ft.cpu(), oracle.get_field(fo["geom"], "final_T")), name
            cnt = hip.state_field("geometry", f_["geom"], "counters", P=P, F=F, dtype=torch.int32, count=8).cpu()
            assert int(cnt[0]) == f_["R"] and int(cnt[3]) == fo["R"], name  # [3]: the reference's num_rendered
        for k in go:
            if go[k].numel() and (grad_keys is None or k in grad_keys):
                if elementwise:  # the north-star statement itself: per element, outliers bounded
                    # composite-level tensors (what the composite kernel produces) are held to a tighter worst element
                    # than the end-to-end ones behind the per-Gaussian chain
                    assert_elementwise(g_[k], go[k], f"{name}:{k}",
                                       composite_worst_bound if k in COMPOSITE_GRADS else worst_bound, log)
                else:
                    r, e = rel_err(g_[k], go[k])
                    assert r <= RTOL, f"{name}: {k}: rel {r:.2e} abs {e:.2e}"
                    # the max-norm says nothing about small elements: what the composite kernel produces is also held to
                    # the north-star criterion per ELEMENT on every scene of the suite
                    if k in COMPOSITE_GRADS:
                        assert_elementwise(g_[k], go[k], f"{name}:{k}", composite_worst_bound, log)
        if ordered and P and grad_keys is None:
            gord = ordered_backward(hip, sc, f_, seed, tile, mode, **kw)
            assert_ordered_equals_oracle(go, gord, where=f"{name}:ordered:")
            gcond = ordered_backward(hip, sc, f_, seed, tile, mode, condition=True, **kw)
            assert_rounding_only(g_, gord, gcond, k_bound=rounding_k_bound, log=log, name=f"{name}:")
            del gord, gcond
        if chain and P:
            gc = oracle.backward_chain(max(F, 0), {k: g_[k] for k in ("dL_dmeans2D", "dL_dconic", "dL_dcolors",
                                                                      "dL_ddepths")}, *fo["bwd_args"])
            for k in oracle.CHAIN_KEYS:
                if gc[k].numel():
                    assert_elementwise(g_[k], gc[k], f"{name}:chain:{k}", chain_worst_bound, log, allow_outliers=2,
                                       min_fraction=chain_min_fraction)
                    if chain_exact:
                        a_, b_ = g_[k].detach().cpu().float(), gc[k].detach().cpu().float()
                        same = (a_ == b_) | (a_.isnan() & b_.isnan())
                        assert bool(same.all()), (f"{name}:chain:{k}: {int((~same).sum())} of {a_.numel()} elements differ from "
                                                  f"the oracle's replay on identical inputs (largest difference "
                                                  f"{float((a_ - b_).abs().nan_to_num(0).max()):.3e})")
        if P and grad_keys is None:
            tau_sum = go["dL_dtau"].double().sum(0).float()
            # six sums over all P Gaussians (millions of cancelling terms at the full configs): held to 1e-4 of the
            # largest of the six — a per-element statement about six numbers would be a statement about summation noise
            r, _ = rel_err(g_["dL_dtau_sum"], tau_sum)
            assert r <= RTOL, f"{name}: dL_dtau_sum: rel {r:.2e}"
            if log is not None:
                log.append(dict(name=f"{name}:dL_dtau_sum", max_norm_rel=r, n=6))
    if fo["R"] > 0:
        pl = hip.state_field("binning", fr["binning"], "point_list", R=fr["R"], F=F, dtype=torch.int32, count=fr["R"])
        assert torch.equal(pl.cpu(), oracle.get_field(fo["geom"], "point_list"))
        nc = hip.state_field("image", fr["img"], "n_contrib", W=W, H=H, dtype=torch.int32, count=W * H)
        assert torch.equal(nc.cpu(), oracle.get_field(fo["geom"], "n_contrib"))
        # exact lists: an order-preserving sub-list of the reference's that keeps every blending instance
        kr, flr, _ = _tile_pairs(hip, fr, sc, tile)
        if fg["R"] > 0:
            ke, fle, _ = _tile_pairs(hip, fg, sc, tile)
            keep = torch.isin(kr, ke)
            assert int(keep.sum()) == ke.numel() and torch.equal(kr[keep], ke)
            assert torch.equal(flr[keep], fle)
        else:
            keep = torch.zeros_like(kr, dtype=torch.bool)
        assert int((flr[~keep] != 0).sum()) == 0
    oracle.release(fo["geom"])
    return fg, gg


@pytest.mark.parametrize("F", [0, 3, 15, 16, 32])
@pytest.mark.parametrize("mode", [_abi.BWD_REFERENCE, _abi.BWD_EXACT])
def test_language_channels_and_modes(hip, oracle, F, mode):
    _check(hip, oracle, make_scene(3000, 160, 120, F, seed=10 + F), seed=F, mode=mode)


@pytest.mark.parametrize("deg,max_deg", [(0, 0), (0, 3), (1, 1), (2, 3), (3, 3)])
def test_sh_degrees(hip, oracle, deg, max_deg):
    sc = make_scene(2500, 128, 128, 15, seed=20 + deg, max_sh_degree=max_deg, sh_degree=deg)
    _check(hip, oracle, sc, seed=deg)


@pytest.mark.parametrize("tile", [15, 16])
@pytest.mark.parametrize("mode", [_abi.BWD_REFERENCE, _abi.BWD_EXACT])
def test_tile_sizes_and_ragged_image(hip, oracle, tile, mode):
    # 157x101 is not a multiple of either tile size: partial tiles on both edges
    _check(hip, oracle, make_scene(4000, 157, 101, 15, seed=31), seed=1, tile=tile, mode=mode)


def test_background_scale_modifier_and_rotated_camera(hip, oracle):
    cam = default_camera(200, 150, yaw_deg=9.0, tx=0.2)
    sc = make_scene(5000, 200, 150, 15, seed=41, bg=torch.tensor([0.3, 0.6, 0.1]), camera=cam)
    _check(hip, oracle, sc, seed=2, scale_modifier=1.7)
    _check(hip, oracle, sc, seed=3, scale_modifier=0.4, mode=_abi.BWD_EXACT)


def test_precomputed_colors_and_covariance(hip, oracle):
    sc = make_scene(3000, 160, 120, 15, seed=51)
    g = torch.Generator().manual_seed(5)
    colors = torch.rand(sc.P, 3, generator=g)
    # cov3D from scale/rotation via the oracle's own preprocess state
    r = oracle.rasterize_language_gaussians(*fwd_args(sc))
    cov = oracle.get_field(r[4], "cov3D").view(sc.P, 6).clone()
    vis = r[3] > 0
    cov[~vis] = torch.tensor([1e-3, 0, 0, 1e-3, 0, 1e-3])
    oracle.release(r[4])
    _check(hip, oracle, sc, seed=4, colors_precomp=colors)
    _check(hip, oracle, sc, seed=5, cov3D_precomp=cov)
    _check(hip, oracle, sc, seed=6, colors_precomp=colors, cov3D_precomp=cov, mode=_abi.BWD_EXACT)


def test_empty_culled_and_single(hip, oracle):
    dev = torch.device(DEV)
    sc = make_scene(0, 64, 48, 15, seed=0)
    sc.language = torch.zeros(0, 15)
    a = fwd_args(sc, dev)
    a[3] = torch.zeros(0, 15, device=dev)
    r = hip.rasterize_language_gaussians(*a)
    assert r[0] == 0 and float(r[1].abs().max()) == 0 and float(r[2].abs().max()) == 0
    assert float(r[8].abs().max()) == 0 and r[3].numel() == 0
    # everything behind the near plane (z <= 0.2)
    sc = make_scene(300, 64, 48, 15, seed=1)
    sc.means3D[:, 2] = 0.2
    fg, gg = _check(hip, oracle, sc, seed=1)
    assert fg["R"] == 0 and float(gg["dL_dmeans3D"].abs().max()) == 0
    # one Gaussian, and a background
    sc = make_scene(1, 64, 48, 15, seed=2, bg=torch.tensor([0.2, 0.4, 0.6]))
    sc.means3D[:] = torch.tensor([0.05, -0.02, 1.5])
    sc.scales[:] = 0.08
    sc.opacities[:] = 0.9
    _check(hip, oracle, sc, seed=2)
    _check(hip, oracle, sc, seed=2, mode=_abi.BWD_EXACT)


def test_huge_and_tiny_gaussians(hip, oracle):
    """Splats covering the whole image (hundreds of tiles each) and sub-pixel splats."""
    sc = make_scene(600, 300, 200, 15, seed=61, scale_mult=12.0)
    _check(hip, oracle, sc, seed=7)
    sc = make_scene(6000, 120, 90, 15, seed=62, scale_mult=0.05)
    _check(hip, oracle, sc, seed=8)


def test_needles_and_faint_splats_exact_binning(hip, oracle):
    """Stress of the exact tile lists: 1000:1 needles at every orientation (the alpha-floor ellipse is a
    sliver of the reference's bounding square) and opacities around the 1/255 floor.  _check asserts that
    no instance that blends a pixel under the reference binning is missing from the exact lists.
    For the needles only the composite's own gradients are compared: behind them the backward of the 2D
    covariance inverse (CR/backward.cu:213-236) cancels terms of order 1e8 down to order 1e3, which turns
    the 1e-7 summation-order noise of ANY implementation (the reference's float atomics included) into
    errors of tens of percent in dL_dmeans3D / dL_dscales — there is no reproducible value to match.
    What CAN be matched is the chain on identical inputs (chain=True): even at 1000:1 two fp32 evaluation orders of it
    agree in 99.9 % of the elements to 1e-4, the worst one to 1.4e-3 (bounds below: 99.5 % / 2e-2)."""
    needle = dict(grad_keys=COMPOSITE_GRADS, chain_worst_bound=2e-2, chain_min_fraction=0.995)
    sc = make_scene(4000, 240, 165, 15, seed=64)
    sc.scales[:, 0] *= 25.0
    sc.scales[:, 1] *= 0.04
    _check(hip, oracle, sc, seed=10, **needle)
    sc = make_scene(4000, 240, 165, 15, seed=65, scale_mult=3.0)
    g = torch.Generator().manual_seed(65)
    sc.opacities[:] = (torch.rand(sc.P, 1, generator=g) * 0.02).reshape(sc.opacities.shape)  # 0 .. 5/255
    _check(hip, oracle, sc, seed=11, mode=_abi.BWD_EXACT)
    cam = default_camera(240, 165, yaw_deg=-14.0, tx=-0.3)
    sc = make_scene(3000, 240, 165, 15, seed=66, camera=cam, scale_mult=6.0)
    sc.scales[:, 2] *= 0.02
    _check(hip, oracle, sc, seed=12, tile=16, **needle)


@pytest.mark.parametrize("seed", range(int(os.environ.get("OLSR_STRESS_SEEDS", "256"))))
def test_exact_binning_never_drops_a_blending_instance(hip, seed):
    """Randomised stress of the conservative interval arithmetic behind the exact tile lists (no oracle needed):
    random anisotropy up to 3000:1, scale, opacity around the alpha floor, off-screen and near-plane splats,
    rotated cameras, both tile sizes.  The exact lists must give bit-identical images / n_touched, and every
    instance that blends a pixel under the reference binning must still be listed."""
    g = torch.Generator().manual_seed(1000 + seed)
    W, H = (211, 143) if seed % 2 else (160, 120)
    tile = 16 if seed % 3 == 0 else 15
    cam = default_camera(W, H, yaw_deg=float(torch.rand(1, generator=g) * 40 - 20), tx=float(torch.rand(1, generator=g) - 0.5))
    sc = make_scene(3000, W, H, 15, seed=2000 + seed, camera=cam, scale_mult=float(10 ** (torch.rand(1, generator=g) * 2 - 1)))
    sc.scales *= torch.exp(torch.randn(sc.P, 3, generator=g) * 2.0)            # anisotropy
    sc.opacities[:] = torch.sigmoid(torch.randn(sc.P, 1, generator=g) * 3 - 2).reshape(sc.opacities.shape)
    sc.opacities[::7] = (torch.rand(len(sc.opacities[::7]), generator=g) * 0.012).reshape(sc.opacities[::7].shape)  # ~ 1/255
    dev = torch.device(DEV)
    fr, gr = run_backend(hip, sc, dev, seed, tile, _abi.BWD_EXACT, binning=_abi.BINNING_RECT)
    fe, ge = run_backend(hip, sc, dev, seed, tile, _abi.BWD_EXACT, binning=_abi.BINNING_ELLIPSE)
    assert fe["R"] <= fr["R"]
    for k in ("color", "language", "depth", "opacity", "radii", "n_touched"):
        assert torch.equal(fe[k], fr[k]), k
    if fr["R"] > 0:
        kr, flr, _ = _tile_pairs(hip, fr, sc, tile)
        if fe["R"] > 0:
            ke, fle, _ = _tile_pairs(hip, fe, sc, tile)
            keep = torch.isin(kr, ke)
            assert int(keep.sum()) == ke.numel() and torch.equal(kr[keep], ke)
            assert torch.equal(flr[keep] & 15, fle & 15)
        else:
            keep = torch.zeros_like(kr, dtype=torch.bool)
        assert int(((flr[~keep] & 15) != 0).sum()) == 0
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dlanguage"):
        r, _ = rel_err(ge[k], gr[k])
        assert r <= 1e-5, (k, r)
    hip.TILE, hip.BWD_MODE, hip.BINNING = 15, 0, 1


@pytest.mark.parametrize("seed", range(int(os.environ.get("OLSR_PARITY_SEEDS", "10"))))
def test_random_scenes_against_the_oracle(hip, oracle, seed):
    """Randomised full parity (both binning modes, lists, images, gradients) over F, tile, backward mode, SH
    degree, camera pose, background, scale and opacity distributions."""
    g = torch.Generator().manual_seed(5000 + seed)
    r = lambda: float(torch.rand(1, generator=g))
    F = [0, 3, 15, 16, 32][seed % 5]
    tile = 15 if seed % 3 else 16
    mode = _abi.BWD_REFERENCE if seed % 2 else _abi.BWD_EXACT
    deg = seed % 4
    W, H = 96 + int(r() * 120), 64 + int(r() * 90)
    cam = default_camera(W, H, yaw_deg=r() * 30 - 15, tx=r() * 0.6 - 0.3)
    sc = make_scene(1500 + int(r() * 2500), W, H, F, seed=6000 + seed, camera=cam, max_sh_degree=deg, sh_degree=deg,
                    bg=torch.tensor([r(), r(), r()]) if seed % 2 else None, scale_mult=10 ** (r() * 1.4 - 0.7))
    sc.scales *= torch.exp(torch.randn(sc.P, 3, generator=g) * 0.7)
    sc.opacities[:] = torch.sigmoid(torch.randn(sc.P, 1, generator=g) * 2.5 - 0.5).reshape(sc.opacities.shape)
    # pose / 3-D gradients of strongly anisotropic splats are ill-conditioned (see the needle test): compare the
    # composite-level gradients for every scene and the full chain where the scene is tame
    _check(hip, oracle, sc, seed=seed, tile=tile, mode=mode, scale_modifier=0.6 + r(), grad_keys=COMPOSITE_GRADS)


def test_transparent_scene_walks_whole_lists(hip, oracle):
    """Low opacities: no pixel saturates, every tile list is walked to its end in both passes."""
    sc = make_scene(4000, 150, 105, 15, seed=71)
    sc.opacities.mul_(0.03)
    _check(hip, oracle, sc, seed=9)
    _check(hip, oracle, sc, seed=9, mode=_abi.BWD_EXACT)


def test_multi_kernel_sort_fallback(hip, oracle):
    """Sorts too large for the one-kernel-per-pass radix passes (> 33 M keys) fall back to histogram -> device-wide
    scan -> scatter launches; olsr_debug_sort_knobs(legacy=1) forces that path at any size so it keeps meeting the
    oracle."""
    from online_lang_splatting_amd._lib import lib
    try:
        lib().olsr_debug_sort_knobs(-1, -1, 1)
        _check(hip, oracle, make_scene(6000, 200, 150, 15, seed=73), seed=4)
        _check(hip, oracle, make_scene(3000, 157, 101, 0, seed=74), seed=5, tile=16, mode=_abi.BWD_EXACT)
    finally:
        lib().olsr_debug_sort_knobs(-1, -1, 0)
    _check(hip, oracle, make_scene(6000, 200, 150, 15, seed=73), seed=4)


def test_emission_of_screen_filling_splats(hip, oracle):
    """The emission expands Gaussians -> tile rows -> tiles through tables in LDS (k_binning.hip:
    emit_balanced_kernel).  A few hundred splats that each cover most of a 640x480 frame give a batch of 256 depth ranks
    more tile rows than one fill of the row table holds (16 x 256), rows of 40 tiles that a wave writes together, and
    Gaussians that straddle many 1024-instance output windows."""
    sc = make_scene(700, 640, 480, 3, seed=91, scale_mult=25.0)
    _check(hip, oracle, sc, seed=9)
    _check(hip, oracle, make_scene(300, 333, 257, 0, seed=92, scale_mult=40.0), seed=10, tile=16)


def test_rect_upper_bound_in_the_references_operation_order(hip, oracle):
    """getRect's upper bound is `p.x + max_radius + BLOCK_X - 1` — three float operations (CR/auxiliary.h:52-55).  Folding
    `BLOCK_X - 1` into one constant changes the last bit when the sum crosses 128, and with it a whole tile column:
    scene 5323 of scripts/oracle_stress.py (a Gaussian with mean x = 58.999992, radius 54, 16-pixel tiles: 64 tiles in
    the reference, 56 with the folded constant)."""
    from online_lang_splatting_amd.scene import default_camera
    g = torch.Generator().manual_seed(77_000 + 5000 + 323)
    r = lambda: float(torch.rand(1, generator=g))  # noqa: E731
    P = int(300 + r() * 9000)
    W, H = int(64 + r() * 400), int(48 + r() * 300)
    tile = 16 if r() < 0.4 else 15
    F = (0, 3, 15, 16, 32)[int(r() * 5) % 5]
    deg = int(r() * 4) % 4
    cam = default_camera(W, H, yaw_deg=r() * 50 - 25, tx=r() - 0.5)
    sc = make_scene(P, W, H, F, seed=900_000 + 5000 + 323, camera=cam, scale_mult=10 ** (r() * 2.2 - 1.2), max_sh_degree=deg)
    assert (P, W, H, tile, F) == (6312, 463, 250, 16, 15)
    fo, _ = run_backend(oracle, sc, None, 1, tile, 0)
    fr, _ = run_backend(hip, sc, torch.device(DEV), 1, tile, 0, binning=_abi.BINNING_RECT)
    assert fr["R"] == fo["R"] == 314790
    tt = hip.state_field("geometry", fr["geom"], "tiles_touched", P=sc.P, F=F, dtype=torch.int32, count=sc.P).cpu()
    assert int(tt[4248]) == 64 and torch.equal(tt, oracle.get_field(fo["geom"], "tiles_touched"))
    oracle.release(fo["geom"])


@pytest.mark.parametrize("kpt", [2, 4, 8, 12, 16])
def test_every_sort_pass_instantiation(hip, oracle, kpt):
    """The radix passes pick their keys-per-thread from the input size (olsr_state.h: sort_plan);
    olsr_debug_sort_knobs pins it, so that every instantiation sorts the same frame — at kpt = 2 its 0.6 M instances
    need more than 256 blocks, which also exercises the ticket order of a pass that is not resident at once."""
    from online_lang_splatting_amd._lib import lib
    try:
        lib().olsr_debug_sort_knobs(kpt, -1, -1)
        sc = make_scene(60000, 640, 480, 15, seed=76)
        _check(hip, oracle, sc, seed=6)
        if kpt in (2, 16):
            _check(hip, oracle, make_scene(2500, 160, 120, 0, seed=77), seed=7, tile=16)
    finally:
        lib().olsr_debug_sort_knobs(0, -1, -1)


@pytest.mark.parametrize("P", [1, 2, 63, 64, 65, 1023, 1024, 1025, 2048, 2049, 4096, 4097, 5000, 8191, 8192, 8193, 10000])
def test_one_launch_depth_sort_equals_the_radix_passes(hip, P):
    """Round 5: up to 8 192 Gaussians are depth-sorted by ONE launch of one workgroup (k_sort.hip: sort_small_kernel — the
    histogram, the frame's bookkeeping and the four 8-bit passes inside the block).  Its order, the frame's counters, the
    emission totals and everything downstream must equal what the histogram launch + four radix passes leave, bit for bit,
    at every block shape (2 / 4 / 8 keys per thread, ragged tails, one Gaussian); beyond 8 192 the passes run either way."""
    from online_lang_splatting_amd._lib import lib
    dev = torch.device(DEV)
    sc = make_scene(P, 200, 150, 3, seed=300 + P % 97)
    # equal depths in quantity: the order among them must be the index order (stable)
    sc.means3D[::3, 2] = sc.means3D[0, 2] if P > 3 else sc.means3D[::3, 2]
    a = fwd_args(sc, dev)
    out = {}
    try:
        lib().olsr_debug_sort_compact(0)   # (the pass kernels' order of EVERY Gaussian is what is compared with the small sort's)
        for small in (1, 0):
            lib().olsr_debug_sort_small(small)
            R, color, lang, radii, geom, binb, img, depth, opac, nt = hip.rasterize_language_gaussians(*a)
            order = hip.state_field("geometry", geom, "depth_order", P=P, F=3, dtype=torch.int32, count=P).clone()
            cnt = hip.state_field("geometry", geom, "counters", P=P, F=3, dtype=torch.int32, count=10).clone()
            # (one 64-bit total per block of 1024 ranks: instances | emitting Gaussians << 40)
            et = hip.state_field("geometry", geom, "emit_totals", P=P, F=3, dtype=torch.int32, count=2 * ((P + 1023) // 1024)).clone()
            pl = hip.state_field("binning", binb, "point_list", R=R, F=3, dtype=torch.int32, count=R).clone() if R else None
            out[small] = (R, order, cnt, et, pl, color.clone(), lang.clone(), depth.clone(), nt.clone())
    finally:
        lib().olsr_debug_sort_small(1)
        lib().olsr_debug_sort_compact(1)
    assert out[1][0] == out[0][0]
    for x, y in zip(out[1][1:], out[0][1:]):
        assert (x is None and y is None) or torch.equal(x, y)
    assert sorted(out[1][1].tolist()) == list(range(P))


@pytest.mark.parametrize("P,frac", [(8193, 0.5), (10000, 0.0), (10000, 1.0), (30000, 0.3), (70001, 0.8), (262144, 0.2)])
def test_visible_set_compaction_of_the_depth_sort_changes_nothing(hip, P, frac):
    """Round 6: the depth sort orders only the Gaussians that emit instances (the histogram kernel compacts them in index
    order).  With a share `frac` of the Gaussians pushed behind the camera — whole blocks of 256 of them, ragged tails, none,
    all — the compacted order equals the emitting subsequence of the uncompacted one, and counters, emission totals, lists,
    images and n_touched are bit-identical."""
    from online_lang_splatting_amd._lib import lib
    dev = torch.device(DEV)
    sc = make_scene(P, 200, 150, 3, seed=500 + P % 89)
    g = torch.Generator().manual_seed(P)
    hide = torch.rand(P, generator=g) < frac
    hide[: min(P, 700)] = frac > 0            # (whole leading blocks without a single emitting Gaussian)
    sc.means3D[hide, 2] = -1.0
    sc.means3D[::5, 2] = sc.means3D[4, 2]     # ties in quantity
    a = fwd_args(sc, dev)
    out = {}
    try:
        for comp in (1, 0):
            lib().olsr_debug_sort_compact(comp)
            R, color, lang, radii, geom, binb, img, depth, opac, nt = hip.rasterize_language_gaussians(*a)
            cnt = hip.state_field("geometry", geom, "counters", P=P, F=3, dtype=torch.int32, count=13).clone()
            tt = hip.state_field("geometry", geom, "tiles_touched", P=P, F=3, dtype=torch.int32, count=P).clone()
            if comp:
                n_ord = int(cnt[12])
                order = hip.state_field("geometry", geom, "depth_order_compacted", P=P, F=3, dtype=torch.int32, count=max(n_ord, 1))[:n_ord].clone()
            else:
                full = hip.state_field("geometry", geom, "depth_order", P=P, F=3, dtype=torch.int32, count=P).clone()
                order = full[tt[full.long()] > 0]
            pl = hip.state_field("binning", binb, "point_list", R=R, F=3, dtype=torch.int32, count=R).clone() if R else None
            out[comp] = (R, order, cnt[:11], pl, color.clone(), lang.clone(), depth.clone(), nt.clone(), radii.clone())
    finally:
        lib().olsr_debug_sort_compact(1)
    assert out[1][0] == out[0][0]
    assert int((tt > 0).sum()) == out[1][1].numel()
    for x, y in zip(out[1][1:], out[0][1:]):
        assert (x is None and y is None) or torch.equal(x, y)


@pytest.mark.parametrize("threads", [256, 1024])
@pytest.mark.parametrize("kpt", [0, 4, 8, 12])
def test_radix_block_shapes_give_the_same_lists(hip, oracle, threads, kpt):
    """Round 5: the radix passes and their histogram kernels run as 1024-thread or — for a scene that says frames are in flight
    on other streams (OLSR_FLAG_FRAMES_IN_FLIGHT) — as four-wave workgroups; olsr_debug_sort_threads forces a shape.  Every
    shape and keys-per-thread leaves the reference's lists and the oracle's images, bit for bit."""
    from online_lang_splatting_amd._lib import lib
    try:
        lib().olsr_debug_sort_threads(threads)
        lib().olsr_debug_sort_knobs(kpt, -1, -1)
        _check(hip, oracle, make_scene(60000, 640, 480, 15, seed=81), seed=8)
        _check(hip, oracle, make_scene(9000, 200, 150, 0, seed=82), seed=9, tile=16)
    finally:
        lib().olsr_debug_sort_threads(0)
        lib().olsr_debug_sort_knobs(0, -1, -1)


def test_frames_in_flight_flag_changes_no_result(hip):
    """The per-call form: a workspace whose scenes carry FLAG_FRAMES_IN_FLIGHT (FrameLanes with more than one lane sets it)
    renders the same frame and the same gradients as one without."""
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket, RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(40000, 400, 300, 15, seed=83)
    cam = sc.camera
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
             viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
             projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
             tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    cot = [t.to(dev) for t in sc.cotangents(2)]
    res = []
    for flags in (0, _abi.FLAG_FRAMES_IN_FLIGHT):
        ws = RasterWorkspace(sc.P, 400, 300, 15, sc.shs.shape[1], 1_500_000, dev, flags=flags)
        b = GradientBucket(sc.P, GradLayout(sc.shs.shape[1], 15), dev)
        ws.set_scene(**g)
        out = {k: v.clone() for k, v in ws.forward().items()}
        ws.backward(*cot, bucket=b, first=True, bucket_only=True)
        torch.cuda.synchronize()
        res.append((out, b.flat.clone(), ws.rendered()))
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    assert torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]


def test_repeated_backward_on_one_forward(hip):
    """The row compaction's look-back state is re-armed by the kernel itself: a backward may be repeated on the same
    forward (autograd's retain_graph, or the test above) and must give the same bits every time."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(30000, 400, 300, 15, seed=75)
    cam = sc.camera
    ws = RasterWorkspace(sc.P, 400, 300, 15, sc.shs.shape[1], 1_500_000, dev)
    ws.set_scene(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
                 rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev),
                 viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
                 projmatrix_raw=cam.projection_matrix.to(dev), campos=cam.camera_center.to(dev), tanfovx=cam.tanfovx,
                 tanfovy=cam.tanfovy, sh_degree=sc.sh_degree)
    dc, dl, dd = (t.to(dev) for t in sc.cotangents(1))
    ws.forward()
    first = {k: v.clone() for k, v in ws.backward(dc, dl, dd).items()}
    st0 = ws.backward_status()
    for _ in range(4):
        again = ws.backward(dc, dl, dd)
        assert ws.backward_status() == st0
        for k in first:
            assert torch.equal(again[k], first[k]), k
    assert st0[0] > 0 and not st0[1]


def test_config1_rgb_forward(hip, oracle):
    """BASELINE.json configs[0]: 10 k Gaussians, 256x256, RGB-only (SH degree 3)."""
    from online_lang_splatting_amd.scene import make_config_scene
    _check(hip, oracle, make_config_scene(1), seed=1)


def test_mark_visible(hip, oracle):
    sc = make_scene(5000, 64, 48, 0, seed=81)
    cam = sc.camera
    dev = torch.device(DEV)
    got = hip.mark_visible(sc.means3D.to(dev), cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev))
    exp = oracle.mark_visible(sc.means3D, cam.world_view_transform, cam.full_proj_transform)
    assert got.dtype == torch.bool and torch.equal(got.cpu(), exp)
    assert 0 < int(exp.sum()) < sc.P


def test_run_to_run_determinism(hip):
    """No float atomics anywhere: gradients are bit-reproducible."""
    sc = make_scene(20000, 320, 240, 15, seed=91)
    dev = torch.device(DEV)
    _, g1 = run_backend(hip, sc, dev, 1, 15, 0)
    _, g2 = run_backend(hip, sc, dev, 1, 15, 0)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("factor", [5000.0, 1.0e6])
def test_depth_sort_fourth_pass_for_far_gaussians(hip, oracle, factor):
    """Depth keys are float bits: a scene scaled by `factor` (positions and scales: the same picture, depths x factor) moves
    the keys into other exponent ranges — every byte of the four 8-bit passes carries information; the instance lists must
    still equal the reference's bit for bit.  (Round 3 tried three 9-bit passes on keys rebased at bits(0.2) with a
    conditional fourth pass for depths beyond 13 107: correct, but 512-digit passes are slower — 76 us against 73 for the
    depth sort of config 3 — and it was not kept; the test stays.)"""
    sc = make_scene(6000, 200, 150, 15, seed=97)
    sc.means3D = (sc.means3D * factor).contiguous()
    sc.scales = (sc.scales * factor).contiguous()
    depths = sc.means3D[:, 2]
    assert float(depths.max()) > 13107.0 * 1.5
    # (the gradients of a scene of this size span 1e-12 .. 1e-2 per tensor: the forward, the lists and the composite-level
    #  gradients are what this test is about)
    _check(hip, oracle, sc, seed=5, grad_keys=COMPOSITE_GRADS)
