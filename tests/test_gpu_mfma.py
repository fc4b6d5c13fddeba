"""OLSR_FLAG_FWD_ACCUM_MFMA: the forward's feature accumulation on the matrix cores (v_mfma_f32_16x16x4_f32).
Every decision stays on the vector ALU: final_T, n_contrib, radii, n_touched, the instance lists and the backward's
liveness flags — hence every gradient — are bit-identical to the default path; the images are rounded as
fma(alpha T, f, C) instead of fma(f alpha, T, C) and must sit within 1e-6 (relative to the image's range) of the oracle."""
import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import default_camera, make_config_scene, make_scene
from parity_common import rel_err, run_backend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IMG_TOL = 1e-6


def _compare(hip, oracle, sc, seed, tile=15, mode=0, binning=_abi.BINNING_ELLIPSE, flag=_abi.FLAG_FWD_ACCUM_MFMA):
    dev = torch.device(DEV)
    fo, go = run_backend(oracle, sc, None, seed, tile, mode)
    fv, gv = run_backend(hip, sc, dev, seed, tile, mode, binning=binning)
    hip.FLAGS = flag
    try:
        fm, gm = run_backend(hip, sc, dev, seed, tile, mode, binning=binning)
    finally:
        hip.FLAGS = 0
    W, H = sc.camera.width, sc.camera.height
    assert fm["R"] == fv["R"]
    for k in ("radii", "n_touched", "opacity"):  # opacity = 1 - final_T: a decision-path value
        assert torch.equal(fm[k], fv[k]), k
    for name in ("final_T", "n_contrib"):
        dt = torch.float32 if name == "final_T" else torch.int32
        a = hip.state_field("image", fm["img"], name, W=W, H=H, dtype=dt, count=W * H)
        b = hip.state_field("image", fv["img"], name, W=W, H=H, dtype=dt, count=W * H)
        assert torch.equal(a, b), name
    if fv["R"] > 0:
        fa = hip.state_field("binning", fm["binning"], "flags", R=fm["R"], F=sc.F, dtype=torch.uint8, count=fm["R"])
        fb = hip.state_field("binning", fv["binning"], "flags", R=fv["R"], F=sc.F, dtype=torch.uint8, count=fv["R"])
        assert torch.equal(fa, fb)
    worst = 0.0
    for k in ("color", "language", "depth"):
        if fo[k] is not None and fo[k].numel():
            r, _ = rel_err(fm[k], fo[k])
            worst = max(worst, r)
            assert r <= IMG_TOL, f"{k}: {r:.2e} of the image's range"
            assert torch.equal(fv[k].cpu(), fo[k])  # (the default path stays bit-identical)
    for k in gv:  # the backward never reads the images: same gradients, bit for bit
        assert torch.equal(gm[k], gv[k]), k
    oracle.release(fo["geom"])
    return worst


@pytest.mark.parametrize("F", [0, 3, 15, 16, 32])
def test_mfma_accumulation_all_channel_counts(hip, oracle, F):
    _compare(hip, oracle, make_scene(3000, 160, 120, F, seed=60 + F), seed=F)


@pytest.mark.parametrize("F", [0, 15, 32])
def test_weight_accumulation_on_the_vector_alu(hip, oracle, F):
    """OLSR_FLAG_FWD_ACCUM_WEIGHT: the MFMA variant's rounding (fma(alpha T, f, C)) on the vector ALU."""
    _compare(hip, oracle, make_scene(3000, 160, 120, F, seed=60 + F), seed=F, flag=_abi.FLAG_FWD_ACCUM_WEIGHT)
    if F == 15:
        _compare(hip, oracle, make_scene(6000, 200, 150, 15, seed=83, scale_mult=6.0), seed=3, tile=16,
                 flag=_abi.FLAG_FWD_ACCUM_WEIGHT)


@pytest.mark.parametrize("tile,mode", [(15, _abi.BWD_EXACT), (16, _abi.BWD_REFERENCE), (16, _abi.BWD_EXACT)])
def test_mfma_accumulation_tiles_and_modes(hip, oracle, tile, mode):
    # 157 x 101: partial tiles on both edges; background colour; rotated camera
    cam = default_camera(157, 101, yaw_deg=7.0, tx=0.1)
    sc = make_scene(4000, 157, 101, 15, seed=71, bg=torch.tensor([0.3, 0.6, 0.1]), camera=cam)
    _compare(hip, oracle, sc, seed=2, tile=tile, mode=mode)


def test_mfma_accumulation_long_lists_and_rect_binning(hip, oracle):
    """Screen-filling splats (lists of thousands of entries per tile, several LDS batches, partial groups at every batch
    end) and the reference's bounding-square lists."""
    sc = make_scene(6000, 200, 150, 15, seed=83, scale_mult=6.0)
    _compare(hip, oracle, sc, seed=3)
    _compare(hip, oracle, sc, seed=4, binning=_abi.BINNING_RECT)


def test_mfma_accumulation_full_config3(hip, oracle):
    worst = _compare(hip, oracle, make_config_scene(3), seed=3)
    print(f"config 3: worst image deviation {worst:.2e} of the image's range")
