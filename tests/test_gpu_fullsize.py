"""Oracle parity at BASELINE.json's FULL sizes (configs 2, 3 and all eight views of config 5).

Everything `_check` asserts on small scenes is asserted here at full size, against the CPU oracle on the same
seeded inputs: images / radii / n_touched / final_T bit-identical in both binning modes, num_rendered, point_list
and n_contrib bit-identical with the reference's bounding-square binning (the first contact of the large-sort
paths — thousands of radix blocks, multi-block scans — with the oracle), exact tile lists an order-preserving
sub-list that keeps every blending instance, and EVERY gradient (composite level, per-Gaussian chain, dL_dtau per
Gaussian and summed) under the element-wise north-star criterion: >= 99.99 % of the elements within 1e-4 relative
(+ 1e-6 of the tensor's largest magnitude), the worst element bounded (WORST_BOUND), both printed and written to
gpurun_out/parity_fullsize.json.  Since late round 4 the per-Gaussian chain on identical inputs (the oracle replaying the
reference's chain on the product's composite-level gradients) must EQUAL the product's outputs element for element
(`_check(chain_exact=True)`): its `:chain:` rows in the record have worst = 0.

The oracle needs ~2 s (config 2/3) to ~10 s (config 5) per frame on the GPU box's host cores.
"""
import json
import os

import pytest
import torch

from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.scene import CONFIGS, arc_cameras, make_config_scene, make_scene
from test_gpu_parity import _check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Round 6: the backward is closed without a tolerance of its own.  (i) The composite backward in the reference's own
# association, on the GPU (olsr_debug_backward_ordered), EQUALS the oracle's composite-level gradients bit for bit at every
# full config; (ii) the product's fast kernel differs from it by ROUNDING ONLY — every element of every composite-level tensor
# within K x 2^-24 x its condition (the same sums over magnitudes), K <= 36 measured here (bound 64; config 3's one dL_dconic
# element at 1.1e-4 of its value sits at K = 3.4: three terms of order 1e8 cancel); (iii) the per-Gaussian chain behind them is
# exact on identical inputs.  What remains below is the north-star criterion itself and a sanity bound on the worst single
# element behind the chain, in the relative units of parity_common.elementwise_report (1e-4 == within tolerance): the chain
# amplifies composite-level rounding behind the inverse of the 2D covariance (measured worst 4.7e-3, config 5 view 2; 1.1e-4
# at composite level, through _check's composite_worst_bound = 2e-4).
WORST_BOUND = 1e-2


def _dump(tag, log):
    out = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(out):
        return
    path = os.path.join(out, "parity_fullsize.json")
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[tag] = log
    json.dump(data, open(path, "w"), indent=1)


@pytest.mark.parametrize("cfg", [2, 3])
def test_full_config_against_the_oracle(hip, oracle, cfg):
    """BASELINE.json configs[1] (100 k Gaussians, 640x480, RGB + depth) and configs[2] (500 k, 1200x680, F = 15),
    forward + reference-mode backward, every output and every gradient."""
    log = []
    _check(hip, oracle, make_config_scene(cfg), seed=cfg, elementwise=True, worst_bound=WORST_BOUND, log=log)
    _dump(f"config{cfg}", log)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("view", range(8))
def test_config5_every_view_against_the_oracle(hip, oracle, view):
    """BASELINE.json configs[4]: 2 M Gaussians, 1920x1080, F = 32 — each of its eight arc views (what the eight ranks of
    the frame-sharded run render; yaw -14 .. +14 degrees, 0.15 m apart), ~15 M bounding-square instances per view: the
    large radix tables, multi-round sort passes and 192-byte gradient rows meet the oracle here."""
    c = CONFIGS[5]
    cam = arc_cameras(c["W"], c["H"], 8)[view]
    sc = make_scene(c["P"], c["W"], c["H"], c["F"], seed=5, max_sh_degree=c["max_sh_degree"], camera=cam)
    log = []
    _check(hip, oracle, sc, seed=5, elementwise=True, worst_bound=WORST_BOUND, log=log)
    _dump(f"config5_view{view}", log)
    torch.cuda.empty_cache()


def test_config3_exact_backward_against_the_oracle(hip, oracle):
    """The true-gradient backward (OLSR_BWD_EXACT: four waves per tile, F extra reduced values) at full config-3
    size."""
    log = []
    _check(hip, oracle, make_config_scene(3), seed=13, mode=_abi.BWD_EXACT, elementwise=True,
           worst_bound=WORST_BOUND, log=log)
    _dump("config3_exact", log)
    torch.cuda.empty_cache()


def test_config3_against_the_contract_everywhere_model_of_the_reference(hip, oracle):
    """Both models of what nvcc makes of the reference are on record every round.  The parity
    oracle contracts the value accumulation only (fma(f alpha, T, C)) and keeps every decision expression un-contracted; nvcc's
    default --fmad=true would contract the decision expressions as well.  `liboracle_contract.so` is the same restatement
    built with -ffp-contract=fast: a proxy for that compiler.  The product cannot be bit-identical to both; against this model
    it must still meet the north-star tolerance — every image and every composite-level gradient per element (a handful of
    blend decisions flip: 12 of 54.9 M between the two oracles, profiles/r2_cuda_sensitivity.json), the tensors behind the
    per-Gaussian chain at >= 99.99 % — and the figures are written to gpurun_out/parity_fullsize.json next to the default
    model's."""
    from parity_common import elementwise_report, run_backend
    from test_gpu_parity import DEV
    sc = make_config_scene(3)
    log = []
    oracle.use_variant("contract_fast")
    try:
        assert oracle.lib().oracle_variant().decode() != "default"
        fo, go = run_backend(oracle, sc, None, 3, 15, _abi.BWD_REFERENCE)
        fg, gg = run_backend(hip, sc, torch.device(DEV), 3, 15, _abi.BWD_REFERENCE, binning=_abi.BINNING_ELLIPSE)
        torch.cuda.synchronize()
        n_rad = int((fg["radii"].cpu() != fo["radii"]).sum())
        log.append(dict(name="contract:radii_different", n=n_rad))
        assert n_rad <= 2, n_rad
        for k in ("color", "language", "depth", "opacity"):
            got = fg[k].cpu().reshape(fo[k].shape)
            r = elementwise_report(got, fo[k])
            r.update(name=f"contract:{k}", elements_not_bit_identical=int((got != fo[k]).sum()))
            log.append(r)
            print(r)
            # a flipped blend decision sits at the alpha floor: it moves a pixel's channel by at most ~(1/255) |feature| T —
            # isolated elements far outside the relative band but small in absolute terms
            assert r["frac_within"] >= 0.9999 and r["worst_abs"] <= 1.5 / 255.0 * max(1.0, r["max_ref"]), r
        for k in go:
            if go[k].numel():
                r = elementwise_report(gg[k], go[k])
                r.update(name=f"contract:{k}")
                log.append(r)
                print(r)
                assert r["frac_within"] >= 0.9999, r
                assert r["worst_abs"] <= 2e-2 * r["max_ref"], r   # (flips are bounded perturbations, not blow-ups)
        oracle.release(fo["geom"])
    finally:
        oracle.use_variant("default")
    assert oracle.lib().oracle_variant().decode() == "default"
    _dump("config3_contract_everywhere_oracle", log)
    torch.cuda.empty_cache()
