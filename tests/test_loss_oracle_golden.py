"""The mapping-loss oracle (oracle/loss_oracle.py) against vectors produced by the reference's own
get_loss_mapping + F.interpolate + l1_loss (tests/golden/make_golden_loss.py -> mapping_loss.npz)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss_oracle  # noqa: E402


def golden_cases():
    z = np.load(os.path.join(ROOT, "tests", "golden", "mapping_loss.npz"))
    for i in range(int(z["n_cases"])):
        yield {k[len(f"c{i}_"):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(f"c{i}_")}


def golden_tracking_cases():
    z = np.load(os.path.join(ROOT, "tests", "golden", "mapping_loss.npz"))
    for i in range(int(z["n_tracking_cases"])):
        yield {k[len(f"t{i}_"):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(f"t{i}_")}


def run_tracking_oracle(c):
    return loss_oracle.tracking_loss_and_grads(
        c["image"], c["depth"], c["opacity"], c["gt_image"], c["gt_depth"], c["grad_mask"].view(1, *c["gt_depth"].shape) > 0.5,
        c["a"], c["b"], alpha=float(c["alpha"]), rgb_boundary_threshold=float(c["thr"]))


def run_oracle(c):
    return loss_oracle.mapping_loss_and_grads(
        c["image"], c["depth"], c["lang"], c["gt_image"], c["gt_depth"], c["gt_lang"], c["a"], c["b"],
        alpha=float(c["alpha"]), rgb_boundary_threshold=float(c["thr"]), lamda_lang=1.0, initialization=bool(int(c["init"])))


def test_oracle_reproduces_the_reference_loss_and_gradients():
    n = 0
    for c in golden_cases():
        o = run_oracle(c)
        # same PyTorch ops in the same order on the same machine: equal to the last bit
        assert torch.equal(o["loss"], c["loss"])
        assert torch.equal(o["rgb"] + o["depth"], c["loss_map"])
        assert torch.equal(o["lang"], c["loss_lang"])
        assert torch.equal(o["dL_dimage"], c["d_image"])
        assert torch.equal(o["dL_ddepth"], c["d_depth"])
        assert torch.equal(o["dL_dlanguage"], c["d_lang"])
        assert torch.equal(o["dL_da"], c["d_a"]) and torch.equal(o["dL_db"], c["d_b"])
        n += 1
    assert n == 3


def test_masks_and_ties_have_zero_gradient():
    c = next(golden_cases())
    H, W = c["image"].shape[1:]
    assert float(c["d_image"][:, : H // 4, : W // 3].abs().max()) == 0      # below the rgb boundary threshold
    assert float(c["d_image"][:, H - 1, W - 1].abs().max()) == 0            # exact tie: d|0| = 0
    assert float(c["d_depth"][0, H // 2:, : W // 5].abs().max()) == 0       # invalid depth
    assert float(c["d_image"].abs().max()) > 0 and float(c["d_lang"].abs().max()) > 0


def test_tracking_oracle_reproduces_the_reference():
    n = 0
    for c in golden_tracking_cases():
        o = run_tracking_oracle(c)
        assert torch.equal(o["loss"], c["loss"])
        assert torch.equal(o["dL_dimage"], c["d_image"])
        assert torch.equal(o["dL_ddepth"], c["d_depth"])
        assert torch.equal(o["dL_da"], c["d_a"]) and torch.equal(o["dL_db"], c["d_b"])
        n += 1
    assert n == 2
