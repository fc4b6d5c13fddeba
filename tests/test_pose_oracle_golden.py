"""oracle/pose_oracle.py against the golden vectors the reference's own pose update produced
(tests/golden/pose.npz <- tests/golden/make_golden_pose.py: torch.optim.Adam as set up in utils/slam_frontend.py,
utils/pose_utils.update_pose, utils/camera_utils.Camera)."""
import os

import numpy as np

from oracle.pose_oracle import PoseOracle, se3_exp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose.npz"))
CONV = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conventions.npz"))


def test_se3_exp_matches_the_reference():
    for tau, want in zip(CONV["se3_tau"], CONV["se3_exp"]):
        np.testing.assert_allclose(se3_exp(tau), want, rtol=0, atol=2e-7)


def test_pose_sequences_match_the_reference():
    for s in range(int(G["num_seq"])):
        lr = G[f"seq{s}_lr"]
        o = PoseOracle(G[f"seq{s}_R0"], G[f"seq{s}_T0"], G[f"seq{s}_proj"], lr_rot=lr[0], lr_trans=lr[1], lr_exposure=lr[2])
        for i, (gt, ge) in enumerate(zip(G[f"seq{s}_grad_tau"], G[f"seq{s}_grad_exposure"])):
            conv = o.step(gt, ge)
            assert conv == bool(G[f"seq{s}_converged"][i]), (s, i)
            np.testing.assert_allclose(o.tau, G[f"seq{s}_tau"][i], rtol=2e-6, atol=2e-6 * float(lr[:2].max()), err_msg=f"tau {s} {i}")  # (absolute part: a moment that nearly cancels)
            np.testing.assert_allclose(o.T_w2c[:3, :3], G[f"seq{s}_R"][i], rtol=0, atol=5e-7 * (i + 1))
            np.testing.assert_allclose(o.T_w2c[:3, 3], G[f"seq{s}_T"][i], rtol=0, atol=5e-7 * (i + 1))
            np.testing.assert_allclose(o.viewmatrix, G[f"seq{s}_view"][i], rtol=0, atol=5e-7 * (i + 1))
            scale = np.abs(G[f"seq{s}_full"][i]).max()
            np.testing.assert_allclose(o.projmatrix, G[f"seq{s}_full"][i], rtol=0, atol=1e-6 * scale * (i + 1))
            np.testing.assert_allclose(o.campos, G[f"seq{s}_campos"][i], rtol=0, atol=1e-6 * (i + 1))
            np.testing.assert_allclose(o.exposure, G[f"seq{s}_exposure"][i], rtol=2e-6, atol=1e-9)
