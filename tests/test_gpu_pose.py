"""olsr_pose_step (csrc/k_pose.hip) against the golden vectors from the reference's own pose update and against the CPU
oracle; PoseState / TrackingLoop of slam_iterations.py end to end (a perturbed pose converges back)."""
import os

import numpy as np
import pytest
import torch

from online_lang_splatting_amd.scene import default_camera, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose.npz"))


def _pose_state(s, dev):
    from online_lang_splatting_amd.slam_iterations import PoseState
    lr = G[f"seq{s}_lr"]
    T = torch.eye(4)
    T[:3, :3] = torch.from_numpy(G[f"seq{s}_R0"])
    T[:3, 3] = torch.from_numpy(G[f"seq{s}_T0"])
    return PoseState(T.to(dev), torch.from_numpy(G[f"seq{s}_proj"]).to(dev), 1.0, 1.0, lr_rot=float(lr[0]),
                     lr_trans=float(lr[1]), lr_exposure=float(lr[2]))


def test_pose_step_matches_the_references_update(hip):
    from oracle.pose_oracle import PoseOracle
    dev = torch.device(DEV)
    for s in range(int(G["num_seq"])):
        ps = _pose_state(s, dev)
        lr = G[f"seq{s}_lr"]
        o = PoseOracle(G[f"seq{s}_R0"], G[f"seq{s}_T0"], G[f"seq{s}_proj"], lr_rot=lr[0], lr_trans=lr[1], lr_exposure=lr[2])
        # the start pose's matrices exist before any step
        np.testing.assert_allclose(ps.viewmatrix.cpu().numpy(), o.viewmatrix, rtol=0, atol=0)
        np.testing.assert_allclose(ps.campos.cpu().numpy(), o.campos, rtol=0, atol=1e-6)
        for i, (gt, ge) in enumerate(zip(G[f"seq{s}_grad_tau"], G[f"seq{s}_grad_exposure"])):
            ps.step(torch.from_numpy(gt).to(dev), torch.from_numpy(ge).to(dev))
            o.step(gt, ge)
            st = ps.status.cpu()
            assert bool(st[0]) == bool(G[f"seq{s}_converged"][i]) and int(st[1]) == i + 1
            tol = 5e-7 * (i + 1)
            np.testing.assert_allclose(ps.last_tau.cpu().numpy(), G[f"seq{s}_tau"][i], rtol=2e-6, atol=2e-6 * float(lr[:2].max()))
            np.testing.assert_allclose(ps.T_w2c.cpu().numpy()[:3, :3], G[f"seq{s}_R"][i], rtol=0, atol=tol)
            np.testing.assert_allclose(ps.T_w2c.cpu().numpy()[:3, 3], G[f"seq{s}_T"][i], rtol=0, atol=tol)
            np.testing.assert_allclose(ps.viewmatrix.cpu().numpy(), G[f"seq{s}_view"][i], rtol=0, atol=tol)
            scale = np.abs(G[f"seq{s}_full"][i]).max()
            np.testing.assert_allclose(ps.projmatrix.cpu().numpy(), G[f"seq{s}_full"][i], rtol=0, atol=1e-6 * scale * (i + 1))
            np.testing.assert_allclose(ps.campos.cpu().numpy(), G[f"seq{s}_campos"][i], rtol=0, atol=1e-6 * (i + 1))
            np.testing.assert_allclose(ps.exposure.cpu().numpy(), G[f"seq{s}_exposure"][i], rtol=2e-6, atol=1e-9)
            # ... and the oracle (the checker of larger runs) stays with the kernel
            np.testing.assert_allclose(ps.T_w2c.cpu().numpy(), o.T_w2c, rtol=0, atol=tol)
        # the last row of T_w2c is exactly [0 0 0 1] (update_RT rebuilds it)
        assert ps.T_w2c.cpu()[3].tolist() == [0.0, 0.0, 0.0, 1.0]


def test_tracking_loop_recovers_a_perturbed_pose(hip):
    """render -> olsr_tracking_loss -> pose-only backward without a language cotangent -> olsr_pose_step: the loss of a
    perturbed start pose falls and the pose moves back towards the one the targets were rendered from; the NULL-language
    path gives the same trajectory as zero-filled cotangents."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
    from oracle.pose_oracle import se3_exp
    dev = torch.device(DEV)
    W, H, F = 320, 240, 15
    sc = make_scene(20000, W, H, F, seed=21)
    cam = default_camera(W, H)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    proj = cam.projection_matrix.to(dev)
    ws = RasterWorkspace(sc.P, W, H, F, sc.shs.shape[1], 2_000_000, dev)
    T_gt = torch.eye(4, device=dev)
    ps = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy, optimise_exposure=False)
    ws.set_scene(sh_degree=sc.sh_degree, **ps.camera(), **g)
    out = ws.forward()
    gt_image, gt_depth = out["color"].clone(), out["depth"][0].clone()
    T0 = torch.from_numpy(se3_exp(np.array([0.02, -0.015, 0.01, 0.004, -0.006, 0.003], dtype=np.float32))).to(dev) @ T_gt
    traj = {}
    for variant in ("null", "zeros"):
        ps.reset(T0)
        loop = TrackingLoop(ws, g, sc.sh_degree, ps, gt_image, gt_depth, language_cotangent=variant)
        losses_, errs = [], []
        for it in range(60):
            loop.iteration()
            losses_.append(float(loop.loss[0]))
            errs.append(float((ps.T_w2c - T_gt).abs().max()))
        # (Adam moves a pose increment by about its learning rate per step: 1 mm / 3 mrad)
        assert losses_[-1] < 0.8 * losses_[0], (variant, losses_[0], losses_[-1])
        assert errs[-1] < 0.8 * errs[0], (variant, errs[0], errs[-1])
        traj[variant] = ps.T_w2c.clone()
    assert torch.equal(traj["null"], traj["zeros"])


def test_tracking_iteration_replays_from_a_hip_graph(hip):
    """The whole iteration — sync-free forward, tracking loss, pose-only backward, pose step — is a fixed sequence of
    launches on device-resident state (the Adam step number included: PoseState(device_step_count=True)), so it can be
    recorded into a HIP graph once and replayed: same pose, same optimiser state, bit for bit, as issuing it eagerly.
    The device-side step count itself gives what the host-side count gives."""
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    from online_lang_splatting_amd.slam_iterations import PoseState, TrackingLoop
    from oracle.pose_oracle import se3_exp
    dev = torch.device(DEV)
    W, H, F = 320, 240, 15
    sc = make_scene(20000, W, H, F, seed=22)
    cam = default_camera(W, H)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    proj = cam.projection_matrix.to(dev)
    ws = RasterWorkspace(sc.P, W, H, F, sc.shs.shape[1], 2_000_000, dev)
    T_gt = torch.eye(4, device=dev)
    ref = PoseState(T_gt, proj, cam.tanfovx, cam.tanfovy)
    ws.set_scene(sh_degree=sc.sh_degree, **ref.camera(), **g)
    out = ws.forward()
    gt_image, gt_depth = out["color"].clone(), out["depth"][0].clone()
    T0 = torch.from_numpy(se3_exp(np.array([0.02, -0.015, 0.01, 0.004, -0.006, 0.003], dtype=np.float32))).to(dev) @ T_gt
    states = {}
    for name, on_device in (("host_count", False), ("device_count", True), ("graph", True)):
        ps = PoseState(T0, proj, cam.tanfovx, cam.tanfovy, device_step_count=on_device)
        loop = TrackingLoop(ws, g, sc.sh_degree, ps, gt_image, gt_depth)
        graph = None
        if name == "graph":  # (the test's own capture: the product no longer offers one, slam_iterations.TrackingLoop)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    loop.iteration()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loop.iteration()
        ps.reset(T0)
        for _ in range(25):
            if graph is not None:
                graph.replay()
            else:
                loop.iteration()
        torch.cuda.synchronize()
        assert ps.status.tolist()[1] == 25
        states[name] = ps.state.clone()
    assert torch.equal(states["graph"], states["device_count"])
    # host- and device-side bias corrections: the same doubles up to the last bit of pow(), cast to float
    assert torch.allclose(states["device_count"], states["host_count"], rtol=1e-6, atol=1e-9)
    assert not torch.equal(states["graph"][:16].view(4, 4), T0)  # (convergence itself: the test above)
