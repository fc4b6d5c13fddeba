"""The surface-structured map generator (online_lang_splatting_amd/scene.py: make_room_scene) against the recipe it
restates — gaussian_splatting/scene/gaussian_model.py:180-281 (depth back-projection per keyframe, random down-sampling by
32 / 64, scale = sqrt(distCUDA2 * point_size) on three equal axes, identity rotation, opacity 0.5, RGB2SH) — and the oracle
rendering it: a map built by back-projecting ray-cast depth must render that depth again."""

import torch

from online_lang_splatting_amd.scene import C0_SH, knn_mean_dist2_host, make_room_scene


def test_recipe_and_determinism(oracle):
    W, H, F = 150, 105, 15
    rs = make_room_scene(4000, W, H, F, views=4, seed=1, knn=knn_mean_dist2_host)
    sc = rs.scene
    assert sc.P == 4000 and rs.keyframes == len(rs.points_per_keyframe)
    n = W * H
    assert rs.points_per_keyframe[0] == n // 32 and rs.points_per_keyframe[1] == n // 64   # pcd_downsample_init / pcd_downsample
    assert sum(rs.points_per_keyframe) == sc.P
    assert torch.equal(sc.opacities, torch.full((sc.P, 1), 0.5))                           # inverse_sigmoid(0.5) activated
    assert torch.equal(sc.rotations, torch.tensor([1.0, 0, 0, 0]).repeat(sc.P, 1))
    assert torch.equal(sc.scales[:, 0], sc.scales[:, 1]) and torch.equal(sc.scales[:, 0], sc.scales[:, 2])
    assert torch.allclose(sc.language.norm(dim=1), torch.ones(sc.P), atol=1e-6)
    rgb = sc.shs[:, 0, :] * C0_SH + 0.5                                                    # SH2RGB of f_dc
    assert float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1 + 1e-6
    # the scale recipe on the first keyframe's cloud, neighbours by the oracle's brute force: sqrt(max(d2, 1e-7) * point_size)
    n0 = rs.points_per_keyframe[0]
    d2 = oracle.distCUDA2(sc.means3D[:n0])
    ratio = sc.scales[:n0, 0] ** 2 / torch.clamp_min(d2, 1e-7)
    assert float(ratio.max() - ratio.min()) <= 1e-6 * float(ratio.max())                   # one point_size per keyframe ...
    assert 0.0 < float(ratio[0]) <= 0.05 + 1e-6                                            # ... min(0.05, 0.05 * median depth)
    # every Gaussian lies on a surface of the room: inside the box, and on the keyframe's ray-cast depth
    half = torch.tensor([3.5, 1.4, 2.5]) + 1e-3
    assert bool((sc.means3D.abs() <= half).all())
    again = make_room_scene(4000, W, H, F, views=4, seed=1, knn=knn_mean_dist2_host).scene
    for a, b in ((sc.means3D, again.means3D), (sc.scales, again.scales), (sc.shs, again.shs), (sc.language, again.language)):
        assert torch.equal(a, b)
    other = make_room_scene(4000, W, H, F, views=4, seed=2, knn=knn_mean_dist2_host).scene
    assert not torch.equal(sc.means3D, other.means3D)


def test_the_oracle_renders_the_depth_the_map_was_built_from(oracle):
    """Back-projection and pose conventions: alpha-normalised rendered depth equals the ray-cast depth of the same view."""
    from parity_common import run_backend
    W, H = 300, 170
    rs = make_room_scene(60_000, W, H, 3, views=3, seed=4, knn=knn_mean_dist2_host)
    for v in (0, 2):
        fo, go = run_backend(oracle, rs.view(v), None, 1, 15, 0)
        op, depth = fo["opacity"][0], fo["depth"][0]
        seen = op > 0.3
        assert float(seen.float().mean()) > 0.5
        rel = ((depth / op.clamp_min(1e-6))[seen] - rs.targets[v][1][seen]).abs() / rs.targets[v][1][seen]
        assert float(rel.median()) < 5e-3, float(rel.median())
        # a surface map: (nearly) every visible Gaussian receives a gradient — unlike the i.i.d. volume, where saturation
        # leaves 2 % of them with one
        vis = fo["radii"] > 0
        live = go["dL_dmeans2D"].abs().sum(1) > 0
        assert float(live.sum()) > 0.9 * float(vis.sum())
        colour_err = (fo["color"] / op.clamp_min(1e-6) - rs.targets[v][0])[:, seen].abs().mean()
        assert float(colour_err) < 0.05, float(colour_err)
        oracle.release(fo["geom"])
