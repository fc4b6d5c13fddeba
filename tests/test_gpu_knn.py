"""olsr_knn_mean_dist2 (drop-in for simple_knn._C.distCUDA2) against the brute-force oracle.
The result is defined by exact 3-NN with one pinned distance expression, so equality is bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cloud(P, seed, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return torch.rand(P, 3, generator=g) * 4 - 2
    if kind == "clusters":  # dense blobs + empty space: pruning has to reach far boxes
        c = torch.randn(16, 3, generator=g) * 5
        return c[torch.randint(0, 16, (P,), generator=g)] + torch.randn(P, 3, generator=g) * 0.05
    if kind == "plane":  # degenerate bounding box on one axis (depth-map back-projection of a wall)
        p = torch.rand(P, 3, generator=g)
        p[:, 2] = 1.5
        return p
    if kind == "duplicates":
        p = torch.rand(max(P // 3, 1), 3, generator=g)
        return p.repeat(3, 1)[:P].contiguous()
    raise ValueError(kind)


@pytest.mark.parametrize("P,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (4, "uniform"), (63, "uniform"),
                                    (64, "uniform"), (65, "clusters"), (4097, "uniform"), (20000, "clusters"),
                                    (10000, "plane"), (3000, "duplicates"), (30000, "uniform")])
def test_equals_brute_force(hip, oracle, P, kind):
    from simple_knn._C import distCUDA2  # the reference's import line (gaussian_model.py:18)
    pts = _cloud(P, P, kind)
    got = distCUDA2(pts.to(DEV)).cpu()
    exp = oracle.distCUDA2(pts)
    assert got.shape == (P,) and got.dtype == torch.float32
    assert torch.equal(got, exp), f"max abs diff {float((got - exp).abs().max())}"
    if P >= 4:
        assert bool(torch.isfinite(got).all())
    else:
        assert bool((got > 1e37).all())  # missing neighbours count as FLT_MAX, as in the reference


def test_large_cloud_against_kdtree(hip):
    """500 k points (too many for the O(P^2) oracle): scipy's exact k-d tree supplies the neighbours, the
    distances are recomputed with the pinned fp32 expression."""
    from scipy.spatial import cKDTree
    from online_lang_splatting_amd.simple_knn import distCUDA2
    P = 500_000
    g = torch.Generator().manual_seed(5)
    pts = torch.rand(P, 3, generator=g) * torch.tensor([8.0, 5.0, 3.0])
    got = distCUDA2(pts.to(DEV)).cpu().numpy()
    x = pts.numpy()
    _, idx = cKDTree(x.astype(np.float64)).query(x.astype(np.float64), k=6)   # self + 5: margin for fp32 ties
    d = x[idx[:, 1:]] - x[:, None, :]
    d2 = np.float32(d[..., 0] * d[..., 0])
    d2 = np.float32(np.float64(d[..., 1]) * np.float64(d[..., 1]) + np.float64(d2))     # fma(dy, dy, dx*dx)
    d2 = np.float32(np.float64(d[..., 2]) * np.float64(d[..., 2]) + np.float64(d2))     # fma(dz, dz, .)
    d2.sort(axis=1)
    exp = ((d2[:, 0] + d2[:, 1]) + d2[:, 2]) / np.float32(3.0)
    assert np.array_equal(got, exp.astype(np.float32))


def test_cpu_tensor_is_refused(hip):
    from online_lang_splatting_amd.simple_knn import distCUDA2
    with pytest.raises(RuntimeError, match="GPU"):
        distCUDA2(torch.rand(10, 3))
    assert distCUDA2(torch.zeros(0, 3, device=DEV)).shape == (0,)
