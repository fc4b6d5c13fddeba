"""CPU checks around row f4 (no GPU): the DGR-D oracle composition (tests/dgrd_oracle.py) and the Python surface of
the disentangled shim package."""
import inspect

import pytest
import torch

import dgrd_oracle as D
from test_gpu_disentangled import _cot, _scene2


def test_radii_rule_and_language_only_gradients():
    s = _scene2(1500, 173, 131, 3, 12, border=True)
    fo, saved = D.forward(s)
    r1, r2 = fo["raw_radii"]
    only2, only1 = (r1 < 0) & (r2 > 0), (r2 < 0) & (r1 > 0)
    assert int(only2.sum()) > 0 and int(only1.sum()) > 0
    # DGR-D forward.cu:391-431: both radii are written as soon as one square covers a tile
    assert bool((fo["radii"][only2] > 0).all()) and bool((fo["radii_lang"][only1] > 0).all())
    assert bool(((fo["radii"] > 0) == (fo["radii_lang"] > 0)).all())
    # a set that covers no tile emits nothing and is touched by no pixel
    assert int(fo["n_touched"][only2].sum()) == 0 and int(fo["n_touched_lang"][only1].sum()) == 0
    dc, dl, dd = _cot(s, 3, 12)
    g = D.backward(s, saved, torch.zeros_like(dc), dl, torch.zeros_like(dd))
    for k in ("means2D", "means3D", "rho", "theta", "scales", "rotations", "opacities", "sh", "colors"):
        assert float(g[k].abs().max()) == 0.0, k  # computeCov2DCUDA_no_tau: the language set moves no mean, no pose
    for k in ("language", "opacities_lang", "scales_lang", "rotations_lang"):
        assert float(g[k].abs().max()) > 0.0, k
    D.release(saved)


def test_disentangled_shim_surface():
    import diff_gaussian_rasterization_disentangle as pkg
    from online_lang_splatting_amd import disentangled
    assert pkg.LanguageGaussianRasterizer is disentangled.LanguageGaussianRasterizer
    # argument order of DGR-D/diff_gaussian_rasterization/__init__.py:528-545
    names = list(inspect.signature(pkg.LanguageGaussianRasterizer.forward).parameters)
    assert names == ["self", "means3D", "means2D", "opacities", "opacities_lang", "shs", "colors_precomp",
                     "language_precomp", "scales", "scales_lang", "rotations", "rotations_lang", "cov3D_precomp",
                     "cov3D_precomp_lang", "theta", "rho"]
    assert pkg.GaussianRasterizationSettings._fields[:5] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg")
    rast = pkg.LanguageGaussianRasterizer(raster_settings=None)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(m, m, m[:, :1], m[:, :1], scales=m, rotations=torch.zeros(4, 4), scales_lang=m,
             rotations_lang=torch.zeros(4, 4), language_precomp=m)
    with pytest.raises(Exception, match="for language"):
        rast(m, m, m[:, :1], m[:, :1], colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), language_precomp=m)
    assert disentangled.TILE == 16


# ---- DGR-D's language loops, restated line by line (numpy, one tile at a time, one array lane per thread of the
# 16x16 block) and run on the oracle's own lists: pins the claim of tests/dgrd_oracle.py and disentangled.py that the
# language rasterizer's reference-mode backward with zero colour / depth cotangents IS DGR-D's language loop.
def _dgrd_language_loops(xy, con_o, feat, point_list, ranges, W, H, dL_dpix_F):
    import numpy as np
    f32 = np.float32
    P, F = feat.shape
    gx, gy = (W + 15) // 16, (H + 15) // 16
    out_L = np.zeros((F, H, W), f32)
    out_T = np.ones((H, W), f32)
    n_contrib = np.zeros((H, W), np.int64)
    n_touched = np.zeros(P, np.int64)
    dconic = np.zeros((P, 3), np.float64)
    dopac = np.zeros(P, np.float64)
    dlang = np.zeros((P, F), np.float64)
    ty, tx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    for t in range(gx * gy):
        bx, by = t % gx, t // gx
        px, py = (bx * 16 + tx).reshape(-1), (by * 16 + ty).reshape(-1)  # thread_rank = ty * 16 + tx
        inside = (px < W) & (py < H)
        pxf, pyf = px.astype(f32), py.astype(f32)
        r0, r1 = int(ranges[2 * t]), int(ranges[2 * t + 1])
        # forward, DGR-D forward.cu:562-633
        T = np.ones(256, f32)
        done = ~inside
        L = np.zeros((256, F), f32)
        contributor = np.zeros(256, np.int64)
        last = np.zeros(256, np.int64)
        for i in range(r0, r1):
            if done.all():
                break
            g = int(point_list[i])
            act = ~done
            contributor[act] += 1
            dx, dy = xy[g, 0] - pxf, xy[g, 1] - pyf
            co = con_o[g]
            power = f32(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
            alpha = np.minimum(f32(0.99), co[3] * np.exp(power, dtype=f32))
            test_T = T * (f32(1) - alpha)
            ok = act & ~(power > 0) & ~(alpha < f32(1.0 / 255.0))
            term = ok & (test_T < f32(0.0001))
            done |= term
            c = ok & ~term
            L[c] += feat[g][None, :] * alpha[c, None] * T[c, None]
            n_touched[g] += int((c & (test_T > 0.5)).sum())
            T[c] = test_T[c]
            last[c] = contributor[c]
        ins = inside
        out_T[py[ins], px[ins]] = T[ins]
        n_contrib[py[ins], px[ins]] = last[ins]
        for ch in range(F):
            out_L[ch, py[ins], px[ins]] = L[ins, ch]
        # backward, DGR-D backward.cu:1337-1428
        Tb = np.where(inside, T, f32(0)).astype(f32)  # T_final_lang
        last_contributor = np.where(inside, last, 0)
        contributor = np.full(256, r1 - r0, np.int64)
        dLF = np.zeros((256, F), f32)
        dLF[ins] = dL_dpix_F[:, py[ins], px[ins]].T
        accum = np.zeros((256, F), f32)
        last_alpha = np.zeros(256, f32)
        last_feat = np.zeros((256, F), f32)
        done_b = ~inside
        for i in range(r1 - 1, r0 - 1, -1):
            g = int(point_list[i])
            contributor = np.where(done_b, contributor, contributor - 1)
            skip = done_b | (contributor >= last_contributor)
            dx, dy = xy[g, 0] - pxf, xy[g, 1] - pyf
            co = con_o[g]
            power = f32(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
            skip |= power > 0
            G = np.exp(power, dtype=f32)
            alpha = np.minimum(f32(0.99), co[3] * G)
            skip |= alpha < f32(1.0 / 255.0)
            if skip.all():  # skip_counter_lang == BLOCK_SIZE
                continue
            Tb = np.where(skip, Tb, Tb / (f32(1) - alpha)).astype(f32)
            dch = alpha * Tb
            # NOT skip-guarded (:1385-1393)
            accum = last_alpha[:, None] * last_feat + (f32(1) - last_alpha)[:, None] * accum
            last_feat = np.broadcast_to(feat[g][None, :], (256, F)).astype(f32)
            dL_dalpha = ((feat[g][None, :] - accum) * dLF).sum(1, dtype=f32)
            coll = np.where(skip[:, None], f32(0), dch[:, None] * dLF)
            dL_dalpha = dL_dalpha * Tb
            last_alpha = np.where(skip, last_alpha, alpha).astype(f32)
            dL_dG = co[3] * dL_dalpha
            gdx, gdy = G * dx, G * dy
            keep = ~skip
            dconic[g, 0] += float((f32(-0.5) * gdx * dx * dL_dG)[keep].sum(dtype=np.float64))
            dconic[g, 1] += float((f32(-0.5) * gdx * dy * dL_dG)[keep].sum(dtype=np.float64))
            dconic[g, 2] += float((f32(-0.5) * gdy * dy * dL_dG)[keep].sum(dtype=np.float64))
            dopac[g] += float((G * dL_dalpha)[keep].sum(dtype=np.float64))
            dlang[g] += coll[0].astype(np.float64)  # thread 0 of the tile only (:1423-1425)
    return out_L, out_T, n_contrib, n_touched, dconic, dopac, dlang


def test_language_loops_of_dgrd_restated_directly():
    import numpy as np
    from oracle import oracle_C as O
    from online_lang_splatting_amd import _abi
    W, H, F, P = 70, 52, 3, 260
    s = _scene2(P, W, H, F, 31)
    cam = s["camera"]
    O.TILE, O.BWD_MODE, O.FLAGS = 16, _abi.BWD_REFERENCE, 0
    zeros = torch.zeros(P, 3)
    common = (cam.world_view_transform, cam.full_proj_transform, cam.projection_matrix, cam.tanfovx, cam.tanfovy)
    R, _c, lang, radii, geom, binb, img, _d, opac, nt = O.rasterize_language_gaussians(
        s["bg"], s["means3D"], zeros, s["language"], s["opacities_lang"], s["scales_lang"], s["rotations_lang"], 1.0,
        torch.empty(0), *common, H, W, torch.empty(0), 0, cam.camera_center, False, False)
    g = torch.Generator().manual_seed(77)
    dl = torch.randn(F, H, W, generator=g) / (H * W)
    grads = O.backward_all(F, s["bg"], s["means3D"], radii, zeros, s["language"], s["scales_lang"], s["rotations_lang"],
                           1.0, torch.empty(0), *common, torch.zeros(3, H, W), dl, torch.zeros(1, H, W), torch.empty(0),
                           0, cam.camera_center, geom, R, binb, img, False)
    xy = O.get_field(geom, "means2D").view(P, 2).numpy()
    con_o = O.get_field(geom, "conic_opacity").view(P, 4).numpy()
    pl = O.get_field(geom, "point_list").numpy()
    rg = O.get_field(geom, "ranges").numpy()
    out_L, out_T, n_contrib, n_touched, dconic, dopac, dlang = _dgrd_language_loops(
        xy, con_o, s["language"].numpy(), pl, rg, W, H, dl.numpy())
    assert R > 1000 and int((n_contrib > 0).sum()) > 0.5 * W * H
    np.testing.assert_allclose(out_L, lang.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(1.0 - out_T, opac.numpy()[0], rtol=2e-5, atol=2e-6)
    assert np.array_equal(n_contrib, O.get_field(geom, "n_contrib").view(H, W).numpy())
    assert np.array_equal(n_touched, nt.numpy())

    def close(a, b, name):
        scale = float(np.abs(b).max())
        assert scale > 0, name
        assert float(np.abs(a - b).max()) <= 2e-4 * scale, (name, float(np.abs(a - b).max()), scale)
    dc = grads["dL_dconic"].numpy().astype(np.float64)
    close(dconic[:, 0], dc[:, 0, 0], "conic.x")
    close(dconic[:, 1], dc[:, 0, 1], "conic.y")
    close(dconic[:, 2], dc[:, 1, 1], "conic.w")
    close(dopac, grads["dL_dopacity"].numpy().astype(np.float64)[:, 0], "opacity")
    close(dlang, grads["dL_dlanguage"].numpy().astype(np.float64), "language")
    O.release(geom)
