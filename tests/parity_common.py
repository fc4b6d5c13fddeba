"""Shared helpers of the parity tests: run the same seeded scene through the CPU oracle
(oracle/oracle_C.py) and through the HIP library (online_lang_splatting_amd/_C.py) at the
`_C` level, i.e. with the reference's own positional signatures (DGR/rasterize_points.h)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from online_lang_splatting_amd.scene import make_scene  # noqa: E402

EMPTY = torch.empty(0)


def fwd_args(sc, dev=None, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0, prefiltered=False,
             debug=False):
    """Positional args of rasterize_language_gaussians / rasterize_gaussians (minus `language` for F == 0)."""
    cam = sc.camera

    def mv(t):
        if t is None:
            return torch.empty(0, device=dev) if dev is not None else torch.empty(0)
        return t.to(dev) if dev is not None else t
    shs = None if colors_precomp is not None else sc.shs
    scales = None if cov3D_precomp is not None else sc.scales
    rots = None if cov3D_precomp is not None else sc.rotations
    a = [mv(sc.bg), mv(sc.means3D), mv(colors_precomp)]
    if sc.F > 0:
        a.append(mv(sc.language))
    a += [mv(sc.opacities), mv(scales), mv(rots), scale_modifier, mv(cov3D_precomp), mv(cam.world_view_transform),
          mv(cam.full_proj_transform), mv(cam.projection_matrix), cam.tanfovx, cam.tanfovy, cam.height, cam.width,
          mv(shs), sc.sh_degree, mv(cam.camera_center), prefiltered, debug]
    return a


def run_backend(C, sc, dev=None, seed=0, tile=15, mode=0, binning=1, **kw):
    """Forward + backward through backend module C (oracle_C or the product _C).  Returns (fwd dict, grads dict).
    `binning` (product only; the oracle always bins like the reference): 1 = exact ellipse lists (the
    product's default), 0 = the reference's bounding-square lists."""
    C.TILE = tile
    C.BWD_MODE = mode
    C.BINNING = binning
    F = sc.F
    a = fwd_args(sc, dev, **kw)
    if F > 0:
        R, color, lang, radii, geom, binb, img, depth, opac, nt = C.rasterize_language_gaussians(*a)
    else:
        R, color, radii, geom, binb, img, depth, opac, nt = C.rasterize_gaussians(*a)
        lang = None
    fwd = dict(R=R, color=color, language=lang, radii=radii, depth=depth, opacity=opac, n_touched=nt, geom=geom,
               binning=binb, img=img)
    dc, dl, dd = sc.cotangents(seed)
    mv = (lambda t: None if t is None else t.to(dev)) if dev is not None else (lambda t: t)
    off = 1 if F > 0 else 0
    # backward positional args (DGR/rasterize_points.h): bg, means3D, radii, colors, [language], scales, rotations,
    # scale_modifier, cov3D_precomp, view, proj, proj_raw, tanx, tany, dL_dcolor, [dL_dlang], dL_ddepth, sh, degree,
    # campos, geom, R, binning, img, debug
    b = [a[0], a[1], radii, a[2]]
    if F > 0:
        b.append(a[3])
    b += [a[4 + off], a[5 + off], a[6 + off], a[7 + off], a[8 + off], a[9 + off], a[10 + off], a[11 + off],
          a[12 + off], mv(dc)]
    if F > 0:
        b.append(mv(dl))
    b += [mv(dd), a[15 + off], a[16 + off], a[17 + off], geom, R, binb, img, False]
    if F <= 0:
        b = list(b)
        b.insert(4, None)   # language slot
        b.insert(15, None)  # dL_dout_language slot
    grads = C.backward_all(max(F, 0), *b)
    fwd["bwd_args"] = b  # (oracle_C.backward_chain replays the per-Gaussian half on other composite-level gradients)
    return fwd, grads


def rel_err(a, b):
    """max |a-b| / max|b| (scale-relative) and the plain max abs error."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.numel() == 0:
        return 0.0, 0.0
    err = (a - b).abs().max().item()
    return err / (b.abs().max().item() + 1e-30), err


# north_star / SURVEY.md section 7: "at least 99.99 % of elements within 1e-4 rel (+1e-6 abs), bounded outliers".
# The absolute term is 1e-6 of the tensor's largest magnitude (gradients span many decades; an absolute 1e-6
# would be meaningless for tensors of order 1e-9).
ELEM_RTOL = 1e-4
ELEM_ATOL_REL = 1e-6
ELEM_MIN_FRACTION = 0.9999


def elementwise_report(a, b, rtol=ELEM_RTOL, atol_rel=ELEM_ATOL_REL):
    """Per-ELEMENT comparison of `a` (tested) with `b` (oracle).
    Returns dict(n, frac_within, worst, worst_abs, max_ref): frac_within = fraction of elements with
    |a-b| <= rtol*|b| + atol_rel*max|b|; worst = the largest |a-b| / (|b| + atol_rel*max|b| / rtol), i.e. the
    worst element's error in units of the same mixed tolerance (worst <= rtol <=> every element passes)."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    if a.numel() == 0:
        return dict(n=0, frac_within=1.0, worst=0.0, worst_abs=0.0, max_ref=0.0)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = b.abs().max().item()
    diff = (a - b).abs()
    if not bool(torch.isfinite(diff).all()):
        return dict(n=a.numel(), frac_within=0.0, worst=float("inf"), worst_abs=float("inf"), max_ref=scale)
    floor = atol_rel * scale
    within = diff <= rtol * b.abs() + floor
    worst = (diff / (b.abs() + floor / rtol + 1e-300)).max().item()
    return dict(n=a.numel(), frac_within=within.double().mean().item(), worst=worst,
                worst_abs=diff.max().item(), max_ref=scale)


def assert_elementwise(a, b, name, worst_bound, log=None, min_fraction=ELEM_MIN_FRACTION, allow_outliers=0):
    """Asserts the element-wise criterion and a bound on the worst element (in relative units, see
    elementwise_report); prints both (pytest -s / the failure message) and appends them to `log`.
    allow_outliers: that many elements may sit outside the band whatever the fraction says.  Default 0: the
    99.99 % criterion stands as written for every composite-level comparison.  Only the per-Gaussian CHAIN comparisons opt in
    (allow_outliers=2 at their call sites): in tensors of fewer than 2 * 10^4 elements 99.99 % means "none or one", and the
    chain amplifies summation-order noise behind the inverse of the 2D covariance; the worst-element bound applies to the
    outliers all the same."""
    r = elementwise_report(a, b)
    line = (f"{name:24s} n={r['n']:>9d} within={100.0 * r['frac_within']:.5f}% worst_rel={r['worst']:.3e} "
            f"worst_abs={r['worst_abs']:.3e} max|ref|={r['max_ref']:.3e}")
    print(line)
    if log is not None:
        log.append(dict(name=name, **r))
    n_out = int(round((1.0 - r["frac_within"]) * r["n"]))
    assert r["frac_within"] >= min_fraction or n_out <= allow_outliers, "element-wise criterion: " + line
    assert r["worst"] <= worst_bound, f"worst element above {worst_bound:g}: " + line
    return r


COMPOSITE_KEYS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dlanguage", "dL_ddepths")


def ordered_backward(hip, sc, fwd, seed, tile, mode, dev=None, cotangents=None, condition=False, **kw):
    """Composite-level gradients of the ORDERED kernel (olsr_debug_backward_ordered: the composite backward in the reference's own
    association, csrc/k_render_bwd_ordered.hip) on the state buffers `fwd` (run_backend's forward dict of the product);
    condition=True: their condition A instead (the same sums over magnitudes).  **kw as run_backend / fwd_args take them."""
    from online_lang_splatting_amd import _abi
    from online_lang_splatting_amd._lib import check, lib
    import ctypes as C
    dev = torch.device(dev or "cuda:0")
    F = max(sc.F, 0)
    a = fwd_args(sc, dev, **kw)
    off = 1 if F > 0 else 0
    bg, means3D, colors = a[0], a[1], a[2]
    language = a[3] if F > 0 else None
    opacity, scales, rotations, scale_modifier, cov3D = a[3 + off], a[4 + off], a[5 + off], a[6 + off], a[7 + off]
    view, proj, proj_raw, tanx, tany, H, W, sh, degree, campos = a[8 + off:18 + off]
    s, keep = hip._scene(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D, view, proj,
                         proj_raw, tanx, tany, H, W, sh, degree, campos, False, False,
                         cfg=(tile, mode, _abi.BINNING_ELLIPSE))
    dc, dl, dd = cotangents if cotangents is not None else sc.cotangents(seed)
    dc, dl, dd = [None if t is None else t.to(dev).contiguous() for t in (dc, dl, dd)]
    P, R = sc.P, int(fwd["R"])
    f32 = dict(dtype=torch.float32, device=dev)
    L = lib()
    scratch = torch.empty(L.olsr_debug_backward_ordered_scratch_bytes(R, F), dtype=torch.uint8, device=dev)
    g = dict(dL_dmeans2D=torch.empty(P, 3, **f32), dL_dconic=torch.empty(P, 2, 2, **f32), dL_dopacity=torch.empty(P, 1, **f32),
             dL_dcolors=torch.empty(P, 3, **f32), dL_dlanguage=torch.empty(P, F, **f32), dL_ddepths=torch.empty(P, 1, **f32))
    p = lambda t: t.data_ptr() if t is not None and t.numel() > 0 else None  # noqa: E731
    check(L.olsr_debug_backward_ordered(
        C.byref(s), fwd["geom"].data_ptr(), R, fwd["binning"].data_ptr(), fwd["img"].data_ptr(), p(dc), p(dl), p(dd),
        scratch.data_ptr(), p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]), p(g["dL_dcolors"]),
        p(g["dL_dlanguage"]), p(g["dL_ddepths"]), 1 if condition else 0,
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize(dev)
    return g


def same_bits(a, b):
    a, b = a.detach().cpu().float().reshape(-1), b.detach().cpu().float().reshape(-1)
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


def assert_ordered_equals_oracle(go, gord, where=""):
    for k in COMPOSITE_KEYS:
        if go[k].numel():
            a, b = gord[k].cpu().reshape(-1), go[k].reshape(-1)
            if not same_bits(a, b):
                bad = (a != b) & ~(a.isnan() & b.isnan())
                i = int(bad.nonzero()[0])
                raise AssertionError(f"{where}{k}: {int(bad.sum())} of {a.numel()} elements differ from the oracle, first at {i}: "
                                     f"{float(a[i])!r} vs {float(b[i])!r}")




def assert_rounding_only(gfast, gord, gcond, k_bound=None, log=None, name=""):
    """The fast composite backward differs from the reference's association BY ROUNDING ONLY: every element of every
    composite-level gradient lies within K x 2^-24 x A of the ordered kernel's, A = the element's condition (the same sums
    over magnitudes, ordered_backward(condition=True)).  Returns / logs the largest K per tensor; k_bound asserts it."""
    eps = 2.0 ** -24
    worst = {}
    for k in COMPOSITE_KEYS:
        if not gord[k].numel():
            continue
        a, b, c = (t.detach().double().cpu().reshape(-1) for t in (gfast[k], gord[k], gcond[k]))
        d = (a - b).abs()
        assert bool(torch.isfinite(d).all()), k
        # (an element whose condition is zero has no term at all: both kernels must give exactly zero there)
        zero = c == 0
        assert bool((d[zero] == 0).all()), f"{name}{k}: a difference where no term contributes"
        K = (d[~zero] / (eps * c[~zero])).max().item() if bool((~zero).any()) else 0.0
        worst[k] = K
        if log is not None:
            log.append(dict(name=f"{name}rounding_only:{k}", K_max=K, n=int(a.numel())))
    line = "rounding-only K: " + " ".join(f"{k[3:]}={v:.1f}" for k, v in worst.items())
    print(line)
    if k_bound is not None:
        assert all(v <= k_bound for v in worst.values()), line
    return worst
