"""CPU oracle with the reference's `_C` surface.  TEST INFRASTRUCTURE ONLY.

Exposes the five functions of DGR/ext.cpp:15-21 with the positional signatures of
DGR/rasterize_points.h:17-152, operating on CPU tensors through liboracle.so
(oracle/oracle.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product never does.

Extra knobs that the reference fixes at compile time are module globals:
TILE (CR/config.h:17-18) and BWD_MODE (reference / exact).
"""
import ctypes as C
import os
import subprocess
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from online_lang_splatting_amd import _abi  # noqa: E402  (struct declarations only)

TILE = 15
BWD_MODE = _abi.BWD_REFERENCE
FLAGS = 0  # _abi.FLAG_* for the forward

_lib = None
_states = {}
_next_id = [1]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "olsr.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


_variant = "default"


def use_variant(name):
    """Switch the library behind this module: "default" (the parity oracle), "libm_exp" or "contract_fast"
    (sensitivity variants built by `make -C oracle variants`; scripts/cuda_sensitivity.py only)."""
    global _lib, _variant
    assert name in ("default", "libm_exp", "contract_fast"), name
    if name != _variant:
        _states.clear()
        _lib, _variant = None, name


def lib():
    global _lib
    if _lib is None:
        so = build()
        if _variant != "default":
            subprocess.check_call(["make", "-C", _HERE, "-s", "variants"])
            so = os.path.join(_HERE, {"libm_exp": "liboracle_libmexp.so", "contract_fast": "liboracle_contract.so"}[_variant])
        L = C.CDLL(so)
        L.oracle_set_record.argtypes = [C.c_int]
        L.oracle_variant.restype = C.c_char_p
        L.oracle_create.restype = C.c_void_p
        L.oracle_destroy.argtypes = [C.c_void_p]
        vp = C.c_void_p
        L.oracle_forward.argtypes = [vp, C.POINTER(_abi.OlsrScene)] + [vp] * 7
        L.oracle_forward.restype = C.c_int
        L.oracle_backward.argtypes = [vp, C.POINTER(_abi.OlsrScene)] + [vp] * 16
        L.oracle_backward.restype = C.c_int
        L.oracle_backward_chain.argtypes = [vp, C.POINTER(_abi.OlsrScene)] + [vp] * 11
        L.oracle_backward_chain.restype = C.c_int
        L.oracle_mark_visible.argtypes = [C.c_int32, vp, vp, vp, vp]
        L.oracle_mark_visible.restype = C.c_int
        L.oracle_get_field.argtypes = [vp, C.c_char_p, vp]
        L.oracle_get_field.restype = C.c_int64
        L.oracle_knn_mean_dist2.argtypes = [C.c_int32, vp, vp]
        L.oracle_knn_mean_dist2.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_get_threads.restype = C.c_int
        L.oracle_expf_probe.argtypes = [C.c_float]
        L.oracle_expf_probe.restype = C.c_float
        _lib = L
        L.oracle_set_threads(usable_cpus())  # (OpenMP's default is every visible CPU, whatever the container may use)
    return _lib


def usable_cpus():
    """CPUs this process may really use: the scheduler affinity capped by the container's CPU quota (cgroup v2 cpu.max or
    v1 cfs quota).  The GPU box shows 256 hardware threads and grants 16: 256 OpenMP threads there run a frame 7x slower
    than 16 do."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def set_threads(n):
    """OpenMP threads of the following oracle calls; returns the previous setting."""
    prev = lib().oracle_get_threads()
    lib().oracle_set_threads(int(n))
    return prev


class _State:
    def __init__(self):
        self.h = lib().oracle_create()

    def __del__(self):
        try:
            lib().oracle_destroy(self.h)
        except Exception:
            pass


def _f(t):
    return t.contiguous().float() if t is not None and t.numel() > 0 else None


def _scene(bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
           projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug, F):
    keep = [_f(x) for x in (bg, means3D, sh, colors, language, opacity, scales, rotations, cov3D_precomp,
                            viewmatrix, projmatrix, projmatrix_raw, campos)]
    bg_, m_, sh_, col_, lang_, op_, sc_, rot_, cov_, v_, p_, pr_, cp_ = keep
    M = sh_.shape[1] if sh_ is not None else 0
    s = _abi.make_scene(P=means3D.shape[0], D=degree, M=M, F=F, width=W, height=H, tile=TILE,
                        prefiltered=prefiltered, debug=debug, bwd_mode=BWD_MODE, tan_fovx=tan_fovx,
                        tan_fovy=tan_fovy, scale_modifier=scale_modifier, flags=FLAGS, background=bg_, means3D=m_, shs=sh_,
                        colors_precomp=col_, language_precomp=lang_, opacities=op_, scales=sc_, rotations=rot_,
                        cov3D_precomp=cov_, viewmatrix=v_, projmatrix=p_, projmatrix_raw=pr_, cam_pos=cp_)
    return s, keep


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def _forward(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered,
             debug):
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    P = means3D.shape[0]
    s, keep = _scene(bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                     viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos,
                     prefiltered, debug, F)
    out_color = torch.zeros(3, H, W)
    out_lang = torch.zeros(max(F, 0), H, W)
    out_depth = torch.zeros(1, H, W)
    out_opacity = torch.zeros(1, H, W)
    radii = torch.zeros(P, dtype=torch.int32)
    n_touched = torch.zeros(P, dtype=torch.int32)
    R = C.c_int32(0)
    st = _State()
    sid = _next_id[0]
    _next_id[0] += 1
    _states[sid] = st
    _check(lib().oracle_forward(st.h, C.byref(s), out_color.data_ptr(), out_lang.data_ptr(), out_depth.data_ptr(),
                                out_opacity.data_ptr(), radii.data_ptr(), n_touched.data_ptr(),
                                C.addressof(R)), "forward")
    geom = torch.tensor([sid], dtype=torch.int64).view(torch.uint8)
    empty = torch.empty(0, dtype=torch.uint8)
    return R.value, out_color, out_lang, radii, geom, empty, empty.clone(), out_depth, out_opacity, n_touched


def state_of(geomBuffer):
    return _states[int(geomBuffer.view(torch.int64)[0])]


def release(geomBuffer):
    _states.pop(int(geomBuffer.view(torch.int64)[0]), None)


_FIELD_DT = {"clamped": torch.uint8, "tiles_touched": torch.int32, "point_offsets": torch.int32,
             "point_list": torch.int32, "keys": torch.int64, "ranges": torch.int32, "n_contrib": torch.int32,
             "contrib_mask": torch.int32}


def get_field(geomBuffer, name):
    st = state_of(geomBuffer)
    n = lib().oracle_get_field(st.h, name.encode(), None)
    if n < 0:
        raise KeyError(name)
    t = torch.zeros(n, dtype=_FIELD_DT.get(name, torch.float32))
    lib().oracle_get_field(st.h, name.encode(), t.data_ptr())
    return t


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                        campos, prefiltered, debug):
    """RasterizeGaussiansCUDA, DGR/rasterize_points.cu:35-123."""
    r = _forward(0, bg, means3D, colors, None, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                 viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                 campos, prefiltered, debug)
    R, color, _lang, radii, geom, binning, img, depth, opacity_out, n_touched = r
    return R, color, radii, geom, binning, img, depth, opacity_out, n_touched


def rasterize_language_gaussians(bg, means3D, colors, language, opacity, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy,
                                 image_height, image_width, sh, degree, campos, prefiltered, debug):
    """RasterizeLanguageGaussiansCUDA, DGR/rasterize_points.cu:125-241."""
    F = language.shape[1]
    return _forward(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                    viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh,
                    degree, campos, prefiltered, debug)


def _backward(F, bg, means3D, radii, colors, language, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
              projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, dL_dout_language, dL_dout_depth, sh,
              degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, keep_internal=False, composite=None):
    """composite: dict with dL_dmeans2D [P,3], dL_dconic [P,2,2], dL_dcolors [P,3], dL_ddepths [P,1] — run only the
    per-Gaussian chain (oracle_backward_chain) on THESE composite-level gradients and return its six outputs."""
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1], dL_dout_color.shape[2]
    st = state_of(geomBuffer)
    # opacities are not an input of the reference backward (they live in conic_opacity); the
    # oracle keeps them in its state, so pass a dummy non-null pointer to satisfy check_scene.
    dummy_op = torch.zeros(max(P, 1))
    s, keep = _scene(bg, means3D, colors, language, dummy_op, scales, rotations, scale_modifier, cov3D_precomp,
                     viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, False,
                     debug, F)
    M = s.M
    g = dict(
        dL_dmeans2D=torch.zeros(P, 3), dL_dconic=torch.zeros(P, 2, 2), dL_dopacity=torch.zeros(P, 1),
        dL_dcolors=torch.zeros(P, 3), dL_dlanguage=torch.zeros(P, max(F, 0)), dL_ddepths=torch.zeros(P, 1),
        dL_dmeans3D=torch.zeros(P, 3), dL_dcov3D=torch.zeros(P, 6), dL_dsh=torch.zeros(P, M, 3),
        dL_dscales=torch.zeros(P, 3), dL_drotations=torch.zeros(P, 4), dL_dtau=torch.zeros(P, 6))
    dc = dL_dout_color.contiguous().float()
    dl = dL_dout_language.contiguous().float() if F > 0 else torch.zeros(1)
    dd = dL_dout_depth.contiguous().float()
    rad = radii.contiguous().to(torch.int32)
    if composite is not None:
        cin = {k: composite[k].detach().cpu().contiguous().float() for k in
               ("dL_dmeans2D", "dL_dconic", "dL_dcolors", "dL_ddepths")}
        assert cin["dL_dmeans2D"].numel() == 3 * P and cin["dL_dconic"].numel() == 4 * P
        assert cin["dL_dcolors"].numel() == 3 * P and cin["dL_ddepths"].numel() == P
        _check(lib().oracle_backward_chain(
            st.h, C.byref(s), rad.data_ptr(), cin["dL_dmeans2D"].data_ptr(), cin["dL_dconic"].data_ptr(),
            cin["dL_dcolors"].data_ptr(), cin["dL_ddepths"].data_ptr(), g["dL_dmeans3D"].data_ptr(),
            g["dL_dcov3D"].data_ptr(), g["dL_dsh"].data_ptr(), g["dL_dscales"].data_ptr(),
            g["dL_drotations"].data_ptr(), g["dL_dtau"].data_ptr()), "backward_chain")
        return {k: g[k] for k in CHAIN_KEYS}
    _check(lib().oracle_backward(
        st.h, C.byref(s), rad.data_ptr(), dc.data_ptr(), dl.data_ptr(), dd.data_ptr(),
        g["dL_dmeans2D"].data_ptr(), g["dL_dconic"].data_ptr(), g["dL_dopacity"].data_ptr(),
        g["dL_dcolors"].data_ptr(), g["dL_dlanguage"].data_ptr(), g["dL_ddepths"].data_ptr(),
        g["dL_dmeans3D"].data_ptr(), g["dL_dcov3D"].data_ptr(), g["dL_dsh"].data_ptr(),
        g["dL_dscales"].data_ptr(), g["dL_drotations"].data_ptr(), g["dL_dtau"].data_ptr()), "backward")
    return g


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depths, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 debug):
    """RasterizeGaussiansBackwardCUDA, DGR/rasterize_points.cu:243-331."""
    g = _backward(0, bg, means3D, radii, colors, None, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, None, dL_dout_depths, sh, degree,
                  campos, geomBuffer, R, binningBuffer, imageBuffer, debug)
    return (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dopacity"], g["dL_dmeans3D"], g["dL_dcov3D"], g["dL_dsh"],
            g["dL_dscales"], g["dL_drotations"], g["dL_dtau"])


def rasterize_language_gaussians_backward(bg, means3D, radii, colors, language, scales, rotations, scale_modifier,
                                          cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy,
                                          dL_dout_color, dL_dout_language, dL_dout_depth, sh, degree, campos,
                                          geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeLanguageGaussiansBackwardCUDA, DGR/rasterize_points.cu:333-455."""
    F = language.shape[1]
    g = _backward(F, bg, means3D, radii, colors, language, scales, rotations, scale_modifier, cov3D_precomp,
                  viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, dL_dout_language,
                  dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug)
    return (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dlanguage"], g["dL_dopacity"], g["dL_dmeans3D"],
            g["dL_dcov3D"], g["dL_dsh"], g["dL_dscales"], g["dL_drotations"], g["dL_dtau"])


CHAIN_KEYS = ("dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dtau")


def backward_chain(F, composite, *args, **kw):
    """The per-Gaussian chain alone on caller-provided composite-level gradients; args as backward_all."""
    return _backward(F, *args, composite=composite, **kw)


def backward_all(F, *args, **kw):
    """Same as the two backward functions but returns every gradient incl. dL_dconic / dL_ddepths."""
    return _backward(F, *args, **kw)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, DGR/rasterize_points.cu:457-476."""
    P = means3D.shape[0]
    present = torch.zeros(P, dtype=torch.bool)
    m, v, p = means3D.contiguous().float(), viewmatrix.contiguous().float(), projmatrix.contiguous().float()
    if P:
        _check(lib().oracle_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), present.data_ptr()),
               "mark_visible")
    return present


def distCUDA2(points):
    """simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:15-26): brute-force exact 3-NN on the CPU."""
    pts = points.contiguous().float()
    out = torch.zeros(pts.shape[0], dtype=torch.float32)
    if pts.shape[0]:
        _check(lib().oracle_knn_mean_dist2(pts.shape[0], pts.data_ptr(), out.data_ptr()), "knn")
    return out


def expf(x):
    return lib().oracle_expf_probe(float(x))


def cov3d_backward(scales, scale_modifier, rotations, dL_dcov3D):
    """computeCov3D backward alone (CR/backward.cu:350-413): (dL_dscales [P,3], dL_drotations [P,4])."""
    sc, ro, dc = (t.contiguous().float() for t in (scales, rotations, dL_dcov3D))
    P = sc.shape[0]
    ds, dr = torch.zeros(P, 3), torch.zeros(P, 4)
    L = lib()
    L.oracle_cov3d_backward.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(L.oracle_cov3d_backward(P, sc.data_ptr(), float(scale_modifier), ro.data_ptr(), dc.data_ptr(), ds.data_ptr(),
                                   dr.data_ptr()), "cov3d_backward")
    return ds, dr
