"""The reference's `_C` extension surface over the MI355X-native library.

Same five functions as DGR/ext.cpp:15-21, same positional signatures and return tuples as
DGR/rasterize_points.h:17-152 — so the autograd layer (rasterizer.py) reads like the
reference's.  Tensors are allocated with torch (device memory and the current HIP stream are
plumbing); all arithmetic happens behind the C-ABI of include/olsr.h in libolsr.so.

Two bindings of the same library entry points live here.  The five `_C` functions go through the compiled torch
extension `_olsr_torch` (csrc/olsr_torch.cpp, built in-tree by build.py) — argument checks, allocation and the call
in C++, like the reference's ext.cpp; the ctypes binding below serves the same calls when that module has not been
built (OLSR_BINDING=ctypes forces it) and everything outside the five functions (state introspection, losses, Adam,
kNN, the benchmark's async entry).  Both end in libolsr.so; neither computes anything itself.

Knobs the reference fixes at compile time (CR/config.h:15-18) are module attributes:
  TILE      logical tile edge, 15 (reference) or 16
  BWD_MODE  _abi.BWD_REFERENCE (bug-compatible, default) or _abi.BWD_EXACT (true gradient)
  BINNING   _abi.BINNING_ELLIPSE (default: exact tile lists, identical outputs) or _abi.BINNING_RECT
            (the reference's bounding-square lists, bit-identical num_rendered / point_list / n_contrib)
The number of language channels is taken from language.shape[1] (supported: 3, 15, 16, 32).
"""
import ctypes as C
import os
import threading

import torch

from . import _abi
from ._lib import check, lib

TILE = 15
BWD_MODE = _abi.BWD_REFERENCE
BINNING = _abi.BINNING_ELLIPSE
FLAGS = 0  # _abi.FLAG_* bits OR-ed into every forward (e.g. _abi.FLAG_FWD_ACCUM_MFMA); OLSR_FWD_ACCUM=mfma sets that one
if os.environ.get("OLSR_FWD_ACCUM", "") == "mfma":
    FLAGS |= _abi.FLAG_FWD_ACCUM_MFMA
elif os.environ.get("OLSR_FWD_ACCUM", "") == "weight":
    FLAGS |= _abi.FLAG_FWD_ACCUM_WEIGHT

_EMPTY = torch.empty(0)
_ext = None
_ext_checked = False


def compiled_binding():
    """The `_olsr_torch` module, or None when it is absent (not built) or OLSR_BINDING=ctypes.  OLSR_BINDING=torch
    makes its absence an error."""
    global _ext, _ext_checked
    want = os.environ.get("OLSR_BINDING", "")
    if want == "ctypes":
        return None
    if not _ext_checked:
        _ext_checked = True
        lib()  # libolsr.so first: a missing library must raise its own message, not a loader error
        try:
            from . import _olsr_torch
            _ext = _olsr_torch
        except ImportError as e:
            _ext = None
            compiled_binding.error = e
    if _ext is None and want == "torch":
        raise ImportError(f"OLSR_BINDING=torch but _olsr_torch is not importable: {getattr(compiled_binding, 'error', None)}; "
                          "build it with `python -m online_lang_splatting_amd.build`")
    return _ext


def _t(x):
    return _EMPTY if x is None else x


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _c(t):
    """contiguous fp32 view, or None for an absent tensor (0 elements -> nullptr)."""
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError("every rasterizer input must live on the GPU (got a CPU tensor): "
                           "this rasterizer has no CPU path")
    return t.contiguous() if t.dtype == torch.float32 else t.contiguous().float()


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU: this rasterizer has no CPU path")


class _Resizer:
    """resizeFunctional, DGR/rasterize_points.cu:27-33: the C side asks for bytes, we allocate a uint8 tensor and
    hand back its data pointer.  ONE ctypes trampoline serves every call (building a CFUNCTYPE object per buffer per
    call costs more than the allocation): the `user` word the library passes back selects the slot."""

    _tls = threading.local()

    def __init__(self, device):
        self.device = device
        self.t = None

    @staticmethod
    def _dispatch(user, nbytes):
        r = _Resizer._tls.slots[int(user or 0)]
        r.t = torch.empty(int(nbytes), dtype=torch.uint8, device=r.device)
        return r.t.data_ptr()

    @classmethod
    def bind(cls, *resizers):
        """Make `resizers` the targets of user words 0, 1, ... for the next library call of this thread."""
        cls._tls.slots = resizers
        return [C.c_void_p(i) for i in range(len(resizers))]


_Resizer.cb = _abi.ALLOC_FN(_Resizer._dispatch)


def current_config():
    """(TILE, BWD_MODE, BINNING) as they stand now.  The autograd layer captures this at forward time and hands it
    to the matching backward (`cfg=`), so that changing a knob between a forward and its backward — a SLAM loop
    renders 12 views before it back-propagates — cannot make the backward carve the state buffers differently."""
    return (TILE, BWD_MODE, BINNING)


def _scene(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
           projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug, cfg=None,
           flags=0):
    tile, bwd_mode, binning = cfg if cfg is not None else current_config()
    keep = [_c(x) for x in (bg, means3D, sh, colors, language, opacity, scales, rotations, cov3D_precomp, viewmatrix,
                            projmatrix, projmatrix_raw, campos)]
    bg_, m_, sh_, col_, lang_, op_, sc_, rot_, cov_, v_, p_, pr_, cp_ = keep
    M = sh_.shape[1] if sh_ is not None else 0
    s = _abi.make_scene(P=means3D.shape[0], D=degree, M=M, F=F, width=W, height=H, tile=tile,
                        prefiltered=prefiltered, debug=debug, bwd_mode=bwd_mode, tan_fovx=tan_fovx, tan_fovy=tan_fovy,
                        scale_modifier=scale_modifier, binning=binning, flags=flags, background=bg_, means3D=m_, shs=sh_, colors_precomp=col_,
                        language_precomp=lang_, opacities=op_, scales=sc_, rotations=rot_, cov3D_precomp=cov_,
                        viewmatrix=v_, projmatrix=p_, projmatrix_raw=pr_, cam_pos=cp_)
    return s, keep


def _forward(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
             projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
             prefiltered, debug, cfg=None, flags=0):
    """`cfg` (tile, bwd_mode, binning) overrides the module knobs for this call; `flags`: _abi.FLAG_*."""
    flags = int(flags) | FLAGS
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # DGR/rasterize_points.cu:159-161
    _require_gpu(means3D, "means3D")
    ext = compiled_binding()
    if ext is not None:
        tile, bwd_mode, binning = cfg if cfg is not None else current_config()
        return ext.forward(F, _t(bg), means3D, _t(colors), _t(language), _t(opacity), _t(scales), _t(rotations),
                           float(scale_modifier), _t(cov3D_precomp), _t(viewmatrix), _t(projmatrix), _t(projmatrix_raw),
                           float(tan_fovx), float(tan_fovy), int(image_height), int(image_width), _t(sh), int(degree),
                           _t(campos), bool(prefiltered), bool(debug), tile, bwd_mode, binning, int(flags))
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    with torch.cuda.device(dev):
        s, keep = _scene(F, bg, means3D, colors, language, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos,
                         prefiltered, debug, cfg=cfg, flags=flags)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        # every output is fully written by the library (no torch::full zero-fill, unlike
        # DGR/rasterize_points.cu:170-175)
        out_color = torch.empty(3, H, W, **f32)
        out_lang = torch.empty(F, H, W, **f32)
        out_depth = torch.empty(1, H, W, **f32)
        out_opacity = torch.empty(1, H, W, **f32)
        radii = torch.empty(P, **i32)
        n_touched = torch.empty(P, **i32)
        geom, binning, img = _Resizer(dev), _Resizer(dev), _Resizer(dev)
        u_geom, u_bin, u_img = _Resizer.bind(geom, binning, img)
        R = C.c_int32(0)
        check(lib().olsr_forward(C.byref(s), _Resizer.cb, u_geom, _Resizer.cb, u_bin, _Resizer.cb, u_img, out_color.data_ptr(),
                                 out_lang.data_ptr() if F > 0 else None, out_depth.data_ptr(),
                                 out_opacity.data_ptr(), radii.data_ptr() if P else None,
                                 n_touched.data_ptr() if P else None, C.byref(R), _stream(dev)))
    return R.value, out_color, out_lang, radii, geom.t, binning.t, img.t, out_depth, out_opacity, n_touched


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """RasterizeGaussiansCUDA, DGR/rasterize_points.cu:35-123."""
    R, color, _l, radii, geom, binning, img, depth, opac, n_touched = _forward(
        0, bg, means3D, colors, None, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
        debug)
    return R, color, radii, geom, binning, img, depth, opac, n_touched


def rasterize_language_gaussians(bg, means3D, colors, language, opacity, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy,
                                 image_height, image_width, sh, degree, campos, prefiltered, debug):
    """RasterizeLanguageGaussiansCUDA, DGR/rasterize_points.cu:125-241."""
    if language is None or language.dim() != 2 or language.shape[1] not in _abi.SUPPORTED_F[1:]:
        raise RuntimeError(f"language_precomp must be [P, F] with F in {_abi.SUPPORTED_F[1:]}")
    return _forward(language.shape[1], bg, means3D, colors, language, opacity, scales, rotations, scale_modifier,
                    cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, image_height,
                    image_width, sh, degree, campos, prefiltered, debug)


# ctypes twin of the compiled binding's g_rows_per_instance / g_rows_redone: [unpacked ratio, packed ratio, backwards redone]
_ROWS_PER_INSTANCE = [0.0, 0.0, 0]


def debug_rows_ratio(packed, ratio=-1.0):
    """(tests) Set (ratio >= 0) / read the rows-per-instance figure the next backward guesses its scratch size from, in
    whichever binding serves the calls.  Returns (ratio in force, backwards that had to be redone exactly so far)."""
    ext = compiled_binding()
    if ext is not None:
        return tuple(ext.debug_rows_ratio(bool(packed), float(ratio)))
    if ratio >= 0:
        _ROWS_PER_INSTANCE[1 if packed else 0] = float(ratio)
    return (_ROWS_PER_INSTANCE[1 if packed else 0], _ROWS_PER_INSTANCE[2])


_GRAD_NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dlanguage", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
               "dL_dscales", "dL_drotations", "dL_dtau", "dL_dtau_sum", "dL_dconic", "dL_ddepths")


def _backward(F, bg, means3D, radii, colors, language, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
              projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, dL_dout_language, dL_dout_depth, sh,
              degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug, want_internal=False, cfg=None,
              rows_token=0):
    """rows_token: last_forward_token() taken right after the matching forward — lets the row scratch be sized by the
    frame's exact gradient-row count (olsr_live_rows) instead of the bound of 2 / 4 rows per instance; 0: the bound."""
    _require_gpu(means3D, "means3D")
    ext = compiled_binding()
    if ext is not None:
        tile, bwd_mode, binning = cfg if cfg is not None else current_config()
        g = ext.backward(F, _t(bg), means3D, radii, _t(colors), _t(language), _t(scales), _t(rotations),
                         float(scale_modifier), _t(cov3D_precomp), _t(viewmatrix), _t(projmatrix), _t(projmatrix_raw),
                         float(tan_fovx), float(tan_fovy), dL_dout_color, _t(dL_dout_language), _t(dL_dout_depth), _t(sh),
                         int(degree), _t(campos), geomBuffer, int(R), binningBuffer, imageBuffer, bool(debug),
                         bool(want_internal), tile, bwd_mode, binning, int(rows_token))
        out = dict(zip(_GRAD_NAMES, g))
        if not want_internal:
            del out["dL_dconic"], out["dL_ddepths"]
        return out
    dev = means3D.device
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1], dL_dout_color.shape[2]
    with torch.cuda.device(dev):
        s, keep = _scene(F, bg, means3D, colors, language, None, scales, rotations, scale_modifier, cov3D_precomp,
                         viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, H, W, sh, degree, campos, False,
                         debug, cfg=cfg)
        M = s.M
        f32 = dict(dtype=torch.float32, device=dev)
        # written exactly once per row by the library: no torch::zeros (DGR/rasterize_points.cu:386-398)
        g = dict(dL_dmeans2D=torch.empty(P, 3, **f32), dL_dcolors=torch.empty(P, 3, **f32),
                 dL_dlanguage=torch.empty(P, F, **f32), dL_dopacity=torch.empty(P, 1, **f32),
                 dL_dmeans3D=torch.empty(P, 3, **f32), dL_dcov3D=torch.empty(P, 6, **f32),
                 dL_dsh=torch.empty(P, M, 3, **f32), dL_dscales=torch.empty(P, 3, **f32),
                 dL_drotations=torch.empty(P, 4, **f32), dL_dtau=torch.empty(P, 6, **f32),
                 dL_dtau_sum=torch.empty(6, **f32))
        if want_internal:
            g["dL_dconic"] = torch.empty(P, 2, 2, **f32)
            g["dL_ddepths"] = torch.empty(P, 1, **f32)
        dc, dl, dd = _c(dL_dout_color), _c(dL_dout_language), _c(dL_dout_depth)
        rad = radii.contiguous()

        def p(name):
            t = g.get(name)
            return t.data_ptr() if t is not None and t.numel() > 0 else None
        # Row scratch: one partial-gradient row per live (instance, 64-pixel slot) pair.  The forward posts their exact
        # number to the host; olsr_backward_rows waits for it when this thread is ahead of the GPU and the bound
        # L <= slots * R (two packed survivor waves per instance in the reference mode of 15x15 tiles, else four slots)
        # would cost more than 64 MB — the GPU is busy with the forward meanwhile.
        tile, bwd_mode, _binning = cfg if cfg is not None else current_config()
        packed = bwd_mode == _abi.BWD_REFERENCE and tile == 15
        # the policy of the compiled binding (csrc/olsr_torch.cpp: backward): the posted count, else a guess from the last
        # verified frame's rows per instance — launched at once, verified while the GPU works, redone exactly if too small
        L = lib()
        bound = max(int(R), 0) * (2 if packed else 4)
        rows = int(L.olsr_live_rows(int(rows_token), 1 if packed else 0))
        guessed = False
        if rows < 0 or rows > bound:
            ratio = _ROWS_PER_INSTANCE[1 if packed else 0]
            if L.olsr_live_rows_overwritten(int(rows_token)):
                rows = bound   # (the slot belongs to a later forward: no count will ever arrive)
            elif rows_token > 0 and ratio > 0 and R > 0 and L.olsr_backward_scratch_bytes(bound, F) > (64 << 20):
                rows = int(1.5 * ratio * R) + 65536          # (a multiple of 128 Ki rows: a stable size for the caching allocator)
                rows, guessed = min(bound, (rows + 131071) // 131072 * 131072), True
            else:
                rows = int(L.olsr_backward_rows(int(rows_token), 1 if packed else 0, int(R), int(F)))

        def launch(nrows):
            scratch = torch.empty(L.olsr_backward_scratch_bytes(nrows, F), dtype=torch.uint8, device=dev)
            check(L.olsr_backward(
                C.byref(s), rad.data_ptr() if P else None, geomBuffer.data_ptr(), int(R), binningBuffer.data_ptr(),
                imageBuffer.data_ptr(), _abi.ALLOC_FN(0), None, scratch.data_ptr(), nrows,
                dc.data_ptr() if dc is not None else None,
                dl.data_ptr() if dl is not None else None, dd.data_ptr() if dd is not None else None,
                p("dL_dmeans2D"), p("dL_dconic"), p("dL_dopacity"), p("dL_dcolors"), p("dL_dlanguage"), p("dL_ddepths"),
                p("dL_dmeans3D"), p("dL_dcov3D"), p("dL_dsh"), p("dL_dscales"), p("dL_drotations"), p("dL_dtau"),
                p("dL_dtau_sum"), None, None, _stream(dev)))
            # the scratch tensor may be released now: later work on this stream is ordered after the kernels
            # that read it, and the caching allocator reuses blocks stream-ordered
        launch(rows)
        exact = -1 if guessed else (rows if (rows < bound or bound == 0) else -1)
        if guessed:
            exact = int(L.olsr_live_rows_wait(int(rows_token), 1 if packed else 0, 20000))
            if exact < 0 or exact > rows:
                _ROWS_PER_INSTANCE[2] += 1
                launch(bound if (exact < 0 or exact > bound) else exact)
        if exact >= 0 and R > 0:  # a slowly decaying maximum (views of one window differ in rows per instance)
            now, old = exact / float(R), _ROWS_PER_INSTANCE[1 if packed else 0]
            _ROWS_PER_INSTANCE[1 if packed else 0] = max(now, 0.9 * old + 0.1 * now)
    return g


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depths, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                 cfg=None, with_tau_sum=False, rows_token=0):
    """RasterizeGaussiansBackwardCUDA, DGR/rasterize_points.cu:243-331.  `cfg`: see current_config().
    with_tau_sum: append the device-reduced sum over P of dL_dtau ([6] = [rho | theta]) to the tuple — what the
    reference's Python layer computes with torch.sum (DGR/diff_gaussian_rasterization/__init__.py:383-385)."""
    g = _backward(0, bg, means3D, radii, colors, None, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color, None, dL_dout_depths, sh, degree,
                  campos, geomBuffer, R, binningBuffer, imageBuffer, debug, cfg=cfg, rows_token=rows_token)
    out = (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dopacity"], g["dL_dmeans3D"], g["dL_dcov3D"], g["dL_dsh"],
           g["dL_dscales"], g["dL_drotations"], g["dL_dtau"])
    return out + (g["dL_dtau_sum"],) if with_tau_sum else out


def rasterize_language_gaussians_backward(bg, means3D, radii, colors, language, scales, rotations, scale_modifier,
                                          cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy,
                                          dL_dout_color, dL_dout_language, dL_dout_depth, sh, degree, campos,
                                          geomBuffer, R, binningBuffer, imageBuffer, debug, cfg=None, with_tau_sum=False,
                                          rows_token=0):
    """RasterizeLanguageGaussiansBackwardCUDA, DGR/rasterize_points.cu:333-455.  `cfg`: see current_config();
    `with_tau_sum`: see rasterize_gaussians_backward."""
    g = _backward(language.shape[1], bg, means3D, radii, colors, language, scales, rotations, scale_modifier,
                  cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, tan_fovx, tan_fovy, dL_dout_color,
                  dL_dout_language, dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                  debug, cfg=cfg, rows_token=rows_token)
    out = (g["dL_dmeans2D"], g["dL_dcolors"], g["dL_dlanguage"], g["dL_dopacity"], g["dL_dmeans3D"],
           g["dL_dcov3D"], g["dL_dsh"], g["dL_dscales"], g["dL_drotations"], g["dL_dtau"])
    return out + (g["dL_dtau_sum"],) if with_tau_sum else out


def last_forward_token():
    """Names the rasterize_*gaussians call this thread made last (olsr_last_forward_token); hand it to the matching
    backward as `rows_token`."""
    return int(lib().olsr_last_forward_token())


def backward_all(F, *args, **kw):
    """Every gradient of the backward incl. the internal dL_dconic / dL_ddepths and the
    device-reduced dL_dtau_sum (parity tests, frame-sharded trainer)."""
    return _backward(F, *args, want_internal=True, **kw)


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, DGR/rasterize_points.cu:457-476."""
    _require_gpu(means3D, "means3D")
    ext = compiled_binding()
    if ext is not None:
        return ext.mark_visible(means3D, _t(viewmatrix), _t(projmatrix))
    dev = means3D.device
    P = means3D.shape[0]
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P:
        with torch.cuda.device(dev):
            m, v, p = _c(means3D), _c(viewmatrix), _c(projmatrix)
            check(lib().olsr_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr() if p is not None else None,
                                          present.data_ptr(), _stream(dev)))
    return present


def state_field(kind, buf, name, *, P=0, F=0, R=0, W=0, H=0, dtype=torch.float32, count=0):
    """View into an opaque state buffer (olsr_*_field) as a tensor of `count` elements.
    "point_list" (sorted position -> Gaussian id, the reference's BinningState::point_list) is not
    materialised by the library; it is composed here from `src` and `inst_gid`."""
    if kind == "binning" and name == "point_list":
        src = state_field("binning", buf, "src", R=R, F=F, dtype=torch.int32, count=count)
        gid = state_field("binning", buf, "inst_gid", R=R, F=F, dtype=torch.int32, count=count)
        return gid[src.long()].to(dtype)
    L = lib()
    if kind == "geometry":
        ptr = L.olsr_geometry_field(buf.data_ptr(), P, F, name.encode())
    elif kind == "binning":
        ptr = L.olsr_binning_field(buf.data_ptr(), R, F, name.encode())
    else:
        ptr = L.olsr_image_field(buf.data_ptr(), W, H, TILE, name.encode())
    if not ptr:
        raise KeyError(name)
    off = ptr - buf.data_ptr()
    nbytes = count * torch.empty(0, dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype)
