"""ctypes mirror of include/olsr.h (struct olsr_scene, callback type, constants).

Pure declarations: importing this module loads no native code.
"""
import ctypes as C

OLSR_OK = 0
OLSR_ERR_ARG = -1
OLSR_ERR_DEVICE = -2
OLSR_ERR_ALLOC = -3
OLSR_ERR_CAPACITY = -4

BWD_REFERENCE = 0
BWD_EXACT = 1

ACT_OPACITY_SIGMOID = 1      # OLSR_ACT_*: the array holds the raw parameter, the kernels apply the activation
ACT_SCALE_EXP = 2
ACT_ROTATION_NORMALIZE = 4
ACT_ALL = 7
FLAG_SIGNED_EMPTY_RADII = 1  # OLSR_FLAG_SIGNED_EMPTY_RADII: radii = -radius for a bounding square that covers no tile
FLAG_FWD_ACCUM_WEIGHT = 4    # OLSR_FLAG_FWD_ACCUM_WEIGHT: fma(alpha T, f, C) on the vector ALU (images to ~1e-7)
FLAG_FRAMES_IN_FLIGHT = 8    # OLSR_FLAG_FRAMES_IN_FLIGHT: several frames in flight on several streams -> four-wave radix blocks
FLAG_FWD_ACCUM_MFMA = 2      # OLSR_FLAG_FWD_ACCUM_MFMA: the forward's feature accumulation on the matrix cores (images to ~1e-7)

BINNING_RECT = 0     # every tile of the reference's bounding square (bit-identical instance lists)
BINNING_ELLIPSE = 1  # only tiles the alpha >= 1/255 ellipse reaches (identical outputs, shorter lists)

SUPPORTED_F = (0, 3, 15, 16, 32)
SUPPORTED_TILES = (15, 16)

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_fp = C.c_void_p  # device (or, for the oracle, host) float pointers travel as raw addresses


class OlsrScene(C.Structure):
    """struct olsr_scene, include/olsr.h."""

    _fields_ = [
        ("P", C.c_int32),
        ("D", C.c_int32),
        ("M", C.c_int32),
        ("F", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("tile", C.c_int32),
        ("prefiltered", C.c_int32),
        ("debug", C.c_int32),
        ("bwd_mode", C.c_int32),
        ("tan_fovx", C.c_float),
        ("tan_fovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("binning", C.c_int32),
        ("background", _fp),
        ("means3D", _fp),
        ("shs", _fp),
        ("colors_precomp", _fp),
        ("language_precomp", _fp),
        ("opacities", _fp),
        ("scales", _fp),
        ("rotations", _fp),
        ("cov3D_precomp", _fp),
        ("viewmatrix", _fp),
        ("projmatrix", _fp),
        ("projmatrix_raw", _fp),
        ("cam_pos", _fp),
        ("activations", C.c_int32),
        ("flags", C.c_int32),
        ("tile_depth_cut", _fp),
        ("backward_row_capacity", C.c_int64),
        ("depth_order_carry", _fp),
    ]


class OlsrGradBucket(C.Structure):
    """struct olsr_grad_bucket, include/olsr.h."""

    _fields_ = [("flat", _fp), ("densify", _fp), ("max_radii", _fp), ("assign", C.c_int32), ("_pad0", C.c_int32),
                ("row_mask", _fp)]


class OlsrAdamParams(C.Structure):
    """struct olsr_adam_params, include/olsr.h."""

    _fields_ = [(n, C.c_double) for n in ("lr_xyz", "lr_sh_dc", "lr_sh_rest", "lr_opacity", "lr_scale", "lr_rotation",
                                           "lr_language", "beta1", "beta2", "eps")] + [("step", C.c_int32), ("_pad0", C.c_int32)]


class OlsrPoseParams(C.Structure):
    """struct olsr_pose_params, include/olsr.h."""

    _fields_ = [(n, C.c_double) for n in ("lr_rot", "lr_trans", "lr_exposure", "beta1", "beta2", "eps",
                                           "converged_threshold")] + [("step", C.c_int32), ("_pad0", C.c_int32)]


class OlsrLossParams(C.Structure):
    """struct olsr_loss_params, include/olsr.h."""

    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("F", C.c_int32), ("lang_width", C.c_int32),
                ("lang_height", C.c_int32), ("initialization", C.c_int32), ("alpha", C.c_float),
                ("rgb_boundary_threshold", C.c_float), ("lamda_lang", C.c_float), ("_pad0", C.c_int32)]


class OlsrLossFusion(C.Structure):
    """struct olsr_loss_fusion, include/olsr.h: the loss evaluated in the forward composite's epilogue."""

    _fields_ = [("params", OlsrLossParams), ("tracking", C.c_int32), ("skip_images", C.c_int32), ("gt_image", _fp),
                ("gt_depth", _fp), ("gt_language", _fp), ("exposure", _fp), ("grad_mask", _fp), ("dL_dimage", _fp),
                ("dL_ddepth", _fp), ("dL_dlanguage", _fp), ("loss", _fp), ("dL_dexposure", _fp), ("scratch", _fp)]


def _ptr(t):
    """data_ptr of a tensor, or None for an absent (None / empty) one — the reference maps
    empty tensors to nullptr the same way (contiguous().data<float>() of a 0-element tensor,
    tested with `!= nullptr` in CR/forward.cu:320,356)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def make_scene(*, P, D, M, F, width, height, tile, prefiltered, debug, bwd_mode, tan_fovx, tan_fovy,
               scale_modifier, binning=BINNING_RECT, activations=0, flags=0, background, means3D, shs, colors_precomp, language_precomp, opacities,
               scales, rotations, cov3D_precomp, viewmatrix, projmatrix, projmatrix_raw, cam_pos, tile_depth_cut=None,
               backward_row_capacity=0, depth_order_carry=None):
    s = OlsrScene()
    s.P, s.D, s.M, s.F = int(P), int(D), int(M), int(F)
    s.width, s.height, s.tile = int(width), int(height), int(tile)
    s.prefiltered, s.debug, s.bwd_mode = int(bool(prefiltered)), int(bool(debug)), int(bwd_mode)
    s.tan_fovx, s.tan_fovy, s.scale_modifier = float(tan_fovx), float(tan_fovy), float(scale_modifier)
    s.binning = int(binning)
    s.activations = int(activations)
    s.flags = int(flags)
    s.background = _ptr(background)
    s.means3D = _ptr(means3D)
    s.shs = _ptr(shs)
    s.colors_precomp = _ptr(colors_precomp)
    s.language_precomp = _ptr(language_precomp)
    s.opacities = _ptr(opacities)
    s.scales = _ptr(scales)
    s.rotations = _ptr(rotations)
    s.cov3D_precomp = _ptr(cov3D_precomp)
    s.viewmatrix = _ptr(viewmatrix)
    s.projmatrix = _ptr(projmatrix)
    s.projmatrix_raw = _ptr(projmatrix_raw)
    s.cam_pos = _ptr(cam_pos)
    s.tile_depth_cut = _ptr(tile_depth_cut)
    s.backward_row_capacity = int(backward_row_capacity)
    s.depth_order_carry = _ptr(depth_order_carry)
    return s
