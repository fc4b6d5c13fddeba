"""Builds libolsr.so (HIP kernels + C-ABI) in-tree for gfx950.

    python -m online_lang_splatting_amd.build [--force] [--keep-temps]

hipcc cross-compiles without a GPU.  Every translation unit is compiled with
-ffp-contract=off: the kernels' fp32 operation order is part of the parity contract
(csrc/olsr_device.h), FMAs appear only where the source writes them.
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libolsr.so")
ARCH = "gfx950"

# (source, object name, extra defines)
UNITS = [
    ("olsr_api.hip", "olsr_api.o", []),
    ("k_preprocess.hip", "k_preprocess.o", []),
    ("k_binning.hip", "k_binning.o", []),
    ("k_sort.hip", "k_sort.o", []),
    ("k_order_carry.hip", "k_order_carry.o", []),
    ("k_render_fwd.hip", "k_render_fwd.o", ["-DOLSR_FWD_TU_LOSS=0"]),
    ("k_render_fwd.hip", "k_render_fwd_loss.o", ["-DOLSR_FWD_TU_LOSS=1"]),
    ("k_render_bwd.hip", "k_render_bwd_ref.o", ["-DOLSR_BWD_TU_MODE=0"]),
    ("k_render_bwd.hip", "k_render_bwd_exact.o", ["-DOLSR_BWD_TU_MODE=1"]),
    ("k_render_bwd_ordered.hip", "k_render_bwd_ordered.o", []),
    ("k_preprocess_bwd.hip", "k_preprocess_bwd.o", []),
    ("k_accumulate.hip", "k_accumulate.o", []),
    ("k_loss.hip", "k_loss.o", []),
    ("k_knn.hip", "k_knn.o", []),
    ("k_adam.hip", "k_adam.o", []),
    ("k_pose.hip", "k_pose.o", []),
]
HEADERS = ["olsr_device.h", "olsr_state.h", "olsr_kernels.h", "olsr_loss_device.h", os.path.join("..", "..", "include", "olsr.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _flags(keep_temps):
    # -fno-slp-vectorize: packed fp32 (v_pk_*_f32) is half rate on gfx950 — two lanes' work in twice the time — so the
    # compiler's own pairing of scalar fp32 buys nothing and costs the moves that line the operands up (round 4, measured:
    # preprocess 29.3 -> 27.0 us, preprocess_backward 72.3 -> 69.7 us, numerically neutral).  Where pairs pay (the forward
    # composite's loop, written on explicit 2-vectors) the source says so itself.
    f = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
    if keep_temps:
        f += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    f += os.environ.get("OLSR_EXTRA_DEFS", "").split()  # (experiments: variant builds of the kernels)
    return f


def _objpath(obj):
    # one directory per object: -save-temps=obj names its files after the source, and
    # k_render_bwd.hip is compiled twice
    return os.path.join(OBJ, obj[:-2], obj)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(unit, keep_temps, force):
    src, obj, defs = unit
    srcp, objp = os.path.join(CSRC, src), _objpath(obj)
    os.makedirs(os.path.dirname(objp), exist_ok=True)
    deps = [srcp] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and not _stale(objp, deps):
        return obj, False, ""
    cmd = [hipcc()] + _flags(keep_temps) + defs + ["-c", srcp, "-o", objp]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src} {defs}:\n{p.stdout}\n{p.stderr}")
    return obj, True, p.stderr


def build(force=False, keep_temps=False, verbose=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or min(len(UNITS), max(1, (os.cpu_count() or 2)))
    rebuilt = False
    logs = []
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did, log in ex.map(lambda u: _compile(u, keep_temps, force), UNITS):
            rebuilt = rebuilt or did
            if log:
                logs.append((obj, log))
            if verbose and did:
                print(f"[olsr build] compiled {obj}", file=sys.stderr)
    objs = [_objpath(u[1]) for u in UNITS]
    if rebuilt or force or _stale(LIB, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"link failed:\n{p.stdout}\n{p.stderr}")
        if verbose:
            print(f"[olsr build] linked {LIB}", file=sys.stderr)
    if keep_temps:
        with open(os.path.join(OBJ, "resource_usage.log"), "w") as f:
            for obj, log in logs:
                f.write(f"==== {obj}\n{log}\n")
    return LIB


TORCH_EXT = os.path.join(HERE, "_olsr_torch.so")


def build_torch_ext(force=False, verbose=False):
    """g++ build of csrc/olsr_torch.cpp, the compiled torch binding of the `_C` surface (host-only C++ against the
    torch headers; it links libolsr.so through an $ORIGIN rpath).  In-tree, like libolsr.so, so that it travels
    with the snapshot.  Takes about a minute the first time (torch's headers)."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "olsr_torch.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "olsr.h"), os.path.abspath(__file__)]
    if not force and not _stale(TORCH_EXT, deps):
        return TORCH_EXT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    tlib = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_olsr_torch",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{d}" for d in ce.include_paths()] + [f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}"]
    cmd += [src, "-o", TORCH_EXT, f"-L{tlib}", f"-L{HERE}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python", "-l:libolsr.so", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"g++ failed for olsr_torch.cpp:\n{p.stdout}\n{p.stderr[-6000:]}")
    if verbose:
        print(f"[olsr build] built {TORCH_EXT}", file=sys.stderr)
    return TORCH_EXT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--keep-temps", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, keep_temps=a.keep_temps, verbose=True))
    print(build_torch_ext(force=a.force, verbose=True))
