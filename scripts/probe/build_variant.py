"""A variant build of libolsr.so for kernel experiments: the composite units (or the units named by --units) recompiled
with extra defines, everything else taken from the regular build.  Select it at run time with OLSR_LIB=<path>.
usage: build_variant.py NAME [--units k_render_fwd.o,k_render_bwd_ref.o] -- -DFOO=1 ...   ->  online_lang_splatting_amd/_variants/libolsr_NAME.so"""
import os
import subprocess
import sys

sys.path.insert(0, ".")
from online_lang_splatting_amd import build as B  # noqa: E402

args = sys.argv[1:]
name = args[0]
units = ["k_render_fwd.o", "k_render_fwd_loss.o", "k_render_bwd_ref.o", "k_render_bwd_exact.o"]
if "--units" in args:
    units = args[args.index("--units") + 1].split(",")
defs = args[args.index("--") + 1:]
B.build()
vdir = os.path.join(B.HERE, "_variants")
odir = os.path.join("/tmp", "olsr_variant_" + name)
os.makedirs(vdir, exist_ok=True)
os.makedirs(odir, exist_ok=True)
objs = []
procs = []
for src, obj, d in B.UNITS:
    if obj in units:
        o = os.path.join(odir, obj)
        procs.append(subprocess.Popen([B.hipcc()] + B._flags(False) + d + defs + ["-c", os.path.join(B.CSRC, src), "-o", o]))
        objs.append(o)
    else:
        objs.append(B._objpath(obj))
for p in procs:
    if p.wait() != 0:
        raise SystemExit("compile failed")
lib = os.path.join(vdir, f"libolsr_{name}.so")
subprocess.check_call([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
