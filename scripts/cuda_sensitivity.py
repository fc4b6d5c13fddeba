#!/usr/bin/env python3
"""How much do the results depend on arithmetic the CUDA toolchain decides and no other toolchain can reproduce?

The parity oracle (oracle/oracle.cpp) fixes two things the reference leaves to nvcc / the CUDA libm: `exp` (a fully
specified fp32 routine instead of CUDA's) and FMA contraction (none on decision paths; the value accumulation
C += f*alpha*T contracted like nvcc's default --fmad=true).  This script measures what changes if those choices are
made differently, by running the oracle's two sensitivity builds on BASELINE.json configs 1-3:

  libm_exp       exp from the host libm (another <= 1-ulp routine)
  contract_fast  every a*b+c the compiler can find contracted (g++ -ffp-contract=fast): a proxy for a compiler that,
                 like nvcc --fmad=true, also contracts the decision expressions (power, det, mid*mid-det, cov2D, ...)

and reports, against the default oracle: the number of (pixel, splat) blend decisions that flip (per-instance
256-bit masks of the tile's thread ranks that blended the instance, aligned by (tile, Gaussian)), Gaussians whose
radius / tile rectangle changes, and the induced image and gradient differences under the north-star criterion.
CPU only (the oracle is test infrastructure); writes profiles/r2_cuda_sensitivity.json.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_C as O  # noqa: E402
from online_lang_splatting_amd.scene import make_config_scene  # noqa: E402
from parity_common import elementwise_report, run_backend  # noqa: E402

POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint8)


def popcount(a):
    return int(POP8[a.view(np.uint8)].sum(dtype=np.int64))


def run(variant, sc, seed):
    O.use_variant(variant)
    O.lib().oracle_set_record(1)
    fo, go = run_backend(O, sc, None, seed, 15, 0)
    gx = (sc.camera.width + 14) // 15
    gy = (sc.camera.height + 14) // 15
    rg = O.get_field(fo["geom"], "ranges").view(-1, 2).long()
    pl = O.get_field(fo["geom"], "point_list").long()
    lens = (rg[:, 1] - rg[:, 0]).clamp(min=0)
    tile_of = torch.repeat_interleave(torch.arange(gx * gy), lens)
    # list positions are sorted by tile, so position k belongs to tile_of[k]
    key = (tile_of * max(sc.P, 1) + pl).numpy()
    mask = O.get_field(fo["geom"], "contrib_mask").view(-1, 8).numpy().astype(np.uint32)
    out = dict(R=fo["R"], key=key, mask=mask, radii=fo["radii"].clone(),
               images={k: fo[k].clone() for k in ("color", "language", "depth", "opacity") if fo[k] is not None and fo[k].numel()},
               grads={k: v.clone() for k, v in go.items() if v.numel()})
    O.release(fo["geom"])
    O.lib().oracle_set_record(0)
    return out


def compare(base, var):
    # align instances by (tile, Gaussian); an instance present on one side only counts all its blends as flips
    ka, kb = base["key"], var["key"]
    oa, ob = np.argsort(ka, kind="stable"), np.argsort(kb, kind="stable")
    ka_s, kb_s = ka[oa], kb[ob]
    common, ia, ib = np.intersect1d(ka_s, kb_s, assume_unique=True, return_indices=True)
    ma, mb = base["mask"][oa][ia], var["mask"][ob][ib]
    flips = popcount(ma ^ mb)
    only_a = np.setdiff1d(np.arange(len(ka_s)), ia)
    only_b = np.setdiff1d(np.arange(len(kb_s)), ib)
    flips += popcount(base["mask"][oa][only_a]) + popcount(var["mask"][ob][only_b])
    decisions = popcount(base["mask"])
    res = dict(instances=int(base["R"]), instances_variant=int(var["R"]), instances_only_in_one=int(len(only_a) + len(only_b)),
               blend_decisions=decisions, flipped_decisions=int(flips),
               flipped_fraction=float(flips) / max(decisions, 1),
               gaussians_with_different_radius=int((base["radii"] != var["radii"]).sum()))
    imgs = {}
    for k, a in base["images"].items():
        r = elementwise_report(var["images"][k], a)
        imgs[k] = dict(max_abs=r["worst_abs"], max_ref=r["max_ref"], frac_within_1e4=r["frac_within"], worst_rel=r["worst"],
                       pixels_changed=int((var["images"][k] != a).sum()))
    grads = {}
    for k, a in base["grads"].items():
        r = elementwise_report(var["grads"][k], a)
        grads[k] = dict(frac_within_1e4=r["frac_within"], worst_rel=r["worst"], max_abs=r["worst_abs"], max_ref=r["max_ref"])
    res["images"], res["gradients"] = imgs, grads
    return res


def main():
    cfgs = [int(c) for c in sys.argv[1:]] or [1, 2, 3]
    report = {"note": __doc__.strip().splitlines()[0], "configs": {}}
    for cfg in cfgs:
        sc = make_config_scene(cfg)
        base = run("default", sc, cfg)
        entry = {}
        for variant in ("libm_exp", "contract_fast"):
            var = run(variant, sc, cfg)
            entry[variant] = compare(base, var)
            e = entry[variant]
            print(f"config {cfg} {variant:14s}: {e['flipped_decisions']} of {e['blend_decisions']} blend decisions flip "
                  f"({100 * e['flipped_fraction']:.5f} %), {e['gaussians_with_different_radius']} radii differ, "
                  f"R {e['instances']} -> {e['instances_variant']}; colour max|d| {e['images']['color']['max_abs']:.2e} "
                  f"({100 * e['images']['color']['frac_within_1e4']:.4f} % of pixels within 1e-4), "
                  f"dL_dmeans3D within 1e-4: {100 * e['gradients']['dL_dmeans3D']['frac_within_1e4']:.4f} %", flush=True)
        report["configs"][str(cfg)] = entry
    O.use_variant("default")
    out = os.path.join(ROOT, "profiles", "r2_cuda_sensitivity.json")
    json.dump(report, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
