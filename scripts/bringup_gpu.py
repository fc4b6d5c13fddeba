"""GPU bring-up: stage-by-stage comparison of the HIP library against the CPU oracle."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_common import run_backend, rel_err, make_scene
from oracle import oracle_C as O
from online_lang_splatting_amd import _C as G

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))

def compare(P, W, H, F, deg=0, seed=1, tile=15, mode=0, bg=None, max_sh=None):
    sc = make_scene(P, W, H, F, seed=seed, max_sh_degree=deg if max_sh is None else max_sh, sh_degree=deg, bg=bg)
    t0 = time.time(); fo, go = run_backend(O, sc, None, seed, tile, mode); t1 = time.time()
    fg, gg = run_backend(G, sc, dev, seed, tile, mode, binning=0); torch.cuda.synchronize(); t2 = time.time()  # reference binning: lists comparable 1:1
    print(f"--- P={P} {W}x{H} F={F} deg={deg} tile={tile} mode={mode}: R oracle={fo['R']} hip={fg['R']}  (oracle {t1-t0:.2f}s, hip {t2-t1:.2f}s)")
    # geometry stage
    for name, cnt, dt in (("depths", P, torch.float32), ("means2D", 2*P, torch.float32), ("conic_opacity", 4*P, torch.float32),
                          ("cov3D", 6*P, torch.float32), ("rgb", 3*P, torch.float32), ("tiles_touched", P, torch.int32)):
        h = G.state_field("geometry", fg["geom"], name, P=P, F=F, dtype=dt, count=cnt).cpu()
        o = O.get_field(fo["geom"], name)
        vis = (fo["radii"] > 0)
        per = cnt // P
        m = vis.repeat_interleave(per)
        if dt == torch.int32:
            print(f"  geom {name:14s} equal={bool((h[m]==o[m]).all())}")
        else:
            print(f"  geom {name:14s} bit-equal={bool((h[m]==o[m]).all())} rel={rel_err(h[m], o[m])[0]:.2e}")
    print(f"  radii equal={bool((fg['radii'].cpu()==fo['radii']).all())}")
    if fo["R"] == fg["R"] and fo["R"] > 0:
        pl = G.state_field("binning", fg["binning"], "point_list", R=fg["R"], F=F, dtype=torch.int32, count=fg["R"]).cpu()
        print(f"  point_list equal={bool((pl==O.get_field(fo['geom'],'point_list')).all())}")
        rg = G.state_field("image", fg["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2*((W+tile-1)//tile)*((H+tile-1)//tile)).cpu()
        ro = O.get_field(fo["geom"], "ranges")
        same = ((rg.view(-1,2)[:,1]-rg.view(-1,2)[:,0]) == (ro.view(-1,2)[:,1]-ro.view(-1,2)[:,0])).all()
        print(f"  ranges lengths equal={bool(same)}")
        nc = G.state_field("image", fg["img"], "n_contrib", W=W, H=H, dtype=torch.int32, count=W*H).cpu()
        print(f"  n_contrib equal={bool((nc==O.get_field(fo['geom'],'n_contrib')).all())}")
    for k in ("color", "language", "depth", "opacity"):
        if fo[k] is None or fo[k].numel() == 0: continue
        r, e = rel_err(fg[k], fo[k])
        print(f"  fwd  {k:14s} bit-equal={bool((fg[k].cpu()==fo[k]).all())} rel={r:.2e} abs={e:.2e}")
    print(f"  n_touched equal={bool((fg['n_touched'].cpu()==fo['n_touched']).all())}")
    for k in go:
        if go[k].numel() == 0: continue
        r, e = rel_err(gg[k], go[k])
        print(f"  bwd  {k:14s} rel={r:.2e} abs={e:.2e}")
    if "dL_dtau_sum" in gg:
        r, e = rel_err(gg["dL_dtau_sum"], go["dL_dtau"].sum(0))
        print(f"  bwd  dL_dtau_sum    rel={r:.2e}")
    O.release(fo["geom"])

compare(300, 45, 30, 15)
compare(300, 45, 30, 15, mode=1)
compare(2000, 128, 128, 15, deg=0, bg=torch.tensor([0.3, 0.1, 0.7]))
compare(2000, 128, 128, 0, deg=3)
compare(3000, 160, 120, 3, deg=1, tile=16, max_sh=2)
compare(3000, 160, 120, 32, mode=1)
compare(3000, 160, 120, 16, mode=0)
compare(20000, 320, 240, 15, seed=3)
