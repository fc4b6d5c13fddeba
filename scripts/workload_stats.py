"""Workload statistics of config 3 on the GPU: active-instance fraction, list lengths, kmax."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_common import run_backend
from online_lang_splatting_amd import _C as G
from online_lang_splatting_amd.scene import make_config_scene
dev = torch.device("cuda:0")
sc = make_config_scene(3); P, W, H, F = sc.P, 1200, 680, 15
fg, gg = run_backend(G, sc, dev, 3, 15, 0)
R = fg["R"]; gx, gy = math.ceil(W/15), math.ceil(H/15)
flags = G.state_field("binning", fg["binning"], "flags", R=R, F=F, dtype=torch.uint8, count=R)
rg = G.state_field("image", fg["img"], "ranges", W=W, H=H, dtype=torch.int32, count=2*gx*gy).view(-1,2).long()
nc = G.state_field("image", fg["img"], "n_contrib", W=W, H=H, dtype=torch.int32, count=W*H).view(H, W).float()
lens = (rg[:,1]-rg[:,0]).float()
print(f"R={R} active instances={int(flags.sum())} ({flags.float().mean().item():.3f})")
print(f"list len mean {lens.mean():.0f} max {lens.max():.0f} p50 {lens.median():.0f}")
# per-tile kmax
pad = torch.zeros(gy*15, gx*15, device=dev); pad[:H,:W] = nc
kmax = pad.view(gy,15,gx,15).permute(0,2,1,3).reshape(gy*gx,-1).max(1).values
print(f"kmax mean {kmax.mean():.0f} max {kmax.max():.0f}; mean n_contrib {nc.mean():.0f}; sum kmax {kmax.sum():.0f}")
tt = G.state_field("geometry", fg["geom"], "tiles_touched", P=P, F=F, dtype=torch.int32, count=P).float()
print(f"tiles_touched mean {tt.mean():.2f} max {tt.max():.0f} p99 {tt.quantile(0.99):.0f}; visible {(fg['radii']>0).float().mean():.3f}")
for thr in (8, 16, 32, 64, 128, 256):
    m = tt > thr
    print(f"n>{thr}: {int(m.sum())} Gaussians ({m.float().mean().item()*100:.1f}%), instances {int(tt[m].sum())} ({tt[m].sum().item()/R*100:.1f}% of R)")
livepairs = 0
fl = flags.to(torch.int32)
for b in range(4): livepairs += int(((fl >> b) & 1).sum())
print("live (instance,slot) pairs", livepairs)
