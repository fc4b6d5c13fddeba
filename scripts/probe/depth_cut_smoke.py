import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from online_lang_splatting_amd.frame_shard import RasterWorkspace
from online_lang_splatting_amd.scene import make_scene, arc_cameras
dev = torch.device("cuda:0")
W, H, F, P = 320, 240, 15, 40000
sc = make_scene(P, W, H, F, seed=5)
g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
         rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
def camd(c): return dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx, tanfovy=c.tanfovy)
c0 = camd(sc.camera)
cot = [t.to(dev) for t in sc.cotangents(1)]
ref = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 2_000_000, dev)
ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 2_000_000, dev, depth_cut=True)
ref.set_scene(sh_degree=sc.sh_degree, **c0, **g); o_ref = {k: v.clone() for k, v in ref.forward().items()}; g_ref = {k: v.clone() for k, v in ref.backward(*cot).items()}
print("ref R", ref.rendered())
for it in range(2):
    ws.set_scene(sh_degree=sc.sh_degree, **c0, **g)
    o = ws.forward(); gr = ws.backward(*cot)
    torch.cuda.synchronize()
    same = all(torch.equal(o[k], o_ref[k]) for k in o_ref)
    gsame = all(torch.equal(gr[k], g_ref[k]) for k in g_ref)
    for k in g_ref:
        if not torch.equal(gr[k], g_ref[k]):
            d=(gr[k]-g_ref[k]).abs(); sc_=g_ref[k].abs().max().item()
            print("   ", k, "max abs diff", d.max().item(), "scale", sc_, "n differing", int((d>0).sum()), "of", d.numel(), "n > 1e-4 rel", int((d > 1e-4*g_ref[k].abs()+1e-6*sc_).sum()))
    print(it, "R", int(ws.num_rendered.cpu()[0]), "status", ws.forward_status(), "images same", same, "grads same", gsame, "finite cuts", int(torch.isfinite(ws.depth_cut).sum()), "of", ws.depth_cut.numel())
