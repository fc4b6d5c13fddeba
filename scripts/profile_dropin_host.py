"""Host-side profile (cProfile) of the drop-in autograd path on a BASELINE config (argv[1], default 1): where the Python /
binding time of a forward + backward goes when the frame is too small to hide it (config 1: ~0.26 ms per frame on the
host against ~0.20 ms on the GPU)."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diff_gaussian_rasterization import GaussianRasterizationSettings, LanguageGaussianRasterizer, GaussianRasterizer
from online_lang_splatting_amd.scene import make_config_scene
cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
sc = make_config_scene(cfgn); cam = sc.camera; H, W = cam.height, cam.width
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0,
    viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev), projmatrix_raw=cam.projection_matrix.to(dev),
    sh_degree=sc.sh_degree, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
F = sc.F
rast = (LanguageGaussianRasterizer if F > 0 else GaussianRasterizer)(raster_settings=settings)
names = ("means3D", "opacities", "scales", "rotations", "shs") + (("language",) if F > 0 else ())
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
theta = torch.zeros(3, device=dev, requires_grad=True); rho = torch.zeros(3, device=dev, requires_grad=True)
dc, dl, dd = (None if t is None else t.to(dev) for t in sc.cotangents(3))
def step():
    if F > 0:
        color, lang, radii, depth, opacity, nt = rast(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], language_precomp=p["language"], scales=p["scales"], rotations=p["rotations"], theta=theta, rho=rho)
        outs, cots = [color, lang, depth], [dc, dl, dd]
    else:
        color, radii, depth, opacity, nt = rast(means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], scales=p["scales"], rotations=p["rotations"], theta=theta, rho=rho)
        outs, cots = [color, depth], [dc, dd]
    for t in list(p.values()) + [means2D, theta, rho]: t.grad = None
    torch.autograd.backward(outs, cots)
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print("ms/frame", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
