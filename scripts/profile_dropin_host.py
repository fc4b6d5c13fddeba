"""Host-side profile of the drop-in autograd path on config 3 (where do the ~0.6 ms of Python / binding time per frame go).
    python scripts/profile_dropin_host.py [steps=200]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from online_lang_splatting_amd.scene import make_config_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, LanguageGaussianRasterizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
sc = make_config_scene(3)
cam = sc.camera
rs = GaussianRasterizationSettings(
    image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev),
    scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
    projmatrix_raw=cam.projection_matrix.to(dev), sh_degree=sc.sh_degree, campos=cam.camera_center.to(dev),
    prefiltered=False, debug=False)
rast = LanguageGaussianRasterizer(raster_settings=rs)
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs", "language")}
means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
theta = torch.zeros(3, device=dev, requires_grad=True)
rho = torch.zeros(3, device=dev, requires_grad=True)
dc, dl, dd = [t.to(dev) for t in sc.cotangents(3)]
leaves = list(p.values()) + [means2D, theta, rho]


def step():
    color, language, radii, depth, opacity, n_touched = rast(
        means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
        rotations=p["rotations"], theta=theta, rho=rho, language_precomp=p["language"])
    for t in leaves:
        t.grad = None
    torch.autograd.backward([color, language, depth], [dc, dl, dd])


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / steps:.4f} ms per frame un-profiled")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
