"""Frames/s of the DROP-IN path: the reference's own API (diff_gaussian_rasterization.LanguageGaussianRasterizer,
autograd forward + backward, torch allocations, the host syncs for R and L) on BASELINE config 3."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diff_gaussian_rasterization import GaussianRasterizationSettings, LanguageGaussianRasterizer
from online_lang_splatting_amd.scene import make_config_scene

dev = torch.device("cuda:0")
sc = make_config_scene(3)
cam = sc.camera
H, W = cam.height, cam.width
settings = GaussianRasterizationSettings(
    image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=sc.bg.to(dev), scale_modifier=1.0,
    viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
    projmatrix_raw=cam.projection_matrix.to(dev), sh_degree=sc.sh_degree, campos=cam.camera_center.to(dev),
    prefiltered=False, debug=False)
rast = LanguageGaussianRasterizer(raster_settings=settings)
p = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "shs", "language")}
means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
theta = torch.zeros(3, device=dev, requires_grad=True)
rho = torch.zeros(3, device=dev, requires_grad=True)
dc, dl, dd = (t.to(dev) for t in sc.cotangents(3))


def step():
    color, lang, radii, depth, opacity, n_touched = rast(
        means3D=p["means3D"], means2D=means2D, opacities=p["opacities"], shs=p["shs"], language_precomp=p["language"],
        scales=p["scales"], rotations=p["rotations"], theta=theta, rho=rho)
    for t in list(p.values()) + [means2D, theta, rho]:
        t.grad = None
    # the image cotangents go straight into autograd (what bench.py's `dropin` leg does): no loss kernels in between
    torch.autograd.backward([color, lang, depth], [dc, dl, dd])


for _ in range(20):  # the first steps of a fresh process page in the image and grow the caching allocator
    step()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"path": "drop-in autograd API (diff_gaussian_rasterization.LanguageGaussianRasterizer)",
                  "config": 3, "frames_per_s": round(n / dt, 1), "ms_per_frame": round(1e3 * dt / n, 3)}))
# per-step wall times (diagnostic)
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", " ".join(f"{t:.2f}" for t in ts), file=sys.stderr)
print("torch allocator:", torch.cuda.memory_stats().get("num_alloc_retries"), torch.cuda.memory_reserved() >> 20, "MiB reserved", file=sys.stderr)
