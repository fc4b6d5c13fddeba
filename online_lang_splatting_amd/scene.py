"""Seeded synthetic workloads for the rasterizer hot path.

Everything is generated on the CPU with a seeded torch.Generator so that the oracle and
the GPU see identical bits.  Camera conventions restate what the reference callers pass
(utils/camera_utils.py:103-118, gaussian_splatting/utils/graphics_utils.py:72-93):
viewmatrix = W2C^T, projmatrix = (P W2C)^T, projmatrix_raw = P^T, campos = inv(W2C^T)[3,:3].
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch

# BASELINE.json configs (index -> P, W, H, F, sh_degree_max).  cfg4 (Replica loop) has no
# synthetic equivalent at this level; see DESIGN.md.
CONFIGS = {
    1: dict(P=10_000, W=256, H=256, F=0, max_sh_degree=3),
    2: dict(P=100_000, W=640, H=480, F=0, max_sh_degree=0),
    3: dict(P=500_000, W=1200, H=680, F=15, max_sh_degree=0),
    5: dict(P=2_000_000, W=1920, H=1080, F=32, max_sh_degree=0),
}


def projection_matrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """getProjectionMatrix2, gaussian_splatting/utils/graphics_utils.py:72-93."""
    left = ((2 * cx - W) / W - 1.0) * W / 2.0
    right = ((2 * cx - W) / W + 1.0) * W / 2.0
    top = ((2 * cy - H) / H + 1.0) * H / 2.0
    bottom = ((2 * cy - H) / H - 1.0) * H / 2.0
    left = znear / fx * left
    right = znear / fx * right
    top = znear / fy * top
    bottom = znear / fy * bottom
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world2view2(R, t):
    """getWorld2View2 with zero translate / unit scale, graphics_utils.py:33-47."""
    Rt = torch.zeros(4, 4)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt


@dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    R: torch.Tensor = field(default_factory=lambda: torch.eye(3))
    T: torch.Tensor = field(default_factory=lambda: torch.zeros(3))
    znear: float = 0.01
    zfar: float = 100.0

    @property
    def tanfovx(self):
        return self.width / (2.0 * self.fx)

    @property
    def tanfovy(self):
        return self.height / (2.0 * self.fy)

    @property
    def world_view_transform(self):
        return world2view2(self.R, self.T).transpose(0, 1).contiguous()

    @property
    def projection_matrix(self):
        return projection_matrix2(self.znear, self.zfar, self.cx, self.cy, self.fx, self.fy,
                                  self.width, self.height).transpose(0, 1).contiguous()

    @property
    def full_proj_transform(self):
        return (self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)

    @property
    def camera_center(self):
        return self.world_view_transform.inverse()[3, :3].contiguous()


def default_camera(W, H, yaw_deg=0.0, tx=0.0):
    """fx = fy = W/2 (Replica: 600 @ W=1200), principal point at the image centre."""
    a = math.radians(yaw_deg)
    R = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    T = torch.tensor([tx, 0.0, 0.0])
    return Camera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, T)


def arc_cameras(W, H, n=8):
    """cfg5: n poses on an arc, yaw = (k - (n-1)/2)*4 deg, t_x = (k - (n-1)/2)*0.15 m."""
    c = (n - 1) / 2.0
    return [default_camera(W, H, (k - c) * 4.0, (k - c) * 0.15) for k in range(n)]


def shard_cameras(W, H, n=8, spacing=0.015):
    """Weak-scaling viewpoints: n poses side by side, `spacing` metres apart, no rotation, centred on the identity pose
    (n == 1: the identity pose).  Distinct views (1.5 cm shift the image by 1.5 - 30 px over the scene's depth range) of EQUAL
    cost: measured alone at config 3 (profiles/r4_exchange_overhead.json), pose 0 of 4 at 3 cm spacing runs at 2 195 fps against
    2 196 for the identity pose, pose 0 of 8 (10.5 cm off centre) at 2 142 - while pose 0 of the 8-pose ARC (rotated by 14
    degrees, it sees past the edge of the synthetic scene: fewer instances, less saturation, more gradient rows) runs at 1 912.
    A weak-scaling curve over the arc would report those 13 % as a scaling loss although they are a different workload."""
    c = (n - 1) / 2.0
    return [default_camera(W, H, 0.0, (k - c) * spacing) for k in range(n)]


@dataclass
class Scene:
    """One rasterizer invocation worth of inputs (CPU tensors)."""
    camera: Camera
    means3D: torch.Tensor
    opacities: torch.Tensor
    scales: torch.Tensor
    rotations: torch.Tensor
    shs: Optional[torch.Tensor]
    language: Optional[torch.Tensor]
    sh_degree: int
    bg: torch.Tensor
    F: int

    @property
    def P(self):
        return self.means3D.shape[0]

    def to(self, device):
        def mv(t):
            return None if t is None else t.to(device)
        return Scene(self.camera, mv(self.means3D), mv(self.opacities), mv(self.scales), mv(self.rotations),
                     mv(self.shs), mv(self.language), self.sh_degree, mv(self.bg), self.F)

    def cotangents(self, seed=0):
        """dL/d(color, language, depth) ~ N(0,1)/(H*W)."""
        g = torch.Generator().manual_seed(1000 + seed)
        H, W = self.camera.height, self.camera.width
        n = float(H * W)
        dc = torch.randn(3, H, W, generator=g) / n
        dl = torch.randn(max(self.F, 0), H, W, generator=g) / n if self.F > 0 else None
        dd = torch.randn(1, H, W, generator=g) / n
        return dc, dl, dd


def make_scene(P, W, H, F, seed=0, max_sh_degree=0, sh_degree=None, camera=None, bg=None,
               scale_mult=1.0):
    """The generator of SURVEY.md §8(d)."""
    g = torch.Generator().manual_seed(seed)
    cam = camera if camera is not None else default_camera(W, H)
    fx = W / 2.0
    tanx, tany = W / (2.0 * fx), H / (2.0 * fx)
    z = torch.rand(P, generator=g) * (6.0 - 0.3) + 0.3
    near = torch.rand(P, generator=g) < 0.02
    z_near = torch.rand(P, generator=g) * (0.2 - (-1.0)) + (-1.0)
    z = torch.where(near, z_near, z)
    x = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * tanx
    y = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * tany
    means3D = torch.stack([x, y, z], dim=1).contiguous()
    s_med = 1.2 * math.sqrt(W * H / max(P, 1)) * 3.15 / fx * scale_mult
    scales = torch.exp(math.log(s_med) + 0.6 * torch.randn(P, 3, generator=g)).contiguous()
    q = torch.randn(P, 4, generator=g)
    rotations = (q / q.norm(dim=1, keepdim=True)).contiguous()
    opacities = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g)).contiguous()
    M = (max_sh_degree + 1) ** 2
    dc = torch.rand(P, 1, 3, generator=g) * 3.0 - 1.5
    rest = 0.3 * torch.randn(P, M - 1, 3, generator=g)
    shs = torch.cat([dc, rest], dim=1).contiguous()
    language = None
    if F > 0:
        l = torch.randn(P, F, generator=g)
        language = (l / l.norm(dim=1, keepdim=True)).contiguous()
    if bg is None:
        bg = torch.zeros(3)
    deg = max_sh_degree if sh_degree is None else sh_degree
    return Scene(cam, means3D, opacities, scales, rotations, shs, language, deg, bg, F)


def make_config_scene(cfg, seed=None, P=None):
    c = dict(CONFIGS[cfg])
    if P is not None:
        c["P"] = P
    return make_scene(c["P"], c["W"], c["H"], c["F"], seed=cfg if seed is None else seed,
                      max_sh_degree=c["max_sh_degree"])


# ---------------------------------------------------------------------------------------------------------------------
# A surface-structured map: what BASELINE.json configs[3] (the Replica `slam.py` loop) renders, as far as it can be built
# without Replica data.  The generator of SURVEY 8(d) above fills a VOLUME with i.i.d. Gaussians: saturation ends 87 % of
# every tile list and 98 % of the Gaussians receive no gradient.  A SLAM map is one surface layer deep.  make_room_scene
# builds such a map the way the reference's back end does (gaussian_splatting/scene/gaussian_model.py:180-281):
#   for every keyframe: the depth image is back-projected with the keyframe's pose (create_from_rgbd_image + extrinsic),
#   randomly down-sampled by pcd_downsample_init = 32 (first keyframe) / pcd_downsample = 64
#   (configs/rgbd/replicav2/base_config.yaml:11-12); colours -> RGB2SH into f_dc, f_rest = 0; scale = sqrt(distCUDA2(points
#   of THIS keyframe) clamped at 1e-7, times point_size) with point_size = min(0.05, 0.05 * median depth)
#   (adaptive_pointsize, :199-203, :252-263), the same value on the three axes (isotropic = False -> repeat(1, 3));
#   rotation = identity quaternion; opacity = 0.5 (:268-275).
# The depth images are ray-cast from a closed box room of Replica scale with a few pieces of box furniture; the keyframes
# stand on a loop inside the room, looking outwards, close enough together for ten neighbours to overlap (the window of
# BackEnd.map, utils/slam_backend.py:499-670).  Language codes are unit-norm (the auto-encoder's codes are,
# language/autoencoder/model.py:52-56): one code per surface, a little noise per Gaussian.
C0_SH = 0.28209479177387814  # RGB2SH, gaussian_splatting/utils/sh_utils.py:114-118

ROOM_HALF = (3.5, 1.4, 2.5)  # half extents in metres, x right / y down / z forward: a 7.0 x 2.8 x 5.0 m room
# furniture: axis-aligned boxes (lo, hi) standing on the floor (y = +1.4) or hanging on a wall
ROOM_BOXES = (((-3.3, 0.55, 1.2), (-1.9, 1.4, 2.3)),     # sofa
              ((-0.8, 0.65, -0.5), (0.9, 0.72, 0.6)),      # table top
              ((-0.75, 0.72, -0.45), (-0.65, 1.4, -0.35)),  # table legs
              ((0.75, 0.72, -0.45), (0.85, 1.4, -0.35)),
              ((-0.75, 0.72, 0.45), (-0.65, 1.4, 0.55)),
              ((0.75, 0.72, 0.45), (0.85, 1.4, 0.55)),
              ((2.6, -0.6, -2.4), (3.4, 1.4, -1.2)),       # cabinet
              ((1.2, 0.9, 1.6), (2.0, 1.4, 2.4)),          # stool
              ((-1.0, -0.9, 2.42), (1.0, 0.3, 2.5)))       # picture on the far wall


def _raycast_room(cam: "Camera", W: int, H: int, pix=None):
    """z-depth [H, W], hit point in the world [H, W, 3] and surface id [H, W] (0..5 the room's faces, 6 + 6 b + face for box
    b) seen by `cam` at resolution W x H (the intrinsics are scaled from the camera's own resolution).  `pix` (flat pixel
    indices v * W + u, int64 [n]): cast only those rays; the results are then [n], [n, 3], [n]."""
    f64 = torch.float64
    sx, sy = W / cam.width, H / cam.height
    fx, fy = cam.fx * sx, cam.fy * sy
    cx, cy = (cam.cx + 0.5) * sx - 0.5, (cam.cy + 0.5) * sy - 0.5
    R, T = cam.R.to(f64), cam.T.to(f64)
    c = -(R.t() @ T)                                       # camera centre in the world
    if pix is None:
        v, u = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    else:
        v, u = torch.div(pix, W, rounding_mode="floor").to(f64), (pix % W).to(f64)
    d_cam = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], dim=-1)
    d = d_cam @ R                                          # R^T d_cam, row-vector form: z-depth = ray parameter
    eps = 1e-12
    d = torch.where(d.abs() < eps, torch.full_like(d, eps), d)
    half = torch.tensor(ROOM_HALF, dtype=f64)
    # the room, from the inside: the nearest of the three walls the ray points at
    t_ax = (torch.where(d > 0, half, -half) - c) / d
    t, ax = t_ax.min(dim=-1)
    sid = 2 * ax + (torch.gather(d, -1, ax.unsqueeze(-1)).squeeze(-1) > 0).long()
    for b, (lo, hi) in enumerate(ROOM_BOXES):              # boxes, from the outside: slab test
        lo_, hi_ = torch.tensor(lo, dtype=f64), torch.tensor(hi, dtype=f64)
        t1, t2 = (lo_ - c) / d, (hi_ - c) / d
        tn, tf = torch.minimum(t1, t2), torch.maximum(t1, t2)
        t_in, ax_in = tn.max(dim=-1)
        t_out = tf.min(dim=-1).values
        hit = (t_out >= t_in) & (t_in > 1e-6) & (t_in < t)
        t = torch.where(hit, t_in, t)
        sid = torch.where(hit, 6 + 6 * b + 2 * ax_in + (torch.gather(d, -1, ax_in.unsqueeze(-1)).squeeze(-1) > 0).long(), sid)
    hitp = c + t.unsqueeze(-1) * d
    return t.to(torch.float32), hitp.to(torch.float32), sid


def _room_colour(hitp, sid):
    """A procedural texture in [0, 1]: one base colour per surface, modulated by a 0.5 m checker and a fine sinusoid."""
    n_s = 6 + 6 * len(ROOM_BOXES)
    g = torch.Generator().manual_seed(77)
    base = 0.25 + 0.6 * torch.rand(n_s, 3, generator=g)
    p = hitp.to(torch.float64)
    checker = ((torch.floor(p[..., 0] * 2) + torch.floor(p[..., 1] * 2) + torch.floor(p[..., 2] * 2)) % 2)
    fine = 0.5 + 0.5 * torch.sin(23.0 * p[..., 0] + 17.0 * p[..., 1] + 29.0 * p[..., 2])
    rgb = base[sid] * (0.75 + 0.2 * checker.unsqueeze(-1).to(torch.float32)) + 0.08 * (fine.unsqueeze(-1).to(torch.float32) - 0.5)
    return rgb.clamp(0.0, 1.0)


def _surface_codes(F, seed=78):
    n_s = 6 + 6 * len(ROOM_BOXES)
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(n_s, F, generator=g)
    return c / c.norm(dim=1, keepdim=True)


def knn_mean_dist2_host(points: torch.Tensor) -> torch.Tensor:
    """distCUDA2 (submodules/simple-knn/simple_knn.cu:185-221) for the scene GENERATOR on a host without a GPU: an exact k-d
    tree finds the neighbours, the distances use the library kernel's pinned expression d = fma(dz, dz, fma(dy, dy, dx dx)),
    mean = ((d0 + d1) + d2) / 3 — the same bits olsr_knn_mean_dist2 returns (tests/test_gpu_knn.py).  Workload plumbing, not a
    product path: the product's distCUDA2 is GPU only."""
    import numpy as np
    from scipy.spatial import cKDTree
    x = points.detach().cpu().to(torch.float32).numpy()
    n = x.shape[0]
    if n < 4:
        raise ValueError("knn_mean_dist2_host needs at least four points")
    k = min(n, 8)  # self + 7: margin for ties at fp32 resolution
    _, idx = cKDTree(x.astype(np.float64)).query(x.astype(np.float64), k=k)
    d = x[idx[:, 1:]] - x[:, None, :]
    d2 = np.float32(d[..., 0] * d[..., 0])
    d2 = np.float32(np.float64(d[..., 1]) * np.float64(d[..., 1]) + np.float64(d2))
    d2 = np.float32(np.float64(d[..., 2]) * np.float64(d[..., 2]) + np.float64(d2))
    d2.sort(axis=1)
    return torch.from_numpy((((d2[:, 0] + d2[:, 1]) + d2[:, 2]) / np.float32(3.0)).astype(np.float32))


def room_keyframe_cameras(W, H, n, radius=0.9, height=0.1):
    """n keyframe poses on a loop of `radius` metres around the room's centre, `height` metres below it, each looking
    outwards along its radius (yaw = 360 k / n degrees); fx = fy = W / 2 as in default_camera (Replica: 600 at W = 1200)."""
    cams = []
    for k in range(n):
        a = 2.0 * math.pi * k / n
        # camera z axis (forward) in the world: (sin a, 0, cos a); W2C rotation about y by -a ... written out
        R = torch.tensor([[math.cos(a), 0.0, -math.sin(a)], [0.0, 1.0, 0.0], [math.sin(a), 0.0, math.cos(a)]])
        c = torch.tensor([radius * math.sin(a) * ROOM_HALF[0] / 3.5, height, radius * math.cos(a) * ROOM_HALF[2] / 3.5])
        cams.append(Camera(W, H, W / 2.0, W / 2.0, (W - 1) / 2.0, (H - 1) / 2.0, R, -(R @ c)))
    return cams


@dataclass
class RoomScene:
    """make_room_scene's result: the map (`scene`, whose camera is the window's first view), the window's cameras, and per
    view the ray-cast targets a mapping iteration fits (gt_image [3,H,W], gt_depth [H,W], gt_language [F,192,192] or None)."""
    scene: Scene
    cameras: list
    targets: list
    keyframes: int
    points_per_keyframe: list

    def view(self, v) -> Scene:
        s = self.scene
        return Scene(self.cameras[v], s.means3D, s.opacities, s.scales, s.rotations, s.shs, s.language, s.sh_degree, s.bg, s.F)


def make_room_scene(P=500_000, W=1200, H=680, F=15, views=10, seed=0, max_sh_degree=0, knn=None, lang_size=192,
                    window_start=0, random_views=0):
    """About P Gaussians (exactly P when the keyframes supply enough) built keyframe by keyframe as the reference's back end
    builds its map, and `views` consecutive keyframe poses to render it from, followed by `random_views` keyframes drawn from
    the rest of the loop (BackEnd.map renders its window plus two random earlier keyframes, utils/slam_backend.py:510-530).  `knn`: points [n,3] (CPU float32) -> mean
    squared distance to the three nearest neighbours [n]; default: the library's olsr_knn_mean_dist2 when a GPU is present,
    else knn_mean_dist2_host (bit-identical, tests/test_gpu_room_scene.py)."""
    if knn is None:
        if torch.cuda.is_available():
            from .simple_knn import distCUDA2

            def knn(p):
                return distCUDA2(p.cuda()).cpu()
        else:
            knn = knn_mean_dist2_host
    g = torch.Generator().manual_seed(seed)
    N = W * H
    n_first, n_next = N // 32, N // 64                     # pcd_downsample_init / pcd_downsample
    K = 1 if P <= n_first else 1 + -(-(P - n_first) // max(n_next, 1))
    K = max(K, views)
    kcams = room_keyframe_cameras(W, H, K)
    codes = _surface_codes(F) if F > 0 else None
    xyz, rgb, scl, lang, per_kf = [], [], [], [], []
    left = P
    for k, cam in enumerate(kcams):
        if left <= 0:
            break
        # random_down_sample(1 / factor) of the back-projected image: the kept pixels are drawn first and only their rays are
        # cast (the depth image of a keyframe is never needed whole; the median depth of adaptive_pointsize is taken over the
        # kept pixels — in this room it is above 1 m from every pose, so point_size = 0.05 either way)
        n_take = min(n_first if k == 0 else n_next, left)
        if 0 < left - n_take < 4:   # (a last keyframe of fewer than four points has no three neighbours: it takes them along)
            n_take = min(left, N)
        pick = torch.randperm(N, generator=g)[:n_take]
        depth, hitp, sid = _raycast_room(cam, W, H, pick)
        col = _room_colour(hitp, sid)
        # create_from_rgbd_image(..., extrinsic = W2C): x = (u - cx) z / fx, y = (v - cy) z / fy, then C2W
        u, v = (pick % W).to(torch.float32), torch.div(pick, W, rounding_mode="floor").to(torch.float32)
        pc = torch.stack([(u - cam.cx) * depth / cam.fx, (v - cam.cy) * depth / cam.fy, depth], dim=-1)
        pts = ((pc - cam.T) @ cam.R).contiguous()          # R^T (p - T)
        point_size = min(0.05, 0.05 * float(depth.median()))
        d2 = torch.clamp_min(knn(pts), 1e-7) * point_size
        xyz.append(pts)
        rgb.append(col)
        scl.append(torch.sqrt(d2))
        if F > 0:
            l = codes[sid] + 0.1 * torch.randn(n_take, F, generator=g)
            lang.append(l / l.norm(dim=1, keepdim=True))
        per_kf.append(n_take)
        left -= n_take
    means3D = torch.cat(xyz).contiguous()
    n = means3D.shape[0]
    M = (max_sh_degree + 1) ** 2
    shs = torch.zeros(n, M, 3)
    shs[:, 0, :] = (torch.cat(rgb) - 0.5) / C0_SH          # RGB2SH
    scales = torch.cat(scl).unsqueeze(1).repeat(1, 3).contiguous()
    rotations = torch.zeros(n, 4)
    rotations[:, 0] = 1.0
    opacities = torch.full((n, 1), 0.5)
    language = torch.cat(lang).contiguous() if F > 0 else None
    window = [kcams[(window_start + i) % K] for i in range(views)]
    rest = [k for k in range(K) if (k - window_start) % K >= views]
    if random_views > 0 and rest:
        for j in torch.randperm(len(rest), generator=g)[:random_views].tolist():
            window.append(kcams[rest[j]])
    targets = []
    for cam in window:
        depth, hitp, sid = _raycast_room(cam, W, H)
        gt_lang = None
        if F > 0:
            _, _, sid_l = _raycast_room(cam, lang_size, lang_size)
            gt_lang = codes[sid_l].permute(2, 0, 1).contiguous()
        targets.append((_room_colour(hitp, sid).permute(2, 0, 1).contiguous(), depth.contiguous(), gt_lang))
    sc = Scene(window[0], means3D, opacities, scales, rotations, shs.contiguous(), language, 0, torch.zeros(3), F)
    return RoomScene(sc, window, targets, len(per_kf), per_kf)
