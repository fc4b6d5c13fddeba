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
