"""Dense, differentiable PyTorch formulation of the rasterizer (test helper).

An INDEPENDENT derivation used to pin the oracle's `exact` backward with autograd: every
(pixel, Gaussian) pair is materialised, compositing is a cumulative product, and no analytic
gradient is written anywhere.  Thresholds (near cull, tile membership, alpha floor,
transmittance floor, the 0.99 alpha cap) are treated as constants of the graph, exactly as
the reference's analytic backward treats them (CR/backward.cu:1081-1096,1155).

Pose gradients: the camera is perturbed on the left, T_CW' = Exp(tau) T_CW with
tau = [rho | theta], which is the parametrisation the reference differentiates
(CR/backward.cu:273-288,602-640; utils/pose_utils.py:76-93).
"""
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def skew(v):
    z = torch.zeros((), dtype=v.dtype)
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def se3_exp_first_order(tau):
    """Exp(tau) to first order around 0 is enough for a gradient at tau = 0, but use the full
    series terms up to 2nd order so autograd sees the right Jacobian."""
    rho, theta = tau[:3], tau[3:]
    W = skew(theta)
    I = torch.eye(3, dtype=tau.dtype)
    R = I + W + 0.5 * W @ W
    V = I + 0.5 * W + W @ W / 6.0
    T = torch.eye(4, dtype=tau.dtype)
    T = T.clone()
    T[:3, :3] = R
    T[:3, 3] = V @ rho
    return T


def eval_sh_color(deg, sh, dirs):
    """computeColorFromSH, CR/forward.cu:23-74. sh [P,M,3], dirs [P,3] (normalised)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def render_dense(means3D, opacities, scales, rotations, shs, colors_precomp, language, W2C, Pmat, tau, *,
                 width, height, tile, sh_degree, bg, scale_modifier=1.0, cov3D_precomp=None):
    """Returns dict(color[3,H,W], language[F,H,W], depth[1,H,W], opacity[1,H,W], final_T, n_contrib, radii).
    All tensor inputs float64; W2C, Pmat are the (untransposed) 4x4 world->camera and projection."""
    dt = means3D.dtype
    P = means3D.shape[0]
    W, H = width, height
    fx = W / 2.0 * Pmat[0, 0]          # P[0,0] = 2 fx / W
    fy = H / 2.0 * Pmat[1, 1]
    tanfovx = W / (2.0 * fx.detach())
    tanfovy = H / (2.0 * fy.detach())
    T_CW = se3_exp_first_order(tau) @ W2C
    full = Pmat @ T_CW
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = (T_CW @ ph.T).T[:, :3]
    p_hom = (full @ ph.T).T
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    visible = p_view[:, 2] > 0.2
    # cov3D
    if cov3D_precomp is None:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)  # [P,3,3] row-major
        S = torch.diag_embed(scale_modifier * scales)
        L = Rm @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1), torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                             torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], -2)
    # cov2D (EWA), CR/forward.cu:77-116
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # The reference masks dL/dt.x (dL/dt.y) when the ratio is clamped (x_grad_mul, y_grad_mul,
    # CR/backward.cu:182-183,269-270) and ignores that the clamped t.x = lim * t.z still depends on
    # t.z; mimic that by detaching the clamped branch.
    rx, ry = p_view[:, 0] / tz, p_view[:, 1] / tz
    in_x = (rx.detach() >= -limx) & (rx.detach() <= limx)
    in_y = (ry.detach() >= -limy) & (ry.detach() <= limy)
    txc = torch.where(in_x, p_view[:, 0], (torch.clamp(rx, -limx, limx) * tz).detach())
    tyc = torch.where(in_y, p_view[:, 1], (torch.clamp(ry, -limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -(fy * tyc) / (tz * tz)], -1)], -2)  # [P,2,3]
    Rcw = T_CW[:3, :3]
    M = J @ Rcw                                                                  # [P,2,3]
    cov2 = M @ Sigma @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    visible = visible & (det != 0)
    det_s = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c_ / det_s, -b / det_s, a / det_s], -1)
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + tile - 1) // tile, (H + tile - 1) // tile

    def rect(v, rad, g):
        lo = torch.clamp(torch.trunc((v - rad) / tile), 0, g)
        hi = torch.clamp(torch.trunc((v + rad + tile - 1) / tile), 0, g)
        return lo, hi
    x0, x1 = rect(px.detach(), radius, gx)
    y0, y1 = rect(py.detach(), radius, gy)
    visible = visible & ((x1 - x0) * (y1 - y0) > 0)
    # colours
    if colors_precomp is None:
        campos = torch.linalg.inv(W2C)[:3, 3].detach()
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh_color(sh_degree, shs, d)
    else:
        rgb = colors_precomp
    depth = p_view[:, 2]
    # order: (depth, index); invisible last
    key = torch.where(visible, depth.detach().float().double(), torch.full_like(depth, float("inf")))
    order = torch.argsort(key, stable=True)

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)                                   # [N]
    tix, tiy = torch.div(pixx, tile, rounding_mode="floor"), torch.div(pixy, tile, rounding_mode="floor")

    o = order
    dx = px[o][None, :] - pixx[:, None]                                           # [N,P]
    dy = py[o][None, :] - pixy[:, None]
    con = conic[o]
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    member = ((tix[:, None] >= x0[o][None, :]) & (tix[:, None] < x1[o][None, :]) &
              (tiy[:, None] >= y0[o][None, :]) & (tiy[:, None] < y1[o][None, :]) & visible[o][None, :])
    G = torch.exp(torch.clamp_max(power, 0.0))
    a_raw = opacities.reshape(-1)[o][None, :] * G
    alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()              # straight-through cap
    valid = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha_v = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha_v
    test_T = torch.cumprod(one_m, dim=1)
    contrib = valid & (test_T.detach() >= 1e-4)
    alpha_c = torch.where(contrib, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(1.0 - alpha_c, dim=1)
    T_before = torch.cat([torch.ones(T_incl.shape[0], 1, dtype=dt), T_incl[:, :-1]], 1)
    w = alpha_c * T_before
    T_final = T_incl[:, -1] if P > 0 else torch.ones(pixx.shape[0], dtype=dt)
    color = (w @ rgb[o]).T + T_final[None, :] * bg.to(dt)[:, None]
    out = dict(color=color.reshape(3, H, W), depth=(w @ depth[o]).reshape(1, H, W),
               opacity=(1.0 - T_final).reshape(1, H, W), final_T=T_final.reshape(H, W))
    if language is not None:
        out["language"] = (w @ language[o]).T.reshape(-1, H, W)
    idx = torch.arange(1, P + 1)[None, :].expand_as(contrib)
    out["n_contrib"] = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=1).values.reshape(H, W)
    out["radii"] = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    out["n_touched"] = (contrib & (T_incl.detach() > 0.5)).sum(0)[torch.argsort(o)]
    return out
