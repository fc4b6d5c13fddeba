import sys, time, torch
sys.path.insert(0, ".")
from online_lang_splatting_amd.frame_shard import FrameLanes
from online_lang_splatting_amd.scene import make_room_scene
from online_lang_splatting_amd.slam_iterations import MappingStep
dev = torch.device("cuda:0")
W, H, F, P = 1200, 680, 15, 500_000
rs = make_room_scene(P, W, H, F, views=10, random_views=2, seed=3)
sc = rs.scene
start = dict(means3D=sc.means3D.to(dev), shs=sc.shs.to(dev), opacities=torch.logit(sc.opacities).to(dev).contiguous(),
             scales=torch.log(sc.scales).to(dev).contiguous(), rotations=sc.rotations.to(dev), language=sc.language.to(dev))
camd = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), projmatrix_raw=c.projection_matrix.to(dev),
             campos=c.camera_center.to(dev), tanfovx=c.tanfovx, tanfovy=c.tanfovy) for c in rs.cameras]
lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
lanes = FrameLanes(4, P, W, H, F, 1, 1_000_000, dev)
for form in (True, False, "auto", True, "auto"):
    p = {k: v.clone() for k, v in start.items()}
    st = MappingStep(lanes, p, sc.bg.to(dev), 0, camd, rs.targets, lrs, exposure=torch.zeros(2, device=dev), fused_loss=form)
    ts = []
    for i in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st.iteration(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    t0 = time.perf_counter()
    for i in range(6):
        st.iteration()
    torch.cuda.synchronize()
    print(form, "per-iteration ms", ts, "pipelined", round((time.perf_counter() - t0) / 6 * 1e3, 3), st.calibration, st.fused)
