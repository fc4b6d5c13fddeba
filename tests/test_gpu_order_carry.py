"""Carried depth order (include/olsr.h "Carried depth order", csrc/k_order_carry.hip): the forward repairs the order its
previous frame left in the caller's array and falls back to the radix passes ON THE DEVICE when it cannot prove the result.
The contract is "the lists never depend on the array": every case compares a RasterWorkspace(carry_order=True) with a plain
one on the same inputs — instance lists, images, radii, n_touched and gradients bit for bit — whatever the array held, and
checks on which frames the repair was (not) enough.  The plain workspace is what tests/test_gpu_parity.py holds against the
oracle; one case here meets the oracle directly."""
import pytest
import torch

from online_lang_splatting_amd import _C, _abi
from online_lang_splatting_amd.scene import default_camera, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cam(c, dev):
    return dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                tanfovy=c.tanfovy)


def _setup(P=40000, W=320, H=240, F=15, seed=5, tile=15, flags=0, capacity=2_000_000):
    from online_lang_splatting_amd.frame_shard import RasterWorkspace
    dev = torch.device(DEV)
    sc = make_scene(P, W, H, F, seed=seed)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev) if F > 0 else None)
    cot = [t.to(dev) if t is not None else None for t in sc.cotangents(1)]
    kw = dict(tile=tile, flags=flags)
    plain = RasterWorkspace(P, W, H, F, sc.shs.shape[1], capacity, dev, **kw)
    carry = RasterWorkspace(P, W, H, F, sc.shs.shape[1], capacity, dev, carry_order=True, **kw)
    return sc, g, cot, plain, carry, dev


def _frame(ws, sc, cam, g, cot):
    ws.set_scene(sh_degree=sc.sh_degree, **cam, **g)
    out = {k: v.clone() for k, v in ws.forward().items()}
    R = ws.rendered()[0]
    pl = (_C.state_field("binning", ws.binning, "point_list", R=ws.capacity, F=ws.F, dtype=torch.int32, count=R).clone()
          if R else torch.empty(0, dtype=torch.int32))
    grads = {k: v.clone() for k, v in ws.backward(*cot).items()}
    return out, grads, pl, R


def _assert_identical(a, b):
    out, grads, pl, R = a
    out_ref, grads_ref, pl_ref, R_ref = b
    assert R == R_ref
    assert torch.equal(pl, pl_ref), "instance lists differ"
    for k in out_ref:
        assert torch.equal(out[k], out_ref[k]), k
    for k in grads_ref:   # the same lists in the same order: the same sums, bit for bit
        assert torch.equal(grads[k], grads_ref[k]) or bool(((grads[k] == grads_ref[k]) | (grads[k].isnan() & grads_ref[k].isnan())).all()), k


def _is_order_of(ws, dev):
    """the carried array is a permutation of [0, P) in ascending (key, index) order of the frame's sort keys"""
    P = ws.P
    o = ws.depth_order_carry.long()
    assert sorted(o.tolist()) == list(range(P))
    if ws.carry_missed():   # (the radix passes ran: their ping-pong has overwritten the keys; the lists were compared)
        return
    keys = _C.state_field("geometry", ws.geom, "sort_keys", P=P, F=ws.F, dtype=torch.int32, count=P).long() & 0xFFFFFFFF
    comp = (keys[o] << 32) | o
    assert bool((comp[1:] > comp[:-1]).all())


@pytest.mark.parametrize("P", [1, 63, 1000, 1024, 1025, 2047, 2048, 2049, 3072, 5000, 8192, 8193, 10000, 40000])
def test_repeated_view_repairs_and_equals_the_sort(hip, P):
    """First frame: the array holds zeros -> miss, the radix passes run and leave the order; from then on the same view is a
    hit.  Every frame equals the plain workspace's, at every window shape (ragged tails, one window, the one-launch sort's
    range, several windows)."""
    sc, g, cot, plain, carry, dev = _setup(P=P, W=200, H=150, F=3, seed=300 + P % 97)
    if P > 3:
        sc.means3D[::3, 2] = sc.means3D[0, 2]   # equal depths in quantity: ties go by index
        g["means3D"] = sc.means3D.to(dev)
    cam = _cam(sc.camera, dev)
    ref = _frame(plain, sc, cam, g, cot)
    for it in range(3):
        got = _frame(carry, sc, cam, g, cot)
        assert carry.carry_missed() == (it == 0 and P > 1), (it, P)   # (one Gaussian: zeros ARE its order)
        _assert_identical(got, ref)
        _is_order_of(carry, dev)


def test_small_pose_steps_hit_large_ones_fall_back_and_all_are_exact(hip):
    sc, g, cot, plain, carry, dev = _setup(P=60000, W=400, H=300, F=15, seed=11)
    W, H = sc.camera.width, sc.camera.height
    steps = [(0.0, 0.0), (0.02, 0.001), (0.05, 0.002), (0.05, 0.002), (0.3, 0.01), (6.0, 0.2), (6.05, 0.2), (-10.0, -0.4),
             (-10.0, -0.4)]
    verdicts = []
    for yaw, tx in steps:
        cam = _cam(default_camera(W, H, yaw, tx), dev)
        ref = _frame(plain, sc, cam, g, cot)
        got = _frame(carry, sc, cam, g, cot)
        verdicts.append(carry.carry_missed())
        _assert_identical(got, ref)
        _is_order_of(carry, dev)
    assert verdicts[0] is True            # nothing to repair yet
    assert verdicts[1] is False and verdicts[2] is False and verdicts[3] is False   # optimiser-sized steps
    assert verdicts[5] is True            # six degrees at once: beyond half a window
    assert verdicts[6] is False and verdicts[8] is False


@pytest.mark.parametrize("fill", ["random", "duplicates", "out_of_range", "reversed", "other_scene"])
def test_any_content_of_the_array_gives_the_same_frame(hip, fill):
    sc, g, cot, plain, carry, dev = _setup(P=30000, W=320, H=240, F=15, seed=21)
    cam = _cam(sc.camera, dev)
    ref = _frame(plain, sc, cam, g, cot)
    P = carry.P
    gen = torch.Generator(device="cpu").manual_seed(3)
    if fill == "random":
        a = torch.randperm(P, generator=gen)
    elif fill == "duplicates":
        a = torch.randint(0, P // 2, (P,), generator=gen)
    elif fill == "out_of_range":
        a = torch.randint(-2**31, 2**31 - 1, (P,), generator=gen)
    elif fill == "reversed":
        _frame(carry, sc, cam, g, cot)
        a = carry.depth_order_carry.cpu().flip(0)
    else:
        _frame(carry, sc, _cam(default_camera(sc.camera.width, sc.camera.height, 170.0, 0.0), dev), g, cot)
        a = carry.depth_order_carry.cpu()
    carry.depth_order_carry.copy_(a.to(torch.int32))
    got = _frame(carry, sc, cam, g, cot)
    assert carry.carry_missed()
    _assert_identical(got, ref)
    _is_order_of(carry, dev)
    got = _frame(carry, sc, cam, g, cot)
    assert not carry.carry_missed()
    _assert_identical(got, ref)


def test_a_duplicate_index_in_a_sorted_looking_array_is_a_miss(hip):
    """An array that is in order except that one index appears twice (and one never): ascending pairs, but not strictly —
    the repair must not accept it."""
    sc, g, cot, plain, carry, dev = _setup(P=20000, W=320, H=240, F=0, seed=23)
    cam = _cam(sc.camera, dev)
    ref = _frame(plain, sc, cam, g, cot)
    _frame(carry, sc, cam, g, cot)
    a = carry.depth_order_carry.clone()
    for pos in (0, 1023, 1024, 2047, 2048, 7000, carry.P - 1):
        b = a.clone()
        b[pos] = b[pos - 1] if pos else b[1]
        carry.depth_order_carry.copy_(b)
        got = _frame(carry, sc, cam, g, cot)
        assert carry.carry_missed(), pos
        _assert_identical(got, ref)


@pytest.mark.parametrize("tile,F,flags", [(16, 0, 0), (15, 15, _abi.FLAG_FRAMES_IN_FLIGHT), (16, 32, _abi.FLAG_FRAMES_IN_FLIGHT)])
def test_block_shapes_tiles_and_channels(hip, tile, F, flags):
    sc, g, cot, plain, carry, dev = _setup(P=25000, W=320, H=240, F=F, seed=31, tile=tile, flags=flags)
    W, H = sc.camera.width, sc.camera.height
    for yaw in (0.0, 0.03, 0.06, 3.0, 3.0):
        cam = _cam(default_camera(W, H, yaw, 0.0), dev)
        _assert_identical(_frame(carry, sc, cam, g, cot), _frame(plain, sc, cam, g, cot))
    assert not carry.carry_missed()


def test_parameter_steps_between_frames(hip):
    """What a mapping iteration does to the order: every Gaussian moves by a fraction of a millimetre."""
    sc, g, cot, plain, carry, dev = _setup(P=50000, W=320, H=240, F=15, seed=41)
    cam = _cam(sc.camera, dev)
    gen = torch.Generator(device="cpu").manual_seed(5)
    missed = []
    for it in range(5):
        g["means3D"] = (g["means3D"].cpu() + 2e-4 * torch.randn(g["means3D"].shape, generator=gen)).to(dev).contiguous()
        _assert_identical(_frame(carry, sc, cam, g, cot), _frame(plain, sc, cam, g, cot))
        missed.append(carry.carry_missed())
    assert missed[0] and not any(missed[1:]), missed


def test_forward_with_a_carried_order_meets_the_oracle(hip, oracle):
    """One direct comparison with the oracle on a frame whose order came from a repair (not from the radix passes)."""
    from parity_common import fwd_args
    sc, g, cot, plain, carry, dev = _setup(P=12000, W=200, H=150, F=15, seed=51, capacity=600_000)
    W, H = sc.camera.width, sc.camera.height
    cam0, cam1 = default_camera(W, H, 0.0, 0.0), default_camera(W, H, 0.04, 0.002)
    _frame(carry, sc, _cam(cam0, dev), g, cot)
    got = _frame(carry, sc, _cam(cam1, dev), g, cot)
    assert not carry.carry_missed()
    sc.camera = cam1
    ro = oracle.rasterize_language_gaussians(*fwd_args(sc, None))
    assert torch.equal(got[0]["color"].cpu(), ro[1]) and torch.equal(got[0]["language"].cpu(), ro[2])
    assert torch.equal(got[0]["radii"].cpu(), ro[3]) and torch.equal(got[0]["depth"].cpu(), ro[7])


def test_mapping_step_and_tracking_loop_carry_orders_per_view(hip):
    """The two loops turn the carried order on by themselves (one array per view in MappingStep); their results equal the
    loops over plain workspaces bit for bit."""
    from online_lang_splatting_amd.frame_shard import FrameLanes
    from online_lang_splatting_amd.slam_iterations import MappingStep
    dev = torch.device(DEV)
    P, W, H, F = 20000, 240, 180, 15
    sc = make_scene(P, W, H, F, seed=61)
    cams = [_cam(default_camera(W, H, yaw, 0.01 * i), dev) for i, yaw in enumerate((0.0, 2.0, -3.0, 5.0, 1.0))]
    gen = torch.Generator(device="cpu").manual_seed(9)
    targets = [(torch.rand(3, H, W, generator=gen), torch.rand(H, W, generator=gen) + 1.0, torch.rand(F, 48, 48, generator=gen))
               for _ in cams]
    lrs = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)
    res = {}
    for carry_on in (False, True):
        params = dict(means3D=sc.means3D.to(dev).clone(), opacities=sc.opacities.to(dev).clone(),
                      scales=sc.scales.to(dev).clone(), rotations=sc.rotations.to(dev).clone(), shs=sc.shs.to(dev).clone(),
                      language=sc.language.to(dev).clone())
        lanes = FrameLanes(2, P, W, H, F, sc.shs.shape[1], 1_500_000, dev)
        ms = MappingStep(lanes, params, sc.bg.to(dev), sc.sh_degree, cams, targets, lrs, activations=0, fused_loss=True,
                         carry_order=carry_on)
        for _ in range(4):
            ms.iteration()
        torch.cuda.synchronize()
        res[carry_on] = {k: v.clone() for k, v in params.items()}
        if carry_on:
            assert len(ms.view_orders) == len(cams)
    for k in res[False]:
        assert torch.equal(res[True][k], res[False][k]), k
