"""FrameShardedStep on the GPU: two ranks (gloo, sharing the one device of the test box — on a multi-GPU node the same
code runs one rank per GPU over RCCL) shard the views of one mapping step; the exchanged bucket must equal the
single-process result of the same views bit for bit (fixed summation order: rank 0's views, then rank 1's), all three
exchange modes must leave the same parameters after the optimiser step, and an overflow on one rank must surface on
both."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
P, W, H, F, V = 6000, 200, 150, 15, 5
LRS = dict(xyz=1.6e-4, sh_dc=2.5e-3, sh_rest=1.25e-4, opacity=0.05, scale=1e-3, rotation=1e-3, language=2.5e-3)


def _inputs(dev, P=P, V=V):
    from online_lang_splatting_amd.scene import arc_cameras, make_scene
    sc = make_scene(P, W, H, F, seed=17)
    g = dict(bg=sc.bg.to(dev), means3D=sc.means3D.to(dev), opacities=sc.opacities.to(dev), scales=sc.scales.to(dev),
             rotations=sc.rotations.to(dev), shs=sc.shs.to(dev), language=sc.language.to(dev))
    cams = [dict(viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev),
                 projmatrix_raw=c.projection_matrix.to(dev), campos=c.camera_center.to(dev), tanfovx=c.tanfovx,
                 tanfovy=c.tanfovy) for c in arc_cameras(W, H, V)]
    cot = [tuple(t.to(dev) for t in sc.cotangents(100 + v)) for v in range(V)]
    return sc, g, cams, cot


def _step(rank, world, exchange, capacity, dev):
    from online_lang_splatting_amd.frame_shard import FrameShardedStep, FusedAdam, GradLayout, RasterWorkspace
    sc, g, cams, cot = _inputs(dev)
    ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], capacity, dev)
    st = FrameShardedStep(ws, rank, world, exchange=exchange)
    bucket = st.run(g, cams, lambda v, out: cot[v], sh_degree=sc.sh_degree)
    params = dict(means3D=g["means3D"].clone(), shs=g["shs"].clone(), opacities=g["opacities"].clone(),
                  scales=g["scales"].clone(), rotations=g["rotations"].clone(), language=g["language"].clone())
    adam = FusedAdam(P, GradLayout(sc.shs.shape[1], F), dev)
    st.optimizer_step(adam, params, LRS)
    torch.cuda.synchronize()
    return st, bucket, params


def _worker(rank, world, port, exchange, capacity, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    try:
        st, bucket, params = _step(rank, world, exchange, capacity, dev)
        ret[f"flat{rank}"] = bucket.flat.cpu()
        ret[f"densify{rank}"], ret[f"radii{rank}"] = bucket.densify.cpu(), bucket.max_radii.cpu()
        ret[f"params{rank}"] = {k: v.cpu() for k, v in params.items()}
        ret[f"owned{rank}"] = st.owned
        ret[f"poses{rank}"] = {v: t.cpu() for v, t in st.pose_grads.items()}
        if st.wire is not None:
            ret[f"wire{rank}"] = st.wire
    except OverflowError as e:
        ret[f"overflow{rank}"] = str(e)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(exchange, capacity=400000):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, exchange, capacity, ret), nprocs=2, join=True)
    return dict(ret)


def _single_process_reference(dev):
    """The same five views in one process, summed in the order the two ranks produce: (v0 + v2 + v4) + (v1 + v3)."""
    from online_lang_splatting_amd.frame_shard import GradLayout, GradientBucket, RasterWorkspace
    sc, g, cams, cot = _inputs(dev)
    ws = RasterWorkspace(P, W, H, F, sc.shs.shape[1], 400000, dev)
    parts, poses = [], {}
    for rank in range(2):
        b = GradientBucket(P, GradLayout(sc.shs.shape[1], F), dev)
        for n, v in enumerate(range(rank, V, 2)):
            ws.set_scene(sh_degree=sc.sh_degree, **cams[v], **g)
            ws.forward()
            gr = ws.backward(*cot[v], bucket=b, first=(n == 0), bucket_only=True)
            poses[v] = gr["dL_dtau_sum"].clone().cpu()
        parts.append(b)
    torch.cuda.synchronize()
    return (parts[0].flat + parts[1].flat).cpu(), (parts[0].densify + parts[1].densify).cpu(), \
        torch.maximum(parts[0].max_radii, parts[1].max_radii).cpu(), poses


def test_two_ranks_equal_the_single_process_sum(hip):
    dev = torch.device("cuda:0")
    flat, densify, radii, poses = _single_process_reference(dev)
    assert float(flat.abs().max()) > 0 and int((densify[:, 1] == V).sum()) > 0
    res = {m: _spawn(m) for m in ("all_reduce", "sparse", "reduce_scatter")}
    for m in ("all_reduce", "sparse"):
        for r in range(2):
            assert torch.equal(res[m][f"flat{r}"], flat), (m, r)
            assert torch.equal(res[m][f"densify{r}"], densify) and torch.equal(res[m][f"radii{r}"], radii), (m, r)
    w = res["sparse"]["wire0"]
    # the sparse exchange carries the gradient rows that are non-zero somewhere: a subset of the Gaussians some view saw
    assert w["active_rows"] == int((flat != 0).any(1).sum()) <= int((densify[:, 1] > 0).sum())
    assert w["bytes_sparse"] < w["bytes_dense"]
    for r in range(2):  # owner-applies: the rows a rank owns hold the total
        r0, r1 = res["reduce_scatter"][f"owned{r}"]
        assert torch.equal(res["reduce_scatter"][f"flat{r}"][r0:r1], flat[r0:r1])
    # pose gradients stay with the rank that rendered the view
    for r in range(2):
        assert sorted(res["all_reduce"][f"poses{r}"]) == list(range(r, V, 2))
        for v, t in res["all_reduce"][f"poses{r}"].items():
            assert torch.equal(t, poses[v])
    # one optimiser step later every rank holds the same parameters, whatever the exchange
    ref = res["all_reduce"]["params0"]
    for m in res:
        for r in range(2):
            for k in ref:
                assert torch.equal(res[m][f"params{r}"][k], ref[k]), (m, r, k)
    assert not torch.equal(ref["means3D"], _inputs(torch.device("cpu"))[1]["means3D"])  # ... and they did move


def test_overflow_on_any_rank_surfaces_on_every_rank(hip):
    res = _spawn("all_reduce", capacity=2000)
    assert "overflow0" in res and "overflow1" in res and "capacity >=" in res["overflow0"]


# ---- eight ranks (the driver's 8-GPU shape) sharing the one device of the test box -------------------------------------
P8, V8 = 6001, 12  # a Gaussian count 8 does not divide (ragged owned rows), the 12 views of a mapping iteration


def _worker8(rank, world, port, ret):
    from online_lang_splatting_amd.frame_shard import FrameLanes, FrameShardedStep, FusedAdam, GradLayout
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    sc, g, cams, cot = _inputs(dev, P8, V8)
    M = sc.shs.shape[1]
    ws = FrameLanes(2, P8, W, H, F, M, 400000, dev)  # a rank's two views are in flight together (ranks 0-3)
    for exchange in ("all_reduce", "sparse", "reduce_scatter"):
        st = FrameShardedStep(ws, rank, world, exchange=exchange)
        params = {k: v.clone() for k, v in g.items() if k != "bg"}
        adam = FusedAdam(P8, GradLayout(M, F), dev)
        for _ in range(3):  # three optimisation steps: the parameters every rank renders from must stay identical
            bucket = st.run(dict(bg=g["bg"], **params), cams, lambda v, out: cot[v], sh_degree=sc.sh_degree)
            st.optimizer_step(adam, params, LRS)
        torch.cuda.synchronize()
        ret[f"{exchange}:params{rank}"] = {k: v.cpu() for k, v in params.items()}
        ret[f"{exchange}:owned{rank}"] = st.owned
        ret[f"{exchange}:views{rank}"] = sorted(st.pose_grads)
        if rank == 0:
            ret[f"{exchange}:flat"] = bucket.flat.cpu()
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_three_steps_every_exchange(hip):
    """12 views on 8 ranks (ranks 0-3 render two views, 4-7 one), ragged owned rows, three optimisation steps per
    exchange mode: every rank holds bit-identical parameters afterwards, the three exchanges agree with each other to
    summation-order noise, each view's pose gradient stays on exactly one rank, and the parameters moved."""
    from online_lang_splatting_amd.frame_shard import GradientBucket
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, port, ret), nprocs=world, join=True)
    res = dict(ret)
    start = _inputs(torch.device("cpu"), P8, V8)[1]
    for m in ("all_reduce", "sparse", "reduce_scatter"):
        ref = res[f"{m}:params0"]
        for r in range(1, world):
            for k in ref:
                assert torch.equal(res[f"{m}:params{r}"][k], ref[k]), (m, r, k)
        assert not torch.equal(ref["means3D"], start["means3D"])
        views = sorted(v for r in range(world) for v in res[f"{m}:views{r}"])
        assert views == list(range(V8))
        assert [len(res[f"{m}:views{r}"]) for r in range(world)] == [2, 2, 2, 2, 1, 1, 1, 1]
    covered = []
    for r in range(world):
        r0, r1 = res[f"reduce_scatter:owned{r}"]
        assert (r0, r1) == GradientBucket.owned_rows(P8, r, world)
        covered += list(range(r0, r1))
    assert covered == list(range(P8)) and P8 % world != 0
    for k in res["all_reduce:params0"]:
        for m in ("sparse", "reduce_scatter"):
            torch.testing.assert_close(res[f"{m}:params0"][k], res["all_reduce:params0"][k], rtol=2e-4, atol=2e-5)


# ---- RCCL itself: every collective of the exchanges in a group of ONE rank over the "nccl" backend --------------------
def _rccl_worker(rank, world, port, ret):
    """A one-GPU box cannot run two RCCL ranks (one device per rank), but it can run ONE: the collectives are identities
    there, and issuing them proves that every call the exchanges make — dtypes (fp32 SUM, int32 MAX, uint8 MAX), the
    reduce_scatter_tensor / all_gather_into_tensor shapes incl. the ragged tail, the async handles — is accepted and
    executed by RCCL on this stack, on the caller's stream, with the result left where the exchange promises it."""
    from online_lang_splatting_amd.frame_shard import FrameShardedStep, FusedAdam, GradientBucket, GradLayout, RasterWorkspace
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    GradientBucket.exchange_single_rank = True
    try:
        sc, g, cams, cot = _inputs(dev, P8, V)   # P8 = 6001: also odd, so every padded path runs
        M = sc.shs.shape[1]
        ws = RasterWorkspace(P8, W, H, F, M, 400000, dev)
        out = {}
        for exchange in ("all_reduce", "sparse", "reduce_scatter", "auto"):
            st = FrameShardedStep(ws, 0, 1, exchange=exchange)
            params = {k: v.clone() for k, v in g.items() if k != "bg"}
            adam = FusedAdam(P8, GradLayout(M, F), dev)
            bucket = st.run(dict(bg=g["bg"], **params), cams, lambda v, o: cot[v], sh_degree=sc.sh_degree)
            st.optimizer_step(adam, params, LRS)
            torch.cuda.synchronize()
            out[exchange] = (bucket.flat.cpu().clone(), {k: v.cpu() for k, v in params.items()})
        # the bucket-level forms bench.py's weak-scaling mode uses
        b = st.bucket
        before = b.flat.clone()
        for w_ in b.all_reduce(async_op=True):
            w_.wait()
        status = b.sparse_all_reduce_capped(4096).cpu()
        b.reduce_scatter_all_gather(0, 1)
        torch.cuda.synchronize()
        ret["weak_forms_identity"] = bool(torch.equal(b.flat, before))
        ret["capped_status"] = status.tolist()
        # ... and the same three forms through RCCL's C API on the caller's stream (rccl_direct.DirectComm): a communicator of its
        # own, bootstrapped through this process group; identities in a group of one rank, every dtype / op the exchanges use
        from online_lang_splatting_amd.rccl_direct import DirectComm
        dc = DirectComm.from_process_group()
        GradientBucket.direct_comm = dc
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):   # (on whatever stream is current: no hop)
                b.all_reduce()
                status_d = b.sparse_all_reduce_capped(4096)
                b.reduce_scatter_all_gather(0, 1)
                x = torch.arange(1024, dtype=torch.float32, device=dev)
                dc.all_reduce(x, "sum")
                y = torch.arange(1024, dtype=torch.int32, device=dev)
                dc.all_reduce(y, "max")
            side.synchronize()
            ret["direct_identity"] = bool(torch.equal(b.flat, before)) and status_d.cpu().tolist() == status.tolist() and \
                bool(torch.equal(x, torch.arange(1024, dtype=torch.float32, device=dev))) and \
                bool(torch.equal(y, torch.arange(1024, dtype=torch.int32, device=dev)))
        finally:
            GradientBucket.direct_comm = None
            dc.destroy()
        ret["nonzero_rows"] = int((before != 0).any(1).sum())
        ret["backend"] = dist.get_backend()
        ret["same_bucket"] = all(torch.equal(out[m][0], out["all_reduce"][0]) for m in out)
        ret["same_params"] = all(torch.equal(out[m][1][k], out["all_reduce"][1][k]) for m in out for k in out[m][1])
        ret["moved"] = not torch.equal(out["all_reduce"][1]["means3D"], g["means3D"].cpu())
    finally:
        GradientBucket.exchange_single_rank = False
        dist.barrier()
        dist.destroy_process_group()


def test_every_exchange_collective_runs_on_rccl_single_rank(hip):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    r = dict(ret)
    assert r["backend"] == "nccl"
    assert r["same_bucket"] and r["same_params"] and r["moved"] and r["weak_forms_identity"] and r["direct_identity"]
    assert r["capped_status"] == [r["nonzero_rows"], int(r["nonzero_rows"] > 4096)] and r["nonzero_rows"] > 0
