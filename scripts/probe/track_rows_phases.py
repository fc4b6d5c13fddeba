"""Probe: one rank (collectives forced on), 2 lanes, three exchange phases on the same lanes: tracked vs untracked flat after every step."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_frame_shard as T
from online_lang_splatting_amd.frame_shard import FrameLanes, FrameShardedStep, FusedAdam, GradLayout, GradientBucket

os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29577"
dist.init_process_group("gloo", rank=0, world_size=1)
GradientBucket.exchange_single_rank = True
dev = torch.device("cuda:0")
sc, g, cams, cot = T._inputs(dev, T.P8, T.V8)
M = sc.shs.shape[1]
res = {}
for track in (False, True):
    ws = FrameLanes(2, T.P8, T.W, T.H, T.F, M, 400000, dev, track_rows=track)
    log = []
    for exchange in ("all_reduce", "sparse", "reduce_scatter"):
        st = FrameShardedStep(ws, 0, 1, exchange=exchange)
        params = {k: v.clone() for k, v in g.items() if k != "bg"}
        adam = FusedAdam(T.P8, GradLayout(M, T.F), dev)
        for it in range(3):
            bucket = st.run(dict(bg=g["bg"], **params), cams[:3], lambda v, out: cot[v], sh_degree=sc.sh_degree)
            torch.cuda.synchronize()
            log.append((exchange, it, bucket.flat.clone(), ws.lanes[1][1].flat.clone()))
            if track:
                for li in (0, 1):
                    b_ = ws.lanes[li][1]
                    nz = (b_.flat != 0).any(dim=1)
                    bits = ((b_.row_mask.view(-1, 1) >> torch.arange(64, device=dev)) & 1).reshape(-1)[:T.P8].bool()
                    print(f"  [{exchange} {it}] lane {li}: nonzero rows {int(nz.sum())}, mask bits {int(bits.sum())}, nonzero rows outside the mask {int((nz & ~bits).sum())}")
            st.optimizer_step(adam, params, T.LRS)
    res[track] = log
for a, b in zip(res[False], res[True]):
    print(a[0], a[1], "total flat equal", torch.equal(a[2], b[2]), float((a[2] - b[2]).abs().max()), "| lane1 flat equal", torch.equal(a[3], b[3]),
          "rows differing", int(((a[2] != b[2]).any(dim=1)).sum()))
# which way do the lane-1 rows differ at the first differing step?
for a, b in zip(res[False], res[True]):
    if not torch.equal(a[3], b[3]):
        d = (a[3] != b[3]).any(dim=1)
        za, zb = (a[3] == 0).all(dim=1), (b[3] == 0).all(dim=1)
        print(a[0], a[1], "lane1 rows differing", int(d.sum()), "| untracked zero & tracked non-zero (stale):", int((d & za & ~zb).sum()),
              "| untracked non-zero & tracked zero (missing):", int((d & ~za & zb).sum()), "| both non-zero:", int((d & ~za & ~zb).sum()))
        idx = torch.nonzero(d).reshape(-1)[:8].tolist()
        print("  first rows:", idx, "blocks of 128:", [i // 128 for i in idx], "lane in wave:", [i % 64 for i in idx])
        break
