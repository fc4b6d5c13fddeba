"""Which rows / columns of the gradient bucket are non-zero after one view (config 3)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from online_lang_splatting_amd import _abi
from online_lang_splatting_amd.frame_shard import GradientBucket, GradLayout, RasterWorkspace
from online_lang_splatting_amd.scene import make_config_scene
dev = torch.device("cuda:0")
sc = make_config_scene(3)
cam = sc.camera
P, W, H, F, M = sc.P, cam.width, cam.height, sc.F, sc.shs.shape[1]
g_dev, c = bench.device_inputs(sc, cam, dev)
R0 = bench._sized_capacity(F, g_dev, c, H, W, sc.sh_degree, dev, (15, _abi.BWD_REFERENCE, _abi.BINNING_ELLIPSE))
ws = RasterWorkspace(P, W, H, F, M, int(1.4 * R0) + (1 << 16), dev)
b = GradientBucket(P, GradLayout(M, F), dev)
dc, dl, dd = [t.to(dev) for t in sc.cotangents(3)]
ws.set_scene(sh_degree=sc.sh_degree, **c, **g_dev)
ws.forward()
ws.backward(dc, dl, dd, bucket=b, first=True, bucket_only=True)
torch.cuda.synchronize()
nz = b.flat != 0
print("rows non-zero:", int(nz.any(1).sum()), "of", P, "| per column:", nz.sum(0).tolist())
print("nan rows:", int(torch.isnan(b.flat).any(1).sum()))
