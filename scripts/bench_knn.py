"""Times olsr_knn_mean_dist2 (drop-in for simple_knn distCUDA2); prints one JSON line per size."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd.simple_knn import distCUDA2

for P in (50_000, 500_000, 2_000_000):
    g = torch.Generator().manual_seed(P)
    pts = (torch.rand(P, 3, generator=g) * torch.tensor([8.0, 5.0, 3.0])).cuda()
    for _ in range(3):
        distCUDA2(pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        distCUDA2(pts)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({"kernel": "olsr_knn_mean_dist2", "P": P, "ms": round(ms, 3), "Mpoints_per_s": round(P / ms / 1e3, 1)}))
