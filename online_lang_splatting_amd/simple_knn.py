"""Host side of olsr_knn_mean_dist2 (include/olsr.h): drop-in for simple_knn._C.distCUDA2
(/root/reference/submodules/simple-knn/spatial.cu:15-26), the reference's second native dependency
(Gaussian scale initialisation, gaussian_splatting/scene/gaussian_model.py:256-263).  GPU only."""
import ctypes as C

import torch

from ._lib import check, lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on the GPU -> [P] mean squared distance to the 3 nearest neighbours."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on the GPU (there is no CPU fallback)")
    pts = points.contiguous().to(torch.float32)
    P = pts.shape[0]
    out = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)  # torch::full({P}, 0.0), spatial.cu:21
    if P == 0:
        return out
    L = lib()
    scratch = torch.empty(L.olsr_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        check(L.olsr_knn_mean_dist2(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return out
