"""`from simple_knn._C import distCUDA2` -- same import path and call as the reference (scene/saro_gaussian.py:21, :187;
also imported by dataset_readers.py:27, helper_model.py:26, helper_train.py:42).

distCUDA2(points [P,3] float32 on the GPU) -> float32 [P]: mean squared distance of every point to its 3 nearest
neighbours.  HIP implementation in libgsrast_hip.so (`gsrast_knn3_mean_dist2`, csrc/gsrast_knn.h).  No fallback."""
import torch

from diff_gaussian_rasterization_ch3 import _C as _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a GPU (HIP) device; there is no CPU fallback")
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("distCUDA2: points must be [P, 3]")
    pts = points.contiguous().float()
    P = int(pts.shape[0])
    out = torch.empty((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = _lib.lib()
    scratch = torch.empty(L.gsrast_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = L.gsrast_knn3_mean_dist2(P, pts.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                      torch.cuda.current_stream(pts.device).cuda_stream)
    if rc != 0:
        raise _lib._err(rc, "gsrast_knn3_mean_dist2")
    return out
