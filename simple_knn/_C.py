"""``distCUDA2(points[N,3]) -> mean squared distance to the 3 nearest neighbours [N]`` -- same name and contract as
the CUDA extension pinned at reference README.md:29 and called at reference scene/gaussian_model.py:105."""
import ctypes as C

import torch

from splatfields_amd import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 has no CPU path: points must be on a HIP ('cuda') device")
    pts = points.detach().to(torch.float32).contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    with torch.cuda.device(pts.device):
        ws = torch.empty(lib.sr_knn_workspace_bytes(n), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.sr_knn3_mean_dist2(n, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return out
