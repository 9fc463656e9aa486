"""Device-side I/O edges of the CLIs (csrc/sn_io.hip): uint8 frames in, uint8 frames and PSNR out.

``ingest_u8``  == ``numpy2tensor(frames).to(device).to(dtype)`` of inference/test_deblur.py:191-200,128,134, bit for bit,
               with 3 instead of 12 bytes per pixel crossing PCIe;
``egress_u8``  == the per-frame ``clamp(0,1) * 255`` -> skimage PSNR(data_range=255) against the uint8 ground truth
               (:139-143) and the rounded uint8 frame cv2.imwrite would store (:152).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from . import lib as L

_CODE = {torch.float32: L.SN_F32, torch.float16: L.SN_F16, torch.bfloat16: L.SN_BF16}


def ingest_u8(frames_u8: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """frames_u8: [T,H,W,3] uint8 on a HIP device -> [1,T,3,H,W] of ``dtype`` in [0,1]."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
    frames_u8 = frames_u8.contiguous()
    T, H, W, _ = frames_u8.shape
    x = torch.empty((1, T, 3, H, W), dtype=dtype, device=frames_u8.device)
    with torch.cuda.device(frames_u8.device):
        L.check(L.load().sn_ingest_u8(frames_u8.data_ptr(), x.data_ptr(), _CODE[dtype], T, H, W,
                                      torch.cuda.current_stream(frames_u8.device).cuda_stream), "sn_ingest_u8")
    return x


def egress_u8(out: torch.Tensor, gt_u8: Optional[torch.Tensor] = None, want_image: bool = True
              ) -> Tuple[Optional[torch.Tensor], Optional[List[float]]]:
    """out: [T,3,H,W] network output (module dtype) on the device; gt_u8: [T,H,W,3] uint8 on the device or None.

    Returns (uint8 frames [T,H,W,3] on the device or None, per-frame PSNR in dB or None)."""
    assert out.is_cuda and out.dim() == 4 and out.shape[1] == 3 and out.dtype in _CODE
    out = out.contiguous()
    T, _, H, W = out.shape
    lib = L.load()
    img = torch.empty((T, H, W, 3), dtype=torch.uint8, device=out.device) if want_image else None
    sse = None
    if gt_u8 is not None:
        assert gt_u8.dtype == torch.uint8 and tuple(gt_u8.shape) == (T, H, W, 3) and gt_u8.device == out.device
        gt_u8 = gt_u8.contiguous()
        sse = torch.empty((T, lib.sn_egress_blocks()), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        L.check(lib.sn_egress_u8(out.data_ptr(), _CODE[out.dtype], gt_u8.data_ptr() if gt_u8 is not None else None,
                                 img.data_ptr() if img is not None else None, sse.data_ptr() if sse is not None else None,
                                 T, H, W, torch.cuda.current_stream(out.device).cuda_stream), "sn_egress_u8")
    psnr = None
    if sse is not None:
        tot = sse.double().cpu().sum(1)             # T x 64 partial sums: summed in float64 on the host
        psnr = [float("inf") if s == 0 else 10.0 * math.log10(255.0 ** 2 / (s / (3 * H * W))) for s in tot.tolist()]
    return img, psnr


def ssim_u8(out: torch.Tensor, gt_u8: torch.Tensor) -> List[float]:
    """The CLIs' SSIM (test_deblur.py:25-49) per frame, on the device.  out: [T,3,H,W] module dtype, gt_u8: [T,H,W,3] uint8."""
    assert out.is_cuda and out.dim() == 4 and out.shape[1] == 3 and out.dtype in _CODE
    out = out.contiguous()
    T, _, H, W = out.shape
    assert gt_u8.dtype == torch.uint8 and tuple(gt_u8.shape) == (T, H, W, 3) and gt_u8.device == out.device
    gt_u8 = gt_u8.contiguous()
    lib = L.load()
    scratch = torch.empty((T, 15, H, W), dtype=torch.float32, device=out.device)
    part = torch.empty((T, lib.sn_ssim_blocks()), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        L.check(lib.sn_ssim_u8(out.data_ptr(), _CODE[out.dtype], gt_u8.data_ptr(), scratch.data_ptr(), part.data_ptr(), T, H, W,
                               torch.cuda.current_stream(out.device).cuda_stream), "sn_ssim_u8")
    return (part.double().cpu().sum(1) / (3.0 * H * W)).tolist()
