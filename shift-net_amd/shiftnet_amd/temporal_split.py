"""Intra-window temporal split: ONE long window sharded over the ranks by contiguous frame ranges.

Clip-parallel inference (clip_parallel.py) gives every GPU its own CLI window, so results equal the single-GPU run with
the same ``one_len``.  To run a LONGER window than one GPU should hold (upstream: "one_len according to GPU memory",
README.md:48) with results equal to the single-GPU long-window run, the window itself is split: rank r holds frames
``[a_r, b_r)`` of every activation.  The network mixes frames only in ``channel_shift`` (gshift_deblur1.py:504-518) and
in Shift_CAB's roll (gshift_denoise1.py:167-179): a forward unit reads the upper half-channels of frame t-1, a reverse
unit the lower half-channels of frame t+1.  So before every shifted unit each rank sends ONE half-frame ``[h, w, C/2]``
of the unit's input to one neighbour and receives one (41.5 MB at level 1 / 1080p / C = 80 in bf16; 56 (48) exchanges per
forward), deblur2's circular roll closing the ring between the last and the first rank (gshift_deblur2.py:504-505).

The received half-frame lands in a HALO slot just outside the local tensor (every activation of a split engine is
allocated with one spare frame on each side), and the kernels are told so through ``sn_unit_src.wrap == 2``: the
neighbour of local frame 0 / T-1 is then frame index -1 / T (csrc/sn_common.h: sn_prev_frame / sn_next_frame).  Nothing
else in the engine changes; stage 2 trims ``past`` frames on the first rank and ``future`` frames on the last only.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def partition(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal frame ranges [start, stop) per rank (earlier ranks take the remainder)."""
    q, r = divmod(n_frames, world)
    out, a = [], 0
    for k in range(world):
        b = a + q + (1 if k < r else 0)
        out.append((a, b))
        a = b
    return out


class TemporalSplit:
    def __init__(self, rank: int, world: int, circular: bool, group=None) -> None:
        assert 0 <= rank < world
        self.rank, self.world, self.circular, self.group = rank, world, circular, group

    # which neighbour exists for this rank (None = the window boundary, where the reference keeps the frame un-rolled)
    def prev_rank(self) -> Optional[int]:
        if self.rank > 0:
            return self.rank - 1
        return self.world - 1 if (self.circular and self.world > 1) else None

    def next_rank(self) -> Optional[int]:
        if self.rank < self.world - 1:
            return self.rank + 1
        return 0 if (self.circular and self.world > 1) else None

    def wrap_flag(self, mode: int) -> int:
        """``sn_unit_src.wrap`` for a unit of direction ``mode`` (1 forward, 2 reverse) on this rank."""
        nb = self.prev_rank() if mode == 1 else self.next_rank()
        if nb is not None:
            return 2                                  # neighbour frame in the halo slot
        return 1 if self.circular else 0              # single-rank circular roll / kept boundary frame

    @staticmethod
    def halo_base(x: torch.Tensor) -> torch.Tensor:
        """The [T+2, h, w, C] allocation a unit input [T, h, w, C] is the middle of."""
        base = x._base
        if base is None or base.dim() != 4 or base.shape[0] != x.shape[0] + 2 or x.storage_offset() != base.storage_offset() + base.stride(0):
            raise RuntimeError("temporal split: unit inputs must come from a halo-padded allocation (Engine._new)")
        return base

    def exchange(self, x: torch.Tensor, mode: int) -> None:
        """Fill this rank's halo slot of ``x`` ([T,h,w,C], NHWC) for a unit of direction ``mode`` and feed the neighbour's.

        forward: the upper half-channels of my LAST frame go to the next rank's slot -1;
        reverse: the lower half-channels of my FIRST frame go to the previous rank's slot T."""
        base = self.halo_base(x)
        T, _, _, C = x.shape
        Ch = C // 2
        if mode == 1:
            dst, src = self.next_rank(), self.prev_rank()
            send = x[T - 1, :, :, Ch:] if dst is not None else None
            slot = base[0, :, :, Ch:] if src is not None else None
        else:
            dst, src = self.prev_rank(), self.next_rank()
            send = x[0, :, :, :Ch] if dst is not None else None
            slot = base[T + 1, :, :, :Ch] if src is not None else None
        if send is None and slot is None:
            return
        staged = dist.get_backend(self.group) == "gloo" and x.is_cuda           # gloo moves host memory: stage through the CPU
        ops, rbuf = [], None
        if send is not None:
            sbuf = send.contiguous()
            if staged:
                sbuf = sbuf.cpu()
            ops.append(dist.P2POp(dist.isend, sbuf, dst, self.group))
        if slot is not None:
            rbuf = torch.empty(slot.shape, dtype=x.dtype, device="cpu" if staged else x.device)
            ops.append(dist.P2POp(dist.irecv, rbuf, src, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if slot is not None:
            slot.copy_(rbuf)
