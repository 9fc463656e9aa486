"""Intra-window temporal split: ONE long window sharded over the ranks by contiguous frame ranges.

Clip-parallel inference (clip_parallel.py) gives every GPU its own CLI window, so results equal the single-GPU run with
the same ``one_len``.  To run a LONGER window than one GPU should hold (upstream: "one_len according to GPU memory",
README.md:48) with results equal to the single-GPU long-window run, the window itself is split: rank r holds frames
``[a_r, b_r)`` of every activation.  The network mixes frames only in ``channel_shift`` (gshift_deblur1.py:504-518) and
in Shift_CAB's roll (gshift_denoise1.py:167-179): a forward unit reads the upper half-channels of frame t-1, a reverse
unit the lower half-channels of frame t+1.  So before every shifted unit each rank sends ONE half-frame ``[h, w, C/2]``
of the unit's input to one neighbour and receives one (41.5 MB at level 1 / 1080p / C = 80 in bf16; 56 (48) exchanges per
forward), deblur2's circular roll closing the ring between the last and the first rank (gshift_deblur2.py:504-505).

The received half-frame is a contiguous ``[h, w, C/2]`` buffer of its own -- the receive buffer of the exchange -- handed to the
kernels through ``sn_unit_src.halo`` together with ``wrap == 2`` (csrc/sn_common.h: sn_unit_slabs): no halo-padded allocations, no copy
into a strided slot.  The sender packs its strided half-frame once (RCCL / gloo send contiguous memory).  Nothing else in the engine
changes; stage 2 trims ``past`` frames on the first rank and ``future`` frames on the last only.

Overlap: a shifted unit needs the neighbour's half-frame only for ONE of its frames (local frame 0 of a forward unit, the last one of a
reverse unit).  The engine therefore posts the exchange on a side stream as soon as the unit's input exists, runs the unit's CAB2 for
all other frames on the compute stream meanwhile, and waits for the halo only before the boundary frame's launches
(``Engine.naf`` with ``split`` set; ``sn_unit_src`` addresses frame ranges through its base pointer, T and wrap flag).
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

    # Which rank owns the frame a boundary frame borrows from (None = the window boundary, where the reference keeps the frame
    # un-rolled).  `circular` is the roll's own rule: GSTS units of deblur2 close the ring (gshift_deblur2.py:504-505), every other unit and
    # Shift_CAB's roll (gshift_denoise1.py:167-179) do not -- also on a module whose GSTS units are circular.
    def prev_rank(self, circular: bool) -> Optional[int]:
        if self.rank > 0:
            return self.rank - 1
        return self.world - 1 if (circular and self.world > 1) else None

    def next_rank(self, circular: bool) -> Optional[int]:
        if self.rank < self.world - 1:
            return self.rank + 1
        return 0 if (circular and self.world > 1) else None

    def wrap_flag(self, mode: int, circular: bool) -> int:
        """``sn_unit_src.wrap`` for an operator of direction ``mode`` (1 forward, 2 reverse) on this rank."""
        nb = self.prev_rank(circular) if mode == 1 else self.next_rank(circular)
        if nb is not None:
            return 2                                  # the neighbour frame's half arrives from that rank
        return 1 if circular else 0                   # (single rank) circular roll / kept boundary frame

    def validate(self, n_local_frames: int) -> None:
        """Collective check, once per forward and BEFORE any exchange: every rank must hold at least one frame.  A rank that raised
        on its own while the others entered the exchanges would leave them hanging."""
        flag = torch.tensor([1 if n_local_frames < 1 else 0], dtype=torch.int32)
        if dist.get_backend(self.group) != "gloo":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if int(flag.item()):
            raise ValueError("temporal split: every rank must hold at least one frame of the window (the halo exchanges are collective)")

    def any_rank(self, flag: int) -> int:
        """Collective OR of a per-rank flag (the engine's range guard: all ranks recompute a window together or none does)."""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend(self.group) != "gloo":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def exchange(self, x: torch.Tensor, mode: int, circular: bool) -> Optional[torch.Tensor]:
        """One halo exchange for a unit of direction ``mode`` on input ``x`` ([T,h,w,C], NHWC): returns the neighbour's half-frame as a
        contiguous [h,w,C/2] tensor (None where this rank has no neighbour on that side) and feeds the other neighbour.

        forward: the upper half-channels of my LAST frame go to the next rank; I receive the previous rank's.
        reverse: the lower half-channels of my FIRST frame go to the previous rank; I receive the next rank's."""
        T, h, w, C = x.shape
        Ch = C // 2
        if mode == 1:
            dst, src = self.next_rank(circular), self.prev_rank(circular)
            send = x[T - 1, :, :, Ch:] if dst is not None else None
        else:
            dst, src = self.prev_rank(circular), self.next_rank(circular)
            send = x[0, :, :, :Ch] if dst is not None else None
        if send is None and src is None:
            return None
        staged = dist.get_backend(self.group) == "gloo" and x.is_cuda            # gloo moves host memory: stage through the CPU (one-GPU tests)
        ops, halo, rbuf = [], None, None
        if send is not None:
            sbuf = send.contiguous()                                             # the one pack: a half-frame is strided in NHWC
            if staged:
                sbuf = sbuf.cpu()
            ops.append(dist.P2POp(dist.isend, sbuf, dst, self.group))
        if src is not None:
            halo = torch.empty((h, w, Ch), dtype=x.dtype, device=x.device)
            rbuf = torch.empty((h, w, Ch), dtype=x.dtype, device="cpu") if staged else halo        # RCCL receives straight into the kernels' buffer
            ops.append(dist.P2POp(dist.irecv, rbuf, src, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if staged and halo is not None:
            halo.copy_(rbuf)
        return halo
