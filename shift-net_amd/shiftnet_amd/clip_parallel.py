"""Clip-parallel inference: one sliding window per GPU, one exchange of the 2-frame temporal halo.

The reference CLI restores frames ``[kL+2, kL+2+L)`` of a clip from input frames ``[kL, kL+L+4)`` (test_deblur.py:
111-120).  Rank r therefore OWNS the L frames it restores and needs the last two owned frames of rank r-1 and the first
two of rank r+1; rank 0 / the last rank additionally hold the clip's first / last two frames, which nobody restores.
That is the only communication on the path: one all-gather of ``[4,3,H,W]`` raw input frames per rank over RCCL (xGMI);
no feature map ever crosses GPUs, because the result must equal the single-GPU CLI run with the same ``one_len``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def window_ranges(n_frames: int, one_len: int) -> List[Tuple[range, range]]:
    """(input range, restored range) per window exactly as the CLI slices them; remainder frames are dropped."""
    k_len = (n_frames - 4) // one_len
    return [(range(k * one_len, k * one_len + one_len + 4), range(k * one_len + 2, k * one_len + 2 + one_len))
            for k in range(k_len)]


def assemble_window(own: torch.Tensor, first_edge: Optional[torch.Tensor], last_edge: Optional[torch.Tensor],
                    rank: int = 0, world: int = 1, group=None, active: Optional[int] = None) -> Optional[torch.Tensor]:
    """own:[L,3,H,W] -> this rank's input window [L+4,3,H,W].

    ``first_edge`` ([2,3,H,W]) is required on rank 0, ``last_edge`` on the last ACTIVE rank (``active`` ranks hold a window in this round,
    default all; the others still take part in the collective -- ``own`` is then any tensor of the right shape -- and get None).  Without a
    process group (plain single-GPU run) there is no communication at all.  The collective runs where ``own`` lives (RCCL for device
    tensors; a gloo group -- several ranks sharing one device in the tests -- needs host tensors: pass them as such).
    """
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return torch.cat((first_edge, own, last_edge), 0)
    active = world if active is None else active
    # (a one-rank process group still goes through the collective: the RCCL path is exercised on a single-GPU box too)
    send = torch.cat((own[:2], own[-2:]), 0).contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(bufs, send, group=group)
    if rank >= active:
        return None
    head = first_edge if rank == 0 else bufs[rank - 1][2:4]
    tail = last_edge if rank == active - 1 else bufs[rank + 1][0:2]
    return torch.cat((head, own, tail), 0)


def rounds(n_windows: int, world: int) -> List[List[int]]:
    """Window indices per lock-step round of a clip-parallel CLI run: round i = windows [i * world, (i + 1) * world), rank r takes the r-th."""
    return [list(range(i, min(i + world, n_windows))) for i in range(0, n_windows, world)]
