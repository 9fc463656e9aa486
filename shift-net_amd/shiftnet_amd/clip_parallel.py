"""Clip-parallel inference: one sliding window per GPU, one exchange of the 2-frame temporal halo.

The reference CLI restores frames ``[kL+2, kL+2+L)`` of a clip from input frames ``[kL, kL+L+4)`` (test_deblur.py:
111-120).  Rank r therefore OWNS the L frames it restores and needs the last two owned frames of rank r-1 and the first
two of rank r+1; rank 0 / the last rank additionally hold the clip's first / last two frames, which nobody restores.
That is the only communication on the path: two raw input frames ``[2,3,H,W]`` to each neighbour (one batched send / recv pair per side,
RCCL point-to-point over the xGMI link to that neighbour; round 4 all-gathered every rank's 4 frames to every rank: 8 x 50 MB at
1080p where 2 x 25 MB suffice); no feature map ever crosses GPUs, because the result must equal the single-GPU CLI run with the
same ``one_len``.
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
                    rank: int = 0, world: int = 1, group=None, active: Optional[int] = None, all_gather: bool = False) -> Optional[torch.Tensor]:
    """own:[L,3,H,W] -> this rank's input window [L+4,3,H,W].

    ``first_edge`` ([2,3,H,W]) is required on rank 0, ``last_edge`` on the last ACTIVE rank (``active`` ranks hold a window in this round,
    default all; the others get None and -- point-to-point -- take no part at all).  Without a process group (plain single-GPU run) there is
    no communication.  The exchange runs where ``own`` lives (RCCL for device tensors; a gloo group -- several ranks sharing one device in the
    tests -- needs host tensors: pass them as such).  L >= 2: a window's halo is two frames of ONE neighbour (a shorter window would need
    the neighbour's neighbour; the CLI refuses one_len < 2 under --gpus).
    all_gather=True keeps round 4's collective form (every rank's 4 edge frames to every rank); a one-rank group uses it so that the RCCL
    path is exercised on a single-GPU box (tests/test_gpu_multirank.py), where point-to-point has nobody to talk to.  `Halo` (below) picks
    the form per process group and falls back from point to point to the collective on its own.
    """
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return torch.cat((first_edge, own, last_edge), 0)
    active = world if active is None else active
    if world > 1 and own.shape[0] < 2:
        raise ValueError("clip-parallel windows need one_len >= 2: the two halo frames of a side come from ONE neighbour rank")
    if all_gather or world == 1:
        send = torch.cat((own[:2], own[-2:]), 0).contiguous()
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(bufs, send, group=group)
        if rank >= active:
            return None
        head = first_edge if rank == 0 else bufs[rank - 1][2:4]
        tail = last_edge if rank == active - 1 else bufs[rank + 1][0:2]
        return torch.cat((head, own, tail), 0)
    if rank >= active:
        return None
    staged = dist.get_backend(group) == "gloo" and own.is_cuda      # gloo moves host memory: stage through the CPU (ranks sharing one device in tests)
    edge = lambda t: (t.cpu() if staged else t.contiguous())         # noqa: E731
    recv = lambda: torch.empty((2,) + tuple(own.shape[1:]), dtype=own.dtype, device="cpu" if staged else own.device)      # noqa: E731
    ops, head, tail = [], first_edge, last_edge
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)      # P2POp addresses GLOBAL ranks
    if rank > 0:                                    # my first two frames are the left neighbour's tail; its last two are my head
        head = recv()
        ops.append(dist.P2POp(dist.isend, edge(own[:2]), peer(rank - 1), group))
        ops.append(dist.P2POp(dist.irecv, head, peer(rank - 1), group))
    if rank < active - 1:
        tail = recv()
        ops.append(dist.P2POp(dist.isend, edge(own[-2:]), peer(rank + 1), group))
        ops.append(dist.P2POp(dist.irecv, tail, peer(rank + 1), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return torch.cat((head.to(own.device), own, tail.to(own.device)), 0)


class Halo:
    """Which form the halo exchange of a clip-parallel run takes, decided once per process group.

    mode "p2p": two raw frames to / from each neighbour (`assemble_window`); "allgather": round 4's collective form; "auto" (default): p2p,
    and if the FIRST exchange raises on any rank -- every rank learns it through one all-reduce of a failure flag -- all ranks switch to the
    all-gather form together, say so once (`log`), and redo that exchange.  The first call must be made by every rank of the group (idle ranks
    of a partial round included; they get None back as from `assemble_window`).  A p2p exchange that neither completes nor raises within
    `timeout_s` cannot be recovered from inside the process (the communicator's stream is blocked): that raises with the advice to run with
    --halo allgather.  `form` is what ran ("p2p" / "allgather"); one-rank groups always use the collective form (nobody to talk to)."""

    def __init__(self, mode: str = "auto", timeout_s: float = 120.0, log=None) -> None:
        if mode not in ("auto", "p2p", "allgather"):
            raise ValueError(f"halo mode {mode!r}: expected auto, p2p or allgather")
        self.mode, self.timeout_s, self.log = mode, timeout_s, log
        self.form: Optional[str] = "allgather" if mode == "allgather" else None
        self.fell_back = False

    def _completed(self, own: torch.Tensor) -> None:
        if not own.is_cuda:
            return
        import time
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(own.device))
        t0 = time.time()
        while not ev.query():
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError(f"clip-parallel halo exchange (point to point) did not complete within {self.timeout_s:.0f} s: "
                                   "re-run with --halo allgather")
            time.sleep(0.002)

    def assemble(self, own: torch.Tensor, first_edge, last_edge, rank: int = 0, world: int = 1, group=None, active: Optional[int] = None):
        if world == 1 or not (dist.is_available() and dist.is_initialized()):
            self.form = self.form or ("allgather" if dist.is_available() and dist.is_initialized() else "none")
            return assemble_window(own, first_edge, last_edge, rank, world, group, active)
        if self.form is not None:
            return assemble_window(own, first_edge, last_edge, rank, world, group, active, all_gather=self.form == "allgather")
        err: Optional[BaseException] = None
        win = None
        try:                                                     # the first exchange of this group: try point to point
            win = assemble_window(own, first_edge, last_edge, rank, world, group, active)
            self._completed(own)
        except TimeoutError:
            raise
        except Exception as e:                                   # noqa: BLE001
            err = e
        on_host = dist.get_backend(group) == "gloo" or not own.is_cuda
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device="cpu" if on_host else own.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            self.form = "p2p"
            return win
        if self.mode == "p2p":
            raise RuntimeError(f"clip-parallel halo exchange (point to point) failed on some rank (this rank: {err!r})")
        self.form, self.fell_back = "allgather", True
        if self.log is not None:
            self.log(f"clip-parallel: the point-to-point halo exchange failed ({'this rank: %r' % (err,) if err is not None else 'on another rank'}); "
                     "every rank now uses the all-gather form")
        return assemble_window(own, first_edge, last_edge, rank, world, group, active, all_gather=True)


def rounds(n_windows: int, world: int) -> List[List[int]]:
    """Window indices per lock-step round of a clip-parallel CLI run: round i = windows [i * world, (i + 1) * world), rank r takes the r-th."""
    return [list(range(i, min(i + world, n_windows))) for i in range(0, n_windows, world)]
