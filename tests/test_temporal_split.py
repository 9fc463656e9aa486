"""Intra-window temporal split (shiftnet_amd/temporal_split.py).

CPU part (gloo, world 2, 3 and 8): the halo exchange and the boundary rules, checked against the CPU oracle -- every rank
rebuilds the gathered 1.5C-channel input of a shifted unit for ITS frames from its local frames plus the received half-frame, and the
concatenation over ranks must equal the oracle's gather on the whole window, bit for bit, for circular (deblur2) and kept
(all other variants) boundaries and both directions; a whole oracle shift block run rank-locally with the exchange before
every unit must equal the single-process block.

GPU part (-m gpu): two processes on the one MI355X of the GPU box, gloo for the exchange (RCCL refuses two ranks on one
device), each running the HIP engine on half of a window: the concatenated output equals the single-process long-window
output BIT FOR BIT for Shift-Net-s (circular ring between the two ranks) and Shift-Net+ (kept boundaries), AND the CPU oracle's
output on the long window (>= 48 dB for bf16 modules, 1e-4 for float32 ones).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _local_gather(O, x_local, halo, mode, flag):
    """The unit's gathered 1.5C-channel input for the LOCAL frames, from the local frames, the received half-frame `halo` ([h,w,C/2], what the
    kernels get as sn_unit_src.halo) and the kernel flag (0 keep, 1 circular, 2 halo) -- the rule csrc/sn_common.h: sn_unit_slabs implements."""
    rev = mode == 2
    if flag != 2:
        return O.gsts_gather(x_local, rev, flag == 1)
    T, C, h, w = x_local.shape
    nb = torch.zeros((1, C, h, w))                                       # the neighbour frame: only the borrowed half exists on this rank
    hh = halo.permute(2, 0, 1)
    if not rev:
        nb[0, C // 2:] = hh                                              # forward units borrow the UPPER half of the previous frame
        g = O.gsts_gather(torch.cat((nb, x_local)), False, False)
        return g[1:]
    nb[0, :C // 2] = hh                                                  # reverse units the LOWER half of the next frame
    g = O.gsts_gather(torch.cat((x_local, nb)), True, False)
    return g[:-1]


def _cpu_worker(rank, world, port, circular):
    _setup(rank, world, port)
    from oracle import shiftnet_oracle as O
    from shiftnet_amd import synth
    from shiftnet_amd.temporal_split import TemporalSplit, partition
    from shiftnet_amd.weights import synth_state_dict
    name = "gshift_deblur2" if circular else "gshift_denoise2"         # both C = 64, 4 units per block
    V = O.VARIANTS[name]
    assert V.wrap == circular
    T, C, h, w = max(7, world + 3), V.c1, 20, 24
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=55))
    a, b = partition(T, world)[rank]
    sp = TemporalSplit(rank, world, circular)
    sp.validate(b - a)
    for mode in (1, 2):
        halo = sp.exchange(_nhwc(x[a:b]), mode, circular)
        flag = sp.wrap_flag(mode, circular)
        assert (halo is not None) == (flag == 2) and (halo is None or (halo.is_contiguous() and tuple(halo.shape) == (h, w, C // 2)))
        mine = _local_gather(O, x[a:b], halo, mode, flag)
        full = O.gsts_gather(x, mode == 2, circular)
        assert torch.equal(mine, full[a:b]), (rank, mode)
        # Shift_CAB's roll never wraps, also on a circular module: the window's outer ranks keep their boundary frame
        flag_nc = sp.wrap_flag(mode, False)
        assert flag_nc == (2 if (rank > 0 if mode == 1 else rank < world - 1) else 0)
        sp.exchange(_nhwc(x[a:b]), mode, False)
    # a whole shift block, rank-local, with one exchange per unit
    if world <= 3:
        sd = synth_state_dict(name)
        blk = "stage1.decoder_level1."
        with torch.no_grad():
            ref = O.shift_block(sd, blk, x, V)
            cur = x[a:b]
            for i in range(V.units):
                mode = 2 if i % 2 else 1
                halo = sp.exchange(_nhwc(cur), mode, circular)
                u = _local_gather(O, cur, halo, mode, sp.wrap_flag(mode, circular))
                pre = f"{blk}{O._UNIT_NAMES[i]}."
                cur = O.cab1(sd, pre + "1.", O.cab2(sd, pre + "0.", u, V), V)
        assert torch.allclose(cur, ref[a:b], rtol=0, atol=2e-6 * float(ref.abs().max())), rank
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,circular", [(2, True), (2, False), (3, True), (3, False), (8, True)])
def test_halo_exchange_and_boundaries_cpu(world, circular):
    """gloo ranks on CPU: the exchange (contiguous receive buffers), the boundary rules and -- world 8, deblur2 -- the closed ring."""
    mp.spawn(_cpu_worker, args=(world, _free_port(), circular), nprocs=world, join=True)


def _empty_rank_worker(rank, world, port):
    _setup(rank, world, port)
    from shiftnet_amd.temporal_split import TemporalSplit
    sp = TemporalSplit(rank, world, False)
    with pytest.raises(ValueError):
        sp.validate(0 if rank == 1 else 3)          # ONE rank has nothing to restore: EVERY rank raises, nobody is left in an exchange
    dist.barrier()
    dist.destroy_process_group()


def test_empty_rank_is_refused_collectively():
    mp.spawn(_empty_rank_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_partition():
    from shiftnet_amd.temporal_split import partition
    assert partition(100, 8) == [(0, 13), (13, 26), (26, 39), (39, 52), (52, 64), (64, 76), (76, 88), (88, 100)]
    assert partition(20, 2) == [(0, 10), (10, 20)]


def _gpu_worker(rank, world, port, name, dt_name, outdir):
    _setup(rank, world, port)
    import importlib
    from oracle import shiftnet_oracle as O
    from shiftnet_amd import synth
    from shiftnet_amd.temporal_split import partition
    from shiftnet_amd.weights import synth_state_dict
    dt = getattr(torch, dt_name)
    mod = importlib.import_module(f"basicsr.models.archs.{name}")
    V = O.VARIANTS[name]
    T, H, W = 10, 48, 64
    blur, _ = synth.blurred_clip(T, H, W, seed=17)
    x = O.frames_to_tensor(list(blur)).to(dt).cuda()
    nm = torch.full((1, T, 1, H, W), 30.0 / 255.0, dtype=dt, device="cuda") if V.denoise else None
    net = mod.GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name), strict=True)
    net = net.to(dt).cuda().eval()
    a, b = partition(T, world)[rank]
    net.set_temporal_split(rank, world)
    with torch.no_grad():
        part = net(x[:, a:b].contiguous(), nm[:, a:b].contiguous()) if V.denoise else net(x[:, a:b].contiguous())
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"part{rank}.npy"), part.float().cpu().numpy())
    if rank == 0:
        net.set_temporal_split(0, 1)
        with torch.no_grad():
            full = net(x, nm) if V.denoise else net(x)
        np.save(os.path.join(outdir, "full.npy"), full.float().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name,dt", [("gshift_deblur2", "bfloat16"), ("gshift_deblur1", "bfloat16"), ("gshift_denoise1", "float32")])
def test_temporal_split_equals_long_window_bit_for_bit(name, dt, tmp_path):
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), name, dt, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    parts = np.concatenate([np.load(tmp_path / f"part{r}.npy") for r in range(world)], 0)
    assert parts.shape == full.shape == (6, 3, 48, 64)
    assert np.array_equal(parts, full)                                   # split HIP == unsplit HIP, bit for bit
    # ... and against the CPU oracle on the long window (the reference restated), not only against ourselves
    for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import shiftnet_oracle as O
    from shiftnet_amd import synth
    from shiftnet_amd.weights import synth_state_dict
    V = O.VARIANTS[name]
    blur, _ = synth.blurred_clip(10, 48, 64, seed=17)
    x = O.frames_to_tensor(list(blur))
    tdt = getattr(torch, dt)
    xq = x.to(tdt).float()                                                # the module input is rounded to its dtype
    nm = torch.full((1, 10, 1, 48, 64), 30.0 / 255.0).to(tdt).float() if V.denoise else None
    with torch.no_grad():
        ref = O.forward(V, synth_state_dict(name), xq, nm, 2, 2).numpy()
    if dt == "float32":
        assert np.abs(parts - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    else:
        mse = float(np.mean((parts - ref) ** 2))
        assert 10 * np.log10(1.0 / mse) >= 48.0
