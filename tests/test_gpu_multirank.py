"""-m gpu: the clip-parallel path on a real device.  The driver's GPU box has ONE MI355X, so the process group here has a
single rank -- but it is a real ``nccl`` (= RCCL) group: the halo all-gather of ``clip_parallel.assemble_window`` runs
through RCCL on device tensors, and the window it assembles feeds the HIP forward (BASELINE config 5's per-GPU window:
Shift-Net+ deblur, 1920x1080, one_len 12 -> T_in 16).  Multi-rank correctness of the exchange itself is covered on CPU by
``tests/test_host_logic.py::test_halo_exchange_gloo`` (gloo, world 2 / 4 / 8).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from shiftnet_amd import synth
from shiftnet_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture
def rccl_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_rccl_halo_allgather_and_config5_window(rccl_world1):
    from basicsr.models.archs.gshift_deblur1 import GShiftNet
    from shiftnet_amd.clip_parallel import assemble_window
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    L, H, W = 12, 1080, 1920
    blur, _ = synth.blurred_clip(L + 4, H, W, seed=21)
    fr = (torch.from_numpy(blur).permute(0, 3, 1, 2).cuda().to(torch.bfloat16) / 255).contiguous()
    own, first, last = fr[2:2 + L].contiguous(), fr[:2].contiguous(), fr[-2:].contiguous()
    win = assemble_window(own, first, last, 0, 1)             # goes through dist.all_gather on the RCCL group
    assert win.is_cuda and torch.equal(win, fr)
    # the collective really ran on the device: gather a second, distinguishable payload and read it back
    probe = torch.arange(4 * 3 * 8 * 8, device="cuda", dtype=torch.float32).reshape(4, 3, 8, 8)
    out = [torch.empty_like(probe)]
    dist.all_gather(out, probe)
    assert torch.equal(out[0], probe)

    net = GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict("gshift_deblur1"), strict=True)
    net = net.to(torch.bfloat16).cuda().eval()
    with torch.no_grad():
        y1 = net(win.unsqueeze(0))
        y2 = net(assemble_window(own, first, last, 0, 1).unsqueeze(0))
    torch.cuda.synchronize()
    assert y1.shape == (L, 3, H, W) and torch.isfinite(y1.float()).all()
    assert torch.equal(y1, y2)                                # deterministic: no atomics anywhere in the path
    # "+" keeps the boundary frames un-rolled (gshift_deblur1.py:513,517): the restored frames stay near the input
    assert (y1.float() - win[2:-2].float()).abs().max().item() < 1.0


def test_bench_two_rank_flow_on_one_device():
    """`bench.py --gpus 2` end to end with both ranks on the box's single GPU (gloo transport, SN_BENCH_SHARED_DEVICE_TEST=1): the
    self-launch through torch.distributed.run, the halo all-gather in every step, the MAX-over-ranks timing, rank 0's local
    profiling step (no collective the other rank is not in) and the final barrier must all complete; the line is marked as a
    shared-device test, never as a 2-GPU measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SN_BENCH_SHARED_DEVICE_TEST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "144",
                        "--width", "256", "--one-len", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["shared_device_test"] is True and d["value"] > 0 and d["config"]["parallelism"] == "clip-parallel x2"
