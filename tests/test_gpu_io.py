"""-m gpu: device-side I/O edges (csrc/sn_io.hip) against the host formulas of the CLI (which restate upstream's)."""
import numpy as np
import pytest
import torch

from shiftnet_amd import cli, synth
from shiftnet_amd.io_edges import egress_u8, ingest_u8, ssim_u8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
def test_ingest_u8_is_numpy2tensor_bit_for_bit(dt):
    blur, _ = synth.blurred_clip(3, 36, 52, seed=31)
    blur[0, 0, :256 // 52 + 1] = 0
    blur[1].reshape(-1)[:256] = np.arange(256, dtype=np.uint8)          # every uint8 value occurs
    ref = cli.numpy2tensor(list(blur)).to("cuda").to(dt)              # what upstream feeds the network (test_deblur.py:128,134)
    got = ingest_u8(torch.from_numpy(blur).cuda(), dt)
    assert got.shape == ref.shape and got.dtype == dt and torch.equal(got, ref)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
def test_egress_u8_matches_host_metrics(dt):
    T, H, W = 3, 40, 56
    g = torch.Generator().manual_seed(5)
    out = (torch.rand(T, 3, H, W, generator=g) * 1.4 - 0.2).to(dt)     # values below 0 and above 1: the clamp matters
    _, sharp = synth.blurred_clip(T, H, W, seed=32)
    img, psnr = egress_u8(out.cuda(), torch.from_numpy(sharp).cuda())
    for e in range(T):
        host = out[e].float().clamp(0, 1.0).permute(1, 2, 0).numpy() * 255       # test_deblur.py:140-141
        assert abs(psnr[e] - cli.psnr_255(host, sharp[e])) < 1e-3
        assert np.array_equal(img[e].cpu().numpy(), np.rint(host).astype(np.uint8))


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
@pytest.mark.parametrize("hw", [(40, 56), (9, 300)])
def test_device_ssim_matches_the_cli_formula(dt, hw):
    """sn_ssim_u8 vs cli.ssim_calculate (scipy gaussian_filter over the (C,H,W) volume, reflect boundaries on all three axes)."""
    T, (H, W) = 2, hw
    _, sharp = synth.blurred_clip(T, H, W, seed=33)
    g = torch.Generator().manual_seed(6)
    out = (torch.from_numpy(sharp).permute(0, 3, 1, 2).float() / 255 + 0.08 * torch.randn(T, 3, H, W, generator=g)).to(dt)
    got = ssim_u8(out.cuda(), torch.from_numpy(sharp).cuda())
    for e in range(T):
        host = out[e].float().clamp(0, 1.0).permute(1, 2, 0).numpy() * 255
        assert abs(got[e] - cli.ssim_calculate(host, sharp[e])) < 2e-5, (e, got[e])


@pytest.mark.parametrize("name,dt", [("gshift_deblur2", torch.bfloat16), ("gshift_deblur2", torch.float16), ("gshift_deblur1", torch.bfloat16)])
def test_cli_psnr_of_a_half_precision_module_is_within_the_contract_of_the_fp32_reference(name, dt):
    """What the deblur CLI reports for a bf16 / fp16 module -- uint8 frames in, float32 restored frames from conv_last's fp32
    accumulators (forward_fp32_out, "+ x" on the exact v / 255), clamp * 255, PSNR vs the uint8 ground truth on the device -- against
    the SAME metric of the fp32 reference forward on the same frames (oracle = the reference restated; inference/test_deblur.py:139-143).
    north_star: within 0.01 dB, with no re-definition of the reference."""
    import importlib
    from oracle import shiftnet_oracle as O
    from shiftnet_amd.weights import synth_state_dict
    blur, sharp = synth.blurred_clip(7, 48, 64, seed=3)
    sd = synth_state_dict(name)
    net = importlib.import_module(f"basicsr.models.archs.{name}").GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(sd, strict=True)
    net = net.to(dt).cuda().eval()
    u8 = torch.from_numpy(blur).cuda()
    with torch.no_grad():
        out = net.forward_fp32_out(ingest_u8(u8, dt), shortcut=ingest_u8(u8, torch.float32))
        ref = O.forward(O.VARIANTS[name], sd, cli.numpy2tensor(list(blur)), None, 2, 2)
    _, psnr = egress_u8(out, torch.from_numpy(sharp[2:5]).cuda(), want_image=False)
    for e in range(3):
        host_ref = ref[e].clamp(0, 1.0).permute(1, 2, 0).numpy() * 255               # test_deblur.py:140-141 on the reference output
        assert abs(psnr[e] - cli.psnr_255(host_ref, sharp[2 + e])) <= 0.01, (name, dt, e, psnr[e], cli.psnr_255(host_ref, sharp[2 + e]))
