"""-m gpu: the fp32 engine (csrc/sn_f32.hip + shiftnet_amd/engine32.py) against the CPU oracle and the reference fixtures.

fp32 storage, fp32 weights, fp32 accumulation: the only differences from the oracle are summation order, the exp / rsqrt
implementations and -- in the default "split" mode -- the ~2^-16 of a product whose fp32 operands are split into bf16 hi + lo parts for
the bf16 matrix cores (csrc/sn_f32.hip: conv32s_kernel); the "exact" mode multiplies in fp32 (v_mfma_f32_16x16x4_f32).  EVERY test runs in
BOTH modes at the same tolerance, max-abs <= 1e-4 * scale (SURVEY.md 8c "HIP fp32 vs oracle"), four orders of magnitude below what a
wrong tap / slab / gate half / weight row would produce.  ``Engine32`` shares every line
of control flow with the bf16 engine (it only overrides the leaf operators), so these tests pin the logic of both; they
also are the float32 path upstream's denoise CLI uses for the "+" model (inference/test_denoise.py:83-85).
"""
import os

import numpy as np
import pytest
import torch

from oracle import shiftnet_oracle as O
from shiftnet_amd import synth
from shiftnet_amd.spec import VARIANTS
from shiftnet_amd.weights import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-4


def to_dev(t_nchw):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)


def to_cpu(t_nhwc):
    return t_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(name, got, ref, tol=TOL):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert np.isfinite(err) and err <= tol * scale, f"{name}: max-abs {err:.3g} > {tol} * {scale:.3g}"


@pytest.fixture(scope="module", params=["exact", "split"], autouse=True)
def conv_mode(request):
    """Engine32.split_bf16: False = exact fp32 products, True (the product default) = bf16 hi / lo operands on the bf16 MFMA."""
    from shiftnet_amd.engine32 import Engine32
    old = Engine32.split_bf16
    Engine32.split_bf16 = request.param == "split"
    yield request.param
    Engine32.split_bf16 = old


@pytest.fixture(scope="module")
def engines(conv_mode):
    from shiftnet_amd.engine import make_engine
    cache = {}

    def get(name):
        if name not in cache:
            sd = synth_state_dict(name)
            cache[name] = (make_engine(VARIANTS[name], sd, DEV, torch.float32), sd)
        return cache[name]
    return get


def act(t):
    from shiftnet_amd.engine import Act
    return Act(t, t.shape[3])


@pytest.mark.parametrize("name", list(VARIANTS))
def test_blocks_fp32(name, engines):
    eng, sd = engines(name)
    V = O.VARIANTS[name]
    C, T, h, w = V.c1, 4, 20, 44
    x = torch.from_numpy(synth.unit_noise((T, C, h, w), seed=81))
    xd = act(to_dev(x))
    blk = "stage1.decoder_level1."
    with torch.no_grad():
        for mode, rev, unit in ((1, False, "encoder_level1."), (2, True, "encoder_level1_1.")):
            pre = blk + unit + "0."
            close(f"cab2_{name}_{mode}", to_cpu(eng.naf(pre, xd, mode).t), O.cab2(sd, pre, O.gsts_gather(x, rev, V.wrap), V))
        pre = blk + "encoder_level1.1."
        close(f"cab1_{name}", to_cpu(eng.naf(pre, xd, 0).t), O.cab1(sd, pre, x, V))
        close(f"unit_rev_{name}", to_cpu(eng.gsts_unit(blk + "encoder_level1_1.", xd, True).t), O.gsts_unit(sd, blk + "encoder_level1_1.", x, True, V))
        close(f"shift_block_{name}", to_cpu(eng.shift_block(blk, xd).t), O.shift_block(sd, blk, x, V))
        x2 = torch.from_numpy(synth.unit_noise((2, C, 13, 70), seed=82))
        close(f"unit_fwd_ragged_{name}", to_cpu(eng.gsts_unit(blk + "encoder_level1.", act(to_dev(x2)), False).t),
              O.gsts_unit(sd, blk + "encoder_level1.", x2, False, V))
        x0 = torch.from_numpy(synth.unit_noise((3, V.c0, 24, 40), seed=91))
        close(f"cab_{name}", to_cpu(eng.cab("stage1.concat.", act(to_dev(x0))).t), O.cab(sd, "stage1.concat.", x0))
        close(f"tfr_unet_{name}", to_cpu(eng.tfr_unet("orb1.", act(to_dev(x0))).t), O.tfr_unet(sd, "orb1.", x0, V))
        close(f"stage1_{name}", to_cpu(eng.stage1(act(to_dev(x0))).t), O.stage1(sd, x0, V))
        if V.shift_cab:
            close(f"shift_cab_{name}", to_cpu(eng.shift_cab("stage1.encoder_level1.", xd, True).t),
                  O.shift_cab(sd, "stage1.encoder_level1.", x, True))


@pytest.mark.parametrize("name", list(VARIANTS))
def test_fused_phase1_fp32_against_one_kernel_per_module(name, engines):
    """SN_FP32_FUSE (sn32_dw_gate, merged RepConv weights with iscale / rscale, sn32_conv1x1_gate2) against the chain that runs one kernel per
    reference module, on a tile whose pixel count is a multiple of 64 (the gated 1x1 needs it; 20 x 44 of test_blocks_fp32 takes the fallback)
    and on a ragged one.  Both against the oracle at the file's tolerance, and against each other ten times tighter: the fusions only
    reorder fp32 additions (merged 5x5 + 3x3 weights, partial sums)."""
    from shiftnet_amd.engine32 import Engine32
    eng, sd = engines(name)
    V = O.VARIANTS[name]
    blk = "stage1.decoder_level1."
    for (T, h, w, seed) in ((3, 16, 48, 83), (2, 13, 70, 84)):
        x = torch.from_numpy(synth.unit_noise((T, V.c1, h, w), seed=seed))
        xd = act(to_dev(x))
        for mode, rev, unit in ((1, False, "encoder_level1.0."), (2, True, "encoder_level1_1.0."), (0, False, "encoder_level1.1.")):
            pre = blk + unit
            ref = O.cab2(sd, pre, O.gsts_gather(x, rev, V.wrap), V) if mode else O.cab1(sd, pre, x, V)
            outs = {}
            for fuse in (True, False):
                old = Engine32.fuse_ops
                Engine32.fuse_ops = fuse
                try:
                    with torch.no_grad():
                        outs[fuse] = to_cpu(eng.naf(pre, xd, mode).t)
                finally:
                    Engine32.fuse_ops = old
                close(f"naf_{name}_{mode}_fuse{int(fuse)}_{h}x{w}", outs[fuse], ref)
            close(f"naf_{name}_{mode}_fused_vs_chain_{h}x{w}", outs[True], outs[False], tol=1e-5)
            if mode:                               # conv1 inside the channel_shift kernel (SN_FP32_SHIFTCONV=1): same taps in the same order
                Engine32.fuse_shiftconv = True
                try:
                    with torch.no_grad():
                        sc = to_cpu(eng.naf(pre, xd, mode).t)
                finally:
                    Engine32.fuse_shiftconv = False
                assert torch.equal(sc, outs[True]), f"naf_{name}_{mode}_shiftconv_{h}x{w}"
    # CAB: the sums of `mid` from conv1's epilogue (sn32_conv_desc.csum) against the sn32_chan_sum pass, ragged tile (partial workgroups)
    x0 = torch.from_numpy(synth.unit_noise((3, V.c0, 23, 41), seed=85))
    outs = {}
    for fuse in (True, False):
        old = Engine32.fuse_ops
        Engine32.fuse_ops = fuse
        try:
            with torch.no_grad():
                outs[fuse] = to_cpu(eng.cab("stage1.concat.", act(to_dev(x0))).t)
        finally:
            Engine32.fuse_ops = old
        close(f"cab_{name}_fuse{int(fuse)}", outs[fuse], O.cab(sd, "stage1.concat.", x0))
    close(f"cab_{name}_fused_vs_chain", outs[True], outs[False], tol=1e-5)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_whole_net_fp32_vs_reference_fixture(name, golden_dir):
    """float32 module through the drop-in class against the REFERENCE's fp32 output (tests/golden/net_*.npz)."""
    import importlib
    mod = importlib.import_module(f"basicsr.models.archs.{name}")
    V = O.VARIANTS[name]
    g = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
    blur, _ = synth.blurred_clip(7, 48, 64, seed=3)
    assert synth.crc(blur) == int(g["in_crc"])
    x = O.frames_to_tensor(list(blur))
    net = mod.GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name), strict=True)
    net = net.to("cuda").eval()
    assert next(net.parameters()).dtype == torch.float32
    nm = torch.full((1, 7, 1, 48, 64), 30.0 / 255.0) if V.denoise else None
    with torch.no_grad():
        out = net(x.cuda(), nm.cuda()) if V.denoise else net(x.cuda())
    assert out.dtype == torch.float32
    close(f"net32_{name}", out, torch.from_numpy(g["p2f2"]))
    from shiftnet_amd.engine32 import Engine32
    assert isinstance(net.prepare(), Engine32)


def test_config1_fp32(golden_dir):
    """BASELINE config 1 (Shift-Net-s, fp32, 1 clip of T=5 256x256): the reference's CPU-runnable case, here on the GPU in fp32."""
    from basicsr.models.archs.gshift_deblur2 import GShiftNet
    name = "gshift_deblur2"
    net = GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(name), strict=True)
    net = net.cuda().eval()
    g = np.load(os.path.join(golden_dir, f"config1_{name}.npz"))
    blur, _ = synth.blurred_clip(5, 256, 256, seed=5)
    assert synth.crc(blur) == int(g["in_crc"])
    with torch.no_grad():
        out = net(O.frames_to_tensor(list(blur)).cuda())
    close("config1_fp32", out, torch.from_numpy(g["out"]))


def test_cli_quadrant_forward_matches_reference_fixture(golden_dir):
    """cli.quadrant_forward (the denoise CLI's 4 overlapping quadrants, inference/test_denoise.py:153-173) on a float32 gshift_denoise1 module --
    upstream's dtype for this model (:83-85) -- with the fixture's FIXED noise tensor, against the reference network run on the same four
    crops and stitched by the CLI's own slice arithmetic (tests/golden/quadrants_gshift_denoise1.npz, made by make_golden.py from the
    reference): <= 1e-4.  Host-side and device-side stitching, both conv arithmetics.  A swapped quadrant or crop offset is off by >= 1.5e-3."""
    from basicsr.models.archs.gshift_denoise1 import GShiftNet
    from shiftnet_amd import cli
    from test_oracle_golden import quadrant_inputs
    g = np.load(os.path.join(golden_dir, "quadrants_gshift_denoise1.npz"))
    x, sigma = quadrant_inputs(g)
    net = GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict("gshift_denoise1"), strict=True)
    net = net.to(DEV).eval()
    with torch.no_grad():
        for on_device in (False, True):
            out = cli.quadrant_forward(net, x.to(DEV), sigma, on_device=on_device)
            assert out.is_cuda == on_device
            close(f"quadrants_on_device_{on_device}", out, torch.from_numpy(g["out"]))
