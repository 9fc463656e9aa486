"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, made by make_golden.py).

Tolerance: fp32 vs fp32 on CPU, max-abs <= 1e-5 relative to the tensor's scale; the GSTS
gather is pure index work and must be bit exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import shiftnet_oracle as O
from shiftnet_amd import synth
from shiftnet_amd.weights import synth_state_dict

NAMES = list(O.VARIANTS)


def t32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def close(a, b, tol=1e-5):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max-abs {err} vs scale {scale}"


@pytest.fixture(scope="module", params=NAMES)
def variant(request):
    name = request.param
    return name, O.VARIANTS[name], synth_state_dict(name)


def test_variant_tables_agree_with_product():
    from shiftnet_amd.spec import VARIANTS as PV, shift_table
    for n, V in O.VARIANTS.items():
        for f in ("in_ch", "c0", "c1", "unet_step", "n_orb", "units", "wrap", "grouped_rep", "denoise", "topo",
                  "hr_cat", "shift_cab", "past", "future"):
            assert getattr(V, f) == getattr(PV[n], f), (n, f)
        assert O.shift_offsets(V.c1) == shift_table(V.c1)


def test_gsts_gather_exact(variant, golden_dir):
    name, V, _ = variant
    g = np.load(os.path.join(golden_dir, f"gsts_gather_{name}.npz"))
    C, T, h, w = V.c1, 4, 24, 24
    idx = (np.arange(T * C * h * w, dtype=np.float32) + 1).reshape(T, C, h, w)
    for tag, rev in (("fwd", False), ("rev", True)):
        u = O.gsts_gather(t32(idx), rev, V.wrap).numpy().astype(np.int32)
        assert np.array_equal(u, g[tag])


def test_spatial_shift_is_onehot_depthwise_conv():
    """Known answer (gshift_deblur1.py:520-525 commented check): shift == depthwise conv with one-hot 17x17 kernels."""
    for C in (64, 80):
        Ch = C // 2
        x = t32(synth.unit_noise((2, Ch, 20, 28), seed=5))
        k = torch.zeros(Ch, 1, 17, 17)
        for c, (dy, dx) in enumerate(O.shift_offsets(C)):
            k[c, 0, 8 + dy, 8 + dx] = 1
        ref = torch.nn.functional.conv2d(x, k, padding=8, groups=Ch)
        assert torch.equal(O.spatial_shift(x), ref)


def test_blocks(variant, golden_dir):
    name, V, P = variant
    g = np.load(os.path.join(golden_dir, f"blocks_{name}.npz"))
    T, h, w, C = 3, 12, 20, V.c1
    x = t32(synth.unit_noise((T, C, h, w), seed=11))
    x0 = t32(synth.unit_noise((T, V.c0, h, w), seed=12))
    x0b = t32(synth.unit_noise((T, V.c0, 16, 24), seed=13))
    assert synth.crc(x.numpy()) == int(g["x_crc"]) and synth.crc(x0.numpy()) == int(g["x0_crc"])
    assert synth.crc(x0b.numpy()) == int(g["x0b_crc"])
    blk = "stage1.decoder_level1."
    with torch.no_grad():
        uf = O.gsts_gather(x, False, V.wrap); ur = O.gsts_gather(x, True, V.wrap)
        close(O.cab2(P, blk + "encoder_level1.0.", uf, V), g["cab2_fwd"])
        close(O.cab2(P, blk + "encoder_level1_1.0.", ur, V), g["cab2_rev"])
        close(O.cab1(P, blk + "encoder_level1.1.", x, V), g["cab1"])
        close(O.gsts_unit(P, blk + "encoder_level1.", x, False, V), g["unit_fwd"])
        close(O.shift_block(P, blk, x, V), g["shift_block"])
        close(O.cab(P, "stage1.concat.", x0), g["cab_c0"])
        close(O.cab(P, "stage1.skip_attn1.", x), g["cab_c1"])
        close(O.tfr_unet(P, "orb1.", x0, V), g["tfr_unet"])
        d = O.down_sample(P, "stage1.down12.", x, V)
        close(d, g["down12_c1"])
        close(O.skip_up_sample(P, "stage1.up21.", d, x), g["up21_c1"])
        close(O.pixel_shuffle_pack(P, "stage1.upsample0.", x), g["pixshuf"])
        close(torch.nn.functional.prelu(torch.nn.functional.conv2d(x0, P["stage1.down01.0.weight"], stride=2),
                                        P["stage1.down01.1.weight"]), g["down01"])
        close(O.stage1(P, x0b, V), g["stage1"])
        if V.shift_cab:
            close(O.shift_cab(P, "stage1.encoder_level1.", x, False), g["shift_cab_fwd"])
            close(O.shift_cab(P, "stage1.encoder_level1_1.", x, True), g["shift_cab_rev"])


def test_whole_net(variant, golden_dir):
    name, V, P = variant
    g = np.load(os.path.join(golden_dir, f"net_{name}.npz"))
    blur, _ = synth.blurred_clip(7, 48, 64, seed=3)
    assert synth.crc(blur) == int(g["in_crc"])
    x = O.frames_to_tensor(list(blur))
    nm = torch.full((1, 7, 1, 48, 64), 30.0 / 255.0) if V.denoise else None
    with torch.no_grad():
        close(O.forward(V, P, x, nm, 2, 2), g["p2f2"])
        close(O.forward(V, P, x, nm), g["default"])


def test_config1_and_windows(golden_dir):
    name = "gshift_deblur2"
    V, P = O.VARIANTS[name], synth_state_dict(name)
    g = np.load(os.path.join(golden_dir, f"config1_{name}.npz"))
    blur, _ = synth.blurred_clip(5, 256, 256, seed=5)
    assert synth.crc(blur) == int(g["in_crc"])
    with torch.no_grad():
        close(O.forward(V, P, O.frames_to_tensor(list(blur)), None, 2, 2), g["out"])
    g = np.load(os.path.join(golden_dir, f"windows_{name}.npz"))
    blur, _ = synth.blurred_clip(12, 32, 40, seed=7)
    assert synth.crc(blur) == int(g["in_crc"])
    wins = O.deblur_windows(12, 4)
    assert [[a.start, a.stop, b.start, b.stop] for a, b in wins] == g["windows"].tolist()
    outs = []
    with torch.no_grad():
        for a, _ in wins:
            outs.append(O.forward(V, P, O.frames_to_tensor(list(blur[a.start:a.stop])), None, 2, 2))
    close(torch.cat(outs, 0), g["out"])
    # T <= past+future yields an empty tensor, not an error (SURVEY.md §8b [probe])
    with torch.no_grad():
        assert O.forward(V, P, O.frames_to_tensor(list(blur[:4])), None, 2, 2).shape[0] == 0


def quadrant_inputs(g):
    """The fixture's input: clean synthetic clip + the fixed noise tensor (tests/golden/make_golden.py: quadrant_case)."""
    T, H, W = (int(v) for v in g["dims"])
    sigma = float(g["sigma"])
    _, sharp = synth.blurred_clip(T, H, W, seed=11)
    assert synth.crc(sharp) == int(g["in_crc"])
    noise = torch.from_numpy(synth.unit_noise((1, T, 3, H, W), seed=12)).float() * sigma
    assert synth.crc(noise.numpy()) == int(g["noise_crc"])
    return O.frames_to_tensor(list(sharp)) + noise, sigma


def test_denoise_quadrant_stitching_matches_reference(golden_dir):
    """The denoise CLI's four overlapping quadrants (inference/test_denoise.py:153-173), restated on the ORACLE, against the reference network
    run on the same four crops and stitched by the CLI's own slice arithmetic (fixture quadrants_gshift_denoise1.npz): fixed noise tensor,
    96 x 128 frames.  A swapped quadrant or a wrong crop offset changes the result by the whole-frame-vs-stitched distance the fixture records
    (1.5e-3) or far more; the bound is 1e-5."""
    name = "gshift_denoise1"
    g = np.load(os.path.join(golden_dir, f"quadrants_{name}.npz"))
    x, sigma = quadrant_inputs(g)
    V, sd = O.VARIANTS[name], synth_state_dict(name)
    B, N, _, H, W = x.shape
    pad_h, pad = 32 - (H // 2 % 16), 32 - (W // 2 % 16)
    hh, ww = H // 2 + pad_h, W // 2 + pad
    std = torch.full((B, N, 1, hh, ww), sigma)
    with torch.no_grad():
        def run(ys, xs):
            return O.forward(V, sd, x[:, :, :, ys, xs].contiguous(), std, 2, 2)
        out = torch.zeros(N - 4, 3, H, W)
        out[..., 0:H // 2, 0:W // 2] = run(slice(0, hh), slice(0, ww))[..., 0:-pad_h, 0:-pad]
        out[..., 0:H // 2, W // 2:] = run(slice(0, hh), slice(W // 2 - pad, W))[..., 0:-pad_h, pad:]
        out[..., H // 2:, 0:W // 2] = run(slice(H // 2 - pad_h, H), slice(0, ww))[..., pad_h:, 0:-pad]
        out[..., H // 2:, W // 2:] = run(slice(H // 2 - pad_h, H), slice(W // 2 - pad, W))[..., pad_h:, pad:]
    close(out.numpy(), g["out"])
    assert float(g["whole_vs_stitched_maxabs"]) > 1e-4      # the case distinguishes stitching from a whole-frame forward
