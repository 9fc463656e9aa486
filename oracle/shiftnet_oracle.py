"""CPU oracle for the Shift-Net inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of what the reference's four
``GShiftNet`` variants compute (``/root/reference/basicsr/models/archs/
gshift_{deblur1,deblur2,denoise1,denoise2}.py``).  It exists so that the HIP
path can be *checked*; it is never the thing shipped or measured:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * the product (``shift-net_amd/``) never imports anything from ``oracle/``.

Pinning: the reference has no tests or golden vectors of its own,
so this oracle is pinned against the reference *itself*, imported on CPU in the
build container, by ``tests/golden/make_golden.py`` (outputs committed under
``tests/golden/``) and re-checked by ``tests/test_oracle_golden.py``.

Design: it is deliberately NOT an ``nn.Module`` tree.  The network is a set of
pure functions over a flat ``{state_dict key: tensor}`` mapping, driven by one
``Variant`` record per model, and the temporal/spatial shift is written as an
explicit gather (index table) rather than as roll/cat/slice-assign, so that it
is an independent statement of the semantics and not a
transliteration.  Citations ``D1:`` = gshift_deblur1.py, ``D2:`` =
gshift_deblur2.py, ``N1:`` = gshift_denoise1.py, ``N2:`` = gshift_denoise2.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# Variant matrix (every field is parity relevant)
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class Variant:
    name: str
    in_ch: int            # 3 (deblur) / 4 = RGB + sigma map (denoise)     D1:740 N1:768
    c0: int               # full-resolution width n_feats0                  D1:738 D2:709
    c1: int               # stage-1 width n_feats2                          D1:733 D2:704
    unet_step: int        # TFR_UNet scale_unetfeats (hard coded)           D1:684 D2:657
    n_orb: int            # TFR_UNets executed in stage0 / stage2           D1:764-768 D2:733-735
    units: int            # (CAB2,CAB1) pairs per Encoder_shift_block       D1:530-547 D2:521-530
    wrap: bool            # circular temporal roll (deblur2 only)           D2:504-505 vs D1:513,517
    grouped_rep: bool     # RepConv groups = C/8 ("+") vs depthwise ("-s")  D1:160-161 D2:162-163
    denoise: bool         # extra CALayer2, biased last 1x1, PReLU'd down   N1:224,229,361-362
    topo: str             # 'plus' (3 levels, CAB encoders) | 'small' (2 levels, 12 shift blocks)
    hr_cat: bool          # conv_hr0(cat(up, skip)) vs conv_hr0(act(up)) + skip   D1:640 D2:611
    shift_cab: bool       # Shift_CAB encoders (denoise1 only)              N1:582-585,645-649
    past: int             # default past_frames / future_frames of the ctor D1:728 N1:758
    future: int


VARIANTS: Dict[str, Variant] = {
    "gshift_deblur1": Variant("gshift_deblur1", 3, 24, 80, 12, 5, 8, False, True, False, "plus", True, False, 1, 1),
    "gshift_deblur2": Variant("gshift_deblur2", 3, 14, 64, 4, 3, 4, True, False, False, "small", False, False, 1, 1),
    "gshift_denoise1": Variant("gshift_denoise1", 4, 24, 80, 12, 5, 8, False, True, True, "plus", True, True, 0, 0),
    "gshift_denoise2": Variant("gshift_denoise2", 4, 14, 64, 4, 3, 4, False, False, True, "small", True, False, 0, 0),
}


# --------------------------------------------------------------------------- #
# Grouped spatial-temporal shift (GSTS)
# --------------------------------------------------------------------------- #
# (dy, dx) of the SOURCE pixel relative to the output pixel, i.e.
# out[h, w] = in[h + dy, w + dx], zero when the source is outside the map.
# Derived from the 24 slice assignments of spatial_shift2 (D1:470-503, D2:465-498):
# e.g. "s_out[:, g, 8:, 8:] = hw[:, g, :-8, :-8]" is out[h,w] = in[h-8, w-8].
_OUTER = [(-8, -8), (-8, -4), (-8, 0), (-8, 4), (-8, 8),
          (8, -8), (8, -4), (8, 0), (8, 4), (8, 8),
          (-4, -8), (-4, 8), (0, -8), (0, 8), (4, -8), (4, 8)]
_INNER = [(-4, -4), (-4, 0), (-4, 4), (0, -4), (0, 4), (4, -4), (4, 0), (4, 4)]


def shift_offsets(c: int) -> List[Tuple[int, int]]:
    """Per-channel (dy, dx) for the C/2 borrowed channels of a width-``c`` block.

    ``number = c//2//8``, outer-ring groups have ``n2=(number-1)//2`` channels,
    inner-ring groups ``n1=number-2*n2`` (D1:447-449,471-472; D2:449,466-467).
    c=80 -> n2=2,n1=1 ; c=64 -> n2=1,n1=2.
    """
    number = c // 2 // 8
    n2 = (number - 1) // 2
    n1 = number - 2 * n2
    table: List[Tuple[int, int]] = []
    for off in _OUTER:
        table += [off] * n2
    for off in _INNER:
        table += [off] * n1
    assert len(table) == 8 * number
    return table


def spatial_shift(hw: Tensor) -> Tensor:
    """Zero-padded per-channel displacement of ``hw:[T,Ch,h,w]`` (spatial_shift2).

    Written as one gather from an 8-pixel zero-padded copy.  Ch must be C/2 of
    a block whose offset table ``shift_offsets(2*Ch)`` has Ch entries.
    """
    T, Ch, h, w = hw.shape
    table = shift_offsets(2 * Ch)
    pad = F.pad(hw, (8, 8, 8, 8))
    out = torch.empty_like(hw)
    for k, (dy, dx) in enumerate(table):
        out[:, k] = pad[:, k, 8 + dy: 8 + dy + h, 8 + dx: 8 + dx + w]
    return out


def temporal_sources(T: int, reverse: bool, wrap: bool) -> List[Tuple[int, int, bool]]:
    """For every output frame t: (frame feeding u[:, :Ch], frame feeding u[:, Ch:C], kept).

    Forward units borrow from t-1, reverse units from t+1.
    ``kept`` marks the boundary frame of the non-circular variants, which is
    passed through un-rolled (D1:513,517); deblur2 wraps instead (D2:504-505).
    """
    rows = []
    for t in range(T):
        if not reverse:
            if t == 0 and not wrap:
                rows.append((0, 0, True))
            else:
                rows.append(((t - 1) % T, t, False))
        else:
            if t == T - 1 and not wrap:
                rows.append((T - 1, T - 1, True))
            else:
                rows.append((t, (t + 1) % T, False))
    return rows


def temporal_roll(x: Tensor, reverse: bool, wrap: bool) -> Tuple[Tensor, Tensor]:
    """Return (y, hw): the rolled tensor y:[T,C,h,w] and the borrowed half hw:[T,C/2,h,w].

    forward  interior: y[t] = cat(x[t-1, Ch:], x[t, :Ch]);  hw = y[:, :Ch]
    reverse  interior: y[t] = cat(x[t, Ch:], x[t+1, :Ch]);  hw = y[:, Ch:]
    kept boundary frame: y[t] = x[t] (so hw is the frame's own lower/upper half).
    (channel_shift, D1:504-528 / D2:499-519; Shift_CAB N1:167-179.)
    """
    T, C, _, _ = x.shape
    Ch = C // 2
    ys = []
    for t, (fa, fb, kept) in enumerate(temporal_sources(T, reverse, wrap)):
        if kept:
            ys.append(x[t])
        else:
            ys.append(torch.cat((x[fa, Ch:], x[fb, :Ch]), dim=0))
    y = torch.stack(ys, 0)
    hw = y[:, Ch:] if reverse else y[:, :Ch]
    return y, hw


def gsts_gather(x: Tensor, reverse: bool, wrap: bool) -> Tensor:
    """The virtual 1.5C-channel input of CAB2: cat(y, spatial_shift(hw)) (D1:528)."""
    y, hw = temporal_roll(x, reverse, wrap)
    return torch.cat((y, spatial_shift(hw.contiguous())), dim=1)


# --------------------------------------------------------------------------- #
# Elementary ops
# --------------------------------------------------------------------------- #
def layer_norm_2d(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    """Per-pixel LayerNorm over channels, biased variance (LayerNormFunction.forward, D1:19-28)."""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + eps).sqrt()
    return y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def _conv(P: Params, pre: str, x: Tensor, stride: int = 1, padding: Optional[int] = None, groups: int = 1) -> Tensor:
    w = P[pre + "weight"]
    if padding is None:
        padding = w.shape[-1] // 2
    return F.conv2d(x, w, P.get(pre + "bias"), stride=stride, padding=padding, groups=groups)


def channel_attention(P: Params, pre: str, x: Tensor) -> Tensor:
    """CALayer / CALayer2: global mean -> 1x1 -> ReLU -> 1x1 -> sigmoid -> scale (D1:54-87)."""
    s = x.mean((2, 3), keepdim=True)
    s = F.relu(_conv(P, pre + "conv_du.0.", s))
    s = torch.sigmoid(_conv(P, pre + "conv_du.2.", s))
    return x * s


def cab(P: Params, pre: str, x: Tensor) -> Tensor:
    """CAB: 3x3 -> PReLU -> 3x3 -> CALayer -> + x (D1:141-156)."""
    r = _conv(P, pre + "body.0.", x)
    r = F.prelu(r, P[pre + "body.1.weight"])
    r = _conv(P, pre + "body.2.", r)
    return channel_attention(P, pre + "CA.", r) + x


def shift_cab(P: Params, pre: str, x: Tensor, reverse: bool) -> Tensor:
    """Shift_CAB (denoise1): non-circular temporal half roll, then a CAB on it (N1:157-186)."""
    y, _ = temporal_roll(x, reverse, wrap=False)
    return cab(P, pre, y)


def _rep_conv(P: Params, pre: str, x: Tensor, groups: int) -> Tensor:
    """RepConv: conv5x5 + conv3x3 + identity, grouped or depthwise (D1:157-165 / D2:159-168)."""
    return _conv(P, pre + "conv_1.", x, groups=groups) + _conv(P, pre + "conv_2.", x, groups=groups) + x


def _naf_body(P: Params, pre: str, v: Tensor, V: Variant) -> Tensor:
    """``body`` of CAB1/CAB2 after the LayerNorm (D1:190-201 / N1:219-229).

    body.0 1x1 -> body.1 RepConv2 (dw3x3 + id) -> body.2 SimpleGate ->
    [denoise: body.3 CALayer2] -> RepConv -> 1x1 -> SimpleGate2 -> CALayer2 -> 1x1.
    """
    i = 0
    a = _conv(P, f"{pre}body.{i}.", v); i += 1
    a = _conv(P, f"{pre}body.{i}.conv_2.", a, groups=a.shape[1]) + a; i += 1          # RepConv2 D1:166-174
    a1, a2 = a.chunk(2, dim=1); g = a1 * a2; i += 1                                   # SimpleGate D1:175-178
    if V.denoise:
        g = channel_attention(P, f"{pre}body.{i}.", g); i += 1                        # N1:224,257
    C = g.shape[1]
    g = _rep_conv(P, f"{pre}body.{i}.", g, groups=(C // 8 if V.grouped_rep else C)); i += 1
    b = _conv(P, f"{pre}body.{i}.", g); i += 1
    b1, b2 = b.chunk(2, dim=1); g2 = b1 * torch.sigmoid(b2); i += 1                   # SimpleGate2 D1:179-182
    g2 = channel_attention(P, f"{pre}body.{i}.", g2); i += 1
    return _conv(P, f"{pre}body.{i}.", g2)


def cab1(P: Params, pre: str, x: Tensor, V: Variant) -> Tensor:
    """CAB1: x + beta * body(LN(x)) (D1:183-211)."""
    res = _naf_body(P, pre, layer_norm_2d(x, P[pre + "norm.weight"], P[pre + "norm.bias"]), V)
    return x + res * P[pre + "beta"]


def cab2(P: Params, pre: str, u: Tensor, V: Variant) -> Tensor:
    """CAB2 on the 1.5C-channel gather: shortcut is the ROLLED tensor u[:, :C] (D1:212-255)."""
    C = P[pre + "beta"].shape[1]
    shortcut, hw = u[:, :C], u[:, C:]
    hw = _conv(P, pre + "conv1.", hw, groups=hw.shape[1])                            # dw3x3 D1:223,251
    v = layer_norm_2d(torch.cat((shortcut, hw), 1), P[pre + "norm.weight"], P[pre + "norm.bias"])
    return shortcut + _naf_body(P, pre, v, V) * P[pre + "beta"]


_UNIT_NAMES = ["encoder_level1", "encoder_level1_1", "encoder_level1_2", "encoder_level1_3",
               "encoder_level1_4", "encoder_level1_5", "encoder_level1_6", "encoder_level1_7"]


def gsts_unit(P: Params, pre: str, x: Tensor, reverse: bool, V: Variant) -> Tensor:
    """One GSTS unit = channel_shift -> CAB2 -> CAB1 (``pre`` ends with the Sequential name + '.')."""
    u = gsts_gather(x, reverse, V.wrap)
    return cab1(P, pre + "1.", cab2(P, pre + "0.", u, V), V)


def shift_block(P: Params, pre: str, x: Tensor, V: Variant) -> Tensor:
    """Encoder_shift_block.forward: V.units units, forward/reverse alternating (D1:530-547 / D2:521-530).

    The ``reverse=`` argument Encoder2 passes is ignored by the reference; so it is here.
    """
    for i in range(V.units):
        x = gsts_unit(P, f"{pre}{_UNIT_NAMES[i]}.", x, reverse=(i % 2 == 1), V=V)
    return x


def down_sample(P: Params, pre: str, x: Tensor, V: Variant) -> Tensor:
    """DownSample: 3x3 stride-2 conv (+bias) ; denoise: no bias, then PReLU (D1:330-340 / N1:356-365)."""
    if V.denoise:
        return F.prelu(_conv(P, pre + "down.0.", x, stride=2), P[pre + "down.1.weight"])
    return _conv(P, pre + "down.", x, stride=2)


def skip_up_sample(P: Params, pre: str, x: Tensor, y: Tensor) -> Tensor:
    """SkipUpSample: bilinear x2 (align_corners=False) -> 1x1 -> + y (D1:341-350)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    return _conv(P, pre + "up.1.", x) + y


def tfr_unet(P: Params, pre: str, x: Tensor, V: Variant) -> Tensor:
    """TFR_UNet.forward (D1:709-722 / D2:682-695): 1/3/3 CAB encoder, 3/3/1 decoder, 2 skip CABs."""
    def seq(name: str, n: int, t: Tensor) -> Tensor:
        for i in range(n):
            t = cab(P, f"{pre}{name}.{i}.", t)
        return t
    enc1 = seq("encoder_level1", 1, x)
    enc2 = seq("encoder_level2", 3, down_sample_plain(P, pre + "down12.", enc1, V))
    enc3 = seq("encoder_level3", 3, down_sample_plain(P, pre + "down23.", enc2, V))
    dec3 = seq("decoder_level3", 3, enc3)
    t = skip_up_sample(P, pre + "up32.", dec3, cab(P, pre + "skip_attn2.", enc2))
    dec2 = seq("decoder_level2", 3, t)
    t = skip_up_sample(P, pre + "up21.", dec2, cab(P, pre + "skip_attn1.", enc1))
    return seq("decoder_level1", 1, t)


def down_sample_plain(P: Params, pre: str, x: Tensor, V: Variant) -> Tensor:
    # TFR_UNet uses the same DownSample class as Encoder2, so it follows the variant too.
    return down_sample(P, pre, x, V)


def pixel_shuffle_pack(P: Params, pre: str, x: Tensor) -> Tensor:
    """PixelShufflePack: 3x3 conv (bias) to 4*C0 channels, pixel_shuffle(2) (D1:256-278)."""
    return F.pixel_shuffle(_conv(P, pre + "upsample_conv.", x), 2)


# --------------------------------------------------------------------------- #
# Stage 1 (Encoder2.forward) for the two topologies
# --------------------------------------------------------------------------- #
def stage1(P: Params, x: Tensor, V: Variant) -> Tensor:
    p = "stage1."
    x = cab(P, p + "concat.", x)
    shortcut = x
    if V.shift_cab:                                                                   # N1:645-646
        x = shift_cab(P, p + "encoder_level0.", x, reverse=False)
        x = shift_cab(P, p + "encoder_level0_1.", x, reverse=True)
    x = F.prelu(_conv(P, p + "down01.0.", x, stride=2, padding=0), P[p + "down01.1.weight"])  # D1:576

    if V.topo == "small":                                                             # D2:587-613
        e = shift_block(P, p + "encoder_level1.", x, V)
        e = shift_block(P, p + "encoder_level1_1.", e, V)
        enc11 = shift_block(P, p + "encoder_level1_2.", e, V)
        e = down_sample(P, p + "down12.", enc11, V)
        e = shift_block(P, p + "encoder_level2.", e, V)
        e = shift_block(P, p + "encoder_level2_1.", e, V)
        e = shift_block(P, p + "encoder_level2_2.", e, V)
        d = shift_block(P, p + "decoder_level2.", e, V)
        d = shift_block(P, p + "decoder_level2_1.", d, V)
        d = shift_block(P, p + "decoder_level2_2.", d, V)
        x = skip_up_sample(P, p + "up21.", d, cab(P, p + "skip_attn1.", enc11))
        d = shift_block(P, p + "decoder_level1.", x, V)
        d = shift_block(P, p + "decoder_level1_1.", d, V)
        dec11 = shift_block(P, p + "decoder_level1_2.", d, V)
    else:                                                                             # D1:613-642 / N1:640-670
        if V.shift_cab:
            e = shift_cab(P, p + "encoder_level1.", x, reverse=False)
            enc11 = shift_cab(P, p + "encoder_level1_1.", e, reverse=True)
        else:
            e = cab(P, p + "encoder_level1.", x)
            enc11 = cab(P, p + "encoder_level1_1.", e)
        e = down_sample(P, p + "down12.", enc11, V)
        e = cab(P, p + "encoder_level2.", e)
        enc22 = cab(P, p + "encoder_level2_1.", e)
        e = down_sample(P, p + "down23.", enc22, V)
        e = cab(P, p + "encoder_level3.", e)
        enc33 = cab(P, p + "encoder_level3_1.", e)
        d = shift_block(P, p + "decoder_level3.", enc33, V)
        d = shift_block(P, p + "decoder_level3_1.", d, V)
        x = skip_up_sample(P, p + "up32.", d, cab(P, p + "skip_attn2.", enc22))
        d = shift_block(P, p + "decoder_level2.", x, V)
        d = shift_block(P, p + "decoder_level2_1.", d, V)
        x = skip_up_sample(P, p + "up21.", d, cab(P, p + "skip_attn1.", enc11))
        d = shift_block(P, p + "decoder_level1.", x, V)
        d = shift_block(P, p + "decoder_level1_1.", d, V)
        dec11 = shift_block(P, p + "decoder_level1_2.", d, V)

    up = pixel_shuffle_pack(P, p + "upsample0.", dec11)
    skip = cab(P, p + "skip_conv.", shortcut)
    if V.hr_cat:                                                                      # D1:640
        out = _conv(P, p + "conv_hr0.", torch.cat((up, skip), 1))
    else:                                                                             # D2:611
        out = _conv(P, p + "conv_hr0.", F.prelu(up, P[p + "act.weight"])) + skip
    return cab(P, p + "out_conv.", out)


# --------------------------------------------------------------------------- #
# Whole network
# --------------------------------------------------------------------------- #
def forward(V: Variant, P: Params, x: Tensor, noise_map: Optional[Tensor] = None,
            past: Optional[int] = None, future: Optional[int] = None) -> Tensor:
    """GShiftNet.forward (D1:783-791, D2:748-756, N1:826-835, N2:744-752).

    x: [B,T,3,H,W] (only x[0] is used, as upstream), noise_map: [B,T,1,H,W] for denoise.
    Returns [T-past-future, 3, H, W].
    """
    past = V.past if past is None else past
    future = V.future if future is None else future
    x = x[0]
    T = x.shape[0]
    inp = torch.cat((x, noise_map[0]), 1) if V.denoise else x
    x0 = cab(P, "feat_extract.1.", _conv(P, "feat_extract.0.", inp))

    # stage0 (D1:762-770 / D2:731-737 / N1:788-796)
    t = x0
    for i in range(1, V.n_orb + 1):
        t = tfr_unet(P, f"orb{i}.", t, V)
    res0 = t if V.denoise else t + x0
    sam = _conv(P, "conv_trans.", res0)

    dec = stage1(P, sam, V)

    # stage2 on the interior frames only (D1:771-781 / N1:797-806)
    lo, hi = past, T - future
    feats = sam if V.denoise else res0                                                # N1:834 vs D1:790
    y = _conv(P, "rconcat.", torch.cat((x0[lo:hi], feats[lo:hi], dec[lo:hi]), 1))
    if V.denoise:
        y = F.prelu(y, P["lrelu.weight"])
    sc = y
    for i in range(1, V.n_orb + 1):
        y = tfr_unet(P, f"rorb{i}.", y, V)
    if not V.denoise:
        y = y + sc
    y = _conv(P, "conv_last.", y)
    return y + x[lo:hi]


# --------------------------------------------------------------------------- #
# Host-side harness logic of inference/test_deblur.py (a16 in SURVEY.md §8a)
# --------------------------------------------------------------------------- #
def deblur_windows(n_frames: int, one_len: int) -> List[Tuple[range, range]]:
    """(input frame range, restored frame range) per window (test_deblur.py:111-120).

    k_len=(N-4)//L ; in [kL, kL+L+4) ; out [kL+2, kL+2+L) ; remainder frames are dropped.
    """
    k_len = (n_frames - 4) // one_len
    return [(range(k * one_len, k * one_len + one_len + 4), range(k * one_len + 2, k * one_len + 2 + one_len))
            for k in range(k_len)]


def frames_to_tensor(frames_u8: Sequence) -> Tensor:
    """numpy2tensor (test_deblur.py:191-200): uint8 HWC -> float32 CHW / 255, stacked to [1,T,3,H,W]."""
    import numpy as np
    ts = [torch.from_numpy(np.ascontiguousarray(np.asarray(f).astype("float64").transpose(2, 0, 1))).float().mul_(1.0 / 255)
          for f in frames_u8]
    return torch.stack(ts).unsqueeze(0)


def psnr_255(out_chw: Tensor, gt_u8_hwc) -> float:
    """PSNR as the CLI computes it: clamp(0,1)*255 un-rounded vs uint8 GT, data_range 255 (test_deblur.py:140-142)."""
    import numpy as np
    img = out_chw.clamp(0, 1.0).permute(1, 2, 0).cpu().numpy() * 255
    gt = np.asarray(gt_u8_hwc).astype(np.float64)
    mse = float(np.mean((img.astype(np.float64) - gt) ** 2))
    return float("inf") if mse == 0 else 10.0 * float(np.log10(255.0 ** 2 / mse))
