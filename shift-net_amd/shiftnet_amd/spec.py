"""Variant records and checkpoint layout of the four Shift-Net architectures.

The drop-in contract is the reference's ``state_dict`` key set:
``inference/test_deblur.py:85`` loads ``torch.load(path)['params']`` with
``strict=True`` into ``GShiftNet``.  This module states, per variant, every key,
its shape and which keys alias one storage (the reference re-uses one
``nn.PReLU`` instance across all CABs built in the same constructor, so the same
scalar appears under several names: ``gshift_deblur1.py:553-556,594-597``,
``TFR_UNet`` ``:685-707``).  ``arch.GShiftNet`` registers its parameters from
this table; ``tests/test_spec_keys.py`` compares it with key lists captured from
the imported reference (``tests/golden/state_keys_*.json``).

Nothing here is copied module code: it is a flat table builder.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple


@dataclass(frozen=True)
class Variant:
    name: str
    in_ch: int          # 3 deblur / 4 denoise (RGB + sigma map)
    c0: int             # full-resolution width (n_feats0)
    c1: int             # stage-1 width (n_feats2)
    unet_step: int      # TFR_UNet per-level channel increment
    n_orb: int          # TFR_UNets executed per stage (5 are always constructed)
    units: int          # GSTS units per Encoder_shift_block
    wrap: bool          # circular temporal roll (deblur2 only)
    grouped_rep: bool   # RepConv groups = C/8 ("+") instead of depthwise ("-s")
    denoise: bool       # extra CALayer2 + biased last 1x1 + PReLU'd DownSample
    topo: str           # 'plus' | 'small'
    hr_cat: bool        # conv_hr0 consumes cat(up, skip) (2*c0 -> c0, bias)
    shift_cab: bool     # Shift_CAB encoders (denoise1)
    ca_red1: bool       # CALayer/CALayer2 force reduction=1 (deblur2)
    last_k: int         # conv_last kernel size (5 deblur / 3 denoise)
    past: int           # ctor defaults
    future: int


VARIANTS: Dict[str, Variant] = {
    "gshift_deblur1": Variant("gshift_deblur1", 3, 24, 80, 12, 5, 8, False, True, False, "plus", True, False, False, 5, 1, 1),
    "gshift_deblur2": Variant("gshift_deblur2", 3, 14, 64, 4, 3, 4, True, False, False, "small", False, False, True, 5, 1, 1),
    "gshift_denoise1": Variant("gshift_denoise1", 4, 24, 80, 12, 5, 8, False, True, True, "plus", True, True, False, 3, 0, 0),
    "gshift_denoise2": Variant("gshift_denoise2", 4, 14, 64, 4, 3, 4, False, False, True, "small", True, False, False, 3, 0, 0),
}

UNIT_NAMES = ["encoder_level1", "encoder_level1_1", "encoder_level1_2", "encoder_level1_3",
              "encoder_level1_4", "encoder_level1_5", "encoder_level1_6", "encoder_level1_7"]

# (dy, dx) of the source pixel relative to the output pixel for the 16 outer-ring and
# 8 inner-ring shift groups, in channel-group order (gshift_deblur1.py:470-503).
OUTER_OFFSETS = [(-8, -8), (-8, -4), (-8, 0), (-8, 4), (-8, 8), (8, -8), (8, -4), (8, 0), (8, 4), (8, 8),
                 (-4, -8), (-4, 8), (0, -8), (0, 8), (4, -8), (4, 8)]
INNER_OFFSETS = [(-4, -4), (-4, 0), (-4, 4), (0, -4), (0, 4), (4, -4), (4, 0), (4, 4)]


def shift_table(c: int) -> List[Tuple[int, int]]:
    """Per borrowed channel k in [0, c/2): (dy, dx). c=80 -> 2 ch per outer / 1 per inner group; c=64 -> 1 / 2."""
    number = c // 2 // 8
    n2 = (number - 1) // 2
    n1 = number - 2 * n2
    out: List[Tuple[int, int]] = []
    for o in OUTER_OFFSETS:
        out += [o] * n2
    for o in INNER_OFFSETS:
        out += [o] * n1
    assert len(out) == c // 2
    return out


class _Table:
    """Ordered (key, shape) list with alias tracking (alias -> canonical key)."""

    def __init__(self) -> None:
        self.entries: List[Tuple[str, Tuple[int, ...]]] = []
        self.alias: Dict[str, str] = {}

    def add(self, key: str, *shape: int) -> None:
        self.entries.append((key, tuple(shape)))

    def add_alias(self, key: str, canonical: str) -> None:
        self.entries.append((key, (1,)))
        if key != canonical:
            self.alias[key] = canonical


class _Act:
    """A shared PReLU scalar: the first key it is registered under is canonical."""

    def __init__(self) -> None:
        self.canonical: Optional[str] = None

    def register(self, tab: _Table, key: str) -> None:
        if self.canonical is None:
            self.canonical = key
        tab.add_alias(key, self.canonical)


def _ca(tab: _Table, pre: str, c: int, reduction: int, V: Variant) -> None:
    r = 1 if V.ca_red1 else reduction
    tab.add(pre + "conv_du.0.weight", c // r, c, 1, 1)
    tab.add(pre + "conv_du.2.weight", c, c // r, 1, 1)


def _cab(tab: _Table, pre: str, c: int, act: _Act, V: Variant) -> None:
    _ca(tab, pre + "CA.", c, 4, V)
    tab.add(pre + "body.0.weight", c, c, 3, 3)
    act.register(tab, pre + "body.1.weight")
    tab.add(pre + "body.2.weight", c, c, 3, 3)


def _naf(tab: _Table, pre: str, c: int, add: int, V: Variant) -> None:
    """CAB2 (add = c/2) or CAB1 (add = 0)."""
    tab.add(pre + "beta", 1, c, 1, 1)
    if add:
        tab.add(pre + "conv1.weight", add, 1, 3, 3)
    tab.add(pre + "norm.weight", c + add)
    tab.add(pre + "norm.bias", c + add)
    i = 0
    tab.add(f"{pre}body.{i}.weight", 2 * c, c + add, 1, 1); i += 1
    tab.add(f"{pre}body.{i}.conv_2.weight", 2 * c, 1, 3, 3); i += 1
    i += 1  # SimpleGate
    if V.denoise:
        _ca(tab, f"{pre}body.{i}.", c, 4, V); i += 1
    gin = 8 if V.grouped_rep else 1
    tab.add(f"{pre}body.{i}.conv_1.weight", c, gin, 5, 5)
    tab.add(f"{pre}body.{i}.conv_2.weight", c, gin, 3, 3); i += 1
    tab.add(f"{pre}body.{i}.weight", 2 * c, c, 1, 1); i += 1
    i += 1  # SimpleGate2
    _ca(tab, f"{pre}body.{i}.", c, 4, V); i += 1
    tab.add(f"{pre}body.{i}.weight", c, c, 1, 1)
    if V.denoise:
        tab.add(f"{pre}body.{i}.bias", c)


def _shift_block(tab: _Table, pre: str, c: int, V: Variant) -> None:
    for u in range(V.units):
        _naf(tab, f"{pre}{UNIT_NAMES[u]}.0.", c, c // 2, V)
        _naf(tab, f"{pre}{UNIT_NAMES[u]}.1.", c, 0, V)


def _down(tab: _Table, pre: str, cin: int, cout: int, V: Variant) -> None:
    if V.denoise:
        tab.add(pre + "down.0.weight", cout, cin, 3, 3)
        tab.add(pre + "down.1.weight", 1)
    else:
        tab.add(pre + "down.weight", cout, cin, 3, 3)
        tab.add(pre + "down.bias", cout)


def _tfr_unet(tab: _Table, pre: str, V: Variant) -> None:
    act = _Act()
    c = [V.c0, V.c0 + V.unet_step, V.c0 + 2 * V.unet_step]
    for lvl, n in ((1, 1), (2, 3), (3, 3)):
        for i in range(n):
            _cab(tab, f"{pre}encoder_level{lvl}.{i}.", c[lvl - 1], act, V)
    _down(tab, pre + "down12.", c[0], c[1], V)
    _down(tab, pre + "down23.", c[1], c[2], V)
    for lvl, n in ((1, 1), (2, 3), (3, 3)):
        for i in range(n):
            _cab(tab, f"{pre}decoder_level{lvl}.{i}.", c[lvl - 1], act, V)
    _cab(tab, pre + "skip_attn1.", c[0], act, V)
    _cab(tab, pre + "skip_attn2.", c[1], act, V)
    tab.add(pre + "up21.up.1.weight", c[0], c[1], 1, 1)
    tab.add(pre + "up32.up.1.weight", c[1], c[2], 1, 1)


def _stage1(tab: _Table, V: Variant) -> None:
    p = "stage1."
    act = _Act()
    act.register(tab, p + "act.weight")
    c0, c1 = V.c0, V.c1
    if V.topo == "small":
        for n in ("encoder_level1", "encoder_level1_1", "encoder_level1_2",
                  "encoder_level2", "encoder_level2_1", "encoder_level2_2"):
            _shift_block(tab, f"{p}{n}.", c1, V)
    else:
        if V.shift_cab:
            _cab(tab, p + "encoder_level0.", c0, act, V)
            _cab(tab, p + "encoder_level0_1.", c0, act, V)
        for n in ("encoder_level1", "encoder_level1_1", "encoder_level2", "encoder_level2_1",
                  "encoder_level3", "encoder_level3_1"):
            _cab(tab, f"{p}{n}.", c1, act, V)
    _cab(tab, p + "concat.", c0, act, V)
    tab.add(p + "down01.0.weight", c1, c0, 2, 2)
    tab.add(p + "down01.1.weight", 1)
    _down(tab, p + "down12.", c1, c1, V)
    if V.topo == "plus":
        _down(tab, p + "down23.", c1, c1, V)
    if V.topo == "small":
        names = ("decoder_level1", "decoder_level1_1", "decoder_level1_2",
                 "decoder_level2", "decoder_level2_1", "decoder_level2_2")
    else:
        names = ("decoder_level1", "decoder_level1_1", "decoder_level1_2",
                 "decoder_level2", "decoder_level2_1", "decoder_level3", "decoder_level3_1")
    for n in names:
        _shift_block(tab, f"{p}{n}.", c1, V)
    _cab(tab, p + "skip_attn1.", c1, act, V)
    if V.topo == "plus":
        _cab(tab, p + "skip_attn2.", c1, act, V)
    tab.add(p + "upsample0.upsample_conv.weight", 4 * c0, c1, 3, 3)
    tab.add(p + "upsample0.upsample_conv.bias", 4 * c0)
    _cab(tab, p + "skip_conv.", c0, act, V)
    _cab(tab, p + "out_conv.", c0, act, V)
    if V.hr_cat:
        tab.add(p + "conv_hr0.weight", c0, 2 * c0, 3, 3)
        tab.add(p + "conv_hr0.bias", c0)
    else:
        tab.add(p + "conv_hr0.weight", c0, c0, 3, 3)
    tab.add(p + "up21.up.1.weight", c1, c1, 1, 1)
    if V.topo == "plus":
        tab.add(p + "up32.up.1.weight", c1, c1, 1, 1)


def param_table(V: Variant) -> _Table:
    """All state_dict entries of ``V`` in the reference's registration order."""
    tab = _Table()
    tab.add("feat_extract.0.weight", V.c0, V.in_ch, 3, 3)
    tab.add("feat_extract.0.bias", V.c0)
    _cab(tab, "feat_extract.1.", V.c0, _Act(), V)
    tab.add("conv_last.weight", 3, V.c0, V.last_k, V.last_k)
    if V.name == "gshift_denoise1":     # only this file constructs lrelu before conv_trans (gshift_denoise1.py:771-772)
        tab.add("lrelu.weight", 1)
        tab.add("conv_trans.weight", V.c0, V.c0, 3, 3)
        tab.add("conv_trans.bias", V.c0)
    else:
        tab.add("conv_trans.weight", V.c0, V.c0, 3, 3)
        tab.add("conv_trans.bias", V.c0)
        tab.add("lrelu.weight", 1)
    _stage1(tab, V)
    for i in range(1, 6):
        _tfr_unet(tab, f"orb{i}.", V)
    for i in range(1, 6):
        _tfr_unet(tab, f"rorb{i}.", V)
    tab.add("rconcat.weight", V.c0, 3 * V.c0, 3, 3)
    if not V.denoise:
        tab.add("rconcat.bias", V.c0)
    return tab
