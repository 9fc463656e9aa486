"""Deterministic synthetic checkpoints.

The reference ships no weights (``pretrained_models/README.md``) and there is no
network here, so tests, ``smoke()`` and ``bench.py`` fill every parameter from
``f(seed, key, shape)``.  The recipe depends only on the key string and the
shape, so the very same tensors can be loaded (``strict=True``) into the
imported reference, into the CPU oracle and into the HIP-backed ``GShiftNet``.

Scales are chosen so that activations stay O(1) through 48..56 residual GSTS
units (checked by ``tests/golden/make_golden.py``): ``beta`` must be non-zero,
otherwise every unit is an identity (it is zero-initialised upstream,
``gshift_deblur1.py:205,240``).

Recipe v2 (round 2).  The first recipe (beta ~ N(0, 0.3), conv_last gain 0.03) made the 48..56-unit residual chain
chaotic: the reference's OWN bf16 CPU run was only 39..42 dB from its fp32 run, so a whole-net parity bound at the
contract's tolerance (PSNR >= 48 dB, |dPSNR| <= 0.01 dB) was unreachable for any bf16 implementation.
With beta ~ N(0, 0.1) and conv_last gain 0.01 the reference's bf16 run is 54..57 dB from fp32 (probed in the build
container on the net fixtures), every unit still contributes a ~10 % residual per CAB, and the restored frame stays a
small correction of the input, as in a trained network.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict

import torch

from .spec import VARIANTS, Variant, param_table


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_state_dict(variant: str | Variant, seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state_dict with the reference's exact key set (aliases share storage)."""
    V = VARIANTS[variant] if isinstance(variant, str) else variant
    tab = param_table(V)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in tab.entries:
        if key in tab.alias:
            out[key] = out[tab.alias[key]]
            continue
        g = _gen(key, seed)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "beta":
            t = torch.randn(shape, generator=g) * 0.1
        elif ".norm." in key and leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                      # PReLU scalar
            t = torch.full(shape, 0.25) + 0.05 * torch.randn(shape, generator=g)
        else:                                      # conv weight [out, in/groups, kh, kw]
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if key.startswith("conv_last"):
                gain = 0.01                         # keep the restored image a small correction of the input
            elif key.endswith("body.2.weight") or key.endswith("up.1.weight"):
                gain = 0.5                          # CAB / SkipUpSample residual branches: no blow-up over 100+ layers
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        out[key] = t.float()
    return out


def alias_groups(variant: str | Variant) -> Dict[str, list]:
    V = VARIANTS[variant] if isinstance(variant, str) else variant
    tab = param_table(V)
    groups: Dict[str, list] = {}
    for k, canon in tab.alias.items():
        groups.setdefault(canon, [canon]).append(k)
    return groups
