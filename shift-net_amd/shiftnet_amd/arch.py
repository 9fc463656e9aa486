"""``GShiftNet``: the drop-in architecture class (same ctor, ``state_dict`` key set and ``forward`` contract as
``basicsr/models/archs/gshift_*.py`` upstream), executing on the MI355X HIP kernels.

Contract mirrored:
  * ``GShiftNet(n_features=48, future_frames=F, past_frames=P)``; ``n_features`` is stored and unused upstream;
  * ``forward(x[, noise_map], k1=None, k2=None, k3=None)``: ``x:[B,T,3,H,W]`` in the parameters' dtype, only
    ``x[0]`` is used, returns ``[T-P-F, 3, H, W]``; ``T <= P+F`` gives an empty tensor;
  * ``load_state_dict(torch.load(p)['params'])`` strict: keys/shapes/aliases come from ``spec.param_table``.
There is no CPU path: calling ``forward`` without a HIP device and the built library raises.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .spec import VARIANTS, Variant, param_table


class _Node(nn.Module):
    """Anonymous container so that dotted checkpoint keys map onto a module tree."""


class GShiftNetBase(nn.Module):
    variant: str = ""

    def __init__(self, n_features: int = 48, future_frames: Optional[int] = None, past_frames: Optional[int] = None):
        super().__init__()
        V = VARIANTS[self.variant]
        self.V: Variant = V
        self.n_feats = n_features
        self.num_ff = V.future if future_frames is None else future_frames
        self.num_fb = V.past if past_frames is None else past_frames
        tab = param_table(V)
        made: Dict[str, nn.Parameter] = {}
        for key, shape in tab.entries:
            canon = tab.alias.get(key, key)
            if canon not in made:
                made[canon] = nn.Parameter(self._init(canon, shape), requires_grad=False)
            self._attach(key, made[canon])
        self._plan = None
        self._plan_sig: Optional[Tuple] = None

    @staticmethod
    def _init(key: str, shape) -> torch.Tensor:
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "beta" or leaf == "bias":
            return torch.zeros(shape)                       # upstream: beta zeros (gshift_deblur1.py:205), conv bias small
        if len(shape) == 1:
            return torch.full(shape, 0.25) if shape[0] == 1 else torch.ones(shape)   # PReLU 0.25 / LN weight 1
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape) * (1.0 / fan_in) ** 0.5

    def _attach(self, key: str, p: nn.Parameter) -> None:
        parts = key.split(".")
        mod: nn.Module = self
        for name in parts[:-1]:
            if name not in mod._modules:
                mod.add_module(name, _Node())
            mod = mod._modules[name]
        mod.register_parameter(parts[-1], p)

    # ------------------------------------------------------------------------------------------------------
    # The device weight plan is rebuilt only when the parameters can have changed: nn.Module funnels .to()/.half()/.cuda()
    # through _apply and checkpoint loading through load_state_dict, so both drop the plan (no per-forward walk over the
    # ~2000 parameters).  Code that edits parameters in place must call invalidate_plan() itself.
    def invalidate_plan(self) -> None:
        self._plan = None

    def _apply(self, fn, *args, **kwargs):
        self._plan = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._plan = None
        return super().load_state_dict(*args, **kwargs)

    def set_temporal_split(self, rank: int, world: int, group=None) -> None:
        """Make this module process frames [a, b) of ONE long window sharded over `world` ranks (temporal_split.py):
        ``forward`` then takes the rank's own frames and returns the frames it restores; results equal the single-GPU run on
        the whole window.  ``world == 1`` switches back to the ordinary behaviour."""
        from .temporal_split import TemporalSplit
        self._split = TemporalSplit(rank, world, self.V.wrap, group) if world > 1 else None
        if self._plan is not None:
            self._plan.split = self._split

    def prepare(self):
        """Build the device weight plan on first use after a load / dtype / device change."""
        from .engine import make_engine
        p0 = next(self.parameters())
        dev = p0.device
        if dev.type != "cuda":
            raise RuntimeError("GShiftNet runs on the MI355X HIP kernels only: move the module to a HIP ('cuda') device. "
                               "There is no CPU fallback.")
        sig = (dev, p0.dtype)
        if self._plan is None or self._plan_sig != sig:
            with torch.cuda.device(dev):
                self._plan = make_engine(self.V, self.state_dict(), dev, p0.dtype)
            self._plan_sig = sig
        self._plan.split = getattr(self, "_split", None)
        return self._plan

    def forward(self, x, noise_map=None, k1=None, k2=None, k3=None):
        eng = self.prepare()
        dt = next(self.parameters()).dtype
        if x.dtype != dt:
            raise RuntimeError(f"Input type ({x.dtype}) and weight type ({dt}) should be the same")
        if self.V.denoise and noise_map is None:
            raise TypeError("noise_map is required by the denoise variants")
        nm = noise_map[0] if noise_map is not None else None
        return eng.forward(x[0], nm, self.num_fb, self.num_ff)


def _make(variant: str):
    V = VARIANTS[variant]

    class GShiftNet(GShiftNetBase):
        def __init__(self, n_features=48, future_frames=V.future, past_frames=V.past):
            super().__init__(n_features, future_frames, past_frames)

        if not V.denoise:
            def forward(self, x, k1=None, k2=None, k3=None):          # deblur signature has no noise_map
                return GShiftNetBase.forward(self, x, None, k1, k2, k3)

    GShiftNet.variant = variant
    GShiftNet.__qualname__ = "GShiftNet"
    return GShiftNet


GShiftNetDeblur1 = _make("gshift_deblur1")
GShiftNetDeblur2 = _make("gshift_deblur2")
GShiftNetDenoise1 = _make("gshift_denoise1")
GShiftNetDenoise2 = _make("gshift_denoise2")
CLASSES = {"gshift_deblur1": GShiftNetDeblur1, "gshift_deblur2": GShiftNetDeblur2,
           "gshift_denoise1": GShiftNetDenoise1, "gshift_denoise2": GShiftNetDenoise2}


def make_model(variant: str, opt) -> nn.Module:
    """``make_model(opt)`` of the reference arch files (gshift_deblur1.py:9-16): touches opt['pretrain_models_dir'] only."""
    _ = opt["pretrain_models_dir"] + "network-default.pytorch"
    return CLASSES[variant]()
