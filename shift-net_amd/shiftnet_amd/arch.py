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
    # The device weight plan (prepacked MFMA fragments etc.) is rebuilt whenever the parameters can have changed:
    #   * .to() / .half() / .cuda() go through _apply, a checkpoint loaded into this module goes through load_state_dict, and one
    #     loaded through a PARENT module reaches this class only as _load_from_state_dict (nn.Module recursion): all three drop it;
    #   * in-place edits (p.data.copy_, an EMA swap, optimizer steps) bump the tensors' version counters: the signature checked per
    #     forward holds their sum and the storage address of every parameter (~0.3 ms for ~2000 parameters, against >= 30 ms windows).
    def invalidate_plan(self) -> None:
        self._plan = None

    def _apply(self, fn, *args, **kwargs):
        self._plan = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._plan = None
        return super().load_state_dict(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._plan = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _param_signature(self) -> Tuple:
        """Identity of the parameter storage the prepared plan was built from: an ORDERED hash of (data_ptr, version) per parameter (a XOR / sum
        would cancel when two parameters swap storage).  Inference tensors (module built or loaded under torch.inference_mode()) have no
        version counter: their in-place edits cannot be seen, `_apply` / `load_state_dict` still invalidate the plan."""
        sig = []
        for p in self.parameters():
            sig.append(p.data_ptr())
            sig.append(-1 if p.is_inference() else p._version)
        p0 = next(self.parameters())
        return (p0.device, p0.dtype, hash(tuple(sig)))

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
        sig = self._param_signature()
        if self._plan is None or self._plan_sig != sig:
            with torch.cuda.device(dev):
                self._plan = make_engine(self.V, self.state_dict(), dev, p0.dtype)
            self._plan_sig = sig
        self._plan.split = getattr(self, "_split", None)
        return self._plan

    def guard_scope(self):
        """Context manager: the forwards inside it share ONE range-guard check at its exit (Engine.guard_scope); `.tripped` then says whether
        to run them again.  Not part of the upstream API; the denoise CLIs wrap the four quadrants of a window in it."""
        return self.prepare().guard_scope()

    def forward_fp32_out(self, x, noise_map=None, shortcut=None):
        """``forward`` with the restored frames returned as float32 straight from the last conv's fp32 accumulators (no rounding to a
        half-precision image tensor in between).  ``shortcut`` ([B,T,3,H,W] float32, optional): the un-rounded input frames for the
        final "+ x" (gshift_deblur1.py:791) when ``x`` itself had to be rounded to the module's half-precision dtype.
        Not part of the upstream API: the CLIs use it because they convert the module output with ``.float()`` before clamp * 255 /
        PSNR / imwrite anyway (inference/test_deblur.py:137-143).  A bf16 image tensor quantises [0.5, 1] to steps of 1/256 on the way
        in and on the way out, which moves PSNR-vs-gt by 0.02-0.06 dB for ANY implementation; with this path a bf16 module stays within
        0.01 dB of the float32 reference (tests/test_gpu_parity.py, tests/test_gpu_io.py)."""
        return self._run(x, noise_map, torch.float32, shortcut)

    def forward(self, x, noise_map=None, k1=None, k2=None, k3=None):
        return self._run(x, noise_map, None, None)

    def _run(self, x, noise_map, out_dtype, shortcut):
        eng = self.prepare()
        dt = next(self.parameters()).dtype
        if x.dtype != dt:
            raise RuntimeError(f"Input type ({x.dtype}) and weight type ({dt}) should be the same")
        if self.V.denoise and noise_map is None:
            raise TypeError("noise_map is required by the denoise variants")
        nm = noise_map[0] if noise_map is not None else None
        return eng.forward(x[0], nm, self.num_fb, self.num_ff, out_dtype, shortcut[0] if shortcut is not None else None)


def _make(variant: str):
    V = VARIANTS[variant]

    class GShiftNet(GShiftNetBase):
        def __init__(self, n_features=48, future_frames=V.future, past_frames=V.past):
            super().__init__(n_features, future_frames, past_frames)

        if not V.denoise:
            def forward(self, x, k1=None, k2=None, k3=None):          # deblur signature has no noise_map
                return self._run(x, None, None, None)

            def forward_fp32_out(self, x, shortcut=None):
                return self._run(x, None, torch.float32, shortcut)

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
