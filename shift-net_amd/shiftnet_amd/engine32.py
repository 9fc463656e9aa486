"""fp32 engine: GShiftNet.forward with fp32 activations and the checkpoint's fp32 weights (csrc/sn_f32.hip).

Why it exists
  * upstream runs the "+" denoiser in float32 (inference/test_denoise.py:83-85: ``.half()`` is commented out), so a
    float32 module must compute in float32, not in bf16 behind a cast;
  * it is the validation build of the engine: ``Engine32`` inherits the whole control flow of ``engine.Engine``
    (stage 0 / 1 / 2, TFR_UNet, shift blocks, unit directions, frame trimming) and replaces only the leaf operators, so
    agreement with the CPU oracle to fp32 round-off (tests: <= 1e-4 of the tensor scale) pins that control flow and the
    weight bookkeeping independently of bf16 rounding noise.

Every activation is NHWC fp32 ``[T, H, W, C]`` with C the logical channel count; weights are the checkpoint's fp32 values
(transposed, and for the split-precision convs also as bf16 hi / lo fragments).  Two operator chains compute the same network:

  * the default (round 4) removes the passes that only re-read and re-write an activation: gates and channel sums
    in the producing conv's epilogue, LayerNorm in the consuming conv's loader, CALayer scales on the next conv's loader /
    residual / epilogue, RepConv2 + SimpleGate in one kernel, RepConv as one merged 5x5;
  * ``SN_FP32_FUSE=0`` (and ``SN_FP32_CAB_TAIL=0``) runs one kernel per reference module, in the reference's order -- the
    validation chain: tests/test_gpu_fp32.py holds both to 1e-4 of the oracle and to 1e-5 of each other.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import lib as L
from . import prep
from .engine import Act, Engine, Plan, _dtype_code, host_f32_copy
from .spec import Variant, shift_table


class Plan32(Plan):
    """fp32 weights on the device in natural (checkpoint) order; convs are looked up by the same names as in ``Plan``."""

    def __init__(self, V: Variant, sd: Dict[str, torch.Tensor], device: torch.device) -> None:
        self.V = V
        self.device = device
        self.sd = host_f32_copy(sd)
        self.dsd = {k: v.to(device) for k, v in self.sd.items()}
        self.convs: Dict[str, Dict[str, object]] = {}
        self.cas: Dict[str, Dict[str, object]] = {}
        self.units: Dict[str, Dict[str, object]] = {}
        self._wt: Dict[str, torch.Tensor] = {}
        self._ws: Dict[str, torch.Tensor] = {}
        self.offs = prep.shift_offsets_i8(shift_table(V.c1)).to(device)
        self._build()

    def add_conv(self, name: str, key: str, cins: Sequence[int], shuffle: bool = False) -> None:      # shuffle: the bf16 plan's row order (unused here)
        self.convs[name] = {"key": key, "cins": list(cins)}

    def add_cab(self, pre: str, c: int) -> None:
        self.add_conv(pre + "body.0", pre + "body.0.", [c])
        self.add_conv(pre + "body.2", pre + "body.2.", [c])
        self.add_ca(pre + "CA", pre + "CA.")
        # [cin][9][cpad] copy of the second conv for the closed-form pooled mean of res (sn32_cab_ca): exact fp32 weights
        cpad = max(16, prep.ceil8(c))
        w2 = torch.zeros((c, 9, cpad), dtype=torch.float32)
        w2[:, :, :c] = self.sd[pre + "body.2.weight"].float().reshape(c, c, 9).permute(1, 2, 0)
        self.cas[pre + "CA"]["w2"] = w2.to(self.device)

    def add_naf(self, pre: str, c: int, with_shift: bool) -> None:
        i = 2                                    # body.0 1x1, body.1 RepConv2, (SimpleGate has no entry in the checkpoint)
        i += 1
        if self.V.denoise:
            self.add_ca(f"{pre}ca1", f"{pre}body.{i}."); i += 1
        rep = i; i += 1
        gate = i; i += 2
        self.add_ca(f"{pre}ca2", f"{pre}body.{i}."); i += 1
        self.units[pre] = {"rep": rep, "gate": gate, "out": i}

    def wt(self, key: str) -> torch.Tensor:
        """conv weight [co][ci/g][k][k] -> [k][k][ci/g][co] (consecutive lanes read consecutive output channels)."""
        if key not in self._wt:
            self._wt[key] = self.dsd[key].permute(2, 3, 1, 0).contiguous()
        return self._wt[key]


    def rep_merged(self, rp: str, identity: bool) -> str:
        """RepConv's parallel branches as ONE 5x5 kernel: conv_1 (5x5) + conv_2 (3x3, zero-padded) [+ the identity at the centre tap]
        (gshift_deblur1.py:143-157 computes the three terms separately and adds the results; merging the WEIGHTS in fp32 changes each output by a
        few ulp of its terms).  The identity is merged only for the depthwise form, whose kernel multiplies in exact fp32: in the split-precision
        grouped kernels a weight 1 + w would keep 16 bits relative to 1, so there the identity stays the epilogue's residual."""
        key = rp + ("merged_id.weight" if identity else "merged.weight")
        if key not in self.sd:
            w5, w3 = self.sd[rp + "conv_1.weight"].float(), self.sd[rp + "conv_2.weight"].float()
            m = w5.clone()
            m[:, :, 1:4, 1:4] += w3
            if identity:
                assert m.shape[1] == 1
                m[:, 0, 2, 2] += 1.0
            self.sd[key] = m
            self.dsd[key] = m.to(self.device)
        return key

    def wsplit(self, key: str, groups: int) -> torch.Tensor:
        """bf16 hi / lo A fragments of a conv weight for the split-precision path (prep.pack_conv32_split), built on first use."""
        if key not in self._ws:
            self._ws[key] = prep.pack_conv32_split(self.sd[key], groups).to(self.device)
        return self._ws[key]


class Engine32(Engine):
    def __init__(self, plan: Plan32) -> None:
        super().__init__(plan, torch.float32)
        self.graph_auto = False            # hipGraph replay stays opt-in (SN_GRAPH=1) for the fp32 engine: measured on the bf16 engine only
        self.schedule = "unit"             # the frame wavefront (Engine.shift_chain) is built on the bf16 engine's frame-range launches
        self.range_guard = False           # no half-precision intermediates: nothing to guard (and no sn_se_fold / sn_ca_mlp flag operand is passed)

    act_dtype = torch.float32
    fused_cab_tail = False
    skip_up_lowres = False             # SkipUpSample stays "bilinear x2 in the 1x1's loader" here (sn_upsample2_add is a bf16 pass)
    # Dense k = 1 / 3 and grouped-by-8 k = 5 convs with their operands split into bf16 hi + lo parts (three bf16 MFMAs per k-step instead of
    # eight fp32 ones; ~2^-16 per product, fp32 accumulation).  False: exact fp32 products everywhere (v_mfma_f32_16x16x4_f32), about half as
    # fast -- SN_FP32_EXACT=1, `--fp32_exact` on the CLIs and bench.py.  Both are within 1e-4 of the reference (tests/test_gpu_fp32.py runs both).
    split_bf16 = os.environ.get("SN_FP32_EXACT", "0") != "1"

    # ---- leaf operators ------------------------------------------------------------------------------------
    _CONV_FNS = ("sn32_conv2d", "sn32_conv1x1_gate2")

    def _call(self, fn: str, label: str, *args, alg_bytes: float = 0.0) -> None:
        """Non-conv launches carry their own minimal HBM bytes for bench.py's per-kernel table (convs: the ("conv32", ...) record of _conv32)."""
        if fn not in self._CONV_FNS:
            self._meta = ("ew32", float(alg_bytes))
        super()._call(fn, label, *args)

    def _conv32(self, wkey: str, bkey: Optional[str], ins: Sequence[torch.Tensor], cins: Sequence[int], *, k: int, stride: int = 1,
                pad: Optional[int] = None, groups: int = 1, prelu: Optional[float] = None, res: Optional[torch.Tensor] = None,
                oscale: Optional[torch.Tensor] = None, oscale_stride: int = 0, iscale: Optional[torch.Tensor] = None,
                rscale: Optional[torch.Tensor] = None, ln: Optional[Sequence[torch.Tensor]] = None, csum: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, out_mode: int = 0,
                in_mode: int = 0, c_out: Optional[int] = None, nchw_out: Optional[torch.Tensor] = None,
                nchw_sc: Optional[torch.Tensor] = None, label: str = "") -> torch.Tensor:
        """ins: NHWC fp32 tensors (possibly channel-slice views of wider tensors: the pixel stride is taken from .stride(2))."""
        P = self.P
        w = P.wt(wkey)
        T, hs, ws = ins[0].shape[:3]
        h_in, w_in = (2 * hs, 2 * ws) if in_mode == 1 else (hs, ws)
        if pad is None:
            pad = k // 2
        h_out = (h_in + 2 * pad - k) // stride + 1
        w_out = (w_in + 2 * pad - k) // stride + 1
        co = int(w.shape[3]) if c_out is None else c_out
        d = L.Conv32Desc()
        for i, (t, c) in enumerate(zip(ins, cins)):
            assert t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2)
            d.inp[i], d.c_in[i], d.cs_in[i] = t.data_ptr(), c, t.stride(2)
        d.n_in, d.T, d.h_in, d.w_in, d.in_mode = len(ins), T, h_in, w_in, in_mode
        d.k, d.stride, d.pad, d.groups, d.h_out, d.w_out, d.c_out = k, stride, pad, groups, h_out, w_out, co
        d.w = w.data_ptr()
        cin = sum(cins)
        # split-precision kernels: stride 1, no bilinear loader, NHWC or pixel-shuffle output; dense k = 1 / 3 with up to three concatenated inputs
        # (conv_hr0, rconcat), grouped-by-8 k = 3 / 5 with one.  conv_last (NCHW + shortcut) and the stride-2 / upsampling convs keep exact products.
        # The bilinear x2 loader (SkipUpSample) exists in the flat 1x1 split kernel; shapes it does not cover fall back to exact products in the library.
        if (self.split_bf16 and out_mode in (0, 1) and stride == 1 and c_out is None
                and all(c % 4 == 0 for c in cins) and all(t.stride(2) % 4 == 0 for t in ins)
                and ((in_mode == 0 and groups == 1 and k in (1, 3) and pad == k // 2 and (len(ins) == 1 or iscale is None))
                     or (in_mode == 0 and len(ins) == 1 and groups > 1 and k in (3, 5) and pad == k // 2 and cin // groups == 8 and co // groups == 8 and co % 16 == 0)
                     or (in_mode == 1 and self.split_bilinear and len(ins) == 1 and groups == 1 and k == 1 and pad == 0 and out_mode == 0 and iscale is None))):
            d.wsplit = P.wsplit(wkey, groups).data_ptr()
        d.bias = P.dsd[bkey].data_ptr() if bkey is not None and bkey in P.dsd else None
        d.act, d.prelu = (1, prelu) if prelu is not None else (0, 0.0)
        if oscale is not None:
            d.oscale, d.oscale_stride = oscale.data_ptr(), oscale_stride
        if iscale is not None:                     # [T][stride] per-frame input scale (CALayer of the producer)
            d.iscale, d.iscale_stride = iscale.data_ptr(), iscale.stride(0)
        if rscale is not None:                     # the same for the residual (RepConv of the denoisers: res = g1 * ca1)
            d.rscale, d.rscale_stride = rscale.data_ptr(), rscale.stride(0)
        if ln is not None:                         # LayerNorm2d of the input while it is loaded (split 1x1 kernel; the library refuses other shapes)
            d.ln_w, d.ln_b = ln[0].data_ptr(), ln[1].data_ptr()
        if csum is not None:                       # [T][tiles][cpad] channel sums of the output (split dense 3x3 kernel; the library refuses other shapes)
            assert csum.shape[0] == T and csum.shape[1] == self.lib.sn32_conv_csum_tiles(h_out, w_out) and csum.is_contiguous()
            d.csum, d.csum_cpad = csum.data_ptr(), csum.shape[2]
        if res is not None:
            d.res, d.cs_res = res.data_ptr(), res.stride(2)
        if out_mode == 2:
            # the fp32 epilogues read the shortcut in the OUTPUT's element type (sn32_conv_desc has no sc_dtype): anything else would be reinterpreted
            if nchw_sc.dtype != nchw_out.dtype:
                nchw_sc = nchw_sc.to(nchw_out.dtype)
            assert nchw_sc.is_contiguous() and nchw_sc.shape == nchw_out.shape, (tuple(nchw_sc.shape), tuple(nchw_out.shape))
            d.out, d.sc, d.nchw_dtype, d.cs_out = nchw_out.data_ptr(), nchw_sc.data_ptr(), _dtype_code(nchw_out.dtype), 0
            o = nchw_out
        else:
            if out is None:
                out = self._new(T, 2 * h_out, 2 * w_out, co // 4) if out_mode == 1 else self._new(T, h_out, w_out, co)
            d.out, d.cs_out = out.data_ptr(), out.stride(2)
            o = out
        d.out_mode = out_mode
        self._meta = ("conv32", T, h_out, w_out, sum(cins), co, k, stride, in_mode, out_mode)
        self._call("sn32_conv2d", f"sn32_conv2d[{label or wkey}]", C.byref(d), self._stream())
        return o

    def conv(self, name: str, ins: Sequence[Act], *, stride: int = 1, pad: Optional[int] = None, prelu: Optional[float] = None,
             res: Optional[Act] = None, out_mode: int = 0, pool: bool = False, in_mode: int = 0, oscale: Optional[torch.Tensor] = None,
             nchw_out: Optional[torch.Tensor] = None, nchw_sc: Optional[torch.Tensor] = None, res2: Optional[Act] = None):
        p = self.P.convs[name]
        key = p["key"]
        k = int(self.P.sd[key + "weight"].shape[-1])
        assert oscale is None and res2 is None
        o = self._conv32(key + "weight", key + "bias", [a.t for a in ins], [a.c for a in ins], k=k, stride=stride, pad=pad,
                         prelu=prelu, res=res.t if res is not None else None, out_mode=out_mode, in_mode=in_mode,
                         nchw_out=nchw_out, nchw_sc=nchw_sc, label=name)
        if out_mode == 2:
            return None
        a = Act(o, o.shape[3])
        if pool:
            T, h, w, c = o.shape
            return a, self._chan_sum(o), h * w
        return a

    # CAB with the CALayer scale known BEFORE the second conv (its pooled input is linear in `mid`: sn32_cab_ca), so that scale and residual ride on
    # conv2's epilogue: chan_sum reads mid instead of res, and the pass of sn32_scale_residual over res and x (3 of a CAB's 9 tensor passes) is gone.
    # SN_FP32_CAB_TAIL=0 restores one kernel per reference module.
    cab_closed_form = os.environ.get("SN_FP32_CAB_TAIL", "1") != "0"

    def cab(self, pre: str, x: Act, extra: Optional[Act] = None) -> Act:
        T, h, w, c = x.dims
        if not self.cab_closed_form or h < 2 or w < 2 or x.t.stride(2) != c:
            return super().cab(pre, x, extra)
        q = self.P.cas[pre + "CA"]
        cpad = max(16, prep.ceil8(c))
        if self.fuse_ops and self.split_bf16 and c % 4 == 0:      # the sums of mid come out of conv1's epilogue (no pass of sn32_chan_sum over mid)
            part = torch.empty((T, self.lib.sn32_conv_csum_tiles(h, w), cpad), dtype=torch.float32, device=self.dev)
            k0 = self.P.convs[pre + "body.0"]["key"]
            mid = Act(self._conv32(k0 + "weight", k0 + "bias", [x.t], [c], k=3, prelu=self.P.scalar(pre + "body.1.weight"), csum=part,
                                   label=pre + "body.0"), c)
        else:
            mid = self.conv(pre + "body.0", [x], prelu=self.P.scalar(pre + "body.1.weight"))
            part = self._chan_sum(mid.t)
        scratch = torch.empty((self.lib.sn_cab_ca_scratch_floats(T),), dtype=torch.float32, device=self.dev)
        ca = torch.empty((T, cpad), dtype=torch.float32, device=self.dev)
        self._call("sn32_cab_ca", f"sn32_cab_ca[{pre}]", part.data_ptr(), part.shape[1], cpad, mid.t.data_ptr(), c, q["cr"], h, w,
                   q["w2"].data_ptr(), q["wa"].data_ptr(), q["wb"].data_ptr(), scratch.data_ptr(), ca.data_ptr(), T, self._stream())
        key = self.P.convs[pre + "body.2"]["key"]
        o = self._conv32(key + "weight", key + "bias", [mid.t], [c], k=3, res=x.t, oscale=ca, oscale_stride=cpad, label=pre + "body.2")
        out = Act(o, c)
        if extra is not None:                     # "+ shortcut" after the last TFR_UNet of a stage (gshift_deblur1.py:769,779)
            ones = torch.ones((T, c), dtype=torch.float32, device=self.dev)
            o2 = self._new(T, h, w, c)
            self._call("sn32_scale_residual", "sn32_add", o.data_ptr(), extra.t.data_ptr(), ones.data_ptr(), c, o2.data_ptr(), T, h * w, c, self._stream())
            out = Act(o2, c)
        return out

    def _chan_sum(self, x: torch.Tensor) -> torch.Tensor:
        T, h, w, c = x.shape
        cpad = max(16, prep.ceil8(c))
        nblk = 64
        part = torch.empty((T, nblk, cpad), dtype=torch.float32, device=self.dev)
        self._call("sn32_chan_sum", "sn32_chan_sum", x.data_ptr(), x.stride(2), c, cpad, T, h * w, nblk, part.data_ptr(), self._stream(),
                   alg_bytes=4.0 * T * h * w * c)
        return part

    def _gate_sum(self, a: torch.Tensor, out: torch.Tensor, mode: int) -> torch.Tensor:
        """out = SimpleGate(a) (mode 0) / SimpleGate2(a) (mode 1) and the partial channel sums of out for the CALayer2 behind it, in one pass."""
        T, h, w, c = out.shape
        cpad = max(16, prep.ceil8(c))
        nblk = 64
        part = torch.empty((T, nblk, cpad), dtype=torch.float32, device=self.dev)
        self._call("sn32_gate_sum", "sn32_gate_sum", a.data_ptr(), c, cpad, mode, out.data_ptr(), T, h * w, nblk, part.data_ptr(), self._stream(),
                   alg_bytes=4.0 * T * h * w * 3 * c)
        return part

    def scale_residual(self, r: Act, x: Optional[Act], ca: torch.Tensor, extra: Optional[Act] = None) -> Act:
        T, h, w, c = r.dims
        o = self._new(T, h, w, c)
        self._call("sn32_scale_residual", "sn32_scale_residual", r.t.data_ptr(), x.t.data_ptr() if x is not None else None, ca.data_ptr(),
                   ca.shape[1], o.data_ptr(), T, h * w, c, self._stream())
        if extra is not None:                     # "+ shortcut" after the last TFR_UNet of a stage (gshift_deblur1.py:769,779)
            ones = torch.ones((T, c), dtype=torch.float32, device=self.dev)
            o2 = self._new(T, h, w, c)
            self._call("sn32_scale_residual", "sn32_add", o.data_ptr(), extra.t.data_ptr(), ones.data_ptr(), c, o2.data_ptr(), T, h * w, c,
                       self._stream())
            o = o2
        return Act(o, c)

    def temporal_roll(self, x: Act, reverse: bool) -> Act:
        T, h, w, c = x.dims
        y = self._new(T, h, w, c)
        mode = 2 if reverse else 1
        for wrap, halo, t0, nt in self._split_pieces(x, mode, False):
            s = self._unit_src(x, mode, wrap=wrap, halo=halo, t0=t0, nt=nt)
            self._call("sn32_gsts_gather", "sn32_temporal_roll", C.byref(s), None, y.data_ptr(), None, self._stream())
        return Act(y, c)

    def _ca(self, name: str, g: torch.Tensor) -> torch.Tensor:
        T, h, w, _ = g.shape
        return self.ca_mlp(name, self._chan_sum(g), h * w)

    def naf(self, pre: str, x: Act, mode: int) -> Act:
        self._unit = tuple(x.dims) + (mode,)
        try:
            return self._naf(pre, x, mode)
        finally:
            self._unit = None

    def _naf(self, pre: str, x: Act, mode: int) -> Act:
        """CAB2 (mode 1/2) / CAB1 (mode 0), operator by operator as the reference module lists them (gshift_deblur1.py:183-255)."""
        P, V, st = self.P, self.V, self._stream()
        T, h, w, c = x.dims
        npix = T * h * w
        u = P.units[pre]
        dsd = P.dsd
        self._meta = ("naf32", T, h, w, c, mode)
        if mode and self.fuse_ops and c % 8 == 0:
            vin = self._new(T, h, w, c + c // 2)                                  # cat(shortcut, conv1(shifted)); roll(x) is written once, here
            w1 = P.wt(pre + "conv1.weight")                                       # [3][3][1][C/2]
            us = None if self.fuse_shiftconv else self._new(T, h, w, c // 2)      # shift(borrowed half)
            for wrap, halo, t0, nt in self._split_pieces(x, mode, V.wrap):
                s = self._unit_src(x, mode, wrap=wrap, halo=halo, t0=t0, nt=nt)
                self._call("sn32_gsts_shiftconv", "sn32_gsts_shiftconv", C.byref(s), P.offs.data_ptr(), w1.data_ptr() if us is None else None,
                           vin.data_ptr(), us.data_ptr() if us is not None else None, self._stream(),
                           alg_bytes=4.0 * (nt or T) * h * w * 2.5 * c)
            if us is not None:
                self._conv32(pre + "conv1.weight", None, [us], [c // 2], k=3, groups=c // 2, out=vin[..., c:])
            shortcut = vin[..., :c]
            kk = c + c // 2
        elif mode:
            ug = self._new(T, h, w, c + c // 2)                                   # cat(roll(x), spatial_shift2(borrowed half))
            vin = self._new(T, h, w, c + c // 2)                                  # cat(shortcut, conv1(shifted)): the gather fills [:c] as well, conv1 writes [c:]
            for wrap, halo, t0, nt in self._split_pieces(x, mode, V.wrap):        # (temporal split: the boundary frame after its halo arrived)
                s = self._unit_src(x, mode, wrap=wrap, halo=halo, t0=t0, nt=nt)
                self._call("sn32_gsts_gather", "sn32_gsts_gather", C.byref(s), P.offs.data_ptr(), ug.data_ptr(), vin.data_ptr(), self._stream(),
                           alg_bytes=4.0 * (nt or T) * h * w * 3.5 * c)
            shortcut = ug[..., :c]
            self._conv32(pre + "conv1.weight", None, [ug[..., c:]], [c // 2], k=3, groups=c // 2, out=vin[..., c:])
            kk = c + c // 2
        else:
            shortcut, vin, kk = x.t, x.t, c
        if self.fuse_ops and self.split_bf16 and kk % 4 == 0 and kk <= 128 and h * w >= 64 and vin.stride(2) % 4 == 0:
            a = self._conv32(pre + "body.0.weight", None, [vin], [kk], k=1, ln=(dsd[pre + "norm.weight"], dsd[pre + "norm.bias"]))   # LayerNorm + 1x1 -> 2C
        else:
            v = self._new(T, h, w, kk)
            self._call("sn32_layernorm", "sn32_layernorm", vin.data_ptr(), vin.stride(2), kk, dsd[pre + "norm.weight"].data_ptr(),
                       dsd[pre + "norm.bias"].data_ptr(), v.data_ptr(), kk, npix, st, alg_bytes=8.0 * npix * kk)
            a = self._conv32(pre + "body.0.weight", None, [v], [kk], k=1)                                         # 1x1 -> 2C
        if self.fuse_ops and c % 4 == 0 and a.stride(2) % 4 == 0:
            return self._naf_tail(pre, u, a, shortcut, T, h, w, c)
        a = self._conv32(pre + "body.1.conv_2.weight", None, [a], [2 * c], k=3, groups=2 * c, res=a)              # RepConv2
        g1 = self._new(T, h, w, c)
        if V.denoise:                                                                                               # SimpleGate + CALayer2 on g1
            g1 = self.scale_residual(Act(g1, c), None, self.ca_mlp(f"{pre}ca1", self._gate_sum(a, g1, 0), h * w)).t
        else:
            self._call("sn32_gate", "sn32_gate", a.data_ptr(), c, 0, g1.data_ptr(), npix, st)                     # SimpleGate
        grp = c // 8 if V.grouped_rep else c
        rp = f"{pre}body.{u['rep']}."
        r = self._conv32(rp + "conv_1.weight", None, [g1], [c], k=5, groups=grp, res=g1)                           # RepConv: 5x5 + id
        r = self._conv32(rp + "conv_2.weight", None, [g1], [c], k=3, groups=grp, res=r)                            #          + 3x3
        b = self._conv32(f"{pre}body.{u['gate']}.weight", None, [r], [c], k=1)                                     # 1x1 -> 2C
        g2 = self._new(T, h, w, c)
        ca2 = self.ca_mlp(f"{pre}ca2", self._gate_sum(b, g2, 1), h * w)                                             # SimpleGate2; CALayer2's scale ...
        ok = f"{pre}body.{u['out']}."
        beta = dsd[pre + "beta"].reshape(1, c)
        y = self._conv32(ok + "weight", ok + "bias", [g2], [c], k=1, oscale=beta, oscale_stride=0, res=shortcut,  # shortcut + res * beta
                         iscale=ca2)                                                                                # ... is applied by this conv's loader
        return Act(y, c)

    # Phase 1 with fewer passes over the activations (SN_FP32_FUSE=0: one kernel per reference module, as _naf lists them):
    #   * RepConv2 + SimpleGate [+ the channel sums for the denoisers' CALayer2] in one kernel (sn32_dw_gate: a' is never written);
    #   * RepConv as one 5x5 conv with the merged weights (Plan32.rep_merged) instead of a 5x5 and a 3x3 pass;
    #   * the denoisers' CALayer2 scale on g1 applied by RepConv's loader and on its residual (iscale / rscale; depthwise: the scale
    #     commutes with the conv, so it is the output scale) -- g1 * ca1 is never materialised.
    fuse_ops = os.environ.get("SN_FP32_FUSE", "1") != "0"
    split_bilinear = os.environ.get("SN_FP32_SPLIT_BILINEAR", "1") != "0"   # SkipUpSample's 1x1 on the split-precision kernel (0: the exact-product kernel of round 3)
    fuse_shiftconv = os.environ.get("SN_FP32_SHIFTCONV", "0") == "1"      # conv1 inside the channel_shift kernel: bit-identical, slower (see sn_f32.hip)

    def _naf_tail(self, pre: str, u: Dict[str, object], a: torch.Tensor, shortcut: torch.Tensor, T: int, h: int, w: int, c: int) -> Act:
        P, V, st = self.P, self.V, self._stream()
        cpad = max(16, prep.ceil8(c))
        nblk = 256 if h * w >= 256 * 64 else 64       # workgroups per frame: T x 256 fills the chip in whole rounds at 480p quadrants (T x 64 left a 25 % tail)
        g1 = self._new(T, h, w, c)
        part = torch.empty((T, nblk, cpad), dtype=torch.float32, device=self.dev) if V.denoise else None
        wdw = P.wt(pre + "body.1.conv_2.weight")                                                                  # [3][3][1][2C]
        self._call("sn32_dw_gate", "sn32_dw_gate", a.data_ptr(), a.stride(2), wdw.data_ptr(), c, cpad, g1.data_ptr(), T, h, w, nblk,
                   part.data_ptr() if part is not None else None, st, alg_bytes=4.0 * T * h * w * 3 * c)
        ca1 = self.ca_mlp(f"{pre}ca1", part, h * w) if V.denoise else None
        rp = f"{pre}body.{u['rep']}."
        if V.grouped_rep and (ca1 is None or self.split_bf16):
            r = self._conv32(P.rep_merged(rp, False), None, [g1], [c], k=5, groups=c // 8, res=g1, iscale=ca1, rscale=ca1, label=rp + "merged")
        elif V.grouped_rep:                       # exact products: the residual scale lives in the split-precision kernels only, g1 * ca1 is a pass of its own
            g1 = self.scale_residual(Act(g1, c), None, ca1).t
            r = self._conv32(P.rep_merged(rp, False), None, [g1], [c], k=5, groups=c // 8, res=g1, label=rp + "merged")
        else:
            r = self._conv32(P.rep_merged(rp, True), None, [g1], [c], k=5, groups=c, oscale=ca1, oscale_stride=ca1.stride(0) if ca1 is not None else 0,
                             label=rp + "merged")
        g2 = self._new(T, h, w, c)
        gk = f"{pre}body.{u['gate']}.weight"
        if self.split_bf16 and (h * w) % 64 == 0 and c % 16 == 0 and r.stride(2) % 4 == 0:
            # 1x1 -> 2C, SimpleGate2 and the sums for CALayer2 in the GEMM's epilogue: the 2C-channel tensor never reaches HBM
            part2 = torch.empty((T, (h * w) // 64, cpad), dtype=torch.float32, device=self.dev)
            self._meta = ("conv32", T, h, w, c, 2 * c, 1, 1, 0, 0)
            self._call("sn32_conv1x1_gate2", f"sn32_conv1x1_gate2[{gk}]", r.data_ptr(), r.stride(2), c, P.wsplit(gk, 1).data_ptr(), c, cpad,
                       g2.data_ptr(), T, h * w, part2.data_ptr(), st)
        else:
            b = self._conv32(gk, None, [r], [c], k=1)                                                               # 1x1 -> 2C
            part2 = self._gate_sum(b, g2, 1)
        ca2 = self.ca_mlp(f"{pre}ca2", part2, h * w)
        ok = f"{pre}body.{u['out']}."
        beta = P.dsd[pre + "beta"].reshape(1, c)
        y = self._conv32(ok + "weight", ok + "bias", [g2], [c], k=1, oscale=beta, oscale_stride=0, res=shortcut, iscale=ca2)
        return Act(y, c)

    def _ingest(self, x: torch.Tensor, noise_map: Optional[torch.Tensor]) -> Act:
        T, cin, H, W = x.shape
        nm_ptr = None
        if self.V.denoise:
            noise_map = noise_map.to(x.dtype).expand(T, 1, H, W).contiguous()
            nm_ptr = noise_map.data_ptr()
        xi = self._new(T, H, W, self.V.in_ch)
        self._call("sn32_ingest", "sn32_ingest", x.data_ptr(), _dtype_code(x.dtype), nm_ptr, xi.data_ptr(), T, cin, H, W, self._stream())
        return Act(xi, self.V.in_ch)
