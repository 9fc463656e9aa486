"""Execution of GShiftNet.forward on the HIP kernels (one process = one GPU = one clip window).

``Plan``   : checkpoint tensors prepared once per device into kernel layouts (prep.py).
``Engine`` : the forward pass as a straight-line sequence of C-ABI calls on torch's current HIP stream.

Activations are NHWC bf16 torch tensors ``[T, H, W, Cs]`` (PyTorch is only the allocator / stream owner here);
no op below has a torch/ATen compute fallback.  The control flow mirrors the reference forward of each variant
(file:line cited per method) so parity can be audited line by line against ``oracle/shiftnet_oracle.py``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as L
from . import prep
from .spec import UNIT_NAMES, Variant, shift_table


@dataclass
class Act:
    """NHWC bf16 activation: t [T,H,W,Cs] with c logical channels (pad channels are zero)."""
    t: torch.Tensor
    c: int

    @property
    def dims(self) -> Tuple[int, int, int, int]:
        return tuple(self.t.shape)  # type: ignore[return-value]


def _dtype_code(dt: torch.dtype) -> int:
    return {torch.float32: L.SN_F32, torch.float16: L.SN_F16, torch.bfloat16: L.SN_BF16}[dt]


def host_f32_copy(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """fp32 CPU copies of a state dict with ONE device-to-host transfer: the ~2000 tensors of a checkpoint are flattened into one
    buffer on their device first (per-key .cpu() calls cost 5755 copyBuffer launches / 27 ms per plan build in the round-2 profile)."""
    keys = list(sd.keys())
    if not keys:
        return {}
    dev_keys = [k for k in keys if sd[k].device.type != "cpu"]
    out = {k: sd[k].detach().float() for k in keys if sd[k].device.type == "cpu"}
    if dev_keys:
        flat = torch.cat([sd[k].detach().reshape(-1).float() for k in dev_keys]).cpu()
        o = 0
        for k in dev_keys:
            n = sd[k].numel()
            out[k] = flat[o:o + n].reshape(sd[k].shape)
            o += n
    return {k: out[k] for k in keys}


class Plan:
    """Device-resident prepared weights of one checkpoint."""

    def __init__(self, V: Variant, sd: Dict[str, torch.Tensor], device: torch.device) -> None:
        self.V = V
        self.device = device
        self.sd = host_f32_copy(sd)
        self.convs: Dict[str, Dict[str, object]] = {}
        self.cas: Dict[str, Dict[str, object]] = {}
        self.units: Dict[str, Dict[str, object]] = {}
        self.offs = prep.shift_offsets_i8(shift_table(V.c1)).to(device)
        # sn_gsts_shiftconv_mfma reads 16 bytes of a channel-planar LDS row at a displaced column: 8-byte aligned only if every displacement is a multiple
        # of 4 pixels (gshift_deblur1.py:411-439: they are 0, +-4, +-8) and within the 34 x 34 window; anything else stays on the VALU kernel
        self.k0_mfma_ok = all(dx % 4 == 0 and abs(dy) <= 8 and abs(dx) <= 8 for dy, dx in shift_table(V.c1))
        self._build()

    # -- helpers ---------------------------------------------------------------------------------------------
    def _dev(self, t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        return None if t is None else t.contiguous().to(self.device)

    def add_conv(self, name: str, key: str, cins: Sequence[int], shuffle: bool = False) -> None:
        w = self.sd[key + "weight"]
        b = self.sd.get(key + "bias")
        cs_in = prep.ceil8(max(cins))
        p = prep.pack_conv(w, b, cins, cs_in, shuffle)
        p["wfrag"] = self._dev(p["wfrag"])
        p["bias"] = self._dev(p["bias"])
        self.convs[name] = p

    def add_ca(self, name: str, key: str) -> None:
        wa = self.sd[key + "conv_du.0.weight"]
        wb = self.sd[key + "conv_du.2.weight"]
        cr, c = wa.shape[0], wa.shape[1]
        self.cas[name] = {"wa": self._dev(wa.reshape(cr, c)), "wb": self._dev(wb.reshape(c, cr)), "c": c, "cr": cr}

    def scalar(self, key: str) -> float:
        return float(self.sd[key].reshape(-1)[0])

    def add_cab(self, pre: str, c: int) -> None:
        self.add_conv(pre + "body.0", pre + "body.0.", [c])
        self.add_conv(pre + "body.2", pre + "body.2.", [c])
        self.add_ca(pre + "CA", pre + "CA.")
        # fp32 [cin][9][cpad] copy of the second conv for the closed-form pooled mean of res (sn_cab_ca); bf16-rounded like the MFMA operand
        cpad = 16 * int(self.convs[pre + "body.2"]["mt"])
        w2 = torch.zeros((c, 9, cpad), dtype=torch.float32)
        w2[:, :, :c] = self.sd[pre + "body.2.weight"].to(torch.bfloat16).float().reshape(c, c, 9).permute(1, 2, 0)
        self.cas[pre + "CA"]["w2"] = self._dev(w2)

    def add_naf(self, pre: str, c: int, with_shift: bool) -> None:
        V, sd = self.V, self.sd
        u: Dict[str, object] = {}
        i = 0
        if with_shift:
            w1 = sd[pre + "conv1.weight"].reshape(c // 2, 9)
            u["w1"] = self._dev((w1.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF).contiguous())   # bf16 in the low half
        g = prep.pack_ln_gemm(sd[f"{pre}body.{i}.weight"], sd[pre + "norm.weight"], sd[pre + "norm.bias"], c); i += 1
        u["w_ln"], u["b_ln"] = self._dev(g["wfrag"]), self._dev(g["bias"])
        w3 = prep.pack_dw3_gate(sd[f"{pre}body.{i}.conv_2.weight"], c); i += 1
        u["w_dw3_h2"] = self._dev(prep.pk_f16_words(w3))        # K12: packed-fp16 stencil, [9][C] words = positions (2k, 2k+1)
        i += 1
        if V.denoise:
            self.add_ca(f"{pre}ca1", f"{pre}body.{i}."); i += 1
        if V.grouped_rep:
            u["w_grp"] = self._dev(prep.pack_grouped_frag(sd[f"{pre}body.{i}.conv_1.weight"], sd[f"{pre}body.{i}.conv_2.weight"]))
        else:
            w5 = prep.pack_dw5(sd[f"{pre}body.{i}.conv_1.weight"], sd[f"{pre}body.{i}.conv_2.weight"])
            u["w_toep5"] = self._dev(prep.pack_toeplitz(w5, 5))           # K3m: 5x5 on the matrix cores
        i += 1
        u["w_gate"] = self._dev(prep.pack_gate_gemm(sd[f"{pre}body.{i}.weight"], c))
        # fused phase 1 (sn_gsts_cab2_phase1 / sn_cab1_phase1, csrc/sn_phase1r.hip); the denoisers run it twice (inner CALayer2: sn_phase1_opts)
        d = {k: self._dev(v) for k, v in prep.pack_phase1r(
            sd[f"{pre}body.0.weight"], sd[pre + "norm.weight"], sd[pre + "norm.bias"], sd[f"{pre}body.1.conv_2.weight"],
            sd[f"{pre}body.{i - 1}.conv_1.weight"], sd[f"{pre}body.{i - 1}.conv_2.weight"], sd[f"{pre}body.{i}.weight"], c).items()}
        d["desc"] = L.Phase1Weights(d["wfrag1"].data_ptr(), d["w3"].data_ptr(), d["wgrp"].data_ptr(), d["wfrag2"].data_ptr())   # the tensors stay referenced in d
        u["p1r"] = d
        i += 1
        i += 1
        self.add_ca(f"{pre}ca2", f"{pre}body.{i}."); i += 1
        o = prep.pack_out_gemm(sd[f"{pre}body.{i}.weight"], sd[pre + "beta"], sd.get(f"{pre}body.{i}.bias"), c)
        u["w_out"], u["b_out"] = self._dev(o["wfrag"]), self._dev(o["bias"])
        self.units[pre] = u

    def add_shift_block(self, pre: str, c: int) -> None:
        for k in range(self.V.units):
            self.add_naf(f"{pre}{UNIT_NAMES[k]}.0.", c, True)
            self.add_naf(f"{pre}{UNIT_NAMES[k]}.1.", c, False)

    def add_down(self, pre: str, cin: int) -> None:
        self.add_conv(pre + "down", pre + ("down.0." if self.V.denoise else "down."), [cin])

    def add_unet(self, pre: str) -> None:
        V = self.V
        c = [V.c0, V.c0 + V.unet_step, V.c0 + 2 * V.unet_step]
        for lvl, n in ((1, 1), (2, 3), (3, 3)):
            for i in range(n):
                self.add_cab(f"{pre}encoder_level{lvl}.{i}.", c[lvl - 1])
                self.add_cab(f"{pre}decoder_level{lvl}.{i}.", c[lvl - 1])
        self.add_down(pre + "down12.", c[0])
        self.add_down(pre + "down23.", c[1])
        self.add_cab(pre + "skip_attn1.", c[0])
        self.add_cab(pre + "skip_attn2.", c[1])
        self.add_conv(pre + "up21", pre + "up21.up.1.", [c[1]])
        self.add_conv(pre + "up32", pre + "up32.up.1.", [c[2]])

    def _build(self) -> None:
        V = self.V
        c0, c1 = V.c0, V.c1
        self.add_conv("feat_extract.0", "feat_extract.0.", [V.in_ch])
        self.add_cab("feat_extract.1.", c0)
        self.add_conv("conv_trans", "conv_trans.", [c0])
        self.add_conv("conv_last", "conv_last.", [c0])
        self.add_conv("rconcat", "rconcat.", [c0, c0, c0])
        for i in range(1, V.n_orb + 1):
            self.add_unet(f"orb{i}.")
            self.add_unet(f"rorb{i}.")
        p = "stage1."
        self.add_cab(p + "concat.", c0)
        self.add_conv(p + "down01", p + "down01.0.", [c0])
        self.add_down(p + "down12.", c1)
        if V.topo == "small":
            blocks = ["encoder_level1", "encoder_level1_1", "encoder_level1_2", "encoder_level2", "encoder_level2_1",
                      "encoder_level2_2", "decoder_level2", "decoder_level2_1", "decoder_level2_2",
                      "decoder_level1", "decoder_level1_1", "decoder_level1_2"]
        else:
            self.add_down(p + "down23.", c1)
            if V.shift_cab:
                self.add_cab(p + "encoder_level0.", c0)
                self.add_cab(p + "encoder_level0_1.", c0)
            for n in ("encoder_level1", "encoder_level1_1", "encoder_level2", "encoder_level2_1",
                      "encoder_level3", "encoder_level3_1"):
                self.add_cab(f"{p}{n}.", c1)
            self.add_cab(p + "skip_attn2.", c1)
            self.add_conv(p + "up32", p + "up32.up.1.", [c1])
            blocks = ["decoder_level3", "decoder_level3_1", "decoder_level2", "decoder_level2_1",
                      "decoder_level1", "decoder_level1_1", "decoder_level1_2"]
        for b in blocks:
            self.add_shift_block(f"{p}{b}.", c1)
        self.add_cab(p + "skip_attn1.", c1)
        self.add_conv(p + "up21", p + "up21.up.1.", [c1])
        self.add_conv(p + "upsample0", p + "upsample0.upsample_conv.", [c1], shuffle=True)
        self.add_cab(p + "skip_conv.", c0)
        self.add_cab(p + "out_conv.", c0)
        self.add_conv(p + "conv_hr0", p + "conv_hr0.", [c0, c0] if V.hr_cat else [c0])


class Engine:
    """bf16-storage engine (fp16 / bf16 modules).  fp32 modules run on engine32.Engine32 (same control flow, fp32 kernels)."""

    def __init__(self, plan: Plan, dtype: torch.dtype = torch.bfloat16) -> None:
        self.P = plan
        self.dtype = dtype
        self.V = plan.V
        self.lib = L.load()
        self.dev = plan.device
        self.prof: Optional[list] = None      # bench.py attaches a list to collect (fn, label, meta, ev0, ev1)
        self._meta: Tuple = ()
        self._unit: Optional[Tuple] = None    # (T, h, w, c, mode) while the launches of a CAB2 / CAB1 of a GSTS unit are being issued
        self.split = None                     # temporal_split.TemporalSplit: this engine holds a frame range of a longer window
        self._side = None                     # side stream of the halo exchanges (created on first use)
        self._tickets = None                  # sn_se_fold frame counters: zero between launches (the kernels re-arm them)
        self._bad = None                      # the range guard's device flag (one u32), created on first use
        self.fallbacks = 0                    # how often the guard moved this engine from the fused phase 1 to the bf16 chain (0 or 1)
        if self.phase1 not in ("auto", "r", "0"):
            raise ValueError(f"SN_PHASE1={self.phase1!r}: expected auto, r (fused kernel) or 0 (two-kernel bf16 chain)")
        # hipGraph replay of the whole forward (~1400 launches per window): the first call with a given input signature runs eagerly, the second
        # one is captured, later ones replay, so the Python / ctypes / allocator work per launch disappears.  Measured on MI355X: neutral at
        # 1280x720 (126.4 vs 126.0 ms per window: the GPU is never starved there, kernels average 90 us), 1.39x on a 64x96 clip where the ~10 us
        # squeeze-excite kernels dominate the launch stream.  SN_GRAPH=1: always, 0: never, default: windows below GRAPH_AUTO_PXF pixel-frames.
        g = os.environ.get("SN_GRAPH", "auto")
        self.use_graph = g == "1"
        self.graph_auto = g not in ("0", "1")
        self._graphs: "OrderedDict[Tuple, object]" = OrderedDict()   # LRU over input signatures, at most GRAPH_SLOTS captured graphs alive

    # ---- low level wrappers --------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _call(self, fn: str, label: str, *args) -> None:
        """One C-ABI launch on the current stream; with a profiler attached, bracketed by stream events."""
        f = getattr(self.lib, fn)
        if self.prof is None:
            L.check(f(*args), label)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(f(*args), label)
        e1.record()
        meta = self._meta
        if self._unit is not None and not (meta and meta[0] == "naf"):       # a kernel of the unit with its own meta (the fp32 engine's operators)
            meta = tuple(meta) + ("unit",) + self._unit
        self.prof.append((fn, label, meta, e0, e1))

    act_dtype = torch.bfloat16

    def _new(self, T: int, h: int, w: int, cs: int) -> torch.Tensor:
        return torch.empty((T, h, w, cs), dtype=self.act_dtype, device=self.dev)

    # single-input 3x3 stride-1 convs: the persistent streaming kernel (csrc/sn_conv3p.hip) or, SN_CONV_TILES=1, the one-workgroup-per-tile kernel
    conv_tiles = os.environ.get("SN_CONV_TILES", "0") == "1"
    conv_stream_all = os.environ.get("SN_CONV_STREAM_ALL", "0") == "1"      # measurements / tests: the streaming kernel also where the library prefers the tile kernel
    conv_res_regs = os.environ.get("SN_CONV_RES_REGS", "0") == "1"          # measurements: the residual operand of the streaming kernel through registers, not LDS
    conv_dbg = int(os.environ.get("SN_CONV_DBG", "0"))                      # measurements (wrong results): 1 no DMA, 2 no B reads / MFMAs, 4 no stores
    conv_depth = int(os.environ.get("SN_CONV_DEPTH", "0"))                  # measurements: 1 / 2 = three / four tiles of prefetch (16 channels)
    conv_wgs = int(os.environ.get("SN_CONV_WGS", "0"))            # measurements: persistent workgroups per CU of the streaming kernel (0: the library's choice)
    conv_s2_small = os.environ.get("SN_CONV_S2_SMALL", "0") == "1"  # measurements: stride-2 convs on 4 x 16 tiles whatever their width (round 5's choice)

    def conv(self, name: str, ins: Sequence[Act], *, stride: int = 1, pad: Optional[int] = None, prelu: Optional[float] = None,
             res: Optional[Act] = None, out_mode: int = 0, pool: bool = False, in_mode: int = 0, oscale: Optional[torch.Tensor] = None,
             nchw_out: Optional[torch.Tensor] = None, nchw_sc: Optional[torch.Tensor] = None, res2: Optional[Act] = None):
        p = self.P.convs[name]
        k, cout = int(p["k"]), int(p["cout"])
        T, hs, ws, cs_in = ins[0].dims
        assert cs_in == p["cs_in"] and len(ins) == p["n_in"], (name, cs_in, p["cs_in"])
        h_in, w_in = (2 * hs, 2 * ws) if in_mode == 1 else (hs, ws)
        if pad is None:
            pad = k // 2
        h_out = (h_in + 2 * pad - k) // stride + 1
        w_out = (w_in + 2 * pad - k) // stride + 1
        d = L.ConvDesc()
        for i in range(3):
            d.inp[i] = ins[i].t.data_ptr() if i < len(ins) else None
        d.n_in, d.cs_in, d.T, d.h_in, d.w_in, d.in_mode = len(ins), cs_in, T, h_in, w_in, in_mode
        d.k, d.stride, d.pad, d.h_out, d.w_out = k, stride, pad, h_out, w_out
        d.wfrag, d.mt, d.ks = p["wfrag"].data_ptr(), int(p["mt"]), int(p["ks"])
        d.bias = p["bias"].data_ptr() if p["bias"] is not None else None
        d.act, d.prelu = (1, prelu) if prelu is not None else (0, 0.0)
        pool_buf = None
        out_act: Optional[Act] = None
        if out_mode == 2:
            assert nchw_out is not None and nchw_sc is not None
            d.out, d.sc, d.cs_out = nchw_out.data_ptr(), nchw_sc.data_ptr(), 8
            d.c_out, d.nchw_dtype, d.sc_dtype = cout, _dtype_code(nchw_out.dtype), _dtype_code(nchw_sc.dtype)
        else:
            c_log = cout // 4 if out_mode == 1 else cout
            cs_out = prep.ceil8(c_log)
            o = self._new(T, 2 * h_out, 2 * w_out, cs_out) if out_mode == 1 else self._new(T, h_out, w_out, cs_out)
            out_act = Act(o, c_log)
            d.out, d.cs_out, d.c_out = o.data_ptr(), cs_out, cout
        d.out_mode = out_mode
        d.flags = L.SN_CONV_TILE_KERNEL if self.conv_tiles else ((self.conv_wgs << 4) | (256 if self.conv_stream_all else 0) | (512 if self.conv_res_regs else 0) | (self.conv_depth << 10) | (self.conv_dbg << 12))
        if self.conv_s2_small:
            d.flags |= 1 << 15
        if res is not None:
            assert out_mode == 0 and res.dims == out_act.dims
            d.res = res.t.data_ptr()
        if res2 is not None:
            assert out_mode == 0 and res2.dims == out_act.dims
            d.res2 = res2.t.data_ptr()
        if oscale is not None:
            d.oscale, d.oscale_stride = oscale.data_ptr(), oscale.shape[1]
        if pool:
            nblk = self.lib.sn_conv_pool_blocks(C.byref(d))
            pool_buf = torch.empty((T, nblk, 16 * d.mt), dtype=torch.float32, device=self.dev)
            d.pool = pool_buf.data_ptr()
        self._meta = ("conv", T, h_out, w_out, len(ins) * cs_in, int(d.cs_out), k, stride, in_mode, out_mode)
        self._call("sn_conv2d", f"sn_conv2d[{name}]", C.byref(d), self._stream())
        if pool:
            return out_act, pool_buf, h_out * w_out
        return out_act

    def ca_mlp(self, name: str, pool: torch.Tensor, npix: int) -> torch.Tensor:
        p = self.P.cas[name]
        T, nblk, cpad = pool.shape
        ca = torch.empty((T, cpad), dtype=torch.float32, device=self.dev)
        self._call("sn_ca_mlp", f"sn_ca_mlp[{name}]", pool.data_ptr(), nblk, cpad, p["c"], p["cr"], 1.0 / npix, p["wa"].data_ptr(),
                                   p["wb"].data_ptr(), ca.data_ptr(), T, None, self._stream())
        return ca

    # ---- blocks (oracle/shiftnet_oracle.py has the same names) ----------------------------------------------
    # K0 = CAB2.conv1(spatial_shift2(borrowed half)): "1" the depthwise 3x3 as a banded GEMM on the matrix cores over a channel-planar LDS window
    # (sn_gsts_shiftconv_mfma, round 6); "0" the VALU kernel of rounds 1-5 (nine 2-byte LDS reads + nine v_dot2c per pixel and channel)
    k0_mfma = os.environ.get("SN_K0_MFMA", "1") != "0"
    # Denoisers (inner CALayer2 on g1): the sums pass of the fused phase 1 stores its g1 rows (fp16, kernel order) and the second pass reads them back
    # instead of running LayerNorm -> 1x1 -> 3x3 -> gate again (sn_phase1_opts.g1_store; bit-identical).  SN_G1_STORE=0: both passes from the input.
    g1_store = os.environ.get("SN_G1_STORE", "1") != "0"
    fold_se = True             # fused phase 1: CALayer2's MLP is finished by the frame's last workgroup (sn_se_fold) instead of an sn_ca_mlp launch
    # Phase 1 of CAB2 / CAB1.  "r": the role-split fused kernel (csrc/sn_phase1r.hip, every variant; `a`, g1 = a1 a2 2^-4 and r are fp16 inside it);
    # "0": the two-kernel chain sn_ln_gemm_gate + sn_dw5m_gemm_gate / sn_grp5_gemm_gate (g1 in bf16 through HBM: a product of two activations
    # cannot leave ITS range).  SN_PHASE1 picks one; "auto" starts fused and lets the range guard decide.
    # C = 64, 20 x 360 x 640, CAB1 / CAB2: "r" 565 / 696 us (round 4), chain 837 / 950 us
    phase1 = os.environ.get("SN_PHASE1", "auto")
    # Range guard of the half-precision intermediates: every squeeze-excite reduction of the GSTS path (the sn_se_fold tail, sn_ca_mlp)
    # raises a device flag when a channel sum is not finite -- what an overflowed fp16 `a`, g1 or r upstream turns into.  forward() reads the flag
    # once per call; a raised flag on the fused path switches THIS engine to the chain for good, warns, and recomputes the window.
    range_guard = os.environ.get("SN_RANGE_GUARD", "1") != "0"
    # "1": forward() reads the flag before it hands the window out (one blocking 4-byte copy: the window can be recomputed).  "async": the flag goes
    # to pinned host memory behind the forward and is looked at when the NEXT forward starts (or by check_range_guard()): no device sync at the end
    # of a window, host preparation of the next one overlaps the GPU -- but a tripped window has already been handed out: it is reported, the
    # engine moves to the chain for the windows that follow, and nothing is recomputed.
    range_guard_async = os.environ.get("SN_RANGE_GUARD", "1") == "async"
    fused_cab_tail = True      # bf16 engine: always.  Engine32 (one kernel per reference module) runs conv, conv, pool, MLP, scale + residual instead

    def cab(self, pre: str, x: Act, extra: Optional[Act] = None) -> Act:
        """CAB: 3x3 -> PReLU -> 3x3 -> CALayer -> +x (gshift_deblur1.py:141-156).

        The CALayer scale is known BEFORE the second conv (its pooled input is linear in `mid`, sn_cab_ca), so scale and residual
        (and the optional second residual `extra`) are applied in conv2's epilogue: two sn_conv2d launches, 5 tensor passes.
        Large 16-channel CABs run the streaming fused form instead (3 passes, `mid` in LDS, bit-identical: `cab_fused` below)."""
        if self.fused_cab_tail:
            p = self.P.cas[pre + "CA"]
            T, h, w, cs = x.dims
            slope = self.P.scalar(pre + "body.1.weight")
            if self.cab_fused in ("8", "16", "s8", "s16") or (self.cab_fused[0] == "p" and extra is None and not self.conv_tiles and
                                                              (self.cab_fused == "p" or (cs == 16 and T * h * w >= self.CAB_FUSED_MIN_PX))):
                out = self._cab_fused(pre, x, extra, slope)
                if out is not None:
                    return out
            scratch = torch.empty((self.lib.sn_cab_ca_scratch_floats(T),), dtype=torch.float32, device=self.dev)
            mid, pool, _ = self.conv(pre + "body.0", [x], prelu=slope, pool=True)
            _, nblk, cpad = pool.shape
            ca = torch.empty((T, cpad), dtype=torch.float32, device=self.dev)
            self._call("sn_cab_ca", f"sn_cab_ca[{pre}]", pool.data_ptr(), nblk, cpad, mid.t.data_ptr(), cs, p["c"], p["cr"], h, w,
                       p["w2"].data_ptr(), p["wa"].data_ptr(), p["wb"].data_ptr(), scratch.data_ptr(), ca.data_ptr(), T, self._stream())
            return self.conv(pre + "body.2", [mid], res=x, oscale=ca, res2=extra)
        # one kernel per reference module (Engine32 provides scale_residual; the bf16 engine never takes this branch)
        r = self.conv(pre + "body.0", [x], prelu=self.P.scalar(pre + "body.1.weight"))
        r, pool, npix = self.conv(pre + "body.2", [r], pool=True)
        return self.scale_residual(r, x, self.ca_mlp(pre + "CA", pool, npix), extra)

    # Fused dense CAB: statistics pass + one kernel with `mid` in LDS, three tensor passes instead of five, bit-identical to the two-launch form.
    # SN_CAB_FUSED: "0" two sn_conv2d launches; "p" the streaming form (csrc/sn_conv3p.hip: cabp_kernel; 16 / 24 channels) wherever it exists;
    # "p16" (default) the streaming form for 16-channel CABs of at least CAB_FUSED_MIN_PX pixel-frames -- the one case it wins: 504 against 606 us
    # per CAB at 20 x 720 x 1280 alone, 496 against 534 inside a config-2 window (-0.75 ms of 101); 24 channels lose (one workgroup per CU);
    # "8" / "16" / "s8" / "s16": the one-workgroup-per-tile form of csrc/sn_cabf.hip with that many tile rows ("s": statistics on the streaming conv).
    cab_fused = os.environ.get("SN_CAB_FUSED", "p16")
    CAB_FUSED_MIN_PX = 4 << 20

    def _conv_desc(self, name: str, x: Act, *, prelu: Optional[float] = None) -> L.ConvDesc:
        """sn_conv_desc of a single-input 3x3 stride-1 conv of a CAB on x (no output, residual or pool operands yet)."""
        p = self.P.convs[name]
        T, h, w, cs = x.dims
        assert cs == p["cs_in"] and p["n_in"] == 1 and int(p["k"]) == 3
        d = L.ConvDesc()
        d.inp[0] = x.t.data_ptr()
        d.n_in, d.cs_in, d.T, d.h_in, d.w_in, d.in_mode = 1, cs, T, h, w, 0
        d.k, d.stride, d.pad, d.h_out, d.w_out = 3, 1, 1, h, w
        d.wfrag, d.mt, d.ks = p["wfrag"].data_ptr(), int(p["mt"]), int(p["ks"])
        d.bias = p["bias"].data_ptr() if p["bias"] is not None else None
        d.act, d.prelu = (1, prelu) if prelu is not None else (0, 0.0)
        d.cs_out, d.c_out, d.out_mode = prep.ceil8(int(p["cout"])), int(p["cout"]), 0
        d.flags = L.SN_CONV_TILE_KERNEL          # the statistics pass and the fused kernel are tile kernels: their pool rows are per tile
        return d

    def _cab_fused(self, pre: str, x: Act, extra: Optional[Act], slope: float) -> Optional[Act]:
        """CAB as sn_cab_stats -> sn_cab_ca_lines -> sn_cab_fused; None when the library has no fused instance for this width."""
        lib = self.lib
        T, h, w, cs = x.dims
        if h < 2 or w < 2:
            return None
        d1 = self._conv_desc(pre + "body.0", x, prelu=slope)
        d2 = self._conv_desc(pre + "body.2", x)
        if self.cab_fused[0] in "sp" and not self.conv_tiles:
            # statistics pass on the streaming kernel (its own pool rows: sn_conv_pool_blocks); the streaming fused kernel reads its plan from conv1's flags
            d1.flags = (self.conv_wgs << 4) | (self.conv_depth << 10) | (0 if self.cab_fused[0] == "p" else self.conv_dbg << 12)
            if self.cab_fused[0] == "p":
                d2.flags = (self.conv_dbg & 1) << 12                  # measurements: conv2's weights in LDS, not registers (16 channels)
        if d1.cs_out != cs or d2.cs_out != cs or not lib.sn_cab_fused_supported(C.byref(d1), C.byref(d2)):
            return None
        p = self.P.cas[pre + "CA"]
        ll = max(h, w)
        lines = torch.empty((T, 4, ll, cs), dtype=self.act_dtype, device=self.dev)
        nblk = lib.sn_conv_pool_blocks(C.byref(d1))
        pool = torch.empty((T, nblk, 16 * d1.mt), dtype=torch.float32, device=self.dev)
        scratch = torch.empty((lib.sn_cab_ca_scratch_floats(T),), dtype=torch.float32, device=self.dev)
        ca = torch.empty((T, 16 * d1.mt), dtype=torch.float32, device=self.dev)
        out = self._new(T, h, w, cs)
        d1.out, d1.pool = lines.data_ptr(), pool.data_ptr()
        d2.out, d2.res, d2.oscale, d2.oscale_stride = out.data_ptr(), x.t.data_ptr(), ca.data_ptr(), ca.shape[1]
        if extra is not None:
            assert extra.dims == x.dims
            d2.res2 = extra.t.data_ptr()
        rows = 0 if self.cab_fused[0] == "p" else 16 if self.cab_fused.endswith("16") else 8      # 0: the streaming form (csrc/sn_conv3p.hip: cabp_kernel)
        st = self._stream()
        self._meta = ("cabf", T, h, w, cs, 1)                   # statistics pass: reads x
        self._call("sn_cab_stats", f"sn_cab_stats[{pre}]", C.byref(d1), ll, st)
        self._meta = ()
        self._call("sn_cab_ca_lines", f"sn_cab_ca[{pre}]", pool.data_ptr(), nblk, 16 * d1.mt, lines.data_ptr(), ll, cs, p["c"], p["cr"], h, w,
                   p["w2"].data_ptr(), p["wa"].data_ptr(), p["wb"].data_ptr(), scratch.data_ptr(), ca.data_ptr(), T, st)
        self._meta = ("cabf", T, h, w, cs, 3 if extra is not None else 2)      # fused pass: reads x (+ extra), writes out
        self._call("sn_cab_fused", f"sn_cab_fused[{pre}]", C.byref(d1), C.byref(d2), rows, st)
        return Act(out, x.c)

    def _wrap_flag(self, mode: int, circular: bool) -> int:
        """sn_unit_src.wrap: 0 keep the window's boundary frame, 1 circular, 2 the neighbour frame's half arrives from the adjacent rank."""
        if self.split is not None and mode:
            return self.split.wrap_flag(mode, circular)
        return 1 if circular else 0

    def _unit_src(self, x: Act, mode: int, *, wrap: Optional[int] = None, halo: Optional[torch.Tensor] = None, t0: int = 0, nt: int = 0) -> L.UnitSrc:
        T, h, w, cs = x.dims
        assert cs == x.c
        wrap = self._wrap_flag(mode, self.V.wrap) if wrap is None else wrap
        return L.UnitSrc(x.t.data_ptr(), T, h, w, x.c, mode, wrap, halo.data_ptr() if halo is not None else None, t0, nt)

    def _split_pieces(self, x: Act, mode: int, circular: bool):
        """Launch plan of a shifted operator on a temporally split window (temporal_split.py): [(wrap, halo, t0, nt), ...].

        Without a split (or where this rank has no neighbour on the borrowing side): one piece, all frames.  Otherwise the frames that
        need no halo are launched FIRST, on the compute stream; the halo exchange runs meanwhile on the engine's side stream (it waits
        only for the producer of x), and the boundary frame -- the only one that reads the neighbour's half -- follows once it arrived."""
        T = x.dims[0]
        flag = self._wrap_flag(mode, circular)
        if self.split is None:
            yield (flag, None, 0, 0)
            return
        main = torch.cuda.current_stream(self.dev)
        ready = main.record_event()                       # x is complete here (stream order)
        recv = flag == 2                                  # this rank's boundary frame borrows from a neighbour rank
        bt = 0 if mode == 1 else T - 1
        if recv and T > 1:
            yield (0, None, 1 if mode == 1 else 0, T - 1)  # the boundary frame is outside the range, so its rule does not matter
        if self._side is None:
            self._side = torch.cuda.Stream(self.dev)
        with torch.cuda.stream(self._side):               # every rank takes part: one that receives nothing may still have to send
            self._side.wait_event(ready)
            x.t.record_stream(self._side)
            halo = self.split.exchange(x.t, mode, circular)
        if recv:
            main.wait_stream(self._side)
            halo.record_stream(main)
            yield (2, halo, bt, 1)
        else:
            yield (flag, None, 0, 0)

    def temporal_roll(self, x: Act, reverse: bool) -> Act:
        T, h, w, cs = x.dims
        assert cs == x.c, "Shift_CAB widths (24, 80) are stored unpadded"
        y = self._new(T, h, w, cs)
        mode = 2 if reverse else 1
        self._meta = ("roll", T, h, w, cs)              # (not a kernel of a GSTS unit: bench.py's unit figure must not inherit the previous launch's record)
        for wrap, halo, t0, nt in self._split_pieces(x, mode, False):
            s = self._unit_src(x, mode, wrap=wrap, halo=halo, t0=t0, nt=nt)
            self._call("sn_temporal_roll", "sn_temporal_roll", C.byref(s), y.data_ptr(), self._stream())
        return Act(y, x.c)

    def shift_cab(self, pre: str, x: Act, reverse: bool) -> Act:
        """Shift_CAB (gshift_denoise1.py:157-186)."""
        return self.cab(pre, self.temporal_roll(x, reverse))

    def _guard_ptr(self) -> Optional[int]:
        if not self.range_guard:
            return None
        if self._bad is None:
            self._bad = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        return self._bad.data_ptr()

    def _fused_phase1(self, T: int) -> bool:
        """Does phase 1 of this engine's CAB2 / CAB1 run as the fused kernel?  (The denoisers' two passes rely on the squeeze-excite tail for
        the inner scale, whose frame counters cover MAX_TICKETS frames.)"""
        if self.phase1 == "0":
            return False
        return not (self.V.denoise and (not self.fold_se or T > self.MAX_TICKETS))

    def naf(self, pre: str, x: Act, mode: int, *, frames: Optional[Tuple[int, int]] = None, bufs: Optional[Dict[str, torch.Tensor]] = None) -> Act:
        """CAB2 (mode 1/2, fed by the GSTS gather of x) or CAB1 (mode 0) (gshift_deblur1.py:183-255).

        Phase 1: [K0 sn_gsts_shiftconv (CAB2 only)] -> sn_gsts_cab2_phase1 / sn_cab1_phase1 (ONE kernel up to g2; the denoisers run it twice, the first
        pass for the channel sums of g1 behind their inner CALayer2), or -- phase1 "0" -- K12 sn_ln_gemm_gate -> K3 (sn_dw5m_gemm_gate depthwise,
        sn_grp5_gemm_gate for the grouped "+" RepConv) with g1 in bf16 through HBM; then the squeeze-excite MLP (folded into phase 1's last
        workgroup per frame, or sn_ca_mlp) and phase 2, K4 sn_gsts_cab2_phase2 / sn_cab1_phase2.  The global average pool of CALayer2 sits
        between the phases and forbids a single pass.  Every frame is independent inside a CAB (the pool is per frame), so
        on a temporally split window the chain runs in two pieces: all frames but the boundary one while the halo exchange is in flight, then
        the boundary frame.  frames = (t0, nt) / bufs: one frame range of a block whose tensors the caller owns (the frame-wavefront schedule)."""
        lib, V, P = self.lib, self.V, self.P
        u = P.units[pre]
        T, h, w, c = x.dims
        self._meta = ("naf", frames[1] if frames is not None else T, h, w, c, mode, T)      # (frames of this launch group, ..., frames of the tensor)
        fused = self._fused_phase1(T)                    # phase 1 in ONE kernel: neither a, g1 nor r leave the CU
        mstencil = not V.grouped_rep                     # chain: depthwise RepConv (C = 64) as a Toeplitz-MFMA 5x5 on a channel-planar g1
        B = bufs if bufs is not None else self.naf_buffers(T, h, w, c, mode)
        hwb, g2, y, pool2, ca2, ca1, g1, pool1, g1s = (B.get(k) for k in ("hwb", "g2", "y", "pool2", "ca2", "ca1", "g1", "pool1", "g1s"))
        nb2 = pool2.shape[1]
        if self._tickets is None and fused and self.fold_se:
            self._tickets = torch.zeros((self.MAX_TICKETS,), dtype=torch.int32, device=self.dev)
        b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
        es = 2                                           # bytes per activation element
        bad = self._guard_ptr()

        def ca_mlp(name: str, pool: torch.Tensor, ca: torch.Tensor, f0: int, n: int) -> None:
            q = P.cas[name]
            nblk = pool.shape[1]
            self._call("sn_ca_mlp", f"sn_ca_mlp[{name}]", pool.data_ptr() + f0 * nblk * c * 4, nblk, c, q["c"], q["cr"], 1.0 / (h * w),
                       q["wa"].data_ptr(), q["wb"].data_ptr(), ca.data_ptr() + f0 * c * 4, n, bad, self._stream())

        if frames is not None:
            assert self.split is None
            pieces = [(self._wrap_flag(mode, V.wrap) if mode else 0, None, frames[0], frames[1])]
        else:
            pieces = self._split_pieces(x, mode, V.wrap) if mode else [(0, None, 0, 0)]
        for wrap, halo, t0, nt in pieces:
            st = self._stream()
            src = self._unit_src(x, mode, wrap=wrap, halo=halo, t0=t0, nt=nt)
            f0, n = (t0, nt) if nt else (0, T)           # frame range of this piece for the operators that take plain pointers
            if mode:
                if self.k0_mfma and P.k0_mfma_ok:
                    self._call("sn_gsts_shiftconv_mfma", "sn_gsts_shiftconv", C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st)
                else:
                    self._call("sn_gsts_shiftconv", "sn_gsts_shiftconv", C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st)
            hw_ptr = hwb.data_ptr() if mode else None
            folded = False
            if fused:
                wt = C.byref(u["p1r"]["desc"])
                sep = None
                if self.fold_se and T <= self.MAX_TICKETS:
                    q = P.cas[pre + "ca2"]
                    se = L.SeFold(q["wa"].data_ptr(), q["wb"].data_ptr(), q["c"], q["cr"], self._tickets.data_ptr(), ca2.data_ptr(), bad)
                    sep = C.byref(se)
                opt = None
                if V.denoise:      # inner CALayer2 on g1 (gshift_denoise1.py:224,257): pass 1 = the channel sums of g1 and, by the tail, its scale ca1
                    q1 = P.cas[pre + "ca1"]
                    se1 = L.SeFold(q1["wa"].data_ptr(), q1["wb"].data_ptr(), q1["c"], q1["cr"], self._tickets.data_ptr(), ca1.data_ptr(), bad)
                    g1s_ptr = g1s.data_ptr() if (g1s is not None and self.g1_store) else None
                    o1 = L.Phase1Opts(None, 1, 0, g1s_ptr)
                    fn1 = "sn_gsts_cab2_phase1" if mode else "sn_cab1_phase1"
                    a1 = (C.byref(src), hw_ptr, wt, None, pool2.data_ptr(), C.byref(se1), C.byref(o1), st) if mode else \
                         (C.byref(src), wt, None, pool2.data_ptr(), C.byref(se1), C.byref(o1), st)
                    self._call(fn1, fn1 + "[g1 sums]", *a1)
                    opt = C.byref(L.Phase1Opts(ca1.data_ptr(), 0, 0, g1s_ptr))
                if mode:
                    self._call("sn_gsts_cab2_phase1", "sn_gsts_cab2_phase1", C.byref(src), hw_ptr, wt, g2.data_ptr(), pool2.data_ptr(), sep, opt, st)
                else:
                    self._call("sn_cab1_phase1", "sn_cab1_phase1", C.byref(src), wt, g2.data_ptr(), pool2.data_ptr(), sep, opt, st)
                folded = sep is not None
            else:
                self._call("sn_ln_gemm_gate", "sn_ln_gemm_gate", C.byref(src), hw_ptr, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(),
                           u["w_dw3_h2"].data_ptr(), g1.data_ptr(), pool1.data_ptr() if pool1 is not None else None, 2 if mstencil else 0, st)
                ca1_ptr = None
                if V.denoise:
                    ca_mlp(pre + "ca1", pool1, ca1, f0, n)
                    ca1_ptr = ca1.data_ptr() + f0 * c * 4
                fr1 = g1.stride(0) * es * f0
                fr2 = g2.stride(0) * es * f0
                k3 = "sn_dw5m_gemm_gate" if mstencil else "sn_grp5_gemm_gate"
                self._call(k3, k3, g1.data_ptr() + fr1, ca1_ptr, u["w_toep5" if mstencil else "w_grp"].data_ptr(), u["w_gate"].data_ptr(),
                           g2.data_ptr() + fr2, pool2.data_ptr() + f0 * nb2 * c * 4, n, h, w, c, st)
            if not folded:
                ca_mlp(pre + "ca2", pool2, ca2, f0, n)
            k4 = "sn_gsts_cab2_phase2" if mode else "sn_cab1_phase2"
            self._call(k4, k4, C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out, y.data_ptr(), st)
        return Act(y, c)

    def naf_buffers(self, T: int, h: int, w: int, c: int, mode: int) -> Dict[str, torch.Tensor]:
        """The tensors one CAB2 / CAB1 writes: hwb (conv1 of the shifted half, CAB2), g2, y, the partial channel sums and the squeeze-excite
        scales; chain only: g1 (bf16, channel-planar for the depthwise RepConv) and the inner pool of the denoisers.  All of them stay
        referenced by the caller until the block's launches are issued: a temporary would go back to the caching allocator at once and a
        tensor K3 WRITES could be carved out of the block K3 still READS its scale from (intermittent wrong frames at the small pyramid
        levels; found by the full-size determinism check of config 4)."""
        lib, V = self.lib, self.V
        fused = self._fused_phase1(T)
        mstencil = not V.grouped_rep
        B: Dict[str, torch.Tensor] = {}
        if mode:
            B["hwb"] = self._new(T, h, w, c // 2)
        B["g2"] = self._new(T, h, w, c)
        B["y"] = self._new(T, h, w, c)
        if fused:
            nb2 = lib.sn_phase1_pool_blocks(T, h, w)
            if nb2 < 1:
                raise L.ShiftNetLibError(f"sn_phase1_pool_blocks failed with code {nb2}")
            if V.denoise and self.g1_store:
                nbytes = C.c_longlong(0)
                L.check(lib.sn_phase1_g1_store_bytes(T, h, w, c, C.byref(nbytes)), "sn_phase1_g1_store_bytes")
                B["g1s"] = torch.empty((nbytes.value,), dtype=torch.uint8, device=self.dev)
        else:
            B["g1"] = (torch.empty((T, h, c, lib.sn_planar_pitch(w)), dtype=torch.bfloat16, device=self.dev) if mstencil else self._new(T, h, w, c))
            nb2 = lib.sn_dw5m_blocks(h, w) if mstencil else lib.sn_grp5_blocks(h, w)
            if V.denoise:
                B["pool1"] = torch.empty((T, lib.sn_lngate_blocks(h, w), c), dtype=torch.float32, device=self.dev)
        if V.denoise:
            B["ca1"] = torch.empty((T, c), dtype=torch.float32, device=self.dev)
        B["pool2"] = torch.empty((T, nb2, c), dtype=torch.float32, device=self.dev)
        B["ca2"] = torch.empty((T, c), dtype=torch.float32, device=self.dev)
        return B

    def gsts_unit(self, pre: str, x: Act, reverse: bool) -> Act:
        return self.naf(pre + "1.", self.naf(pre + "0.", x, 2 if reverse else 1), 0)

    def shift_block(self, pre: str, x: Act) -> Act:
        """Encoder_shift_block.forward (gshift_deblur1.py:530-547 / gshift_deblur2.py:521-530)."""
        for i in range(self.V.units):
            x = self.gsts_unit(f"{pre}{UNIT_NAMES[i]}.", x, reverse=(i % 2 == 1))
        return x

    # layer-frame wavefront.  "unit" (default): every unit runs over all T frames before the next one starts (the reference's
    # order, gshift_deblur1.py:530-547).  "frame": the units of consecutive Encoder_shift_blocks of one pyramid level are issued per FRAME GROUP in
    # dependency order -- unit k needs frames {t, t -/+ 1} of unit k - 1 (forward / reverse shift, :504-518), so a group of unit k can start as
    # soon as its own and one neighbouring group of unit k - 1 are done and a frame's working set moves through many units while it is still in
    # the Infinity Cache.  Results are bit-identical (every kernel honours frame ranges and the pool rows do not depend on them).  Measured:
    # DESIGN.md section 3.3 -- no gain while phase 1 is not bandwidth-bound; SN_SCHEDULE=frame / bench.py --schedule frame keep it one flag away.
    # "streams" (round 6): the window's frames are cut into stream_groups contiguous frame groups and every group runs ITS chain of launches
    # (K0 -> phase 1 -> K4 -> phase 1 -> K4, unit after unit) on its own HIP stream; the only cross-stream edges are the one boundary frame a
    # shifted unit borrows from the neighbouring group (events).  The groups drift out of phase by themselves -- two phase-1 launches cannot
    # share a CU (131 / 149 KB of LDS each), so the second one waits and from then on its memory-bound K4 runs UNDER the other group's
    # issue-bound phase 1 (K4 has no LDS and 76 VGPRs: it fits beside phase 1's 12 waves).  Bit-identical to unit-major by construction.
    schedule = os.environ.get("SN_SCHEDULE", "unit")
    frame_group = int(os.environ.get("SN_FRAME_GROUP", "4"))
    stream_groups = int(os.environ.get("SN_STREAM_GROUPS", "2"))
    STREAMS_MIN_PXF = 400_000     # frames x pixels of a GROUP below which the streams schedule is not used (launches too small to overlap usefully)

    def shift_chain(self, pres: Sequence[str], x: Act) -> Act:
        """Consecutive Encoder_shift_blocks at one level (Encoder2.forward, gshift_deblur1.py:623-637 / gshift_deblur2.py:594-609)."""
        T = x.dims[0]
        if self.schedule not in ("unit", "frame", "streams") or self.frame_group < 1 or self.stream_groups < 1:
            raise ValueError(f"SN_SCHEDULE={self.schedule!r} / SN_FRAME_GROUP={self.frame_group} / SN_STREAM_GROUPS={self.stream_groups}: "
                             "expected unit, frame or streams, and groups of >= 1 frames")
        ev = None
        if self.prof is not None:             # wall time of the whole chain on the launching stream (bench.py: GSTS time of the window under any schedule)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        y = self._shift_chain(pres, x)
        if ev is not None:
            ev[1].record()
            self.prof.append(("gsts_chain", "gsts_chain", ("chain",) + tuple(x.dims) + (len(pres) * self.V.units,), ev[0], ev[1]))
        return y

    def _shift_chain(self, pres: Sequence[str], x: Act) -> Act:
        T = x.dims[0]
        if self.schedule == "streams" and self.split is None and not torch.cuda.is_current_stream_capturing():
            ng = min(self.stream_groups, T // 2)
            if ng >= 2 and (T // ng) * x.dims[1] * x.dims[2] >= self.STREAMS_MIN_PXF:
                return self._shift_chain_streams(pres, x, ng)
        if self.schedule != "frame" or self.split is not None or T <= self.frame_group:
            for pre in pres:
                x = self.shift_block(pre, x)
            return x
        _, h, w, c = x.dims
        units = [(f"{pre}{UNIT_NAMES[i]}.", i % 2 == 1) for pre in pres for i in range(self.V.units)]
        G = self.frame_group
        groups = [(t0, min(G, T - t0)) for t0 in range(0, T, G)]
        # every unit owns its tensors for the whole chain (no reuse games: 12 units x 3 GB at level 1 of config 2, out of 288 GB)
        bufs = [(self.naf_buffers(T, h, w, c, 2 if rev else 1), self.naf_buffers(T, h, w, c, 0)) for _, rev in units]
        ins = [x] + [Act(b1["y"], c) for _, b1 in bufs]           # ins[u]: input of unit u; ins[u + 1]: its output
        for u, j in wavefront_order([rev for _, rev in units], T, G, self.V.wrap):
            pre, rev = units[u]
            y2 = self.naf(pre + "0.", ins[u], 2 if rev else 1, frames=groups[j], bufs=bufs[u][0])
            self.naf(pre + "1.", y2, 0, frames=groups[j], bufs=bufs[u][1])
        return ins[-1]

    STREAM_RING = 3            # buffer sets of the streams schedule: unit u writes set u % STREAM_RING

    def _shift_chain_streams(self, pres: Sequence[str], x: Act, ng: int) -> Act:
        """The chain with one HIP stream per frame group (see `schedule`).  Group g's stream runs, unit after unit, CAB2 then CAB1 on ITS frames
        (frame-range launches: every GSTS kernel honours sn_unit_src.t0 / nt and their results do not depend on the range).  stream_plan gives
        the cross-stream edges: before CAB2 of unit u a group waits for the group that owns the ONE frame its boundary frame borrows from to
        have finished unit u - 1 (gshift_deblur1.py:504-518), and before it overwrites the ring slot of unit u - STREAM_RING for the groups
        that borrowed from it there."""
        T, h, w, c = x.dims
        units = [(f"{pre}{UNIT_NAMES[i]}.", i % 2 == 1) for pre in pres for i in range(self.V.units)]
        R = min(self.STREAM_RING, max(len(units), 2))
        groups, plan = stream_plan([rev for _, rev in units], T, ng, self.V.wrap, R)
        if getattr(self, "_gstreams", None) is None or len(self._gstreams) < ng:
            self._gstreams = [torch.cuda.Stream(self.dev) for _ in range(ng)]
        streams = self._gstreams[:ng]
        main = torch.cuda.current_stream(self.dev)
        sets = [(self.naf_buffers(T, h, w, c, 1), self.naf_buffers(T, h, w, c, 0)) for _ in range(R)]
        if self._tickets is None and self._fused_phase1(T) and self.fold_se:      # allocated on the launching stream, not on a group's
            self._tickets = torch.zeros((self.MAX_TICKETS,), dtype=torch.int32, device=self.dev)
        self._guard_ptr()
        ready = main.record_event()
        done: List[List[Optional[torch.cuda.Event]]] = [[None] * ng for _ in units]      # unit u's output frames of group g are written
        cab2: List[List[Optional[torch.cuda.Event]]] = [[None] * ng for _ in units]      # group g no longer reads unit u's input
        cur = x
        for u, (pre, rev) in enumerate(units):
            b2, b1 = sets[u % R]
            for g, fr in enumerate(groups):
                raw, war = plan[u][g]
                with torch.cuda.stream(streams[g]):
                    s = streams[g]
                    if u == 0:
                        s.wait_event(ready)
                    for uu, gg in raw:
                        s.wait_event(done[uu][gg])
                    for uu, gg in war:
                        s.wait_event(cab2[uu][gg])
                    y2 = self.naf(pre + "0.", cur, 2 if rev else 1, frames=fr, bufs=b2)
                    cab2[u][g] = s.record_event()
                    self.naf(pre + "1.", y2, 0, frames=fr, bufs=b1)
                    done[u][g] = s.record_event()
            cur = Act(b1["y"], c)
        for g in range(ng):
            main.wait_event(done[-1][g])
        return cur

    def down(self, pre: str, x: Act) -> Act:
        """DownSample (gshift_deblur1.py:330-340 / gshift_denoise1.py:356-365)."""
        if self.V.denoise:
            return self.conv(pre + "down", [x], stride=2, prelu=self.P.scalar(pre + "down.1.weight"))
        return self.conv(pre + "down", [x], stride=2)

    # SkipUpSample: "1" (default) the 1x1 at LOW resolution, then one streaming pass bilinear x2 + skip (sn_upsample2_add: conv and interpolation
    # commute); "0" round 2's form, the interpolation in the loader of a full-resolution conv (1.3-1.8 TB/s of its bytes)
    skip_up_lowres = os.environ.get("SN_SKIPUP_LOWRES", "1") != "0"

    def skip_up(self, name: str, x: Act, y: Act) -> Act:
        """SkipUpSample: bilinear x2 -> 1x1 -> + y (gshift_deblur1.py:341-350)."""
        if not self.skip_up_lowres:
            return self.conv(name, [x], in_mode=1, res=y)
        lo = self.conv(name, [x])                             # [T][hs][ws][cs_out]
        T, hs, ws, cs = lo.dims
        assert y.dims == (T, 2 * hs, 2 * ws, cs), (y.dims, lo.dims)
        out = self._new(T, 2 * hs, 2 * ws, cs)
        self._meta = ("up2add", T, hs, ws, cs)
        self._call("sn_upsample2_add", f"sn_upsample2_add[{name}]", lo.t.data_ptr(), y.t.data_ptr(), out.data_ptr(), T, hs, ws, cs, self._stream())
        return Act(out, lo.c)

    def tfr_unet(self, pre: str, x: Act, extra: Optional[Act] = None) -> Act:
        """TFR_UNet.forward (gshift_deblur1.py:709-722).  `extra` is added to the result in the epilogue of the last CAB's
        second conv: the "+ shortcut" that follows the last orb / rorb (:769,779) costs no pass of its own."""
        def seq(nm: str, n: int, t: Act, ex: Optional[Act] = None) -> Act:
            for i in range(n):
                t = self.cab(f"{pre}{nm}.{i}.", t, ex if i == n - 1 else None)
            return t
        enc1 = seq("encoder_level1", 1, x)
        enc2 = seq("encoder_level2", 3, self.down(pre + "down12.", enc1))
        enc3 = seq("encoder_level3", 3, self.down(pre + "down23.", enc2))
        dec3 = seq("decoder_level3", 3, enc3)
        t = self.skip_up(pre + "up32", dec3, self.cab(pre + "skip_attn2.", enc2))
        dec2 = seq("decoder_level2", 3, t)
        t = self.skip_up(pre + "up21", dec2, self.cab(pre + "skip_attn1.", enc1))
        return seq("decoder_level1", 1, t, extra)

    def stage1(self, x: Act) -> Act:
        """Encoder2.forward (gshift_deblur1.py:613-642, gshift_deblur2.py:587-613, gshift_denoise1.py:640-670)."""
        V, p = self.V, "stage1."
        x = self.cab(p + "concat.", x)
        shortcut = x
        if V.shift_cab:
            x = self.shift_cab(p + "encoder_level0.", x, False)
            x = self.shift_cab(p + "encoder_level0_1.", x, True)
        x = self.conv(p + "down01", [x], stride=2, pad=0, prelu=self.P.scalar(p + "down01.1.weight"))
        if V.topo == "small":
            enc11 = self.shift_chain([p + "encoder_level1.", p + "encoder_level1_1.", p + "encoder_level1_2."], x)
            e = self.down(p + "down12.", enc11)
            e = self.shift_chain([f"{p}{n}." for n in ("encoder_level2", "encoder_level2_1", "encoder_level2_2",
                                                       "decoder_level2", "decoder_level2_1", "decoder_level2_2")], e)
            x = self.skip_up(p + "up21", e, self.cab(p + "skip_attn1.", enc11))
        else:
            if V.shift_cab:
                e = self.shift_cab(p + "encoder_level1.", x, False)
                enc11 = self.shift_cab(p + "encoder_level1_1.", e, True)
            else:
                enc11 = self.cab(p + "encoder_level1_1.", self.cab(p + "encoder_level1.", x))
            e = self.down(p + "down12.", enc11)
            enc22 = self.cab(p + "encoder_level2_1.", self.cab(p + "encoder_level2.", e))
            e = self.down(p + "down23.", enc22)
            enc33 = self.cab(p + "encoder_level3_1.", self.cab(p + "encoder_level3.", e))
            d = self.shift_chain([p + "decoder_level3.", p + "decoder_level3_1."], enc33)
            x = self.skip_up(p + "up32", d, self.cab(p + "skip_attn2.", enc22))
            d = self.shift_chain([p + "decoder_level2.", p + "decoder_level2_1."], x)
            x = self.skip_up(p + "up21", d, self.cab(p + "skip_attn1.", enc11))
        dec11 = self.shift_chain([p + "decoder_level1.", p + "decoder_level1_1.", p + "decoder_level1_2."], x)
        skip = self.cab(p + "skip_conv.", shortcut)
        if V.hr_cat:
            up = self.conv(p + "upsample0", [dec11], out_mode=1)
            out = self.conv(p + "conv_hr0", [up, skip])
        else:   # conv_hr0(act(upsample0(x))) + skip: the PReLU commutes with pixel_shuffle, so it is the conv epilogue
            up = self.conv(p + "upsample0", [dec11], out_mode=1, prelu=self.P.scalar(p + "act.weight"))
            out = self.conv(p + "conv_hr0", [up], res=skip)
        return self.cab(p + "out_conv.", out)

    def _ingest(self, x: torch.Tensor, noise_map: Optional[torch.Tensor]) -> Act:
        """x = x[0]; cat((x, noise_map), 1) into the kernel layout (gshift_deblur1.py:784-787, gshift_denoise1.py:828-831)."""
        T, cin, H, W = x.shape
        nm_ptr = None
        if self.V.denoise:
            noise_map = noise_map.to(x.dtype).expand(T, 1, H, W).contiguous()
            nm_ptr = noise_map.data_ptr()
        x8 = self._new(T, H, W, 8)
        self._call("sn_ingest", "sn_ingest", x.data_ptr(), _dtype_code(x.dtype), nm_ptr, x8.data_ptr(), T, cin, H, W, self._stream())
        return Act(x8, self.V.in_ch)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, noise_map: Optional[torch.Tensor], past: int, future: int,
                out_dtype: Optional[torch.dtype] = None, shortcut: Optional[torch.Tensor] = None) -> torch.Tensor:
        """GShiftNet.forward; x:[T,C,H,W] (already x[0]) on this engine's device, any of fp32/fp16/bf16.  out_dtype (default: x.dtype, the
        module contract) = torch.float32 returns the restored frames from conv_last's fp32 accumulators without the rounding to a
        half-precision image tensor: the CLIs' metrics / PNG path (inference/test_deblur.py:137-143 converts with .float() anyway).
        shortcut: optional [T,3,H,W] tensor added instead of x in "return output_features + shortcut[...]" (gshift_deblur1.py:791): the
        un-rounded float32 frames when x had to be rounded to a half-precision module dtype."""
        with torch.cuda.device(self.dev):      # launches, events and allocations all belong to the engine's device
            graphed = self.use_graph or (self.graph_auto and x.shape[0] * x.shape[2] * x.shape[3] <= self.GRAPH_AUTO_PXF)
            eager = (not graphed or self.split is not None or self.prof is not None or out_dtype is not None or shortcut is not None
                     or torch.cuda.is_current_stream_capturing())
            if self.range_guard_async:
                self.check_range_guard()               # the PREVIOUS window's flag (its copy has long landed: no wait in the steady state)
            out = self._forward(x, noise_map, past, future, out_dtype, shortcut) if eager else self._forward_graphed(x, noise_map, past, future)
            if self._scope is not None:                # guard_scope(): one check when the scope closes
                self._scope.frames = max(self._scope.frames, int(x.shape[0]))
            elif self.range_guard_async:
                self._post_guard_copy()
            elif self._guard_tripped():
                out = self._recover(lambda: self._forward(x, noise_map, past, future, out_dtype, shortcut), out, x.shape[0])
            return out

    _scope = None

    class GuardScope:
        """Result of Engine.guard_scope(): `tripped` is set when the scope closes."""
        tripped = False
        frames = 1

    @contextlib.contextmanager
    def guard_scope(self):
        """Several forwards whose results are consumed TOGETHER (the denoise CLIs' four quadrants of a window, inference/test_denoise.py:153-173) under
        ONE range-guard check: inside the scope no forward reads the flag -- no device sync between them, the launches of the next one queue up
        behind the previous one's kernels -- and the scope's exit reads it once.  If it tripped, the engine has moved to the two-kernel chain and
        `scope.tripped` tells the caller to run the scope's forwards again (cli.quadrant_forward does).  Measured on config 4 (bf16, 4 quadrants x ~2900
        launches per window): 111-114 frames/s with a check per forward (round 5), 118 without any (round 4's library) -- VERDICT r05 item 3."""
        sc = Engine.GuardScope()
        outer, self._scope = self._scope, sc
        try:
            yield sc
        finally:
            self._scope = outer
            if outer is None and not self.range_guard_async:
                sc.tripped = self._guard_tripped()
                if sc.tripped:
                    self._recover(None, None, sc.frames)
            elif outer is not None:
                outer.frames = max(outer.frames, sc.frames)
            else:
                self._post_guard_copy()

    def _post_guard_copy(self) -> None:
        """async guard: flag -> pinned host word behind this forward's launches, then re-arm on the device (stream order)."""
        if self._bad is None or torch.cuda.is_current_stream_capturing():
            return
        if getattr(self, "_bad_host", None) is None:
            self._bad_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
            self._bad_event = torch.cuda.Event()
        self._bad_host.copy_(self._bad, non_blocking=True)
        self._bad.zero_()
        self._bad_event.record()
        self._bad_pending = True

    def check_range_guard(self) -> bool:
        """async guard: has a window since the last check tripped the guard?  Waits for the pending flag copy only (a finished forward: no wait).
        A tripped flag warns and moves the engine to the two-kernel chain for the windows that follow (unless SN_PHASE1=r was asked for)."""
        if not getattr(self, "_bad_pending", False):
            return False
        self._bad_event.synchronize()
        self._bad_pending = False
        v = int(self._bad_host.item())
        if self.split is not None:
            v = self.split.any_rank(v)
        if v:
            self._recover(None, None, 1)
        return bool(v)

    def _guard_tripped(self) -> bool:
        """Reads (and re-arms) the range guard's flag: one 4-byte device-to-host copy per forward.  On a temporally split window the ranks
        agree first -- a rank that recomputed on its own would leave the others alone in the halo exchanges."""
        if self._bad is None or torch.cuda.is_current_stream_capturing():
            return False
        v = int(self._bad.item())
        if self.split is not None:
            v = self.split.any_rank(v)
        if v:
            self._bad.zero_()
        return bool(v)

    def _recover(self, rerun, out, T: int = 1):
        """A tripped range guard.  T: frames of the window that tripped it (did THAT forward run the fused kernel?  The denoisers' long windows
        already run the chain).  rerun None (async guard): report and switch only -- the window has been handed out."""
        import warnings
        if self._fused_phase1(T) and self.phase1 == "r":           # an explicit SN_PHASE1=r is honoured: warn, do not switch
            warnings.warn(f"shiftnet_amd: {self.V.name}: a channel sum of the GSTS path is not finite -- an activation left the fp16 range of the fused "
                          "phase-1 kernel (`a`, g1, r).  SN_PHASE1=r was asked for explicitly, so the module stays on it: the result of this window is "
                          "not reliable (SN_PHASE1=auto would have moved to the two-kernel chain).")
            return out
        if self._fused_phase1(T) and self.phase1 != "0":
            warnings.warn(f"shiftnet_amd: {self.V.name}: a channel sum of the GSTS path is not finite -- an activation left the fp16 range of the fused "
                          "phase-1 kernel (`a`, g1, r).  This module now runs phase 1 as the two-kernel chain with g1 in bf16 (SN_PHASE1=0)"
                          + (" and the window is recomputed." if rerun is not None else "; the results since the previous check (a guard_scope, or the "
                             "window already handed out under SN_RANGE_GUARD=async) are NOT reliable: run them again."))
            self.phase1 = "0"
            self.fallbacks += 1
            for v in self._graphs.values():
                if isinstance(v, tuple):
                    v[0].reset()
            self._graphs.clear()
            if rerun is None:
                return out
            out = rerun()
            if not self._guard_tripped():
                return out
        warnings.warn(f"shiftnet_amd: {self.V.name}: non-finite channel sums on the bf16 chain as well (its `a` is fp16): the result of this window is "
                      "not reliable -- run the module in float32 (net.float(): the fp32 engine has no half-precision intermediates).")
        return out

    def _forward_graphed(self, x, noise_map, past, future):
        key = (tuple(x.shape), x.dtype, None if noise_map is None else (tuple(noise_map.shape), noise_map.dtype), past, future)
        ent = self._graphs.get(key)
        if ent is None:                         # first sight of this signature: eager (also the warm-up the capture needs)
            self._graphs[key] = "seen"
            self._evict_graphs()
            return self._forward(x, noise_map, past, future)
        self._graphs.move_to_end(key)
        if ent == "eager":
            return self._forward(x, noise_map, past, future)
        if ent == "seen":
            try:
                sx = x.clone()
                sn = noise_map.clone() if noise_map is not None else None
                g = torch.cuda.CUDAGraph()
                # thread_local: an allocation or a sync on ANOTHER host thread (pin-memory workers, a second engine) must neither fail nor
                # invalidate this capture; __exit__ ends the capture on an exception too: the stream is never left capturing
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    so = self._forward(sx, sn, past, future)
                ent = self._graphs[key] = (g, sx, sn, so)
                self._evict_graphs()
            except Exception as e:                                  # noqa: BLE001  (capture unsupported here: stay eager, say so once)
                import warnings
                if torch.cuda.is_current_stream_capturing():        # belt and braces: a capture that __exit__ could not end
                    try:
                        g.capture_end()
                    except Exception:                               # noqa: BLE001
                        pass
                warnings.warn(f"shiftnet_amd: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                self._graphs[key] = "eager"
                torch.cuda.synchronize(self.dev)
                return self._forward(x, noise_map, past, future)
        g, sx, sn, so = ent
        sx.copy_(x)
        if sn is not None:
            sn.copy_(noise_map)
        g.replay()
        return so.clone()                       # the graph owns `so`: hand out a copy, like the fresh tensor upstream returns

    MAX_TICKETS = 4096       # frames per tensor the squeeze-excite fold has counters for (longer windows fall back to sn_ca_mlp)
    GRAPH_AUTO_PXF = 500_000   # T x H x W below which the launch stream, not the GPU, paces a forward (a 20 x 144 x 256 window: 0.74 M is GPU-bound)
    GRAPH_SLOTS = 2          # captured graphs kept per engine: each pins its whole activation pool (GBs at 720p), so a client that varies
    #                          the window shape must not accumulate them

    def _evict_graphs(self) -> None:
        live = [k for k, v in self._graphs.items() if isinstance(v, tuple)]
        while len(live) > self.GRAPH_SLOTS:
            k = live.pop(0)                     # least recently used
            g = self._graphs.pop(k)[0]
            g.reset()                           # frees the graph's private memory pool
        while len(self._graphs) > 16:           # the "seen" / "eager" markers are bounded too
            k = next(iter(self._graphs))
            v = self._graphs.pop(k)
            if isinstance(v, tuple):
                v[0].reset()

    def _forward(self, x: torch.Tensor, noise_map: Optional[torch.Tensor], past: int, future: int,
                 out_dtype: Optional[torch.dtype] = None, shortcut: Optional[torch.Tensor] = None) -> torch.Tensor:
        V = self.V
        x = x.contiguous()
        T, cin, H, W = x.shape
        div = 8 if V.topo == "plus" else 4
        if H % div or W % div:
            raise ValueError(f"{V.name}: H and W must be multiples of {div} (got {H}x{W})")
        sp = self.split
        lo = past if (sp is None or sp.rank == 0) else 0                         # a split window trims on its outer ranks only
        hi = T - (future if (sp is None or sp.rank == sp.world - 1) else 0)
        n_out = max(hi - lo, 0)
        out = torch.empty((n_out, 3, H, W), dtype=out_dtype or x.dtype, device=x.device)
        if sp is not None:
            sp.validate(n_out)              # collective: all ranks raise together, none is left waiting in an exchange
        if n_out == 0:
            return out                      # T <= past+future yields an empty tensor upstream as well
        x0 = self.cab("feat_extract.1.", self.conv("feat_extract.0", [self._ingest(x, noise_map)]))
        t = x0
        for i in range(1, V.n_orb + 1):     # deblur: res0 = orbN(..) + shortcut (gshift_deblur1.py:769), folded into the last CAB
            t = self.tfr_unet(f"orb{i}.", t, x0 if (i == V.n_orb and not V.denoise) else None)
        res0 = t
        sam = self.conv("conv_trans", [res0])
        dec = self.stage1(sam)
        feats = sam if V.denoise else res0

        def cut(a: Act) -> Act:
            return Act(a.t[lo:hi], a.c)
        y = self.conv("rconcat", [cut(x0), cut(feats), cut(dec)], prelu=self.P.scalar("lrelu.weight") if V.denoise else None)
        sc = y
        for i in range(1, V.n_orb + 1):     # deblur: "+ shortcut" after the last rorb (:779), same folding
            y = self.tfr_unet(f"rorb{i}.", y, sc if (i == V.n_orb and not V.denoise) else None)
        sc = x if shortcut is None else shortcut
        assert sc.shape[0] == T and tuple(sc.shape[2:]) == (H, W) and sc.shape[1] >= 3 and sc.device == x.device
        self.conv("conv_last", [y], out_mode=2, nchw_out=out, nchw_sc=sc[lo:hi, :3].contiguous() if sc.shape[1] > 3 else sc[lo:hi].contiguous())
        return out


def wavefront_order(revs: Sequence[bool], T: int, G: int, circular: bool) -> List[Tuple[int, int]]:
    """Launch order (unit, frame group) of the frame wavefront: sweeps over the units, every unit advancing by at most one group per sweep --
    the lowest-numbered group whose inputs exist.  Unit u needs, of unit u - 1, the group itself and the group of the ONE frame its boundary
    frame borrows from: frame t0 - 1 for a forward unit, t0 + nt for a reverse one (gshift_deblur1.py:504-518); outside [0, T) that frame wraps
    (deblur2, gshift_deblur2.py:504-505) or does not exist (kept boundary frame).  On the ring the first group of a forward unit therefore comes
    last.  Pure host logic: tests/test_host_logic.py checks every order against the dependency rule."""
    ng = -(-T // G)
    groups = [(t0, min(G, T - t0)) for t0 in range(0, T, G)]
    done = [[False] * ng for _ in revs]

    def ready(u: int, j: int) -> bool:
        if u == 0:
            return True
        t0, nt = groups[j]
        tb = t0 + nt if revs[u] else t0 - 1
        if tb < 0 or tb >= T:
            tb = (tb % T) if circular else None
        need = {j} | ({tb // G} if tb is not None else set())
        return all(done[u - 1][g] for g in need)
    order: List[Tuple[int, int]] = []
    left = len(revs) * ng
    while left:
        progressed = False
        for u in range(len(revs)):
            j = next((j for j in range(ng) if not done[u][j] and ready(u, j)), None)
            if j is None:
                continue
            order.append((u, j))
            done[u][j] = True
            left -= 1
            progressed = True
        assert progressed, "frame wavefront: dependency cycle"
    return order


def stream_plan(revs: Sequence[bool], T: int, ng: int, circular: bool, ring: int):
    """Frame groups and cross-stream edges of the streams schedule (Engine._shift_chain_streams).  Returns (groups, plan): groups[g] = (t0, nt),
    ng contiguous ranges as equal as possible; plan[u][g] = (raw, war), lists of (unit, group) whose events group g's stream waits for before it
    launches unit u.  raw: the group that owns the ONE frame g's boundary frame borrows from -- frame t0 - 1 for a forward unit, t0 + nt for a
    reverse one (gshift_deblur1.py:504-518), wrapped on the ring (gshift_deblur2.py:504-505), nobody when that frame does not exist (kept
    boundary) -- must have WRITTEN unit u - 1's output.  war: unit u overwrites ring slot u % ring, last used by unit u - ring, whose output the
    groups that borrow from g were still READING in CAB2 of unit u - ring + 1.  Same-group edges are stream order.  Pure host logic
    (tests/test_host_logic.py replays every plan against the dependency rule)."""
    assert ng >= 1 and T >= ng and ring >= 2
    bounds = [T * j // ng for j in range(ng + 1)]
    groups = [(bounds[j], bounds[j + 1] - bounds[j]) for j in range(ng)]

    def lender(u: int, g: int) -> Optional[int]:
        t0, nt = groups[g]
        tb = t0 + nt if revs[u] else t0 - 1
        if tb < 0 or tb >= T:
            if not circular:
                return None
            tb %= T
        o = max(j for j in range(ng) if bounds[j] <= tb)
        return None if o == g else o
    plan = []
    for u in range(len(revs)):
        row = []
        for g in range(ng):
            ln = lender(u, g)
            raw = [(u - 1, ln)] if (u > 0 and ln is not None) else []
            uu = u - ring + 1
            war = [(uu, gg) for gg in range(ng) if gg != g and lender(uu, gg) == g] if uu >= 1 else []
            row.append((raw, war))
        plan.append(row)
    return groups, plan


def make_engine(V: Variant, sd: Dict[str, torch.Tensor], device: torch.device, dtype: torch.dtype):
    """Engine for a module of `dtype`: fp32 modules compute in fp32 end to end (what upstream's denoise CLI does for the
    "+" model, inference/test_denoise.py:83-85); fp16 / bf16 modules run the bf16-storage MFMA kernels."""
    if dtype == torch.float32:
        from .engine32 import Engine32, Plan32
        return Engine32(Plan32(V, sd, device))
    return Engine(Plan(V, sd, device), dtype)
