"""Earlier generations of the GSTS chain, for A/B measurements only (needs the -DSN_EXPERIMENTAL library:
``python shift-net_amd/build.py --experimental`` and ``SN_EXPERIMENTAL=1`` in the environment).  Not on any product path.

gsts_v: 0 = round-1 five-kernel chain, 1 = fused K12 + LDS-staged VALU 5x5 (K3'), 3 = K12m (3x3 on the matrix cores) + K3m;
2 = the production chain (engine.Engine.naf).
"""
from __future__ import annotations

import ctypes as C

import torch

from .engine import Act, Engine


class ExperimentalEngine(Engine):
    gsts_v = 2

    def naf(self, pre: str, x: Act, mode: int) -> Act:
        """CAB2 (mode 1/2) / CAB1 (mode 0) through one of the superseded kernel chains (gshift_deblur1.py:183-255)."""
        if self.gsts_v == 2:
            return Engine.naf(self, pre, x, mode)
        lib, st, V, P = self.lib, self._stream(), self.V, self.P
        u = P.units[pre]
        T, h, w, c = x.dims
        self._meta = ("naf", T, h, w, c, mode)
        if mode and self.split is not None:
            self.split.exchange(x.t, mode)              # neighbour rank's half-frame into the halo slot (one per shifted unit)
        src = self._unit_src(x, mode)
        hw_ptr = None
        if mode:
            hwb = self._new(T, h, w, c // 2)
            self._call("sn_gsts_shiftconv", "sn_gsts_shiftconv", C.byref(src), P.offs.data_ptr(), u["w1"].data_ptr(), hwb.data_ptr(), st)
            hw_ptr = hwb.data_ptr()
        ca1_ptr = None
        pool1 = None
        mstencil = self.gsts_v >= 2 and c == 64 and not V.grouped_rep                 # K3m: Toeplitz-MFMA 5x5 on planar g1
        blocked = 1 if (self.gsts_v >= 1 and c == 64 and not V.grouped_rep and not mstencil) else 0   # g1 [T][4][h][w][16] for K3'
        if mstencil:
            blocked = 2                                                                # g1 channel-planar [T][h][C][wr]
            g1 = torch.empty((T, h, c, lib.sn_planar_pitch(w)), dtype=torch.bfloat16, device=self.dev)
        else:
            g1 = self._new(T, h, w, c)
        if mstencil and self.gsts_v >= 3:     # K12m: the depthwise 3x3 on the matrix cores as well, planar g1 without a transpose epilogue
            if V.denoise:
                pool1 = torch.empty((T, lib.sn_lngatem_blocks(h, w), c), dtype=torch.float32, device=self.dev)
            self._call("sn_ln_gemm_gate_m", "sn_ln_gemm_gate_m", C.byref(src), hw_ptr, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(),
                       u["w_toep3"].data_ptr(), g1.data_ptr(), pool1.data_ptr() if pool1 is not None else None, st)
        elif self.gsts_v >= 1:      # fused LN + 1x1 + dw3x3 + gate: the 2C tensor stays in LDS
            if V.denoise:
                pool1 = torch.empty((T, lib.sn_lngate_blocks(h, w), c), dtype=torch.float32, device=self.dev)
            self._call("sn_ln_gemm_gate", "sn_ln_gemm_gate", C.byref(src), hw_ptr, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(),
                       u["w_dw3_h2"].data_ptr(), g1.data_ptr(), pool1.data_ptr() if pool1 is not None else None, blocked, st)
        else:
            a = self._new(T, h, w, 2 * c)
            self._call("sn_ln_gemm", "sn_ln_gemm", C.byref(src), hw_ptr, u["w_ln"].data_ptr(), u["b_ln"].data_ptr(), a.data_ptr(), st)
            if V.denoise:
                pool1 = torch.empty((T, lib.sn_dwgate_blocks(h, w), c), dtype=torch.float32, device=self.dev)
            self._call("sn_dw_gate", "sn_dw_gate", a.data_ptr(), u["w_dw3"].data_ptr(), g1.data_ptr(),
                       pool1.data_ptr() if pool1 is not None else None, T, h, w, c, st)
        if V.denoise:
            ca1 = self.ca_mlp(pre + "ca1", pool1, h * w)
            ca1_ptr = ca1.data_ptr()
        if V.grouped_rep and self.gsts_v >= 1:
            # "+" RepConv (groups = C/8) as a block-diagonal MFMA GEMM fused with the 1x1 / SimpleGate2 that follow it
            g2 = self._new(T, h, w, c)
            pool2 = torch.empty((T, lib.sn_grp5_blocks(h, w), c), dtype=torch.float32, device=self.dev)
            self._call("sn_grp5_gemm_gate", "sn_grp5_gemm_gate", g1.data_ptr(), ca1_ptr, u["w_grp"].data_ptr(), u["w_gate"].data_ptr(),
                       g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st)
            ca2 = self.ca_mlp(pre + "ca2", pool2, h * w)
            y = self._new(T, h, w, c)
            b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
            self._call("sn_scale_gemm_res", "sn_scale_gemm_res", C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out,
                       y.data_ptr(), st)
            return Act(y, c)
        if V.grouped_rep:
            # SN_GSTS_V=0: the grouped 5x5 as a block-diagonal dense conv on the MFMA conv kernel (10x redundant MFMA work);
            # the CALayer2 scale of the denoise variant must precede it, so it is applied by a scale pass first.
            g1a = Act(g1, c)
            if ca1_ptr is not None:
                zero = torch.zeros_like(g1)
                g1a = self.scale_residual(g1a, Act(zero, c), ca1)
                ca1_ptr = None
            g1 = self.conv(pre + "rep", [g1a]).t
        if mstencil:
            g1p = g1
            g2 = self._new(T, h, w, c)
            pool2 = torch.empty((T, lib.sn_dw5m_blocks(h, w), c), dtype=torch.float32, device=self.dev)
            self._call("sn_dw5m_gemm_gate", "sn_dw5m_gemm_gate", g1p.data_ptr(), ca1_ptr, u["w_toep5"].data_ptr(), u["w_gate"].data_ptr(),
                       g2.data_ptr(), pool2.data_ptr(), T, h, w, c, st)
            ca2 = self.ca_mlp(pre + "ca2", pool2, h * w)
            y = self._new(T, h, w, c)
            b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
            self._call("sn_scale_gemm_res", "sn_scale_gemm_res", C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out,
                       y.data_ptr(), st)
            return Act(y, c)
        g2 = self._new(T, h, w, c)
        k3 = "sn_dw5_gemm_gate" if (self.gsts_v >= 1 and c == 64 and not V.grouped_rep) else "sn_dw_gemm_gate"
        nb = lib.sn_dw5_blocks(h, w) if k3 == "sn_dw5_gemm_gate" else lib.sn_dwgemm_blocks(h, w)
        pool2 = torch.empty((T, nb, c), dtype=torch.float32, device=self.dev)
        self._call(k3, k3, g1.data_ptr(), ca1_ptr, u["w_dw5_d2" if k3 == "sn_dw5_gemm_gate" else "w_dw5"].data_ptr(),
                   u["w_gate_blk" if k3 == "sn_dw5_gemm_gate" else "w_gate"].data_ptr(), g2.data_ptr(),
                                    pool2.data_ptr(), T, h, w, c, st)
        ca2 = self.ca_mlp(pre + "ca2", pool2, h * w)
        y = self._new(T, h, w, c)
        b_out = u["b_out"].data_ptr() if u["b_out"] is not None else None
        self._call("sn_scale_gemm_res", "sn_scale_gemm_res", C.byref(src), g2.data_ptr(), ca2.data_ptr(), u["w_out"].data_ptr(), b_out, y.data_ptr(), st)
        return Act(y, c)

