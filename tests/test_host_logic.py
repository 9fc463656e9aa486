"""CPU tests of host logic: C-ABI library exports, drop-in class contract, clip-parallel halo exchange (gloo, world 2)."""
import ctypes
import os
import re
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sn_build", os.path.join(ROOT, "shift-net_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    mod.build()                                              # hipcc cross-compiles gfx950 without a GPU
    from shiftnet_amd import lib as L
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "shiftnet_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|void\*) (sn\d*_\w+)\(", header, flags=re.M)))
    assert declared == sorted(L.SYMBOLS), (declared, sorted(L.SYMBOLS))
    for s in declared:
        assert hasattr(lib, s)
    assert lib.sn_abi_version() == L.ABI_VERSION
    m = re.search(r"#define SN_ABI_VERSION (\d+)", header)
    assert m and int(m.group(1)) == L.ABI_VERSION          # header, library and binding agree (a stale .so fails load())
    # the ctypes mirrors of the ABI structs have the size the C compiler gives the header's structs
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "shiftnet_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(sn_conv_desc), '
                             'sizeof(sn32_conv_desc), sizeof(sn_unit_src), sizeof(sn_phase1_weights), sizeof(sn_se_fold), sizeof(sn_phase1_opts));return 0;}\n')
        exe = os.path.join(td, "sz")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.Conv32Desc), ctypes.sizeof(L.UnitSrc), ctypes.sizeof(L.Phase1Weights),
                     ctypes.sizeof(L.SeFold), ctypes.sizeof(L.Phase1Opts)], sizes
    # argument validation is host side and must not need a GPU
    assert lib.sn_conv2d(None, None) == -22
    d = L.ConvDesc(); d.stride, d.mt, d.n_in, d.cs_in, d.h_out, d.w_out = 1, 1, 1, 16, 720, 1280
    assert lib.sn_conv_pool_blocks(ctypes.byref(d)) == 90 * 40
    d.cs_in, d.mt = 64, 4
    assert lib.sn_conv_pool_blocks(ctypes.byref(d)) == 90 * 40            # 8x32 tiles


P1R_RB = 8        # csrc/sn_phase1r.hip: rows per pool block


def _p1r_walks(plan, nfr, h, w, lib):
    """The walks every workgroup of a phase-1 launch makes, re-derived from the plan exactly as cab_phase1r_kernel does (csrc/sn_phase1r.hip:
    team chunk of the (strip, frame block) list of row BLOCKS -> (frame, strip, rows))."""
    nsx, sd, sr, F, nfb, q, nteam = plan
    nbh = -(-h // P1R_RB)
    blocks_all = nsx * nfb * nbh
    plan7 = (ctypes.c_int * 7)(*plan)
    out = []
    for team in range(nteam):
        r0, r1 = team * q, min(team * q + q, blocks_all)
        for fm in range(F):
            for u in range(r0 // nbh, (r1 + nbh - 1) // nbh):
                s, fb = divmod(u, nfb)
                f = fb * F + fm
                if f >= nfr:
                    continue
                b0, b1 = lib.sn_p1r_strip_begin(plan7, s, w), lib.sn_p1r_strip_begin(plan7, s + 1, w)
                xo = 0 if s == 0 else b0 - 3
                cu = ((u + 1) * nbh - 1) // q - (u * nbh) // q + 1
                B0, B1 = max(r0 - u * nbh, 0), min(r1 - u * nbh, nbh)
                out.append(dict(f=f, s=s, Y0=B0 * P1R_RB, Y1=min(B1 * P1R_RB, h), cu=cu, xo=xo, olo=b0 - xo, ohi=b1 - xo))
    return out


def test_g1_store_size_is_strips_times_rows_times_ring_row():
    """sn_phase1_g1_store_bytes (host only): the scratch of the denoisers' two passes holds one kernel-order g1 row (C / 16 waves x 2 groups x 4 planes
    x 256 B = 128 C bytes: 64 region columns of fp16) per (frame, column strip, image row); bad arguments are refused."""
    from shiftnet_amd import lib as L
    lib = L.load()
    n = ctypes.c_longlong(-1)
    for (T, h, w, C) in ((2, 10, 70, 64), (5, 270, 480, 80), (1, 1, 1, 80), (3, 136, 224, 64)):
        o = (ctypes.c_int * 7)()
        assert lib.sn_p1r_plan(1, h, w, 8, 1, o) == 0
        assert lib.sn_phase1_g1_store_bytes(T, h, w, C, ctypes.byref(n)) == 0
        assert n.value == T * o[0] * h * 128 * C, (T, h, w, C, n.value, o[0])
    assert lib.sn_phase1_g1_store_bytes(1, 8, 8, 48, ctypes.byref(n)) == -22
    assert lib.sn_phase1_g1_store_bytes(0, 8, 8, 64, ctypes.byref(n)) == -22
    assert lib.sn_phase1_g1_store_bytes(1, 8, 8, 64, None) == -22


@pytest.mark.parametrize("ncu", [256, 304, 64, 8])
def test_phase1_work_plan_covers_every_row_once(ncu):
    """sn_p1r_plan (the launch's own decomposition, csrc/sn_phase1r.hip): for the level sizes of every BASELINE config, ragged and tiny maps and
    every team size, each row of each strip of each frame is walked exactly once, in whole row blocks (a pool row = one (frame, strip, block)
    has exactly one writer); strips tile the width with own columns the 64-pixel region can produce (3 halo columns towards every neighbour
    strip, none towards the image edge); the number of walks per (frame, strip) is what the squeeze-excite tail counts as arrivals."""
    from shiftnet_amd import lib as L
    lib = L.load()
    sizes = [(20, 360, 640), (20, 180, 320), (52, 360, 640), (52, 180, 320), (52, 90, 160), (16, 540, 960), (16, 270, 480), (16, 135, 240),
             (36, 136, 224), (36, 68, 112), (36, 34, 56), (1, 360, 640), (19, 360, 640), (3, 7, 21), (2, 5, 9), (5, 21, 122), (2, 9, 123), (1, 3, 40),
             (4, 33, 64), (3, 40, 65), (7, 1, 700), (4097, 8, 8)]
    for (nfr, h, w) in sizes:
        for team in (0, 1, 2, 4, 8):
            o = (ctypes.c_int * 7)()
            rc = lib.sn_p1r_plan(nfr, h, w, ncu, team, o)
            if team and (team > max(ncu // 8, 1) or max(ncu // 8, 1) % team):
                assert rc == -22, (nfr, h, w, ncu, team)
                continue
            assert rc == 0, (nfr, h, w, ncu, team)
            plan = list(o)
            nsx, sd, sr, F, nfb, q, nteam = plan
            assert lib.sn_phase1_pool_blocks(nfr, h, w) == nsx * -(-h // P1R_RB)
            assert F in (1, 2, 4, 8) and (team == 0 or F == team) and nfb == -(-nfr // F) and F * nteam <= 8 * max(ncu // 8, 1)
            b = [lib.sn_p1r_strip_begin(o, s, w) for s in range(nsx + 1)]
            assert b[0] == 0 and b[-1] == w and all(b[i] < b[i + 1] for i in range(nsx))
            seen, walks = {}, {}
            for wk in _p1r_walks(plan, nfr, h, w, lib):
                assert 0 <= wk["Y0"] < wk["Y1"] <= h and wk["Y0"] % P1R_RB == 0 and (wk["Y1"] % P1R_RB == 0 or wk["Y1"] == h)
                assert 0 <= wk["olo"] < wk["ohi"] <= 64 and wk["xo"] >= 0
                assert wk["olo"] == (0 if wk["s"] == 0 else 3)                                          # halo towards the left neighbour
                assert wk["ohi"] <= (64 if wk["s"] == nsx - 1 else 61)                                   # ... and towards the right one
                key = (wk["f"], wk["s"])
                rows = seen.setdefault(key, [0] * h)
                for y in range(wk["Y0"], wk["Y1"]):
                    rows[y] += 1
                walks[key] = walks.get(key, 0) + 1
                assert wk["cu"] >= walks[key]
                walks[key + ("cu",)] = wk["cu"]
            assert len(seen) == nfr * nsx, (nfr, h, w, ncu, team, len(seen))
            assert all(all(c == 1 for c in rows) for rows in seen.values()), (nfr, h, w, ncu, team)
            assert all(walks[k] == walks[k + ("cu",)] for k in seen)                                      # arrivals the tail waits for == walks made


@pytest.mark.parametrize("circular", [False, True])
def test_frame_wavefront_order_respects_the_shift_dependencies(circular):
    """engine.wavefront_order: every (unit, frame group) exactly once, and never before the groups of the previous unit that
    hold its own frames and the one frame its boundary frame borrows from (t - 1 forward, t + 1 reverse, wrapped on deblur2's ring)."""
    from shiftnet_amd.engine import wavefront_order
    for n_units in (1, 2, 4, 12, 24):
        revs = [i % 2 == 1 for i in range(n_units)]
        for T, G in ((20, 4), (20, 1), (7, 2), (7, 3), (52, 4), (5, 8), (16, 5)):
            order = wavefront_order(revs, T, G, circular)
            ng = -(-T // G)
            assert sorted(order) == [(u, j) for u in range(n_units) for j in range(ng)]
            pos = {k: i for i, k in enumerate(order)}
            for (u, j), i in pos.items():
                if u == 0:
                    continue
                frames = set(range(j * G, min(j * G + G, T)))
                for t in list(frames):
                    nb = t + 1 if revs[u] else t - 1
                    if 0 <= nb < T:
                        frames.add(nb)
                    elif circular:
                        frames.add(nb % T)
                assert all(pos[(u - 1, t // G)] < i for t in frames), (n_units, T, G, u, j)
            if n_units > 1 and ng > 2:                      # it IS a wavefront: the second unit starts before the first one is through
                assert pos[(1, order[[k[0] for k in order].index(1)][1])] < max(pos[(0, j)] for j in range(ng))


@pytest.mark.parametrize("circular", [False, True])
def test_streams_schedule_orders_every_cross_stream_access(circular):
    """engine.stream_plan (one HIP stream per frame group): with nothing but stream order and the plan's event waits, (a) every frame a group's
    CAB2 of unit u reads -- its own and the one its boundary frame borrows, t - 1 forward / t + 1 reverse, wrapped on deblur2's ring -- was written
    by unit u - 1 before, and (b) nobody still reads the ring slot a unit overwrites.  Happens-before is the transitive closure of
    'same stream, earlier' and the waited events; events: done(u, g) after unit u's CAB1 on g, cab2(u, g) after its CAB2."""
    from shiftnet_amd.engine import stream_plan
    for n_units in (1, 2, 3, 5, 12, 24):
        revs = [i % 2 == 1 for i in range(n_units)]
        for T, ng, ring in ((20, 2, 3), (20, 3, 3), (20, 4, 2), (7, 2, 3), (7, 3, 2), (52, 2, 3), (52, 3, 3), (5, 2, 2), (16, 5, 4)):
            groups, plan = stream_plan(revs, T, ng, circular, ring)
            assert [t for t0, nt in groups for t in range(t0, t0 + nt)] == list(range(T)) and all(nt >= 1 for _, nt in groups)
            owner = {t: g for g, (t0, nt) in enumerate(groups) for t in range(t0, t0 + nt)}
            # node (u, g, k): k = 0 CAB2 of unit u on group g, k = 1 its CAB1.  hb[n] = set of nodes that happened before n STARTS.
            hb = {}
            for u in range(n_units):
                for g in range(ng):
                    raw, war = plan[u][g]
                    before = set()
                    if u > 0:
                        before |= hb[(u - 1, g, 1)] | {(u - 1, g, 1)}
                    for uu, gg in raw:
                        assert uu < u
                        before |= hb[(uu, gg, 1)] | {(uu, gg, 1)}
                    for uu, gg in war:
                        assert uu < u
                        before |= hb[(uu, gg, 0)] | {(uu, gg, 0)}
                    hb[(u, g, 0)] = before
                    hb[(u, g, 1)] = before | {(u, g, 0)}
            for u in range(n_units):
                for g, (t0, nt) in enumerate(groups):
                    reads = set(range(t0, t0 + nt))
                    for t in range(t0, t0 + nt):
                        nb = t + 1 if revs[u] else t - 1
                        if 0 <= nb < T:
                            reads.add(nb)
                        elif circular:
                            reads.add(nb % T)
                    if u > 0:                                   # (a) read after write: unit u - 1's CAB1 of every owner of a frame CAB2 reads
                        assert all((u - 1, owner[t], 1) in hb[(u, g, 0)] for t in reads), (n_units, T, ng, ring, u, g)
                    # (b) write after read: unit u's CAB1 writes slot u % ring over unit u - ring's output, read by CAB2 of unit u - ring + 1
                    uo = u - ring
                    if uo >= 0:
                        for g2, (s0, sn) in enumerate(groups):
                            rd = set(range(s0, s0 + sn))
                            for t in range(s0, s0 + sn):
                                nb = t + 1 if revs[uo + 1] else t - 1
                                if 0 <= nb < T:
                                    rd.add(nb)
                                elif circular:
                                    rd.add(nb % T)
                            if rd & set(range(t0, t0 + nt)):
                                assert (uo + 1, g2, 0) in hb[(u, g, 0)] | {(u, g, 0)}, (n_units, T, ng, ring, u, g, g2)


def test_fp32_entry_points_refuse_what_their_kernels_do_not_implement():
    """The optional operands of sn32_conv_desc exist in specific kernels only; any other shape must come back SN_EINVAL (-22) from the host-side
    checks (no GPU needed) instead of being silently ignored -- a LayerNorm / residual scale / channel sum that is not applied is a wrong result."""
    from shiftnet_amd import lib as L
    lib = L.load()
    buf = (ctypes.c_char * 4096)()
    p = (ctypes.addressof(buf) + 63) & ~63                     # a non-NULL, 64-byte aligned address; never dereferenced: validation precedes any launch

    def desc(**kw):
        d = L.Conv32Desc()
        d.inp[0], d.c_in[0], d.cs_in[0], d.n_in = p, 80, 80, 1
        d.T, d.h_in, d.w_in, d.h_out, d.w_out = 2, 16, 48, 16, 48
        d.k, d.stride, d.pad, d.groups, d.c_out = 1, 1, 0, 1, 160
        d.w, d.out, d.cs_out, d.wsplit = p, p, 160, p
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    EINVAL = -22
    # LayerNorm on load: split 1x1 only -- not without wsplit, not for k = 3, not without its bias vector, not together with an input scale
    assert lib.sn32_conv2d(ctypes.byref(desc(ln_w=p, ln_b=p, wsplit=None)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(ln_w=p, ln_b=p, k=3, pad=1)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(ln_w=p, ln_b=None)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(ln_w=p, ln_b=p, iscale=p, iscale_stride=80)), None) == EINVAL
    # residual scale: grouped-by-8 only, and only with a residual
    assert lib.sn32_conv2d(ctypes.byref(desc(rscale=p, rscale_stride=160, res=p, cs_res=160)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(rscale=p, rscale_stride=80, k=5, pad=2, groups=10, c_out=80, cs_out=80)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(rscale=p, rscale_stride=80, k=5, pad=2, groups=10, c_out=80, cs_out=80, res=p, cs_res=80, wsplit=None)), None) == EINVAL   # exact products
    # channel sums: split dense 3x3 only, cpad within the kernel's M-tiles
    assert lib.sn32_conv2d(ctypes.byref(desc(csum=p, csum_cpad=160)), None) == EINVAL
    assert lib.sn32_conv2d(ctypes.byref(desc(csum=p, csum_cpad=8, k=3, pad=1, c_out=24, cs_out=24, c_in=(ctypes.c_int * 3)(24, 0, 0), cs_in=(ctypes.c_int * 3)(24, 0, 0))), None) == EINVAL
    assert lib.sn32_conv_csum_tiles(272, 448) == 68 * 14
    # fused operators
    assert lib.sn32_dw_gate(p, 120, p, 80, 80, p, 2, 16, 48, 64, None, None) == EINVAL            # pixel stride below 2C
    assert lib.sn32_dw_gate(p, 160, p, 78, 80, p, 2, 16, 48, 64, None, None) == EINVAL            # C not a multiple of 4
    assert lib.sn32_dw_gate(p, 160, p, 80, 64, p, 2, 16, 48, 64, p, None) == EINVAL               # sums asked with cpad < C
    assert lib.sn32_conv1x1_gate2(p, 80, 80, p, 80, 80, p, 2, 16 * 48 + 8, p, None) == EINVAL       # a workgroup would span two frames
    assert lib.sn32_conv1x1_gate2(p, 80, 80, p, 72, 72, p, 2, 16 * 48, p, None) == EINVAL           # C not a multiple of 16
    assert lib.sn32_conv1x1_gate2(p, 80, 80, None, 80, 80, p, 2, 16 * 48, p, None) == EINVAL        # exact-fp32 arithmetic has no gated form
    s = L.UnitSrc(); s.x, s.T, s.h, s.w, s.C, s.mode = p, 2, 16, 48, 80, 1
    assert lib.sn32_gsts_shiftconv(ctypes.byref(s), p, p, p, p, None) == EINVAL                    # both forms at once
    assert lib.sn32_gsts_shiftconv(ctypes.byref(s), p, None, p, None, None) == EINVAL              # neither
    s.C = 76
    assert lib.sn32_gsts_shiftconv(ctypes.byref(s), p, None, p, p, None) == EINVAL                 # C % 8


def test_pmc_summaries_are_bound_to_the_sources_of_their_kernels():
    """bench.py reports PMC traffic only while every translation unit that defines a kernel of the trace (and the shared headers) is unchanged."""
    import copy
    import json
    sys.path.insert(0, ROOT)
    import bench
    ks = bench.kernel_sources()
    assert "cab_phase1r_kernel" in ks["sn_phase1r.hip"] and "scale_gemm_res_kernel" in ks["sn_gsts.hip"] and "conv32s_kernel" in ks["sn_f32.hip"]
    assert not set.intersection(*[ks["sn_f32.hip"], set().union(*(v for f, v in ks.items() if f != "sn_f32.hip"))])     # a kernel name maps to one unit
    doc = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_hbm_traffic_cfg2.json")))      # a real summary: its kernel list, re-bound to today's sources
    doc["csrc_files"] = bench.csrc_files(doc["kernels_per_window"].keys())
    files = doc["csrc_files"]
    assert {"sn_phase1r.hip", "sn_gsts.hip", "sn_conv.hip", "sn_common.h", "shiftnet_hip.h"} <= set(files)
    assert "sn_f32.hip" not in files                       # a bf16 window launches nothing from the fp32 engine's unit
    assert bench.pmc_sources_changed(doc) == []
    stale = copy.deepcopy(doc)
    stale["csrc_files"]["sn_gsts.hip"] = "0" * 16
    assert bench.pmc_sources_changed(stale) == ["sn_gsts.hip"]
    old = {k: v for k, v in doc.items() if k != "csrc_files"}           # summaries from before the per-unit record: whole tree or nothing
    old["csrc_hash"] = "0" * 16
    assert bench.pmc_sources_changed(old) == ["(whole tree: csrc_hash)"]


def test_dropin_class_contract():
    from basicsr.models.archs import gshift_deblur1, gshift_deblur2, gshift_denoise1, gshift_denoise2
    from shiftnet_amd.weights import synth_state_dict
    for mod, name, pf in ((gshift_deblur1, "gshift_deblur1", (1, 1)), (gshift_deblur2, "gshift_deblur2", (1, 1)),
                          (gshift_denoise1, "gshift_denoise1", (0, 0)), (gshift_denoise2, "gshift_denoise2", (0, 0))):
        net = mod.make_model({"pretrain_models_dir": "None"})
        assert type(net).__name__ == "GShiftNet" and (net.num_fb, net.num_ff) == pf
        net.load_state_dict(synth_state_dict(name), strict=True)
        bad = dict(synth_state_dict(name)); bad.pop("conv_last.weight")
        with pytest.raises(RuntimeError):
            net.load_state_dict(bad, strict=True)
        net2 = mod.GShiftNet(future_frames=2, past_frames=2).half()
        assert next(net2.parameters()).dtype == torch.float16
        # shared PReLU aliases stay shared through state_dict round trips
        sd = net.state_dict()
        assert sd["stage1.act.weight"].data_ptr() == sd["stage1.concat.body.1.weight"].data_ptr()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net(torch.zeros(1, 5, 3 if "deblur" in name else 3, 8, 8), *([torch.zeros(1, 5, 1, 8, 8)] if "denoise" in name else []))


def test_weight_plan_is_dropped_when_parameters_change():
    """Every way the parameters can change must invalidate the prepared device weights (shiftnet_amd/arch.py): a checkpoint loaded
    through a PARENT module reaches the class only as _load_from_state_dict, in-place edits only bump version counters."""
    from basicsr.models.archs import gshift_deblur2
    from shiftnet_amd.engine import host_f32_copy
    net = gshift_deblur2.GShiftNet()
    sig0 = net._param_signature()
    assert sig0 == net._param_signature()
    net._plan = object()
    parent = torch.nn.Module(); parent.net = net
    parent.load_state_dict(parent.state_dict())
    assert net._plan is None                                     # recursive load
    sig1 = net._param_signature()
    p = next(net.parameters())
    p.data.mul_(1.0)
    with torch.no_grad():
        p.add_(0.0)                                              # in-place edit: version counter
    assert net._param_signature() != sig1
    net._plan = object(); net.half(); assert net._plan is None   # _apply
    net._plan = object(); net.load_state_dict(net.state_dict()); assert net._plan is None
    sd = {"a": torch.arange(6.0).reshape(2, 3).half(), "b": torch.ones(4)}
    cp = host_f32_copy(sd)
    assert list(cp) == ["a", "b"] and cp["a"].dtype == torch.float32 and torch.equal(cp["a"], sd["a"].float()) and cp["b"].shape == (4,)


def test_window_ranges_match_cli_arithmetic():
    from shiftnet_amd.clip_parallel import window_ranges
    # test_deblur.py:111-120 with N=100, one_len=48: k_len = 96//48 = 2
    r = window_ranges(100, 48)
    assert [(a.start, a.stop, b.start, b.stop) for a, b in r] == [(0, 52, 2, 50), (48, 100, 50, 98)]
    assert window_ranges(30, 16) == [(range(0, 20), range(2, 18))]      # remainder frames dropped
    assert window_ranges(4, 16) == []


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _halo_worker(rank, world, port, L, active, all_gather):
    sys.path.insert(0, os.path.join(ROOT, "shift-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shiftnet_amd.clip_parallel import assemble_window, window_ranges
    n = active * L + 4                              # a round in which only the first `active` ranks hold a window (the last round of a clip)
    # a 1080p-shaped (16:9) clip, small; frame f is identifiable from any of its pixels
    clip = torch.arange(n * 3 * 9 * 16, dtype=torch.float32).reshape(n, 3, 9, 16)
    r = rank if rank < active else 0                # idle ranks hold filler frames, as the CLI does
    own = clip[r * L + 2: r * L + 2 + L]
    win = assemble_window(own, clip[:2] if rank == 0 else None, clip[-2:] if rank == active - 1 else None, rank, world, active=active,
                          all_gather=all_gather)
    if rank >= active:
        assert win is None
    else:
        # window r of the clip-parallel run == window r of the single-GPU CLI loop (inference/test_deblur.py:111-120)
        rin, rout = window_ranges(n, L)[rank]
        assert torch.equal(win, clip[rin.start:rin.stop]), rank
        assert torch.equal(win[2:-2], clip[rout.start:rout.stop]) and torch.equal(win[2:-2], own), rank
    if rank == 0 and world > 1:
        with pytest.raises(ValueError):             # one_len 1: the halo would span two neighbours
            assemble_window(own[:1], clip[:2], None, rank, world)
    dist.barrier()
    dist.destroy_process_group()


# 8 x one_len 12 = BASELINE config 5's partition of 96 frames; (4, 16, 3): a last round with an idle rank; the last case: round 4's all-gather form
@pytest.mark.parametrize("world,L,active,all_gather", [(2, 5, 2, False), (4, 16, 4, False), (8, 12, 8, False), (4, 16, 3, False), (3, 2, 3, False),
                                                       (4, 6, 4, True)])
def test_halo_exchange_gloo(world, L, active, all_gather):
    """clip_parallel.assemble_window: two raw frames to / from each neighbour (batch_isend_irecv), world 2 / 3 / 4 / 8."""
    port = _free_port()
    mp.spawn(_halo_worker, args=(world, port, L, active, all_gather), nprocs=world, join=True)


def _halo_form_worker(rank, world, port, L, mode, break_p2p, active):
    sys.path.insert(0, os.path.join(ROOT, "shift-net_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shiftnet_amd import clip_parallel as CP
    if break_p2p and rank in break_p2p:              # the point-to-point launch raises on these ranks BEFORE anything is posted (what an
        def boom(ops):                               # unsupported transport does); the other ranks may or may not have failed too
            raise RuntimeError("simulated: point-to-point transport unavailable")
        CP.dist.batch_isend_irecv = boom
    logs = []
    halo = CP.Halo(mode, timeout_s=60, log=logs.append)
    n = world * L + 4
    clip = torch.arange(n * 3 * 4 * 6, dtype=torch.float32).reshape(n, 3, 4, 6)
    try:
        for rnd, act in enumerate((world, active, world)):   # three rounds, the middle one possibly partial (idle ranks); the form is decided in the first
            r = rank if rank < act else 0
            own = clip[r * L + 2: r * L + 2 + L]
            win = halo.assemble(own, clip[:2] if rank == 0 else None, clip[act * L + 2: act * L + 4] if rank == act - 1 else None, rank, world, active=act)
            if rank >= act:
                assert win is None
            else:
                assert torch.equal(win, clip[rank * L: rank * L + L + 4]), (rank, rnd)
        want = "allgather" if (mode == "allgather" or break_p2p) else "p2p"
        assert halo.form == want and halo.fell_back == bool(break_p2p and mode == "auto"), (halo.form, halo.fell_back)
        assert bool(logs) == halo.fell_back
        failed = False
    except RuntimeError as e:
        failed = "point to point" in str(e)
    assert failed == (mode == "p2p" and bool(break_p2p)), (mode, break_p2p, failed)       # a forced p2p run fails on EVERY rank, nobody hangs
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,break_p2p,active", [(2, "auto", None, 2), (3, "p2p", None, 2), (2, "allgather", None, 2), (4, "auto", (0, 1, 2, 3), 3),
                                                         (3, "p2p", (0, 1, 2), 3)])
def test_halo_form_selection_and_fallback_gloo(world, mode, break_p2p, active):
    """clip_parallel.Halo (bench.py / the CLIs' --halo): point to point by default; when the first exchange raises, ALL ranks learn it through one
    all-reduce and continue with the all-gather form (logged once); --halo p2p turns the same failure into an error on every rank; --halo allgather
    never tries point to point.  Every later round -- including one with idle ranks -- uses the form chosen in the first."""
    mp.spawn(_halo_form_worker, args=(world, _free_port(), 4, mode, break_p2p, active), nprocs=world, join=True)


def test_cli_refuses_clip_parallel_modes_it_cannot_run():
    """--gpus N with one_len < 2 (a halo would span two neighbour windows) or with --host_io (a single-process mode) must stop in the argument
    parser of EVERY rank, before any process group exists (the single-process CLI accepts both)."""
    from shiftnet_amd import cli
    for extra in (["--one_len", "1"], ["--one_len", "4", "--host_io"]):
        with pytest.raises(SystemExit) as e:
            cli.main("gshift_deblur2", ["--synthetic", "32", "32", "12", "--gpus", "2"] + extra)
        assert e.value.code == 2


def test_cli_spawn_stops_all_ranks_when_one_fails():
    """cli._spawn_ranks (the `--gpus N` launcher of the drop-in CLIs): ranks are started as `python -m shiftnet_amd.cli <variant> ...` whatever
    program called it (pytest here), and a rank that exits non-zero ends the launch at once -- its siblings are terminated instead of waiting in
    a collective until the process-group timeout.  An unknown variant makes every rank fail at start-up: no GPU needed."""
    import time
    from shiftnet_amd import cli
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        cli._spawn_ranks("no_such_variant", 2, ["--synthetic", "32", "32", "12", "--one_len", "4"])
    assert "exited with codes" in str(e.value) and time.time() - t0 < 600


def test_bench_refuses_fewer_gpus_than_asked():
    """`bench.py --gpus N` must never print an N-GPU line from fewer devices (this container has none)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip().startswith("{")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_cli_window_arithmetic_and_metrics():
    import numpy as np
    from shiftnet_amd import cli
    # denoise: whole clip minus 4 in one window; > 100 frames -> halved, residual appended to the last window
    assert cli.denoise_windows(54) == [(0, 50, 0)]
    assert cli.denoise_windows(131) == [(0, 63, 0), (63, 64, 1)]          # 127 -> one_len 63, k_len 2, residual 1
    a = np.full((8, 8, 3), 100.0); b = np.full((8, 8, 3), 110, np.uint8)
    assert abs(cli.psnr_255(a, b) - 10 * np.log10(255.0 ** 2 / 100.0)) < 1e-9
    assert abs(cli.ssim_calculate(b.astype(np.float64), b) - 1.0) < 1e-6
    x = cli.numpy2tensor([b, b])
    assert x.shape == (1, 2, 3, 8, 8) and abs(float(x.max()) - 110 / 255) < 1e-7


def test_xcd_tile_walk_is_a_bijection(tmp_path):
    """The XCD-aware tile walk of K0 / K12 / the dense convs (csrc/sn_common.h: sn_xcd_tiles, sn_xcd_grid, sn_xcd_decode) maps the launch
    grid onto (frame, tile row, tile column): every tile exactly once, padding workgroups rejected, for 729 grid shapes including the
    remainder chunks and nty == 1.  The decode function is __host__ __device__, so the very code the kernels run is checked here with a
    host-only hipcc build (no GPU)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "xcd_tiles_check")
    for flags in ([], ["-DSN_XCD_TILES=0"]):
        r = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-host-only", "-I", os.path.join(root, "shift-net_amd", "csrc")] + flags +
                           [os.path.join(root, "tests", "host", "xcd_tiles_check.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "xcd tiles ok" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]


def test_failed_graph_capture_falls_back_to_eager_without_poisoning_later_calls(monkeypatch):
    """hipGraph capture is automatic for small windows.  A capture that fails (another thread allocating, an unsupported call) must leave
    the engine eager for that signature, warn once, and keep serving this and later calls; the capture itself asks for thread-local error mode."""
    from collections import OrderedDict
    from shiftnet_amd.engine import Engine
    eng = object.__new__(Engine)                                     # no device: only the capture bookkeeping is under test
    eng._graphs, eng.dev = OrderedDict(), torch.device("cpu")
    calls = []
    eng._forward = lambda x, n, p, f: (calls.append("eager"), x + 1)[1]
    modes = []

    class FakeGraph:
        def reset(self):
            pass

    class FailingCapture:
        def __init__(self, g, capture_error_mode="global"):
            modes.append(capture_error_mode)

        def __enter__(self):
            raise RuntimeError("operation not permitted when stream is capturing")

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "graph", FailingCapture)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    x = torch.zeros(2, 3, 4, 4)
    assert torch.equal(eng._forward_graphed(x, None, 2, 2), x + 1)   # first sight: eager
    with pytest.warns(UserWarning, match="capture failed"):
        assert torch.equal(eng._forward_graphed(x, None, 2, 2), x + 1)      # second: capture attempted, fails, eager result
    assert modes == ["thread_local"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                               # later calls: eager, quiet, no new capture attempt
        assert torch.equal(eng._forward_graphed(x, None, 2, 2), x + 1)
    assert modes == ["thread_local"] and calls == ["eager"] * 3
    y = torch.zeros(1, 3, 4, 4)                                      # another signature is not poisoned by the failure: it gets its own attempt
    eng._forward_graphed(y, None, 2, 2)
    with pytest.warns(UserWarning, match="capture failed"):
        eng._forward_graphed(y, None, 2, 2)
    assert len(modes) == 2
