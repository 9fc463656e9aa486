#!/usr/bin/env python3
"""Headline benchmark: restored frames/s of the Shift-Net deblur hot path on MI355X (BASELINE.json configs[1]).

One "step" = one clip window per GPU: Shift-Net-s (gshift_deblur2), 1280x720, one_len = 16 restored frames from
T_in = 20 input frames, bf16 storage / fp32 accumulation, synthetic GoPro-shaped frames and a synthetic checkpoint
(no datasets or weights ship with the reference).  The frames a rank owns are resident in HBM when the timed region
starts; a step is  [halo exchange (N>1 only)] -> GShiftNet.forward -> restored frames in HBM.

N > 1: launched by torch.distributed.run, one rank per GPU; windows are independent (weak scaling), the only
communication is the all-gather of the 2+2 halo frames (shiftnet_amd/clip_parallel.py).

Prints ONE JSON line on rank 0 (see the repository prompt / DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "shift-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it); must be set before the HIP runtime starts
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command line with --no-parity --no-cpu-baseline (tools/make_profiles_r04.sh), one file
# per preset; every file records the hash of the kernel sources it was measured on and is ignored when they have changed since.
PMC_FILES = {(2, "bf16"): "r06_pmc_hbm_traffic_cfg2.json", (3, "bf16"): "r06_pmc_hbm_traffic_cfg3.json",
             # config 4: ONE of the CLI's four quadrants traced (bench.py --config 4 --one-quadrant: a quarter of the ~2900 launches, the four differ
             # only in their crop); tools/pmc_summary.py scales the per-window totals by 4 (PMC_WINDOW_FRACTION=0.25, recorded in the file)
             (4, "bf16"): "r06_pmc_hbm_traffic_cfg4_bf16.json"}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
VARIANT = "gshift_deblur2"
H, W, ONE_LEN = 720, 1280, 16
# elements per full-res pixel: stage0+stage1 per INPUT frame, stage2 per OUTPUT frame
ELEMS = {"gshift_deblur2": (765.2 + 2180.0, 765.2), "gshift_deblur1": (2268.0 + 2866.5, 2268.0),
         "gshift_denoise2": (766.2 + 2194.0, 765.2), "gshift_denoise1": (2269.0 + 2962.5, 2268.0)}


# BASELINE.json configs as bench presets (config 1 is the CPU case: it is the `cpu_baseline` leg and tests/test_gpu_fp32.py)
CONFIGS = {
    2: dict(variant="gshift_deblur2", height=720, width=1280, one_len=16, dtype="bf16", quadrants=False),
    3: dict(variant="gshift_deblur1", height=720, width=1280, one_len=48, dtype="bf16", quadrants=False),
    # config 4: davis_denoise sigma 30, 854x480 T=32: the CLI crops to 852x480 and runs 4 overlapping quadrants of 272x448
    # (inference/test_denoise.py:153-173); upstream keeps this model in float32 (:83-85) -> --dtype fp32 is the reference
    # arithmetic, bf16 the fast path
    4: dict(variant="gshift_denoise1", height=480, width=852, one_len=32, dtype="bf16", quadrants=True),
    # config 5's per-GPU window: 1920x1080, 96 restored frames over 8 GPUs = one_len 12 each (weak scaling with --gpus N)
    5: dict(variant="gshift_deblur1", height=1080, width=1920, one_len=12, dtype="bf16", quadrants=False),
    # north_star's scaling workload: Shift-Net+ on 1080p windows of one_len 16, one per GPU, at --gpus 1 / 2 / 4 / 8
    6: dict(variant="gshift_deblur1", height=1080, width=1920, one_len=16, dtype="bf16", quadrants=False),
}
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def algorithmic_bytes_window(variant, h, w, t_in, t_out, s=2):
    e_in, e_out = ELEMS[variant]
    return h * w * s * (t_in * e_in + t_out * e_out)


def csrc_hash():
    """Identity of the kernels a PMC file was measured on: sha256 over the HIP sources and the C header, in name order."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "shift-net_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "shiftnet_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_sources():
    """{source file: names of the __global__ kernels it defines} for shift-net_amd/csrc/*.hip (template kernels by their base name)."""
    import re
    d = os.path.join(ROOT, "shift-net_amd", "csrc")
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip"):
            out[f] = set(re.findall(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", open(os.path.join(d, f)).read(), flags=re.S))
    return out


def csrc_files(kernel_names):
    """Identity of the sources a set of traced kernels was compiled from: sha256[:16] of every translation unit that defines one of them, plus
    the headers every unit includes.  Each .hip is compiled on its own (build.py), so a change in a unit none of whose kernels ran in the
    trace cannot change what the trace measured -- and a change anywhere else voids it."""
    import hashlib
    import re
    d = os.path.join(ROOT, "shift-net_amd", "csrc")
    base = {re.sub(r"^.*::", "", k.split("<")[0].split("(")[0]).strip() for k in kernel_names}
    files = [f for f, ks in kernel_sources().items() if ks & base]
    files += [f for f in sorted(os.listdir(d)) if f.endswith(".h")] + [os.path.join("..", "..", "include", "shiftnet_hip.h")]
    return {os.path.basename(f): hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()[:16] for f in files}


def pmc_sources_changed(doc):
    """Source files of the kernels in a PMC summary (tools/pmc_summary.py) whose content differs from what the summary was measured on."""
    want = doc.get("csrc_files")
    if not want:                                   # files written before the per-unit record: the whole tree must match
        return [] if doc.get("csrc_hash") == csrc_hash() else ["(whole tree: csrc_hash)"]
    have = csrc_files(doc["kernels_per_window"].keys())
    return sorted(f for f in set(want) | set(have) if want.get(f) != have.get(f))


def symbol_key(sym):
    """rocprofv3 kernel name -> the key bench.py aggregates the same kernel under (None: not one of ours)."""
    import re
    s = sym.replace("(anonymous namespace)::", "")
    m = re.search(r"cab_phase1r_kernel<\d+, (true|false)", s)
    if m:
        return "sn_gsts_cab2_phase1" if m.group(1) == "true" else "sn_cab1_phase1"
    m = re.search(r"cab_phase1_kernel<(\d)", s)
    if m:
        return "sn_gsts_cab2_phase1" if m.group(1) == "3" else "sn_cab1_phase1"
    m = re.search(r"ln_gemm_gate_kernel<\d+, (true|false)>", s)
    if m:
        return "sn_ln_gemm_gate<cab2>" if m.group(1) == "true" else "sn_ln_gemm_gate<cab1>"
    for pat, key in (("scale_gemm_res_kernel", "sn_cab_phase2"), ("shiftconv_kernel", "sn_gsts_shiftconv"), ("shiftconv_mfma_kernel", "sn_gsts_shiftconv"), ("shiftconv_mfma_walk_kernel", "sn_gsts_shiftconv"), ("grp5p_gemm_gate_kernel", "sn_grp5_gemm_gate"),
                     ("dw5m_gemm_gate_kernel", "sn_dw5m_gemm_gate"), ("ca_mlp_kernel", "sn_ca_mlp"), ("cab_ca", "sn_cab_ca"), ("gather_kernel", "sn_temporal_roll"),
                     ("ingest_kernel", "sn_ingest"), ("upsample2_add_kernel", "sn_upsample2_add")):
        if pat in s:
            return key
    m = re.search(r"conv3p_kernel<(\d+), \d+, \d+, \d+, (\d+)", s)      # <M-tiles, channels, tile rows, depth, MODE, ...>: MODE 3 = the fused CAB's statistics pass
    if m:
        return f"sn_cab_stats<mt{m.group(1)}>" if m.group(2) == "3" else f"sn_conv2d<mt{m.group(1)},8x32>"
    m = re.search(r"(?:cab_fused|cabp)_kernel<(\d+), ", s)
    if m:
        return f"sn_cab_fused<mt{m.group(1)}>"
    m = re.search(r"conv3_fast_kernel<(\d+), \d+, \d+, true>", s)
    if m:
        return f"sn_cab_stats<mt{m.group(1)}>"
    m = re.search(r"conv3_fast_kernel<(\d+), ", s)
    if m:
        return f"sn_conv2d<mt{m.group(1)},8x32>"
    m = re.search(r"conv_mfma_kernel<(\d+), (\d+), (\d+)>", s)
    if m:
        return f"sn_conv2d<mt{m.group(1)},{m.group(2)}x{m.group(3)}>"
    m = re.search(r"(sn32_\w+|conv32\w*_kernel|\w+32\w*_kernel)", s)
    if m:
        return "fp32:" + m.group(1)
    return None


def kernel_alg_bytes(fn, meta):
    """Minimal HBM bytes of ONE launch given its interface (each distinct input read once, output written once)."""
    if meta and meta[0] == "ew32":             # fp32 engine, non-conv operator: the engine states its own minimal bytes
        return meta[1]
    if meta and meta[0] == "conv32":           # fp32 engine: NHWC float32, channel counts as given
        _, T, ho, wo, cin, co, k, stride, in_mode, out_mode = meta[:10]
        pin = T * ho * wo * stride * stride / (4 if in_mode == 1 else 1)
        return 4 * (pin * cin + T * ho * wo * (co // 4 if out_mode == 1 else co))
    if meta and meta[0] == "naf":
        _, T, h, w, c, mode = meta[:6]
        px = T * h * w * 2
        p1 = px * (2.5 * c if mode else 2 * c)                                       # phase 1: read x (+ hw), write g2 (or g1)
        return {"sn_gsts_shiftconv": px * c, "sn_gsts_cab2_phase2": px * 3 * c, "sn_cab1_phase2": px * 3 * c, "sn_ln_gemm_gate": p1,
                "sn_dw5m_gemm_gate": px * 2 * c, "sn_grp5_gemm_gate": px * 2 * c, "sn_gsts_cab2_phase1": p1, "sn_cab1_phase1": p1}.get(fn, 0)
    if meta and meta[0] == "up2add":           # SkipUpSample's tail: low-resolution conv result in, skip in, result out
        return 2 * meta[1] * meta[2] * meta[3] * meta[4] * (1 + 4 + 4)
    if meta and meta[0] == "roll":             # Shift_CAB's temporal roll: one read, one write
        return 2 * meta[1] * meta[2] * meta[3] * meta[4] * 2
    if meta and meta[0] == "cabf":             # fused dense CAB: statistics pass reads x; fused pass reads x (+ the second residual) and writes out
        return meta[5] * meta[1] * meta[2] * meta[3] * meta[4] * 2
    if meta and meta[0] == "conv":
        _, T, ho, wo, cin, cs_out, k, stride, in_mode, out_mode = meta
        pin = T * ho * wo * stride * stride / (4 if in_mode == 1 else 1)
        return 2 * (pin * cin + T * ho * wo * cs_out)
    return 0


def usable_cores():
    """Cores this process may really use: affinity mask, cgroup v2 quota, capped at 64 (one socket's physical cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_subprocess(timeout_s=240):
    """Run the CPU leg in a child with a hard timeout so that the bench line is always produced."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "restored frames/s", "cores": usable_cores(), "kind": "port",
                "sample": "cpu leg failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "restored frames/s", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu leg exceeded {timeout_s} s and was stopped"}


def cpu_baseline(budget_s=6.0):
    """The CPU oracle (a port of the reference forward, kind 'port') timed on this host's cores on a bounded sample."""
    from oracle import shiftnet_oracle as O
    from shiftnet_amd import synth
    from shiftnet_amd.weights import synth_state_dict
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth_state_dict(VARIANT)
    V = O.VARIANTS[VARIANT]

    def run(t_in, h, w):
        blur, _ = synth.blurred_clip(t_in, h, w, seed=2)
        x = O.frames_to_tensor(list(blur))
        t0 = time.time()
        with torch.no_grad():
            O.forward(V, sd, x, None, 2, 2)
        return time.time() - t0
    run(5, 64, 64)                                   # warm up the thread pool / mkldnn primitives
    probe = run(5, 128, 128)
    per_pxf = probe / (5 * 128 * 128)
    side = int(min(720, max(128, (budget_s / per_pxf / 5) ** 0.5)) // 8 * 8)
    hh, ww = side, min(1280, side * 16 // 9 // 8 * 8)
    dt = run(5, hh, ww)
    per_pxf = dt / (5 * hh * ww)
    fps = ONE_LEN / ((ONE_LEN + 4) * H * W * per_pxf)
    c1 = run(5, 256, 256)          # BASELINE config 1 in full: Shift-Net-s, fp32, one clip of T=5 256x256 (1 restored frame)
    return {"value": fps, "unit": "restored frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 forward of {VARIANT} on T_in=5 {ww}x{hh} ({dt:.1f} s), cost/pixel/frame extrapolated "
                      f"linearly to T_in=20 1280x720 (torch {torch.__version__}, {cores} threads)",
            "config1_full": {"workload": "BASELINE config 1: Shift-Net-s fp32, 1 clip of T=5 256x256, past/future 2/2", "seconds": round(c1, 3),
                             "restored_frames_per_s": round(1.0 / c1, 4), "input_frames_per_s": round(5.0 / c1, 3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the small parity sample (rocprofv3 / PMC runs: only full windows in the trace)")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--one-len", type=int, default=ONE_LEN)
    ap.add_argument("--variant", default=VARIANT, choices=list(ELEMS),
                    help="default gshift_deblur2 = BASELINE config 2; the other variants are extra measurements")
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS), help="BASELINE.json config preset (default: 2)")
    ap.add_argument("--dtype", default=None, choices=list(DTYPES), help="module dtype (fp32 runs the fp32 engine)")
    ap.add_argument("--quadrants", action="store_true", help="denoise CLI tiling: 4 overlapping quadrants per step")
    ap.add_argument("--one-quadrant", action="store_true", help="profiling only: run the FIRST of the four quadrants per step (a quarter of a config-4 window; "
                                                                 "the line is marked and is not a throughput figure)")
    ap.add_argument("--fp32_exact", action="store_true", help="--dtype fp32: exact fp32 products (v_mfma_f32_16x16x4_f32) instead of the default bf16 hi + lo "
                                                               "split products on the bf16 matrix cores (both within 1e-4 of the reference)")
    ap.add_argument("--schedule", default=None, choices=["unit", "frame", "streams"],
                    help="GSTS launch order: unit-major, the frame-group wavefront on one stream (SURVEY 8 f2), or frame groups on concurrent HIP streams "
                         "(K4 of one group under phase 1 of another); default: the engine's")
    ap.add_argument("--frame-group", type=int, default=None, help="--schedule frame: frames per group (default 4)")
    ap.add_argument("--stream-groups", type=int, default=None, help="--schedule streams: frame groups = HIP streams (default 2)")
    ap.add_argument("--halo", default="auto", choices=["auto", "p2p", "allgather"],
                    help="N > 1: form of the halo exchange -- point to point to the two neighbour ranks, the all-gather of every rank's edge frames, or "
                         "(default) point to point with an automatic, logged fall-back to the all-gather form if the first exchange fails")
    ap.add_argument("--lib", default=None, help="A/B measurements only: another build of libshiftnet_hip.so (the line then carries its path)")
    args = ap.parse_args()
    if args.lib:
        from shiftnet_amd import lib as _L
        _L.LIB_PATH = os.path.abspath(args.lib)
    if args.config is not None:
        c = CONFIGS[args.config]
        args.variant, args.height, args.width, args.one_len = c["variant"], c["height"], c["width"], c["one_len"]
        args.quadrants = c["quadrants"]
        args.dtype = args.dtype or c["dtype"]
    args.dtype = args.dtype or "bf16"
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if torch.cuda.device_count() < args.gpus and os.environ.get("SN_BENCH_SHARED_DEVICE_TEST") != "1":
            sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} HIP devices on this node, {torch.cuda.device_count()} visible: "
                     "refusing to report a multi-GPU number from fewer GPUs")
        # self-launch: one rank per GPU under torch.distributed.run (what the driver does explicitly for N > 1)
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 "(or run plain `python bench.py --gpus N`, which launches the ranks itself)")
    ndev = torch.cuda.device_count()
    # Rank-flow self test (tests/test_gpu_multirank.py on the one-GPU box): every rank on device 0, gloo transport.  The line it prints
    # is marked "shared_device_test" and is NOT a multi-GPU measurement; without this variable fewer devices than ranks is fatal.
    shared = os.environ.get("SN_BENCH_SHARED_DEVICE_TEST") == "1"
    if shared:
        local = 0
    if ndev < (1 if shared else world) or local >= ndev:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {world} HIP devices on this node, {ndev} visible: refusing to report a "
                 f"{world}-GPU number from fewer GPUs")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and shared:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # every rank must sit on its own physical GPU: gather (device index, PCI bus id) and compare
        pr = torch.cuda.get_device_properties(local)
        me = torch.tensor([local, int(getattr(pr, "pci_bus_id", local)), int(getattr(pr, "pci_domain_id", 0))], device=dev, dtype=torch.int64)
        allv = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(allv, me)
        ids = {tuple(v.tolist()) for v in allv}
        if (len(ids) != world and not shared) or dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: {world} ranks share {len(ids)} device(s): not a {world}-GPU run")

    import importlib
    GShiftNet = importlib.import_module(f"basicsr.models.archs.{args.variant}").GShiftNet
    from shiftnet_amd import synth
    from shiftnet_amd.clip_parallel import Halo
    from shiftnet_amd.weights import synth_state_dict

    h, w, L = args.height, args.width, args.one_len
    dt = DTYPES[args.dtype]
    if args.fp32_exact:
        from shiftnet_amd.engine32 import Engine32
        Engine32.split_bf16 = False
    net = GShiftNet(future_frames=2, past_frames=2)
    net.load_state_dict(synth_state_dict(args.variant), strict=True)
    net = net.to(dt).to(dev).eval()
    denoise = "denoise" in args.variant
    if args.schedule or args.frame_group or args.stream_groups:
        e_ = net.prepare()
        e_.schedule = args.schedule or e_.schedule
        e_.frame_group = args.frame_group or e_.frame_group
        e_.stream_groups = args.stream_groups or e_.stream_groups

    # this rank's slice of one long synthetic clip: L owned frames (+ the clip edges on the first / last rank)
    blur, _ = synth.blurred_clip(L + 4, h, w, seed=100 + rank)
    fr = (torch.from_numpy(blur).permute(0, 3, 1, 2).to(dev).to(dt) / 255).contiguous()
    own, first_edge, last_edge = fr[2:2 + L].contiguous(), fr[:2].contiguous(), fr[-2:].contiguous()
    if args.quadrants:      # the denoise CLI's tiling (inference/test_denoise.py:153-173): 4 overlapping quadrants per window
        pad_h, pad = 32 - (h // 2 % 16), 32 - (w // 2 % 16)
        hh, ww = h // 2 + pad_h, w // 2 + pad
        quads = [(0, hh, 0, ww), (0, hh, w // 2 - pad, w), (h // 2 - pad_h, h, 0, ww), (h // 2 - pad_h, h, w // 2 - pad, w)]
        assert all(b - a == hh and d - c == ww for a, b, c, d in quads), (h, w, hh, ww)
    else:
        hh, ww, quads = h, w, [(0, h, 0, w)]
    if args.one_quadrant:
        quads = quads[:1]
    sigma_map = torch.full((1, L + 4, 1, hh, ww), 30.0 / 255.0, dtype=dt, device=dev) if denoise else None

    gather_ms = []          # per step: time of the halo exchange (N > 1), measured with events on the stream it is issued on
    # The exchange of a window's raw edge frames depends on nothing the previous window computes: it is issued on a side stream, where it
    # overlaps the tail of the previous window's kernels, and the compute stream waits for it only before the window's first launch.
    side = torch.cuda.Stream(dev) if world > 1 else None
    halo = Halo(args.halo, log=lambda m: log(f"rank {rank}: {m}"))
    assemble_window = halo.assemble

    def step(local_only=False, timed=False):
        # local_only: rank 0's extra per-kernel profiling step must not enter a collective the other ranks are not in
        if local_only or world == 1:
            win = fr if local_only else assemble_window(own, first_edge, last_edge, rank, world)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            main_stream = torch.cuda.current_stream(dev)
            with torch.cuda.stream(side):
                e0.record()
                win = assemble_window(own, first_edge, last_edge, rank, world)
                e1.record()
            main_stream.wait_stream(side)
            win.record_stream(main_stream)
            if timed:
                gather_ms.append((e0, e1))
        outs = []
        with net.guard_scope():      # the quadrants of a window share one range-guard check, as in cli.quadrant_forward (one forward: no difference)
            for a, b, c, d in quads:
                xq = win[:, :, a:b, c:d].contiguous() if args.quadrants else win
                outs.append(net(xq.unsqueeze(0), sigma_map) if denoise else net(xq.unsqueeze(0)))
        return outs if args.quadrants else outs[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model and clip ready, warming up")
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        barrier()
        log("warm-up done, timing")
        t0 = time.perf_counter()
        step_ev = []                                    # per-step GPU time: events on the launching stream around every step (the median is reported beside the mean)
        for _ in range(args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = step(timed=True)
            e1.record()
            step_ev.append((e0, e1))
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0          # this rank alone, before it waits for the others
        barrier()
        elapsed = time.perf_counter() - t0
    for o in (out if isinstance(out, list) else [out]):
        assert o.shape == (L, 3, hh, ww) and torch.isfinite(o.float()).all()
    per_rank = None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        g_ms = sum(a.elapsed_time(b) for a, b in gather_ms) / max(len(gather_ms), 1)
        mine = torch.tensor([own_elapsed / args.steps * 1e3, g_ms], device=dev, dtype=torch.float64)
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = {"ms_per_step": [round(v[0].item(), 3) for v in allv], "halo_exchange_ms": [round(v[1].item(), 3) for v in allv],
                    "halo_form": halo.form, "halo_fell_back_from_p2p": halo.fell_back,
                    "halo_note": "the exchange is issued on a side stream and overlaps the previous window's tail here; the CLI (cli.infer_clip_parallel) issues it "
                                 "on the compute stream after decoding: its rate per window is this ms_per_step + halo_exchange_ms"}

    result = None
    if rank == 0:
        log(f"timed {args.steps} steps in {elapsed:.3f} s; profiling one extra step with stream events")
        ms = elapsed / args.steps * 1e3
        fps = world * L * args.steps / elapsed
        step_ms = sorted(a.elapsed_time(b) for a, b in step_ev)
        ms_median = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
        # Host <-> device transfers of a window as the CLIs ship it (uint8 frames in, uint8 restored frames out; pinned host buffers), measured here,
        # NOT part of `value` ("excluded but reported")
        hin = torch.empty((L + 4, h, w, 3), dtype=torch.uint8).pin_memory()
        hout = torch.empty((L, h, w, 3), dtype=torch.uint8).pin_memory()
        din, dout = torch.empty_like(hin, device=dev), torch.empty_like(hout, device=dev)
        xfer = {}
        for nm, src, dst in (("h2d_uint8_in_ms", hin, din), ("d2h_uint8_out_ms", dout, hout)):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dst.copy_(src, non_blocking=True); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            xfer[nm] = round(sorted(ts)[2], 3)
        xfer["bytes_in"], xfer["bytes_out"] = hin.numel(), hout.numel()
        xfer["frames_per_s_including_transfers"] = round(world * L / ((ms + xfer["h2d_uint8_in_ms"] + xfer["d2h_uint8_out_ms"]) * 1e-3), 3)
        del hin, hout, din, dout
        # ---- per-kernel durations, live, from events on the launch stream (one extra, untimed step) -----------
        eng = net.prepare()
        eng.prof = []
        with torch.no_grad():
            step(local_only=True)
        torch.cuda.synchronize()
        agg = {}
        unit_ms, unit_bytes, n_cabs, chain_ms = 0.0, 0.0, 0, 0.0
        for fn, label, meta, e0, e1 in eng.prof:
            if fn == "gsts_chain":                      # not a kernel: wall time of one chain of Encoder_shift_blocks on the launching stream
                chain_ms += e0.elapsed_time(e1)
                continue
            if fn == "sn_gsts_shiftconv_mfma":          # K0 on the matrix cores: the same operator (and the same key in the PMC files) as the VALU kernel
                fn = "sn_gsts_shiftconv"
            key = fn
            if fn == "sn_conv2d":                       # one GPU kernel per (M-tiles, tile shape): key by template instance
                mt = -(-max(meta[5], 1) // 16) if meta[9] != 1 else -(-meta[5] * 4 // 16)
                key = f"sn_conv2d<mt{mt},{'8x32' if meta[7] == 1 else '4x16'}>"
            elif fn in ("sn_cab_stats", "sn_cab_fused"):
                key = f"{fn}<mt{-(-meta[4] // 16)}>"
            elif fn == "sn_cab_ca_lines":
                key = "sn_cab_ca"
            elif fn == "sn_ln_gemm_gate":
                key = f"{fn}<{'cab2' if meta[5] else 'cab1'}>"
            elif fn in ("sn_gsts_cab2_phase2", "sn_cab1_phase2"):
                key = "sn_cab_phase2"                    # one GPU kernel behind both entry points
            in_unit = bool(meta) and (meta[0] == "naf" or "unit" in meta)      # a kernel of a CAB2 / CAB1 of a GSTS unit
            if fn.startswith("sn32_"):
                key = f"{fn}<k{meta[6]}{'g' if 'unit' in meta and meta[6] > 1 and meta[4] == meta[5] else ''}>" if meta and meta[0] == "conv32" else fn
                if in_unit:
                    key += "@unit"
            a = agg.setdefault(key, {"ms": 0.0, "n": 0, "bytes": 0.0, "gsts": in_unit})
            d = e0.elapsed_time(e1)
            a["ms"] += d; a["n"] += 1; a["bytes"] += kernel_alg_bytes(fn, meta)
            if in_unit:
                unit_ms += d
                if meta[0] == "naf" and fn in ("sn_gsts_cab2_phase2", "sn_cab1_phase2"):      # one CAB finished: its fused-unit bytes = read x + write y
                    unit_bytes += 2 * meta[1] * meta[2] * meta[3] * meta[4] * 2
                    n_cabs += meta[1] / meta[6]               # a frame-group launch of the wavefront schedule is that fraction of a CAB
                elif "unit" in meta and label.endswith("weight]") and ".body." in label and meta[0] == "conv32" and meta[6] == 1 and meta[4] == meta[5]:
                    ui = meta.index("unit")                                                     # the fp32 CAB's last 1x1 (C -> C): same accounting, 4-byte elements
                    unit_bytes += 2 * meta[ui + 1] * meta[ui + 2] * meta[ui + 3] * meta[ui + 4] * 4
                    n_cabs += 1
        eng.prof = None
        kernel_sum_unit_ms = unit_ms
        streams = getattr(eng, "schedule", "unit") == "streams"
        if streams and chain_ms > 0:                    # concurrent streams: the kernels' own durations overlap, the chains' wall time is what the unit costs
            unit_ms = chain_ms
        # the dominant kernel BY GPU TEMPLATE: both phase-1 entry points run cab_phase1r_kernel, every sn_conv2d instance is
        # one of two conv templates; the per-entry-point rows stay in "kernels"
        def template_of(k):
            if k in ("sn_gsts_cab2_phase1", "sn_cab1_phase1"):
                return "cab_phase1r_kernel (sn_gsts_cab2_phase1 + sn_cab1_phase1)"
            if k.startswith("sn_conv2d<") or k.startswith("sn_cab_stats<") or k.startswith("sn_cab_fused<"):
                return "dense convs (conv3p_kernel + conv3_fast_kernel + conv_mfma_kernel, every instance)"
            return k
        groups = {}
        for k, v in agg.items():
            gk = groups.setdefault(template_of(k), {"ms": 0.0, "n": 0, "bytes": 0.0, "members": []})
            gk["ms"] += v["ms"]; gk["n"] += v["n"]; gk["bytes"] += v["bytes"]; gk["members"].append(k)
        total_kernel_ms = sum(v["ms"] for v in agg.values())
        dom = max(groups, key=lambda k: groups[k]["ms"])
        # HBM traffic from the committed PMC passes of THIS command line run with --no-parity --no-cpu-baseline (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate passes, tools/make_profiles_r04.sh -> tools/pmc_summary.py): bytes per WINDOW per kernel = sum over the kernel's
        # launches / windows in the trace, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.  The file
        # names the sources its kernels were compiled from (csrc_files: per translation unit + shared headers); after a change to any of them the
        # figure is reported as null until the passes are re-run.
        # A figure below 0.9 x the algorithmic bytes cannot be right either (every input is read at least once).
        pmc, pmc_note = None, "no PMC passes for this configuration under profiles/"
        preset = args.config if args.config is not None else (2 if (args.variant, h, w, L, len(quads)) == (VARIANT, H, W, ONE_LEN, 1) else None)
        pmc_file = PMC_FILES.get((preset, args.dtype)) if not args.fp32_exact else None
        if pmc_file:
            try:
                doc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                changed = pmc_sources_changed(doc)
                if changed:
                    pmc_note = f"profiles/{pmc_file} was measured on other sources of its kernels (changed since: {', '.join(changed)}): re-run tools/final_r06.sh"
                else:
                    pmc = {}
                    for sym, v in doc["kernels_per_window"].items():
                        k = symbol_key(sym)
                        if k is not None:
                            e = pmc.setdefault(k, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
                            e["FETCH_SIZE_KB"] += v.get("FETCH_SIZE_KB", 0.0); e["WRITE_SIZE_KB"] += v.get("WRITE_SIZE_KB", 0.0)
                    pmc_note = (f"profiles/{pmc_file}: 2 x FETCH_SIZE + WRITE_SIZE summed over the kernel's launches of one window "
                                f"({doc.get('windows_in_trace')} full windows in the trace, no parity sample, sources of its kernels unchanged: "
                                f"{', '.join(sorted(doc.get('csrc_files', {'csrc_hash': 0})))})")
            except Exception as e:                                      # noqa: BLE001
                pmc_note = f"PMC file unreadable: {e}"

        def window_gb(key):
            if pmc is None:
                return None
            if key.startswith("sn32_"):                                 # the fp32 engine's kernels: all of them together, under the unit / non-unit split
                return None
            v = pmc.get(key)
            return (2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024 / 1e9 if v else None

        def checked(traffic_gb, alg_gb, what):
            if traffic_gb is None:
                return None, pmc_note
            if traffic_gb < 0.9 * alg_gb:
                return None, f"PMC file stale or diluted: {what} measured {traffic_gb:.4f} GB < 0.9 x algorithmic {alg_gb:.4f} GB"
            return round(traffic_gb, 4), pmc_note
        G = groups[dom]
        dom_alg = G["bytes"] / G["n"] / 1e9
        dws = [window_gb(k) for k in G["members"]]
        dw = None if any(v is None for v in dws) else sum(dws)
        dom_traffic, dom_note = checked(None if dw is None else dw / G["n"], dom_alg, dom)
        n_units = max(int(round(n_cabs / 2)), 1)
        gw = [window_gb(k) for k, v in agg.items() if v["gsts"]]
        if pmc is not None and args.dtype == "fp32":      # fp32 engine: the unit's kernels are generic operators, told apart by launch order, not by symbol:
            gw = []                                         # the unit share of the window's traffic is not separable -> whole-net traffic only (below)
        unit_alg = unit_bytes / n_units / 1e9
        unit_traffic, unit_note = checked(None if (not gw or any(g is None for g in gw)) else sum(gw) / n_units, unit_alg, "GSTS kernels per unit")
        ach_dom = G["bytes"] / (G["ms"] * 1e-3) / 1e9
        ach_unit = unit_bytes / max(unit_ms, 1e-9) / 1e6
        kernels = {k: {"ms_total": round(v["ms"], 3), "launches": v["n"],
                       "gbps_algorithmic": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in agg.items()}
        sbytes = 4 if dt == torch.float32 else 2
        win_bytes = len(quads) * algorithmic_bytes_window(args.variant, hh, ww, L + 4, L, s=sbytes)
        dlabel = {"bf16": "bf16", "fp16": "fp16", "fp32": "fp32"}[args.dtype]
        result = {
            "metric": f"restored frames/sec at {w}x{h} T={L} {dlabel}", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "ms_per_step_median": round(ms_median, 3),
            "ms_per_step_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)], "host_transfers": xfer,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            # module dtype; fp16 / bf16 modules both compute with bf16 storage + fp32 accumulation
            "dtype": {"bf16": "bf16", "fp16": "fp16 module on bf16 storage",
                      "fp32": "f32 exact (fp32 products)" if args.fp32_exact else "f32 (bf16 hi+lo split products, fp32 accumulation)"}[args.dtype],
            "config": {"workload": f"{'Shift-Net-s' if args.variant.endswith('2') else 'Shift-Net+'} ({args.variant}), {w}x{h}, one_len={L} (T_in={L + 4}), "
                                   + ("as the denoise CLI's 4 quadrants of %dx%d, " % (ww, hh) if args.quadrants else "")
                                   + (f"module dtype {args.dtype}, " if args.dtype != "bf16" else "")
                                   + "one window per GPU, synthetic checkpoint", "parallelism": f"clip-parallel x{world}",
                       "baseline_config": args.config if args.config is not None else (2 if (args.variant, h, w, L) == (VARIANT, H, W, ONE_LEN) else None),
                       **({"ab_library": args.lib} if args.lib else {}),
                       **({"one_quadrant_only": "profiling run: a quarter of the window's work per step"} if args.one_quadrant else {}),
                       "gsts_schedule": {"unit": "unit-major, one stream", "frame": f"frame wavefront on one stream, groups of {net.prepare().frame_group}",
                                         "streams": f"{net.prepare().stream_groups} frame groups on concurrent HIP streams"}[getattr(net.prepare(), "schedule", "unit")]},
            # the roofline this path is graded on is the FUSED GSTS UNIT (channel_shift + CAB2 + CAB1: read x, write y per
            # CAB = 4 T C h w s bytes) over the time of every GSTS kernel; intermediates count zero bytes.
            "roofline": {"bound": "hbm", "scope": "fused GSTS unit (SURVEY.md 8d), all pyramid levels of one window", "achieved": round(ach_unit, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_unit / HBM_PEAK_GBS, 4), "traffic": unit_traffic,
                         "traffic_note": "GB per unit (average over the window's units): " + unit_note,
                         "algorithmic_gb_per_unit": round(unit_alg, 4), "avg_unit_ms": round(unit_ms / n_units, 4), "units": n_units,
                         # wall time of the window's chains of Encoder_shift_blocks (events on the launching stream around each chain) next to the sum of
                         # their kernels' own durations; with the streams schedule the kernels overlap and `achieved` is taken over the wall time
                         "gsts_chain_wall_ms": round(chain_ms, 3), "gsts_kernel_sum_ms": round(kernel_sum_unit_ms, 3),
                         "time_base": "chain wall time (concurrent streams)" if (streams and chain_ms > 0) else "sum of the GSTS kernels' durations (one stream)",
                         # what the two-phase structure could reach at the measured copy rate: CALayer2's global pool splits every CAB in two
                         # passes, K0 writes hw: 11.5 C bytes per pixel and unit against the model's 4 C, x 6.3 / 8 TB/s
                         "ceiling_frac": round(4.0 / 11.5 * 6300.0 / HBM_PEAK_GBS, 4),
                         "ceiling_note": "bf16 two-phase CAB structure: 11.5 C bytes per pixel-unit floor vs 4 C algorithmic, at the 6.3 TB/s copy rate"},
            "dominant_kernel": {"kernel": dom, "bound": "hbm", "achieved": round(ach_dom, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(ach_dom / HBM_PEAK_GBS, 4), "traffic": dom_traffic, "traffic_note": "GB per launch: " + dom_note,
                                "algorithmic_gb_per_launch": round(dom_alg, 4),
                                "avg_launch_ms": round(G["ms"] / G["n"], 4), "launches": G["n"], "ms_per_window": round(G["ms"], 3),
                                "share_of_kernel_time": round(G["ms"] / max(total_kernel_ms, 1e-9), 4),
                                "by_template": {k: {"ms_per_window": round(v["ms"], 3), "share": round(v["ms"] / max(total_kernel_ms, 1e-9), 4),
                                                    "gbps_algorithmic": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                                                for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:6]},
                                "note": "kernel-local figure of the GPU template with the largest share of the window (averaged over its launches at all pyramid "
                                        "levels): its inputs / outputs include intermediates that the fused-unit model counts as zero bytes"},
            "whole_net_roofline": {"achieved": round(win_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(win_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_gb_per_window": round(win_bytes / 1e9, 1),
                                   "traffic": (round(sum((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) for v in pmc.values()) * 1024 / 1e9, 1) if pmc else None),
                                   "traffic_note": "GB per window, all of this library's kernels: " + pmc_note},
            "kernels": kernels,
            # high-water mark of this rank's device allocations over warm-up, timed steps and the profiling step (PyTorch caching allocator: every
            # activation of a window is a torch tensor): how much of the 288 GB a window of this configuration occupies
            "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
        }
        if per_rank is not None:
            result["per_rank"] = per_rank
        if world > 1 and os.environ.get("SN_BENCH_SHARED_DEVICE_TEST") == "1":
            result["shared_device_test"] = True        # all ranks on one GPU: a rank-flow test, not a measurement
        if world == 1 and args.variant == VARIANT and args.dtype == "bf16" and not args.no_parity:
            # parity sample next to the throughput: the same module on a small clip vs the CPU oracle (checker only)
            from oracle import shiftnet_oracle as O
            from shiftnet_amd import cli as CLI
            sd = synth_state_dict(VARIANT)
            blur_s, sharp_s = synth.blurred_clip(7, 96, 128, seed=4)
            xs = O.frames_to_tensor(list(blur_s))
            with torch.no_grad():
                o_hip = net(xs.to(torch.bfloat16).to(dev)).float().cpu()
                o_ref = O.forward(O.VARIANTS[VARIANT], sd, xs, None, 2, 2)
            gt = torch.from_numpy(sharp_s[2:5]).permute(0, 3, 1, 2).float() / 255

            def psnr(a, b):
                m = (a - b).pow(2).mean().item()
                return 99.0 if m == 0 else 10 * __import__("math").log10(1.0 / m)
            par = {"sample": "Shift-Net-s, 7x96x128 synthetic clip, 3 restored frames, bf16 module vs CPU fp32 oracle (the reference restated)",
                   "psnr_hip_vs_fp32_db": round(psnr(o_hip, o_ref), 2), "max_abs": round((o_hip - o_ref).abs().max().item(), 5),
                   # the module's bf16 OUTPUT TENSOR against the fp32 reference: includes the 1/256 quantisation of a bf16 image tensor
                   "delta_psnr_vs_gt_bf16_tensor_db": round(abs(psnr(o_hip.clamp(0, 1), gt) - psnr(o_ref.clamp(0, 1), gt)), 4)}
            # the CLI path (what test_deblur.py measures): PSNR from the fp32 accumulators of conv_last, no bf16 image tensor in between
            try:
                with torch.no_grad():
                    o32 = net.forward_fp32_out(xs.to(torch.bfloat16).to(dev), shortcut=xs.to(dev)).float().cpu()
                par["delta_psnr_vs_gt_db"] = round(abs(psnr(o32.clamp(0, 1), gt) - psnr(o_ref.clamp(0, 1), gt)), 4)
                par["psnr_fp32_out_vs_fp32_db"] = round(psnr(o32, o_ref), 2)
                par["tolerance"] = ("psnr_hip_vs_fp32_db >= 48 and delta_psnr_vs_gt_db <= 0.01 against the fp32 reference directly (restored frame "
                                    "taken from conv_last's fp32 accumulators and the un-rounded input, as the CLI does); asserted in "
                                    "tests/test_gpu_parity.py, tests/test_gpu_io.py")
            except AttributeError:
                par["tolerance"] = "psnr_hip_vs_fp32_db >= 48"
            result["parity"] = par
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on a bounded sample (child process, hard timeout)")
            result["cpu_baseline"] = cpu_baseline_subprocess()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
