"""The reference's inference CLIs (inference/test_{deblur,deblur_small,denoise,denoise_small}.py) on the HIP path.

Same flags, directory layout, window arithmetic, metric definitions and log line formats as upstream
(test_deblur.py:91-177,271-291 ; test_denoise.py:91-232,318-351), plus:
  --synthetic H W N   run on an in-memory synthetic clip (no dataset ships with the reference, none can be fetched here)
  --checkpoint PATH   override the checkpoint; 'synthetic' uses the deterministic synthetic weights
  --dtype {fp16,bf16,fp32}  dtype of the module (upstream: fp16 except the "+" denoiser, which stays float32): fp16 / bf16
                      modules run the bf16-storage MFMA kernels, fp32 modules the fp32 kernels (engine32.py)
  --gpus N            clip-parallel (deblur CLIs): N processes, one per GPU, take the windows of every clip N at a time (window k of a round on
                      rank k); a rank decodes only the one_len frames it restores and receives the 2 + 2 halo frames of its window from its
                      neighbours in ONE all-gather of raw uint8 frames (RCCL over xGMI; shiftnet_amd/clip_parallel.py); rank 0 writes the log,
                      whose lines equal the single-process run's.  Also honoured when launched by torchrun (WORLD_SIZE set).
  --host_io           convert uint8 <-> float on the host exactly like upstream (default: on the device, csrc/sn_io.hip:
                      same values bit for bit, 3 instead of 12 bytes per pixel over PCIe, PSNR and SSIM reduced on the GPU; the
                      denoise CLIs then also draw their AWGN on the device -- upstream draws it unseeded on the host,
                      test_denoise.py:145-147, so the realisation is not part of the contract -- and stitch the quadrants there)
The restored frames are taken as float32 straight from the last conv's fp32 accumulators (GShiftNet.forward_fp32_out): upstream
converts the half-precision module output with .float() before clamp * 255 / PSNR / imwrite (test_deblur.py:137-143), and a bf16
image tensor would quantise [0.5, 1] to 1/256 steps first.
Image I/O uses PIL (imageio / cv2 / skimage are not in this image); PSNR / SSIM restate the upstream formulas.
"""
from __future__ import annotations

import argparse
import glob
import math
import os
import sys
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import synth
from .arch import CLASSES
from .clip_parallel import Halo, assemble_window, rounds, window_ranges
from .io_edges import egress_u8, ingest_u8, ssim_u8
from .weights import synth_state_dict

DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


class TraverseLogger:
    def __init__(self, result_dir: str, filename: str, quiet: bool = False) -> None:
        self.path = os.path.join(result_dir, filename)
        self.f = None if quiet else open(self.path, "a" if os.path.exists(self.path) else "w")     # quiet: ranks > 0 of a --gpus run

    def write_log(self, log: str) -> None:
        if self.f is None:
            return
        print(log)
        self.f.write(log + "\n")
        self.f.flush()


def read_image(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def write_image(path: str, img_rgb_float: np.ndarray) -> None:
    from PIL import Image
    # cv2.imwrite converts float arrays with saturate_cast<uchar> = round to nearest (half to even), not truncation
    Image.fromarray(np.rint(np.clip(img_rgb_float, 0, 255)).astype(np.uint8)).save(path)


def psnr_255(img: np.ndarray, gt: np.ndarray) -> float:
    """skimage PSNR with data_range=255 on an un-rounded float image vs the uint8 GT (test_deblur.py:142)."""
    mse = float(np.mean((img.astype(np.float64) - gt.astype(np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 ** 2 / mse)


def ssim_calculate(img1: np.ndarray, img2: np.ndarray, sd: float = 1.5, c1: float = 0.01 ** 2, c2: float = 0.03 ** 2) -> float:
    """The CLI's own SSIM (test_deblur.py:25-49): Gaussian statistics over the (C,H,W) volume, inputs / 255."""
    from scipy.ndimage import gaussian_filter
    a = np.array(img1, dtype=np.float32).transpose(2, 0, 1) / 255
    b = np.array(img2, dtype=np.float32).transpose(2, 0, 1) / 255
    mu1, mu2 = gaussian_filter(a, sd), gaussian_filter(b, sd)
    s1 = gaussian_filter(a * a, sd) - mu1 * mu1
    s2 = gaussian_filter(b * b, sd) - mu2 * mu2
    s12 = gaussian_filter(a * b, sd) - mu1 * mu2
    return float(np.mean(((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))))


def numpy2tensor(frames: Sequence[np.ndarray]) -> torch.Tensor:
    """uint8 HWC frames -> [1,T,3,H,W] float32 in [0,1] (numpy2tensor, test_deblur.py:191-200)."""
    ts = [torch.from_numpy(np.ascontiguousarray(np.asarray(f).astype("float64").transpose(2, 0, 1))).float().mul_(1.0 / 255)
          for f in frames]
    return torch.stack(ts).unsqueeze(0)


def denoise_windows(n_frames: int) -> List[Tuple[int, int, int]]:
    """(first input frame, number of restored frames, residual appended) per window (test_denoise.py:111-133)."""
    one_len = n_frames - 4
    if one_len > 100:
        one_len //= 2
    k_len = (n_frames - 4) // one_len
    k_res = (n_frames - 4) % one_len
    return [(kk * one_len, one_len + (k_res if kk == k_len - 1 else 0), k_res if kk == k_len - 1 else 0) for kk in range(k_len)]


def quadrant_forward(net, x: torch.Tensor, sigma: float, on_device: bool = False, x32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The denoise CLI's 4 overlapping quadrants (test_denoise.py:153-173); x:[1,N,3,H,W] on device.  The stitched float32 frames are
    returned on the host like upstream's, or stay on the device (on_device) for the device-side metrics."""
    B, N, _, H, W = x.shape
    pad_h, pad = 32 - (H // 2 % 16), 32 - (W // 2 % 16)
    hh, ww = H // 2 + pad_h, W // 2 + pad
    std = torch.full((1, 1, 1, 1, 1), sigma, dtype=x.dtype, device=x.device).expand(B, N, 1, hh, ww)
    out = torch.zeros(N - 4, 3, H, W, device=x.device if on_device else "cpu")

    def run(ys, xs):        # x32: the un-rounded float32 frames for the final "+ x" of a half-precision module
        sc = x32[:, :, :, ys, xs].contiguous() if x32 is not None and x32.dtype != x.dtype else None
        o = net.forward_fp32_out(x[:, :, :, ys, xs].contiguous(), std, shortcut=sc)
        return o if on_device else o.cpu()
    for _ in range(2):      # one range-guard check for the four quadrants (no device sync between them); a tripped guard moved the module to the
        with net.guard_scope() as gs:      # bf16 chain: run them once more
            o1 = run(slice(0, hh), slice(0, ww))
            o2 = run(slice(0, hh), slice(W // 2 - pad, W))
            o3 = run(slice(H // 2 - pad_h, H), slice(0, ww))
            o4 = run(slice(H // 2 - pad_h, H), slice(W // 2 - pad, W))
        if not gs.tripped:
            break
    out[..., 0:H // 2, 0:W // 2] = o1[..., 0:-pad_h, 0:-pad]
    out[..., 0:H // 2, W // 2:] = o2[..., 0:-pad_h, pad:]
    out[..., H // 2:, 0:W // 2] = o3[..., pad_h:, 0:-pad]
    out[..., H // 2:, W // 2:] = o4[..., pad_h:, pad:]
    return out


class Inference:
    def __init__(self, args, variant: str) -> None:
        self.args = args
        self.variant = variant
        self.denoise = "denoise" in variant
        self.result_path = args.result_path
        os.makedirs(self.result_path, exist_ok=True)
        now = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())
        self.rank, self.world = getattr(args, "rank", 0), getattr(args, "world", 1)
        self.device = torch.device("cuda", getattr(args, "local_device", torch.cuda.current_device() if torch.cuda.is_available() else 0))
        self.logger = TraverseLogger(self.result_path, "inference_log_{}.txt".format(now), quiet=self.rank != 0)
        for k, v in (("Inference -", now), ("save_image:", args.save_image), ("border:", args.border), ("model_path:", args.model_path),
                     ("data_path:", args.data_path), ("result_path:", args.result_path), ("n_seq:", 4 if self.denoise else 5),
                     ("size_must_mode:", 4), ("device:", "cuda")):
            self.logger.write_log("{} {}".format(k, v))
        self.net = CLASSES[variant](future_frames=2, past_frames=2)
        if args.model_path == "synthetic":
            self.net.load_state_dict(synth_state_dict(variant), strict=True)
        else:
            self.net.load_state_dict(torch.load(args.model_path, map_location="cpu")["params"])
        self.dtype = DTYPES[args.dtype]
        self.net = self.net.to(self.dtype).to(self.device).eval()
        self.logger.write_log("Loading model from {}".format(args.model_path))

    # --- clip sources ---------------------------------------------------------------------------------------
    def videos(self):
        a = self.args
        if a.synthetic:
            h, w, n = a.synthetic
            blur, sharp = synth.blurred_clip(n, h, w, seed=0)
            yield "synthetic", (list(sharp) if self.denoise else list(blur)), list(sharp)
            return
        in_dir = a.data_path if self.denoise else os.path.join(a.data_path, "blur")
        for v in sorted(os.listdir(in_dir)):
            ins = sorted(glob.glob(os.path.join(in_dir, v, "*")))
            gts = ins if self.denoise else sorted(glob.glob(os.path.join(a.data_path, "gt", v, "*")))
            yield v, ins, gts

    @staticmethod
    def _load(items):
        return [read_image(p) if isinstance(p, str) else p for p in items]

    # --- main loop (test_deblur.py:91-177 / test_denoise.py:91-232) ---------------------------------------------
    def _summary(self, total_psnr, total_ssim) -> Tuple[float, float]:
        sp = ss = sp2 = ss2 = 0.0
        n = n2 = 0
        for k in total_psnr:
            self.logger.write_log("# Video:{} AVG-PSNR={:.5}, AVG-SSIM={:.4}".format(
                k, sum(total_psnr[k]) / len(total_psnr[k]), sum(total_ssim[k]) / len(total_ssim[k])))
            sp += sum(total_psnr[k]); ss += sum(total_ssim[k]); n += len(total_psnr[k])
            sp2 += sum(total_psnr[k]) / len(total_psnr[k]); ss2 += sum(total_ssim[k]) / len(total_ssim[k]); n2 += 1
        if n:
            self.logger.write_log("# Total AVG-PSNR={:.5}, AVG-SSIM={:.4}".format(sp / n, ss / n))
            if self.denoise:       # the denoise CLIs also log the mean of the per-video means (test_denoise.py:222-223)
                self.logger.write_log("# Total AVG-PSNR={:.5}, AVG-SSIM={:.4}".format(sp2 / n2, ss2 / n2))
        return (sp / n, ss / n) if n else (float("nan"), float("nan"))

    @torch.no_grad()
    def infer_clip_parallel(self) -> Tuple[float, float]:
        """The deblur main loop (test_deblur.py:91-177) with the windows of a clip taken `world` at a time, one per rank: a rank
        decodes the one_len frames it restores (+ the clip-side edge frames on the first / last active rank of a round), the 2 + 2 halo frames
        arrive from the two neighbour ranks as raw uint8 frames (clip_parallel.Halo: point to point, or -- `--halo allgather` / automatic fall-back --
        one all-gather), every rank restores and scores its window on its device, and rank 0 logs the gathered lines in window order -- the same
        lines (times aside) as the single-process run with the same one_len.  Idle ranks of a partial last round decode nothing and, once the
        exchange form is settled, take no part in a point-to-point exchange."""
        import torch.distributed as dist
        a, rank, world = self.args, self.rank, self.world
        on_host = dist.get_backend() == "gloo"                 # several ranks on one device (tests): the collective moves host tensors
        halo = Halo(getattr(a, "halo", "auto"), log=lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True))
        total_psnr, total_ssim = {}, {}
        for v, ins, gts in self.videos():
            vp, vs = [], []
            wins = window_ranges(len(ins), a.one_len)
            for rnd in rounds(len(wins), world):
                t0 = time.time()
                act = len(rnd)
                idle = rank >= act                                 # a partial last round: this rank holds no window
                mine = rnd[rank] if not idle else rnd[0]
                r_in, r_out = wins[mine]
                to = (lambda t: t) if on_host else (lambda t: t.to(self.device))
                if idle and halo.form == "p2p":                    # nothing to decode, nobody to talk to
                    win = None
                else:
                    # (an idle rank of the collective form contributes filler frames of the right shape: the first frame of the round, repeated)
                    own_np = self._load(ins[r_out.start:r_out.stop] if not idle else [ins[r_out.start]] * len(r_out))
                    h, w, _ = own_np[0].shape
                    nh, nw = h - h % 4, w - w % 4
                    crop = lambda ims: torch.from_numpy(np.stack([im[:nh, :nw] for im in ims])).permute(0, 3, 1, 2).contiguous()      # noqa: E731
                    own = to(crop(own_np))
                    first = to(crop(self._load(ins[r_in.start:r_in.start + 2]))) if rank == 0 else None
                    last = to(crop(self._load(ins[r_in.stop - 2:r_in.stop]))) if rank == act - 1 else None
                    win = halo.assemble(own, first, last, rank, world, active=act)      # uint8 [L+4,3,H,W]
                rec = None
                if win is not None:
                    u8 = win.to(self.device).permute(0, 2, 3, 1).contiguous()        # HWC frames, what ingest_u8 takes
                    gtf = [im[:nh, :nw] for im in self._load(gts[r_out.start:r_out.stop])]
                    x = ingest_u8(u8, self.dtype)
                    x32 = ingest_u8(u8, torch.float32) if self.dtype != torch.float32 else None
                    t1 = time.time()
                    output = self.net.forward_fp32_out(x, shortcut=x32)
                    torch.cuda.synchronize(self.device)
                    t2 = time.time()
                    gt_dev = torch.from_numpy(np.stack(gtf)).to(self.device)
                    img_u8, psnrs = egress_u8(output, gt_dev, want_image=a.save_image)
                    ssims = ssim_u8(output, gt_dev)
                    if a.save_image:
                        from PIL import Image
                        os.makedirs(os.path.join(self.result_path, v), exist_ok=True)
                        imgs = img_u8.cpu().numpy()
                        for e in range(len(r_out)):
                            Image.fromarray(imgs[e]).save(os.path.join(self.result_path, v, "%03d.png" % (r_out.start - 2 + e)))
                    name = os.path.basename(ins[r_in.start + 2]).split(".")[0] if isinstance(ins[r_in.start + 2], str) else "%05d" % (r_in.start + 2)
                    t3 = time.time()
                    rec = (mine, name, [float(p) for p in psnrs], [float(q) for q in ssims], t1 - t0, t2 - t1, t3 - t2, t3 - t0)
                    del output, x, x32
                    torch.cuda.empty_cache()
                recs = [None] * world
                dist.all_gather_object(recs, rec)
                for r in sorted((r for r in recs if r is not None), key=lambda r: r[0]):
                    _, name, ps, ss, pre, fwd, post, tot = r
                    vp += ps; vs += ss
                    self.logger.write_log(
                        "> {}-{} PSNR={:.5}, SSIM={:.4} pre_time:{:.3}s, forward_time:{:.3}s, post_time:{:.3}s, total_time:{:.3}s"
                        .format(v, name, ps[-1], ss[-1], pre, fwd, post, tot))
            if vp:
                total_psnr[v], total_ssim[v] = vp, vs
        return self._summary(total_psnr, total_ssim)

    @torch.no_grad()
    def infer(self) -> Tuple[float, float]:
        a = self.args
        if self.world > 1:
            return self.infer_clip_parallel()
        total_psnr, total_ssim = {}, {}
        for v, ins, gts in self.videos():
            vp, vs = [], []
            index = 0
            if self.denoise:
                wins = [(s, n, s + 2) for s, n, _ in denoise_windows(len(ins))]
            else:
                wins = [(r_in.start, len(r_out), r_out.start) for r_in, r_out in window_ranges(len(ins), a.one_len)]
            for start, n_out, gt0 in wins:
                t0 = time.time()
                inputs = self._load(ins[start:start + n_out + 4])
                gtf = self._load(gts[gt0:gt0 + n_out])
                h, w, _ = inputs[2].shape
                nh, nw = h - h % 4, w - w % 4
                inputs = [im[:nh, :nw] for im in inputs]
                gtf = [im[:nh, :nw] for im in gtf]
                name = os.path.basename(ins[start + 2]).split(".")[0] if isinstance(ins[start + 2], str) else "%05d" % (start + 2)
                dev_io = not a.host_io
                x32 = None
                if self.denoise and dev_io:     # uint8 frames up, float32 conversion and the AWGN on the device, quadrants stitched there
                    sigma = a.sigma / 255.0
                    x32 = ingest_u8(torch.from_numpy(np.stack(inputs)).to(self.device), torch.float32)
                    x32 = x32 + torch.empty_like(x32).normal_(mean=0, std=sigma)
                    x = x32.to(self.dtype)
                    t1 = time.time()
                    output = quadrant_forward(self.net, x, sigma, on_device=True, x32=x32)
                elif self.denoise:              # upstream's host path (test_denoise.py:145-173)
                    x32 = numpy2tensor(inputs)
                    sigma = a.sigma / 255.0
                    x32 = (x32 + torch.empty_like(x32).normal_(mean=0, std=sigma)).to(self.device)
                    x = x32.to(self.dtype)
                    t1 = time.time()
                    output = quadrant_forward(self.net, x, sigma, x32=x32)
                elif dev_io:
                    u8 = torch.from_numpy(np.stack(inputs)).to(self.device)
                    x = ingest_u8(u8, self.dtype)
                    x32 = ingest_u8(u8, torch.float32) if self.dtype != torch.float32 else None     # exact v / 255 for the final "+ x"
                    t1 = time.time()
                    output = self.net.forward_fp32_out(x, shortcut=x32)
                else:
                    x32 = numpy2tensor(inputs).to(self.device)
                    x = x32.to(self.dtype)
                    t1 = time.time()
                    output = self.net.forward_fp32_out(x, shortcut=x32 if self.dtype != torch.float32 else None)
                torch.cuda.synchronize()
                t2 = time.time()
                psnr = ssim = float("nan")
                img_u8 = psnrs = ssims = None
                if dev_io:      # clamp * 255, rounding, the PSNR sums and the SSIM statistics all on the device: no float frame goes back
                    gt_dev = torch.from_numpy(np.stack(gtf)).to(self.device)
                    img_u8, psnrs = egress_u8(output, gt_dev, want_image=a.save_image)
                    ssims = ssim_u8(output, gt_dev)
                    img_u8 = img_u8.cpu().numpy() if img_u8 is not None else None
                for e in range(n_out):
                    if dev_io:
                        psnr, ssim = psnrs[e], ssims[e]
                    else:
                        img = output[e].clamp(0, 1.0).permute(1, 2, 0).cpu().numpy() * 255
                        psnr, ssim = psnr_255(img, gtf[e]), ssim_calculate(img, gtf[e])
                    vp.append(psnr); vs.append(ssim)
                    if a.save_image:
                        os.makedirs(os.path.join(self.result_path, v), exist_ok=True)
                        if img_u8 is not None:
                            from PIL import Image
                            Image.fromarray(img_u8[e]).save(os.path.join(self.result_path, v, "%03d.png" % index))
                        else:
                            write_image(os.path.join(self.result_path, v, "%03d.png" % index), img)
                    index += 1
                t3 = time.time()
                del output, x, x32
                torch.cuda.empty_cache()
                self.logger.write_log(
                    "> {}-{} PSNR={:.5}, SSIM={:.4} pre_time:{:.3}s, forward_time:{:.3}s, post_time:{:.3}s, total_time:{:.3}s"
                    .format(v, name, psnr, ssim, t1 - t0, t2 - t1, t3 - t2, t3 - t0))
            if vp:
                total_psnr[v], total_ssim[v] = vp, vs
        return self._summary(total_psnr, total_ssim)


def main(variant: str, argv: Optional[Sequence[str]] = None) -> Tuple[float, float]:
    denoise = "denoise" in variant
    small = variant.endswith("2")
    ap = argparse.ArgumentParser(description="Inference")
    ap.add_argument("--save_image", action="store_true", default=False, help="save image if true")
    ap.add_argument("--border", action="store_true", help="restore border images of video if true")
    ap.add_argument("--default_data", type=str, default=".", help="quick test, optional: " + ("DAVIS, Set8" if denoise else "DVD, GOPRO"))
    if denoise:
        ap.add_argument("--sigma", type=int, default=10, help="sigma")
        ap.add_argument("--one", type=int, default=10, help="unused upstream")
    else:
        ap.add_argument("--one_len", type=int, default=96 if small else 48)
    ap.add_argument("--synthetic", type=int, nargs=3, metavar=("H", "W", "N"), default=None)
    ap.add_argument("--checkpoint", type=str, default=None)
    ap.add_argument("--dtype", choices=list(DTYPES), default="fp32" if variant == "gshift_denoise1" else "fp16")
    ap.add_argument("--result_path", type=str, default=None)
    ap.add_argument("--host_io", action="store_true", help="uint8<->float conversion and PSNR on the host, as upstream")
    ap.add_argument("--fp32_exact", action="store_true",
                    help="float32 modules: exact fp32 products on the fp32 matrix-core instructions instead of the default bf16 hi + lo split "
                         "products (~2^-16 per product; both within 1e-4 of the reference, the exact mode is about half as fast)")
    if not denoise:
        ap.add_argument("--gpus", type=int, default=1, help="clip-parallel: one process per GPU, the windows of a clip N at a time")
        ap.add_argument("--halo", choices=["auto", "p2p", "allgather"], default="auto",
                        help="clip-parallel halo exchange: point to point, all-gather, or point to point with an automatic fall-back (default)")
    a = ap.parse_args(argv)
    if a.fp32_exact:
        os.environ["SN_FP32_EXACT"] = "1"
    a.rank, a.world = 0, 1
    ngpus = getattr(a, "gpus", 1)
    if (ngpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1) and not denoise:
        # clip-parallel takes its halo from ONE neighbour window and keeps the frames on the device
        if a.one_len < 2:
            ap.error("--gpus N / a multi-rank launch needs --one_len >= 2 (a window's two halo frames come from one neighbour window)")
        if a.host_io:
            ap.error("--host_io is a single-process mode (upstream's host path); it cannot be combined with --gpus / a multi-rank launch")
    if ngpus > 1 and "RANK" not in os.environ:
        return _spawn_ranks(variant, ngpus, sys.argv[1:] if argv is None else argv)      # this process only launches and waits (it never touches a GPU); rank 0 prints the log
    # single process: the device an embedding process may have chosen; ranks: LOCAL_RANK, set below before anything touches a device
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 and not denoise
    a.local_device = 0 if multi else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if multi:
        import torch.distributed as dist
        a.rank, a.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        ndev = torch.cuda.device_count()
        a.local_device = int(os.environ.get("LOCAL_RANK", a.rank)) % max(ndev, 1)
        torch.cuda.set_device(a.local_device)
        # one GPU per rank -> RCCL; fewer devices than ranks (tests: two ranks on one device, which RCCL refuses) -> gloo with host staging
        backend = os.environ.get("SN_CLI_BACKEND", "nccl" if ndev >= a.world else "gloo")
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=a.rank, world_size=a.world)
    sfx = "_small" if small else ""
    a.data_path, a.model_path, rp = ".", "synthetic" if a.synthetic else "", "infer_results/synthetic"
    if denoise:
        if a.default_data in ("DAVIS", "Set8"):
            a.data_path = "./dataset/DAVIS-test" if a.default_data == "DAVIS" else "./dataset/Set8"
            a.model_path = "pretrained_models/net_denoise%s.pth" % sfx
            rp = "infer_results/%s%s/sigma%d" % (a.default_data, "_2" if small else "", a.sigma)
    else:
        if a.default_data == "DVD":
            a.data_path, a.model_path, rp = "./dataset/DVD/test", "pretrained_models/net_dvd_deblur%s.pth" % sfx, "infer_results/DVD"
        elif a.default_data == "GOPRO":
            a.data_path, a.model_path, rp = "./dataset/GOPRO/test", "pretrained_models/net_gopro_deblur%s.pth" % sfx, "infer_results/gopro"
    if a.checkpoint:
        a.model_path = a.checkpoint
    a.result_path = a.result_path or rp
    if not a.model_path:
        ap.error("choose --default_data, or --synthetic H W N, or give --checkpoint")
    import torch.distributed as dist
    try:
        res = Inference(a, variant).infer()
    except BaseException:
        # This rank failed while the others may sit in a collective: no barrier here (it would never complete and hide the error behind the
        # process-group timeout).  Leaving the group un-synchronised makes the launcher / _spawn_ranks see a non-zero exit at once.
        # Nor destroy_process_group(): on an RCCL group whose peers sit in a collective it can block itself.  Log and leave at once.
        if a.world > 1 and dist.is_initialized():
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        raise
    if a.world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return res


def _spawn_ranks(variant: str, n: int, argv: Sequence[str]) -> Tuple[float, float]:
    """--gpus N outside a launcher: start N ranks of THIS CLI (``python -m shiftnet_amd.cli <variant> <argv>``: independent of how the caller was
    started -- the drop-in script, pytest, bench.py), one rank each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment,
    rendezvous on 127.0.0.1), and wait for them.  As soon as one rank exits non-zero the others are stopped: they would otherwise wait in a
    collective until the process-group timeout.  The metrics are in rank 0's log."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "shiftnet_amd.cli", variant] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env["PYTHONPATH"] = os.pathsep.join([pkg_parent] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
        procs.append(subprocess.Popen(cmd, env=env))
    rc = [None] * n
    while any(c is None for c in rc):
        for i, p in enumerate(procs):
            if rc[i] is None:
                rc[i] = p.poll()
        if any(c not in (None, 0) for c in rc):
            for i, p in enumerate(procs):
                if rc[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rc[i] is None:
                    try:
                        rc[i] = p.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rc[i] = p.wait()
            break
        time.sleep(0.2)
    if any(rc):
        raise SystemExit("clip-parallel ranks exited with codes %s" % rc)
    return float("nan"), float("nan")


if __name__ == "__main__":          # python -m shiftnet_amd.cli <variant> [flags]: the rank entry point of _spawn_ranks
    main(sys.argv[1], sys.argv[2:])
