"""Harness counterpart of the reference's `inference_SPEINet.py`: same flags, data layout,
selection logic, uint8 round trip, PSNR/SSIM and log-line format — with the forward pass on the HIP path, an explicit
`--device`, presets that no longer clobber explicit paths, and clip sharding over ranks instead of DataParallel.

    <data_path>/{blur,gt}/<clip>/<frame>.png       [<data_path>/label/<clip>.npy  (0/1 per frame, 1 = sharp)]

Without a label file the LD detector (speinet_amd.detector, row a11) labels the frames on the GPU.
File I/O is off the critical path (the authors' logs show ~0.2 s of pre/post time per frame, a 5 frames/s cap): every
frame file is decoded once (windows overlap by n_sequence-1 frames) by a pool of prefetch threads that runs `--prefetch`
windows ahead of the GPU, and uint8 conversion, PSNR/SSIM and the PNG encode of a finished frame run on worker threads
while the GPU deblurs the next one; log lines are still written in frame order.
Reference: inference_SPEINet.py:193-237 (__init__), :338-429 (infer), :484-543 (metrics), :610-700 (flags / presets).
SSIM is restated with numpy (cv2 is absent): Gaussian 11x11, sigma 1.5, valid region — parity unpinned.
"""
from __future__ import annotations

import argparse
import collections
import glob
import os
import time
from concurrent.futures import Future, ThreadPoolExecutor

import numpy as np
import torch

from . import checkpoint, detector, ops, selection
from .dist import gather_metrics, shard_clips_by_length
from .speinet import EncoderCache, SPEINet, default_args


def _imread(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _imwrite(path: str, img: np.ndarray) -> None:
    """Lossless PNG at zlib level 1: the reference writes with cv2.imwrite's default, OpenCV's "best speed" PNG setting
    (inference_SPEINet.py:415-417); PIL's own default (level 6) costs twice the encode time for 10 % smaller files."""
    from PIL import Image
    Image.fromarray(img).save(path, compress_level=1)


def calc_ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    """inference_SPEINet.py:502-543: the reference averages the same 3-channel value three times.  The 11x11 Gaussian
    window is outer(k, k): applied as two 1-D passes (scipy.ndimage, float64, GIL released), valid region only."""
    from scipy.ndimage import correlate1d
    ax = np.arange(11) - 5
    k = np.exp(-(ax ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)

    def filt(x):   # valid region only ([5:-5, 5:-5] of the reference's same-size filter2D)
        y = correlate1d(correlate1d(x, k, axis=0, mode="constant"), k, axis=1, mode="constant")
        return y[5:-5, 5:-5]

    mu1, mu2 = filt(a), filt(b)
    s1, s2, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


_BANDS: dict = {}


def _gauss_band(n: int, device) -> torch.Tensor:
    """[n-10, n] float64 band matrix of the 11-tap Gaussian (sigma 1.5): valid-region filtering along one axis as a matmul."""
    key = (n, str(device))
    if key not in _BANDS:
        ax = torch.arange(11, dtype=torch.float64) - 5
        k = torch.exp(-(ax ** 2) / (2 * 1.5 ** 2))
        k /= k.sum()
        m = torch.zeros(n - 10, n, dtype=torch.float64)
        for i in range(n - 10):
            m[i, i:i + 11] = k
        _BANDS[key] = m.to(device)
    return _BANDS[key]


def metrics_gpu(out_u8: torch.Tensor, gt_u8: torch.Tensor):
    """PSNR and SSIM of two uint8 [H,W,3] frames on the GPU in float64 (same formulas as calc_psnr / calc_ssim; the
    separable Gaussian is two band-matrix products).  Returns two 0-d device tensors: no host sync here."""
    a, b = out_u8.to(torch.float64), gt_u8.to(torch.float64)
    mse = ((a - b) ** 2).mean()
    psnr = 20.0 * torch.log10(255.0 / torch.sqrt(mse))
    kh, kw = _gauss_band(a.shape[0], a.device), _gauss_band(a.shape[1], a.device)

    def filt(x):
        return torch.einsum("jw,iwc->ijc", kw, torch.einsum("ih,hwc->iwc", kh, x))

    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = filt(a), filt(b)
    s1, s2, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
    ssim = (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean()
    return psnr, ssim


class Logger:
    def __init__(self, result_dir: str, name: str, echo: bool = True):
        os.makedirs(result_dir, exist_ok=True)
        self.f = open(os.path.join(result_dir, name), "a")
        self.echo = echo

    def write_log(self, line: str) -> None:
        if self.echo:
            print(line)
        self.f.write(line + "\n")
        self.f.flush()


class FrameCache:
    """Decode-once, upload-once cache of frame files.  `request(paths)` schedules decodes on the thread pool, `get(path)`
    blocks until that file is decoded (host uint8 [H,W,3]); `get_dev(path)` returns the frame on the device, uploaded once
    however many windows share it (a frame is a neighbour twice, a middle frame once and often a reference).  Uploads go
    through a small ring of page-locked staging buffers allocated once: a copy from pageable memory makes the host wait for
    everything queued on the stream before it (one full GPU drain per window), and allocating page-locked memory per frame
    synchronises the device.  Bounded LRUs (a 720p RGB frame is 2.8 MB)."""
    RING = 8

    def __init__(self, pool: ThreadPoolExecutor, capacity: int = 96, device=None):
        self.pool, self.capacity, self.device = pool, capacity, device
        self.items: "collections.OrderedDict[str, Future]" = collections.OrderedDict()
        self.dev: "collections.OrderedDict[str, torch.Tensor]" = collections.OrderedDict()
        self._ring, self._events, self._n = [], [], 0

    def request(self, paths) -> None:
        for p in paths:
            if p in self.items:
                self.items.move_to_end(p)
            elif p not in self.dev:
                self.items[p] = self.pool.submit(_imread, p)
        while len(self.items) > self.capacity:
            self.items.popitem(last=False)

    def get(self, path: str) -> np.ndarray:
        self.request([path])
        return self.items[path].result()

    def _upload(self, arr: np.ndarray) -> torch.Tensor:
        n = arr.size
        if not self._ring or self._ring[0].numel() < n:
            self._ring = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(self.RING)]
            self._events = [None] * self.RING
        i = self._n % self.RING
        self._n += 1
        if self._events[i] is not None:
            self._events[i].synchronize()                 # the copy issued RING uploads ago: long finished
        stage = self._ring[i][:n].view(arr.shape)
        stage.numpy()[...] = arr
        t = stage.to(self.device, non_blocking=True)
        self._events[i] = torch.cuda.Event()
        self._events[i].record()
        return t

    def get_dev(self, path: str) -> torch.Tensor:
        t = self.dev.get(path)
        if t is None:
            self.request([path])
            t = self.dev[path] = self._upload(self.items.pop(path).result())
            while len(self.dev) > self.capacity:
                self.dev.popitem(last=False)
        else:
            self.dev.move_to_end(path)
        return t


class Inference:
    def __init__(self, args):
        self.args = args
        self.n_seq = args.n_sequence
        self.device = torch.device(args.device)
        if self.device.type == "cuda":
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.set_device(self.device)      # worker threads and torch helpers default to the current device
        for k in ("data_path", "result_path"):
            if not getattr(args, k, None):
                raise ValueError(f"--{k} is not set and --default_data {getattr(args, 'default_data', None)!r} has no preset for it")
        self.rank, self.world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        now = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime())
        self.logger = Logger(args.result_path, f"inference_log_{now}_rank{self.rank}.txt", echo=self.rank == 0)
        for k in ("save_image", "border", "model_path", "data_path", "result_path"):
            self.logger.write_log(f"{k}: {getattr(args, k)}")
        self.net = SPEINet(in_channels=3, n_sequence=self.n_seq, out_channels=3, n_resblock=3, n_feat=32, device=str(self.device), args=args)
        if args.model_path and args.model_path != "synthetic":
            checkpoint.load_into(self.net, args.model_path, strict=True)                                      # strict, like :232
        else:
            from .synth import state_dict_template, synth_state_dict
            self.net.load_state_dict(synth_state_dict(state_dict_template(), seed=0))
        self.net = self.net.to(self.device).eval()
        self.net.precision = args.precision
        self.net.corr_precision = {"f32": "bf16x3", "bf16x3": "bf16x3", "bf16": "top2", "f16": "top2"}[args.precision]
        self.net.use_graph = bool(getattr(args, "graph", True))      # one hipGraph per frame shape / routing
        self.net.streams = int(getattr(args, "streams", 1))
        workers = max(2, min(8, (os.cpu_count() or 4) // max(1, self.world)))
        self.io_pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="speinet-io")
        self.post_pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="speinet-post")
        self.prefetch = max(1, int(getattr(args, "prefetch", 4)))

    def labels_for(self, clip: str, frames: list) -> np.ndarray:
        p = os.path.join(self.args.data_path, "label", clip + ".npy")
        if os.path.exists(p):
            return np.load(p)                                   # allow_pickle=False: plain int arrays only
        imgs = torch.from_numpy(np.stack([_imread(f) for f in frames]).astype(np.float32)).permute(0, 3, 1, 2)
        feats = torch.cat([detector.focus_measures(imgs[i:i + 16].to(self.device), 11) for i in range(0, len(imgs), 16)])
        return detector.predict(feats)

    def _result_slot(self, shape):
        """One page-locked landing buffer per in-flight result: the frame (uint8 [H,W,3]) and [finite, PSNR, SSIM].  The main
        thread queues the two device->host copies behind the window's kernels and records an event; the worker waits on THAT
        event only.  (A `.item()` / `.cpu()` issued from the worker is a synchronous copy on the compute stream: it waits for
        every window queued after its own as well, four times per frame — the post pool spent its time blocked, 150 ms per
        frame, and throttled the loop to 36 ms per window on a GPU that needs 29.)  Allocated once per frame shape:
        allocating page-locked memory synchronises the device."""
        ring = self.__dict__.setdefault("_ring", {"shape": None, "slots": [], "n": 0})
        if ring["shape"] != tuple(shape):
            for sl in ring["slots"]:
                if sl["fut"] is not None:
                    sl["fut"].result()
            ring["shape"] = tuple(shape)
            ring["slots"] = [{"out": torch.empty(tuple(shape), dtype=torch.uint8, pin_memory=True),
                              "met": torch.empty(3, dtype=torch.float64, pin_memory=True), "fut": None} for _ in range(4 * self.prefetch)]
        sl = ring["slots"][ring["n"] % len(ring["slots"])]
        ring["n"] += 1
        if sl["fut"] is not None:
            sl["fut"].result()                      # the worker that last read this slot (long done: the ring is 2x the pending bound)
        return sl

    def _post(self, slot: dict, ready, save_to: str):
        """Worker thread: wait for the window's results to land in the slot, check them, encode the PNG."""
        t0 = time.time()
        ready.synchronize()
        finite, psnr, ssim = (float(v) for v in slot["met"].tolist())
        if not finite:
            return None, None, time.time() - t0         # an activation left the range of the 16-bit operand format: `flush` re-runs the window
        if save_to:
            _imwrite(save_to, slot["out"].numpy())
        return (float("inf") if psnr != psnr or psnr == float("inf") else psnr), ssim, time.time() - t0

    def _redo_window(self, x, zero_ref: bool, gt_u8, save_to: str):
        """One window again in split-bf16 arithmetic (fp32 exponent range), synchronously: the fallback for a frame whose half-precision
        pass produced a non-finite value.  Returns (PSNR, SSIM, seconds)."""
        t0 = time.time()
        keep = (self.net.precision, self.net.corr_precision, self.net.use_graph)
        self.net.precision, self.net.corr_precision, self.net.use_graph = "bf16x3", "bf16x3", False
        try:
            with torch.no_grad():
                out = self.net(x, routing=[zero_ref])
        finally:
            self.net.precision, self.net.corr_precision, self.net.use_graph = keep
        if not bool(torch.isfinite(out).all()):
            raise FloatingPointError(f"non-finite values in the deblurred frame {save_to or ''} in bf16x3 arithmetic as well: the input or the "
                                     "checkpoint is at fault")
        out_u8 = out.mul(255.0).clamp(0, 255).round()[0].to(torch.uint8).permute(1, 2, 0).contiguous()
        psnr, ssim = metrics_gpu(out_u8[4:-4, 4:-4], gt_u8[4:-4, 4:-4])
        if save_to:
            _imwrite(save_to, out_u8.cpu().numpy())
        psnr = float(psnr)
        return (float("inf") if psnr != psnr or psnr == float("inf") else psnr), float(ssim), time.time() - t0

    def infer(self):
        a = self.args
        self.range_retries = 0
        clips = sorted(os.listdir(os.path.join(a.data_path, "blur")))
        lengths = [len(glob.glob(os.path.join(a.data_path, "blur", c, "*"))) for c in clips]
        mine = shard_clips_by_length(lengths, self.world)[self.rank]
        stats = torch.zeros(3, dtype=torch.float64)             # sum psnr, sum ssim, frames
        cache = FrameCache(self.io_pool, device=self.device)
        with torch.no_grad():
            for ci in mine:
                t_clip = time.time()
                clip = clips[ci]
                blur = sorted(glob.glob(os.path.join(a.data_path, "blur", clip, "*")))
                gts = sorted(glob.glob(os.path.join(a.data_path, "gt", clip, "*")))
                wins = selection.assemble_windows(blur, self.labels_for(clip, blur), self.n_seq, a.border)
                gt_seqs, _ = selection.gene_seq(gts, self.n_seq, a.border)
                needs = [w["window"] + [w["pre"], w["sub"], g[self.n_seq // 2]] for w, g in zip(wins, gt_seqs)]
                if a.save_image:
                    os.makedirs(os.path.join(a.result_path, clip), exist_ok=True)
                vp, vs = [], []
                enc_cache = EncoderCache()                      # per-clip: encoder results of frames shared by overlapping windows
                pending = collections.deque()                   # (name, future, pre_time, forward_time, t_start), frame order

                def flush(block: bool):
                    while pending and (block or pending[0][1].done()):
                        name, fut, t_pre, t_fwd, t_start, redo = pending.popleft()
                        psnr, ssim, t_post = fut.result()
                        if psnr is None:
                            # half operands do not saturate (+-65504): a frame with a non-finite value is recomputed in split-bf16
                            # arithmetic (fp32 range, f32-grade) instead of aborting the clip; counted and logged
                            psnr, ssim, t_redo = self._redo_window(*redo)
                            t_post += t_redo
                            self.range_retries += 1
                            self.logger.write_log(f"# {clip}-{name}: non-finite value in the {a.precision} frame, recomputed in bf16x3 "
                                                  f"({self.range_retries} so far)")
                        vp.append(psnr)
                        vs.append(ssim)
                        self.logger.write_log('> {}-{} PSNR={:.5}, SSIM={:.4} pre_time:{:.3}s, forward_time:{:.3}s, post_time:{:.3}s, total_time:{:.3}s'
                                              .format(clip, name, psnr, ssim, t_pre, t_fwd, t_post, t_pre + t_fwd + t_post))

                inflight = collections.deque()                  # one event per enqueued window
                # windows alternate over a few launch streams (each replays its own captured instance of the fuse-and-decode graph): the
                # kernels of window k + 1 fill the tails and launch gaps of window k's, as bench.py's frames in flight do
                home = torch.cuda.current_stream(self.device)
                # (the home stream — where the inputs are assembled — is a lane only when it is the ONLY one: a lane waits for the home
                # stream before every window, and would wait for the window graphs queued there)
                nl = max(1, int(getattr(a, "lanes", 2)))
                lanes = [home] if nl == 1 else [torch.cuda.Stream(device=self.device) for _ in range(nl)]
                t_loop = time.time()
                def prepare(k):
                    """Inputs of window k on the device: (x, keys, gt, nh, nw, window, seconds spent)."""
                    t0 = time.time()
                    w = wins[k]
                    for ahead in needs[k:k + 1 + self.prefetch]:
                        cache.request(ahead)
                    imgs = [cache.get_dev(p) for p in needs[k]]            # uint8 [H,W,3] on the device, uploaded once per file
                    gt = imgs.pop()
                    h, wd = imgs[self.n_seq // 2].shape[:2]
                    nh, nw = h - h % 20, wd - wd % 20           # the model needs multiples of 20 (reference crops to 4)
                    imgs = [im[:nh, :nw] for im in imgs]
                    if w["zero_pre"]:
                        imgs[-2] = torch.zeros_like(imgs[-2])
                    if w["zero_sub"]:
                        imgs[-1] = torch.zeros_like(imgs[-1])
                    x = selection.uint8_frames_to_input(imgs)
                    keys = list(needs[k][:self.n_seq]) + [("zero", nh, nw) if w["zero_pre"] else w["pre"],
                                                          ("zero", nh, nw) if w["zero_sub"] else w["sub"]]
                    return x, keys, gt, nh, nw, w, time.time() - t0

                nxt = prepare(0) if wins else None
                for k in range(len(wins)):
                    # keep the host at most two windows ahead of the GPU: enough slack to hide its own work, and a worker that
                    # fetches a finished frame waits ~2 windows, not the whole queue (the post pool would otherwise spend its
                    # time blocked on results instead of encoding PNGs)
                    if len(inflight) >= max(2, len(lanes)):
                        inflight.popleft().synchronize()
                    x, keys, gt, nh, nw, w, t_prep = nxt
                    t0 = time.time()
                    # one window of lookahead: the NEXT window's inputs go to the device and its missing encoder passes start on the
                    # model's prefetch stream before this window's fuse-and-decode graph is queued, so they run underneath it
                    nxt = prepare(k + 1) if k + 1 < len(wins) else None
                    t1 = time.time()
                    gt_u8 = gt[:nh, :nw].contiguous()
                    lane = lanes[k % len(lanes)]
                    if lane is not home:
                        lane.wait_stream(home)                 # the window's inputs were assembled on the home stream
                    with torch.cuda.stream(lane):
                        for t_ in (x, gt_u8):
                            t_.record_stream(lane)
                        if a.reuse and nxt is not None:
                            nxt[0].record_stream(lane)
                            self.net.prefetch_window(nxt[0], nxt[1], enc_cache, zero_ref=bool(nxt[5]["zero_pre"]))
                        if a.reuse:
                            out = self.net.forward_window(x, keys, enc_cache, zero_ref=bool(w["zero_pre"]))
                        else:
                            out = self.net(x, routing=[bool(w["zero_pre"])])
                        # tensor2numpy, the finite check (half operands do not saturate: the worker looks at it), PSNR and SSIM on the
                        # 4-pixel-cropped frame (inference_SPEINet.py:405-410): three HIP launches (csrc/metrics.hip)
                        out_u8, met = ops.frame_post(out[0], gt_u8, 4)
                        slot = self._result_slot(out_u8.shape)
                        slot["out"].copy_(out_u8, non_blocking=True)
                        slot["met"].copy_(met, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                    inflight.append(ev)
                    t2 = time.time()
                    save_to = os.path.join(a.result_path, clip, w["name"] + ".png") if a.save_image else ""
                    slot["fut"] = self.post_pool.submit(self._post, slot, ev, save_to)
                    pending.append((w["name"], slot["fut"], t_prep, t2 - t1, t0, (x, bool(w["zero_pre"]), gt_u8, save_to)))   # pre_time = this window's own input preparation
                    flush(block=len(pending) > 2 * self.prefetch)
                t_drain = time.time()
                flush(block=True)
                self.logger.write_log("# timing {}: setup {:.3f}s, {} windows enqueued in {:.3f}s, drain {:.3f}s".format(
                    clip, t_loop - t_clip, len(wins), t_drain - t_loop, time.time() - t_drain))
                self.logger.write_log("# Video:{} AVG-PSNR={:.5}, AVG-SSIM={:.4}".format(clip, sum(vp) / len(vp), sum(vs) / len(vs)))
                stats += torch.tensor([sum(vp), sum(vs), float(len(vp))], dtype=torch.float64)
        dist = None
        if self.world > 1:
            import torch.distributed as dist
        allm = gather_metrics(stats.to(self.device) if self.world > 1 else stats, dist)
        tot = allm.sum(dim=0)
        if self.rank == 0 and tot[2] > 0:
            self.logger.write_log("# Total AVG-PSNR={:.5}, AVG-SSIM={:.4}".format(tot[0].item() / tot[2].item(), tot[1].item() / tot[2].item()))
        return tot


def synth_clip(root: str, n: int = 40, h: int = 720, w: int = 1280, seed: int = 5) -> str:
    """Write a synthetic n-frame clip in the reference's data layout (<root>/data/{blur,gt}/clip0/%06d.png + label/clip0.npy,
    every 6th frame labelled sharp) and return <root>/data: the input of `harness_throughput` and of tools/harness_bench.py."""
    from PIL import Image
    from .synth import synth_frames
    x = synth_frames(1, h, w, seed=seed)[0]
    for sub in ("blur", "gt"):
        os.makedirs(os.path.join(root, "data", sub, "clip0"), exist_ok=True)
    for i in range(n):
        img = (torch.roll(x[i % 5], shifts=(3 * i, -5 * i), dims=(1, 2)).permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
        for sub in ("blur", "gt"):
            Image.fromarray(img).save(os.path.join(root, "data", sub, "clip0", f"{i:06d}.png"), compress_level=1)
    os.makedirs(os.path.join(root, "data", "label"), exist_ok=True)
    np.save(os.path.join(root, "data", "label", "clip0.npy"), np.asarray([1 if i % 6 == 0 else 0 for i in range(n)]))
    return os.path.join(root, "data")


def harness_throughput(frames: int = 100, precision: str = "f16", h: int = 720, w: int = 1280, extra_args=()) -> dict:
    """End-to-end frames/s of this harness on a synthetic clip ON DISK: PNG decode -> selection -> upload -> forward (with
    cross-window encoder reuse) -> uint8 -> PSNR / SSIM -> PNG encode, everything the reference's loop does per frame
    (inference_SPEINet.py:364-429).  One untimed pass first (graph capture, page cache), then one timed pass."""
    import re
    import shutil
    import tempfile
    root = tempfile.mkdtemp(prefix="speinet_clip_")
    try:
        data = synth_clip(root, frames, h, w)
        a = build_args(["--data_path", data, "--model_path", "synthetic", "--result_path", os.path.join(root, "res"), "--precision", precision,
                        *extra_args])
        inf = Inference(a)
        inf.logger.echo = False
        inf.infer()
        torch.cuda.synchronize()
        t0 = time.time()
        tot = inf.infer()
        torch.cuda.synchronize()
        dt = time.time() - t0
        n = int(tot[2])
        lines = [ln for f in glob.glob(os.path.join(root, "res", "inference_log*")) for ln in open(f) if ln.startswith(">")][-n:]
        mean = lambda key: sum(float(re.search(key + r":([\d.e-]+)s", ln).group(1)) for ln in lines) / max(1, len(lines))
        timing = [ln.strip()[2:] for f in glob.glob(os.path.join(root, "res", "inference_log*")) for ln in open(f) if ln.startswith("# timing")][-1:]
        return {"value": n / dt, "unit": "frames/s", "frames": n, "seconds": dt, "precision": precision, "timing": timing,
                "mean_ms": {k: 1e3 * mean(k) for k in ("pre_time", "forward_time", "post_time")},
                "what": f"speinet_amd.inference on a synthetic {w}x{h} clip on disk, PNG decode/encode, PSNR and SSIM included, "
                        "cross-window encoder reuse on"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


PRESETS = {   # inference_SPEINet.py:626-697 (paths only fill in what the command line left at its default)
    "REDS": dict(data_path="./data/deblur/REDS_8x_Random/test", model_path="../experiment/model/model_best.pt", result_path="../infer_results/bsdtest_reds"),
    "GOPRO": dict(data_path="./data/deblur/GOPRO/test", model_path="../experiment/model/model_best.pt", result_path="../infer_results/gopro"),
    "BSD": dict(data_path="./data/deblur/BSDtest", model_path="./model/model_best.pt", result_path="../infer_results/BSDtest_finetune"),
    "BSDtest_all": dict(data_path="./data/deblur/BSDtest_all/BSD_3ms24ms", model_path="./model/model_best.pt", result_path="../infer_results/BSD_1ms8ms"),
}


def build_args(argv=None):
    p = argparse.ArgumentParser(description="SPEINet inference (MI355X)")
    p.add_argument("--save_image", action="store_true", default=True)
    p.add_argument("--no_save_image", dest="save_image", action="store_false")
    p.add_argument("--border", action="store_true", default=True)
    p.add_argument("--default_data", type=str, default="BSDtest_all")
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--model_path", type=str, default=None)
    p.add_argument("--result_path", type=str, default=None)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--precision", choices=["f32", "bf16x3", "bf16", "f16"], default="f32",
                   help="arithmetic of the GEMM-shaped kernels: f32 exact; bf16x3 f32-grade; f16 the throughput mode that holds "
                        "the 1e-3 dB PSNR bound; bf16 8-bit significands (speinet_amd/ops.py)")
    p.add_argument("--prefetch", type=int, default=4, help="windows decoded ahead of the GPU")
    p.add_argument("--no_reuse", dest="reuse", action="store_false", default=True,
                   help="recompute every encoder pass per window instead of reusing the per-frame results of overlapping windows")
    p.add_argument("--streams", type=int, default=1,
                   help="HIP streams for the independent branches of a frame (round 4: 1 — the frame's Swin calls run as one batch, the "
                        "second in-frame stream has nothing left to carry, and every extra stream competes for the 4 hardware queues)")
    p.add_argument("--lanes", type=int, default=2,
                   help="windows in flight: consecutive windows of a clip alternate over this many launch streams (2: +4.6 %% over 1 on a "
                        "100-frame 720p clip; 3 and 4 measured 8-25 %% SLOWER — three window graphs, the prefetch stream's encoder "
                        "passes and two correlation kernels at a time leave every kernel a third of the chip)")
    p.add_argument("--no_graph", dest="graph", action="store_false", default=True, help="launch kernels eagerly (no hipGraph replay)")
    p.add_argument("--n_GPUs", type=int, default=1,
                   help="ranks, one per GPU, clips sharded over them (the reference's preset attribute n_GPUs, inference_SPEINet.py:626-697, "
                        "there DataParallel); > 1 without RANK in the environment: this process starts the ranks and waits for them")
    a = p.parse_args(argv)
    for k, v in PRESETS.get(a.default_data, {}).items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    n_gpus = a.n_GPUs
    for k, v in vars(default_args()).items():
        setattr(a, k, v)
    a.n_GPUs = n_gpus
    return a


def main(argv=None):
    a = build_args(argv)
    if "RANK" not in os.environ and a.n_GPUs > 1:
        # one command for N GPUs: the parent (no GPU call) starts `python -m torch.distributed.run ... -m speinet_amd.inference <args>`
        import sys
        from .dist import launch_ranks
        raise SystemExit(launch_ranks("speinet_amd.inference", list(sys.argv[1:] if argv is None else argv), a.n_GPUs, module=True))
    if a.n_GPUs != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"--n_GPUs {a.n_GPUs} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
    if "RANK" in os.environ:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        a.device = f"cuda:{os.environ.get('LOCAL_RANK', '0')}"
        dist.init_process_group("nccl")
        try:
            Inference(a).infer()
        finally:
            dist.destroy_process_group()
        return
    Inference(a).infer()


if __name__ == "__main__":
    main()
