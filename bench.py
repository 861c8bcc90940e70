#!/usr/bin/env python3
"""Headline benchmark: BT.709 NV12 -> sRGB BGRA decode throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|8k-half]

A "step" is one pass of the hot path over one batch of synthetic frames already
resident in HBM: the whole ring of `--ring` distinct frames (default 64 x 4K =
0.8 GB in + 2.1 GB out, far beyond the 256 MB Infinity Cache, so the kernel streams
from and to HBM), issued as ring/32 launches of 32 frames (grid.z = frame; 1080p: 128 per launch).
For N > 1 the driver starts one process per GPU (torch.distributed.run); every rank
owns a ring on its own GPU and decodes it with no data-path collective (frames are
independent); rank 0 prints ONE JSON line with the whole-job Gpixel/s.

Timing: barrier + stream sync, K steps, stream sync + barrier, MAX over ranks.
roofline.achieved comes from HIP events recorded on the launch stream around the
same K steps.  cpu_baseline (rank 0, N=1 only) times the reference's own per-pixel
function (oracle/_ref, kind "reference") or, when that library is absent, the CPU
oracle (kind "port") on a bounded sample of the same frames over the host cores.

--dry-run replaces the GPU work by a sleep so the multi-process control flow
(rendezvous, barriers, max over ranks, single JSON line) can be tested on CPU.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (width, height, half_scale, default ring, frames per launch)
    # The ring is one evenly spaced slab, so a launch may hold any number of frames (grid.z = frame);
    # ~1.5 GB of traffic per launch measured best (4K: 32 frames 1 % faster than 64; 1080p: 128 frames
    # 5 % faster than 32).
    "4k": (3840, 2160, False, 64, 32),
    "1080p": (1920, 1080, False, 256, 128),
    "8k-half": (7680, 4320, True, 16, 16),
}
GAMMAS = {"apple": 0, "srgb": 1, "linear": 2, "itu709": 3}
TRANSFER_TAG = {0: 1, 1: 2, 2: 3, 3: 1}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a copy reaches)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--ring", type=int, default=0, help="distinct frames resident per GPU (0 = workload default)")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="0 = workload default (the whole ring)")
    ap.add_argument("--gamma", default="apple", choices=sorted(GAMMAS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work for the baseline sample")
    ap.add_argument("--content", default="random", choices=["random", "smooth"],
                    help="random: uniform bytes (headline; worst case for the LDS table). smooth: video-like "
                         "low-frequency planes + small noise (neighbouring pixels share table buckets)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: sleep instead of decoding (control-flow test)")
    return ap.parse_args(argv)


def geometry(workload, ring_arg, max_batch, per_launch_arg=0):
    """Per-GPU plan of one step: ring size, frames per launch, launches, byte counts."""
    W, H, half, ring_default, per_launch = WORKLOADS[workload]
    per_launch = per_launch_arg or per_launch
    ring = ring_arg or ring_default
    per_launch = max(1, min(per_launch, ring, max_batch))
    ring -= ring % per_launch
    OW, OH = (W // 2, H // 2) if half else (W, H)
    return {
        "W": W, "H": H, "OW": OW, "OH": OH, "half": half, "ring": ring, "per_launch": per_launch,
        "launches": ring // per_launch,
        "y_bytes": W * H, "c_bytes": W * (H // 2), "o_bytes": OW * OH * 4,
        # algorithmic bytes: 1.5 B read per source pixel + 4 B written per output pixel
        "bytes_per_frame": W * H * 3 // 2 + OW * OH * 4,
    }


class GpuRunner:
    """Owns the per-rank context, decoder and resident ring; launches through the C ABI."""

    def __init__(self, args, g, rank, local_rank):
        import numpy as np
        import metalbt709decoder_amd as mb
        from metalbt709decoder_amd import _capi
        from metalbt709decoder_amd._capi import Frame, Surface
        self.np, self._capi, self.g = np, _capi, g
        gamma = GAMMAS[args.gamma]
        ndev = mb.load_library().bt709hip_device_count()
        if ndev <= 0:
            sys.exit("no HIP device: the product has no CPU fallback")
        # one rank per GPU; ranks beyond the device count (only in functional tests on a
        # smaller box) wrap around
        self.ctx = mb.MetalRenderContext(local_rank % ndev)
        if not self.ctx.setupMetal():
            sys.exit("HIP device %d could not be set up" % (local_rank % ndev))
        self.lib, self.h = self.ctx.lib, self.ctx.handle
        info = self.ctx.info()
        self.device, self.arch = info.name.decode(), info.arch.decode()
        self.props_gbps = 2 * info.memory_clock_khz * 1e3 * info.memory_bus_width_bits / 8 / 1e9
        self.dec = mb.MetalBT709Decoder()
        self.dec.metalRenderContext = self.ctx
        self.dec.gamma = gamma
        assert self.dec.setupMetal(), self.dec.lastStatus

        lib, h = self.lib, self.h
        ring, W, H, OW, OH = g["ring"], g["W"], g["H"], g["OW"], g["OH"]
        in_stride = (g["y_bytes"] + g["c_bytes"] + 255) // 256 * 256
        out_stride = (g["o_bytes"] + 255) // 256 * 256
        self.d_in, self.d_out = C.c_void_p(), C.c_void_p()
        _capi.check(lib.bt709hip_malloc(h, in_stride * ring, C.byref(self.d_in)), "malloc in")
        _capi.check(lib.bt709hip_malloc(h, out_stride * ring, C.byref(self.d_out)), "malloc out")
        self.host_frames = {}
        for i in range(ring):  # uploads are outside the timed region
            rng = np.random.default_rng(0x709 + i + 1000 * rank)
            if args.content == "random":  # full byte range: exercises saturation
                buf = rng.integers(0, 256, (1, g["y_bytes"] + g["c_bytes"]), dtype=np.uint8)
            else:
                buf = smooth_frame(np, rng, g, i).reshape(1, -1)
            _capi.check(lib.bt709hip_upload(h, self.d_in.value + i * in_stride, buf.shape[1], buf.ctypes.data,
                                            buf.shape[1], buf.shape[1], 1, None), "upload")
            _capi.check(lib.bt709hip_stream_synchronize(h, None))
            if i == 0:
                self.host_frames[0] = buf.reshape(-1)
        self.frames = (Frame * ring)()
        self.surfs = (Surface * ring)()
        for i in range(ring):
            base = self.d_in.value + i * in_stride
            self.frames[i] = Frame(base, W, base + g["y_bytes"], W, W, H, 1, TRANSFER_TAG[gamma])
            self.surfs[i] = Surface(self.d_out.value + i * out_stride, OW * 4, OW, OH)
        self.ev0, self.ev1 = C.c_void_p(), C.c_void_p()
        _capi.check(lib.bt709hip_event_create(h, C.byref(self.ev0)))
        _capi.check(lib.bt709hip_event_create(h, C.byref(self.ev1)))
        self.Frame, self.Surface = Frame, Surface
        # Untimed pre-warm: the device sits in a low-power state between jobs and needs
        # ~20-50 launches (tens of ms) before its clocks settle (measured: 308 -> 248 us per
        # launch, tools/launchprobe.py).  Done here, before the W warmup steps, so that a
        # small --warmup still times the settled kernel.
        t_end = time.perf_counter() + float(os.environ.get("BT709_BENCH_PREWARM_S", "0.4"))
        while time.perf_counter() < t_end:
            self.step()
            self.sync()

    def step(self):
        g, lib = self.g, self.lib
        n = g["per_launch"]
        for j in range(g["launches"]):
            fp = C.cast(C.byref(self.frames, j * n * C.sizeof(self.Frame)), C.POINTER(self.Frame))
            sp = C.cast(C.byref(self.surfs, j * n * C.sizeof(self.Surface)), C.POINTER(self.Surface))
            if g["half"]:
                rc = lib.bt709hip_decode_half_batch(self.dec._handle, n, fp, sp, None, 0)
            else:
                rc = lib.bt709hip_decode_batch(self.dec._handle, n, fp, None, sp, None, 0)
            if rc != 0:
                raise self._capi.Bt709Error(rc, "decode")

    def sync(self):
        self._capi.check(self.lib.bt709hip_stream_synchronize(self.h, None), "sync")

    def mark(self, which):
        self._capi.check(self.lib.bt709hip_event_record(self.h, self.ev1 if which else self.ev0, None))

    def event_ms(self):
        ms = C.c_float()
        self._capi.check(self.lib.bt709hip_event_elapsed_ms(self.h, self.ev0, self.ev1, C.byref(ms)))
        return ms.value

    def kernel_name(self):
        return self.lib.bt709hip_last_kernel_name().decode()

    def spot_check(self, gamma):
        """Untimed: the first 16 output rows of ring frame 0 against the oracle."""
        import oracle_lib
        np, g = self.np, self.g
        rows = 16
        got = np.empty((rows, g["OW"] * 4), np.uint8)
        self._capi.check(self.lib.bt709hip_download(self.h, got.ctypes.data, got.shape[1], self.surfs[0].bgra,
                                                    self.surfs[0].stride, got.shape[1], rows, None))
        self.sync()
        y, c = split_planes(self.host_frames[0], g)
        o = oracle_lib.Oracle()
        want = (o.decode_nv12_half(gamma, y[:2 * rows], c[:rows]) if g["half"]
                else o.decode_nv12(gamma, y[:rows], c[:rows // 2]))
        return "ok" if np.array_equal(got, want) else "MISMATCH"


class DryRunner:
    """CPU stand-in used only by --dry-run (tests of the N>1 control flow)."""
    device, arch, props_gbps, host_frames = "dry-run", "none", 0.0, {}

    def __init__(self):
        self.t = [0.0, 0.0]

    def step(self):
        time.sleep(0.002)

    def sync(self):
        pass

    def mark(self, which):
        self.t[which] = time.perf_counter()

    def event_ms(self):
        return (self.t[1] - self.t[0]) * 1e3

    def kernel_name(self):
        return "dry-run"


def smooth_frame(np, rng, g, i):
    """Video-like synthetic frame: low-frequency luma/chroma fields plus +-2 noise, video-legal range."""
    W, H = g["W"], g["H"]
    xx = np.arange(W, dtype=np.float32)[None, :]
    yy = np.arange(H, dtype=np.float32)[:, None]
    y = 126 + 95 * np.sin(xx / 97.0 + i) * np.cos(yy / 61.0 + 0.5 * i) + rng.integers(-2, 3, (H, W))
    cx, cy = xx[:, ::2], yy[::2]
    cb = 128 + 80 * np.sin(cx / 151.0 + 0.3 * i) * np.sin(cy / 83.0) + rng.integers(-2, 3, (H // 2, W // 2))
    cr = 128 + 80 * np.cos(cx / 131.0) * np.sin(cy / 113.0 + 0.7 * i) + rng.integers(-2, 3, (H // 2, W // 2))
    c = np.empty((H // 2, W), np.uint8)
    c[:, 0::2] = np.clip(cb, 16, 240).astype(np.uint8)
    c[:, 1::2] = np.clip(cr, 16, 240).astype(np.uint8)
    return np.concatenate([np.clip(y, 16, 235).astype(np.uint8).reshape(-1), c.reshape(-1)])


def split_planes(buf, g):
    y = buf[:g["y_bytes"]].reshape(g["H"], g["W"])
    c = buf[g["y_bytes"]:].reshape(g["H"] // 2, g["W"])
    return y, c


def main(argv=None):
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    if world > 1:
        # Control plane only (barrier + max of two scalars): gloo on CPU tensors.  The data path
        # has no exchange step -- frames are independent -- so no RCCL collective exists.
        # torch is imported BEFORE the product library so the process holds one HIP runtime.
        import torch  # noqa: F401
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    g = geometry(args.workload, args.ring, 65535, args.frames_per_launch)
    runner = DryRunner() if args.dry_run else GpuRunner(args, g, rank, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        runner.step()
    runner.sync()
    barrier()
    t0 = time.perf_counter()
    runner.mark(0)
    for _ in range(args.steps):
        runner.step()
    runner.mark(1)
    runner.sync()
    t1 = time.perf_counter()
    barrier()

    elapsed, ev_ms = t1 - t0, runner.event_ms()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, ev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(t[0]), float(t[1])

    out_px_per_step = g["ring"] * g["OW"] * g["OH"]
    value = world * args.steps * out_px_per_step / elapsed / 1e9
    bytes_per_launch = g["bytes_per_frame"] * g["per_launch"]
    avg_launch_s = (ev_ms / 1e3) / (args.steps * g["launches"])
    achieved = bytes_per_launch / avg_launch_s / 1e9
    read_gbps = (g["W"] * g["H"] * 3 // 2) * g["per_launch"] / avg_launch_s / 1e9

    result = {
        "metric": "Gpixel/s, %s NV12->sRGB BGRA decode (output pixels)" % args.workload,
        "value": round(value, 3),
        "unit": "Gpixel/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",  # fp32 arithmetic on u8 samples, exact-table transfer, u8 out
        "data": "synthetic",
        "config": {
            "workload": "%dx%d NV12 BT.709 -> %dx%d BGRA8 sRGB, gamma=%s%s; per GPU a ring of %d distinct frames "
                        "(%s, seed 0x709+i) resident in HBM; one step = the whole ring = "
                        "%d launches x %d frames"
                        % (g["W"], g["H"], g["OW"], g["OH"], args.gamma, ", fused 2:1 rescale" if g["half"] else "",
                           g["ring"], "uniform random bytes" if args.content == "random" else "smooth video-like planes",
                           g["launches"], g["per_launch"]),
            "frames_per_step_per_gpu": g["ring"],
            "sharding": "independent frames per GPU, no collective",
            "device": runner.device,
            "arch": runner.arch,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": load_traffic(args.workload),
            "kernel": runner.kernel_name(),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_us": round(avg_launch_s * 1e6, 3),
            "read_GBps": round(read_gbps, 1),
            "props_memclk_x2_x_buswidth_GBps": round(runner.props_gbps, 1),
        },
    }

    if rank == 0 and world == 1 and not args.dry_run:
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(runner.host_frames[0], g, GAMMAS[args.gamma], args.cpu_seconds)
        result["parity_spot_check"] = runner.spot_check(GAMMAS[args.gamma])
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(workload):
    """HBM bytes per launch from the PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from separate rocprofv3 --pmc runs); None if not collected."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]["hbm_bytes_per_launch"]
    except Exception:
        return None


def usable_cores():
    """Threads worth starting: affinity mask, capped by a cgroup CPU quota and by 64."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(frame0, g, gamma, target_seconds):
    """Bounded sample of the same workload on the host cores.  Checker code, timed only:
    this is the one place bench.py touches oracle/."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import oracle_lib
    W, half = g["W"], g["half"]
    kind, impl = "port", oracle_lib.Oracle()
    if not half:
        try:
            impl, kind = oracle_lib.Reference(), "reference"
        except Exception:
            pass
    y, c = split_planes(frame0, g)
    chunk = 128  # source rows decoded per call

    def run_chunk(out):
        if half:
            impl.decode_nv12_half(gamma, y[:chunk], c[:chunk // 2])
        else:
            impl.decode_nv12(gamma, y, c, rows=(0, chunk), out=out)

    cores = usable_cores()
    scratch = np.zeros((chunk, W * 4), np.uint8)
    run_chunk(scratch)  # page in
    t0 = time.perf_counter()
    run_chunk(scratch)
    one = (time.perf_counter() - t0) / (chunk * W)  # seconds per source pixel on one thread

    deadline = time.perf_counter() + target_seconds

    def work(_):
        mine = np.zeros((chunk, W * 4), np.uint8)
        done = 0
        while time.perf_counter() < deadline:
            run_chunk(mine)
            done += chunk
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        per_thread = list(ex.map(work, range(cores)))
    rows_done = sum(per_thread)
    chunks_per_thread = max(per_thread) // chunk
    dt = time.perf_counter() - t0
    src_px = rows_done * W
    out_px = src_px // 4 if half else src_px
    return {
        "value": round(out_px / dt / 1e9, 5),
        "unit": "Gpixel/s",
        "cores": cores,
        "kind": kind,
        "sample": "top %d rows of seeded ring frame 0 (%dx%d), decoded %d times by each of %d threads "
                  "(%.0f Mpx source in %.1f s)" % (chunk, g["W"], g["H"], chunks_per_thread, cores, src_px / 1e6, dt),
        "single_thread_value": round((0.25 if half else 1.0) * 1e-9 / one, 6),
    }


if __name__ == "__main__":
    main()
