#!/usr/bin/env python3
"""Headline benchmark: BT.709 NV12 -> sRGB BGRA decode throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|8k-half|4k-batch8]

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM: the whole ring of
`--ring` distinct frames (default 256 x 4K = 3.2 GB in + 8.5 GB out, far beyond the 256 MB Infinity Cache, so the kernel
streams from and to HBM), issued as ONE bt709hip_decode_batch launch (grid.z = frame; the XCD-aware work map).  The ring is
the product's: bt709hip_ring_create allocates it and hunts for a fast-streaming placement (untimed set-up); the ring this
process allocated first, without a hunt, is timed too and reported as roofline.first_allocation_frac.

--gpus N > 1: one process per GPU.  Either the driver starts them (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE in
the environment) or -- a plain `python3 bench.py --gpus N` -- this script starts its own N ranks (self_launch) and relays rank
0's single JSON line.  Every rank owns a ring on its own GPU and decodes it with no data-path collective (frames are
independent); torch.distributed (gloo, CPU tensors) carries the barrier and the MAX over ranks only.

--workload 4k-batch8 is BASELINE config 5 as written: a step is 8 x 4K frames in total, frame i goes to rank i mod N, so a
rank decodes 8/N frames per step in one launch (N = 8: one 4K frame = one ~8 us kernel per step, launch-bound) -- strong
scaling.  --share M runs a single rank with the share of an 8/M-GPU job; --graph replays the K steps from one recorded HIP
graph; --coalesce n turns on the decoder's coalescing submit (one-frame calls gathered n to a launch).

Timing: W warmup steps, then the region of EXACTLY K steps -- barrier + stream sync, K steps, stream sync + barrier, MAX
over ranks -- is timed (`k_step_region_ms`).  With the driver's K = 20 that is ~36 ms of work, so the reported figure comes
from regions of m * K steps bracketed the same way, m the smallest integer that makes a region >= 100 ms (`region_steps`),
measured `repeats` times (>= 5); the MEDIAN is reported (min / max beside it) and `ms_per_step` is per step.
roofline.achieved comes from HIP events recorded on the launch stream around the same steps of the median region;
roofline.same_run_copy_GBps is a 16-byte-per-lane non-temporal copy over the same slabs, timed in the same process.  Every
region that counts is bracketed by two sentinel dispatches outside both clocks, so a rocprofv3 kernel trace of this command
can be cut down to the timed launches (tools/pmc_summary.py).  cpu_baseline (rank 0, N=1 only) times the reference's own
per-pixel function (oracle/_ref, kind "reference") or, when that library is absent, the CPU oracle (kind "port") on WHOLE
seeded frames: a 1080p and a 4K frame on one thread, then the workload's frame row-partitioned over the host cores.

--y4m FILE: the ring holds the frames of a YUV4MPEG2 clip (repeated to the ring length) instead of PRNG bytes: `data` = "y4m",
config.workload names the file's sha256 and geometry; the spot check compares the sampled ring frames with the oracle's decode of
the clip's own planes.

--dry-run replaces the GPU work by a sleep so the multi-process control flow (self-launch or rendezvous, barriers, max over
ranks, single JSON line) can be tested on CPU.
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
    # The ring is one evenly spaced slab, so a launch may hold any number of frames (grid.z = frame).  Rounds 1-2: ~1.5 GB
    # of traffic per launch measured best with the plain work map (4K: 32 frames; longer launches got slower: 64 / 128 /
    # 256 frames 0.75 / 0.71 / 0.70).  Round 3: with the XCD-aware work map (each XCD class a contiguous band of the launch's
    # frames, csrc/bt709_kernels.hip) a launch GAINS with its length -- no tail, no boundary -- so a step is ONE launch over
    # a ring of 256 distinct 4K frames (11.7 GB in + out; 1080p: 1024 frames): 0.77-0.81 against 0.75-0.77, same allocation.
    "4k": (3840, 2160, False, 256, 256),
    "1080p": (1920, 1080, False, 1024, 1024),
    "8k-half": (7680, 4320, True, 16, 16),
    # BASELINE config 5: 8 frames per step over ALL ranks; per_launch is replaced by the rank's share
    "4k-batch8": (3840, 2160, False, 64, 8),
}
BATCH8_FRAMES = 8
GAMMAS = {"apple": 0, "srgb": 1, "linear": 2, "itu709": 3}
TRANSFER_TAG = {0: 1, 1: 2, 2: 3, 3: 1}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a copy reaches)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--y4m", default=None, metavar="FILE",
                    help="fill the ring from this YUV4MPEG2 C420jpeg clip (the reference's on-disk format, Renderer/y4m_writer.h:61-241; "
                         "tools/make_y4m_clip.py writes one) instead of synthetic frames: geometry from the header, planar chroma interleaved on "
                         "the device (bt709hip_interleave_cbcr), the clip's frames repeated to the ring length; everything else as --workload 4k")
    ap.add_argument("--ring", type=int, default=0, help="distinct frames resident per GPU (0 = workload default)")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="0 = workload default (the whole ring)")
    ap.add_argument("--gamma", default="apple", choices=sorted(GAMMAS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work for the baseline sample")
    ap.add_argument("--content", default="random", choices=["random", "smooth", "flat"],
                    help="random: uniform bytes (headline; worst case for the LDS table). smooth: video-like "
                         "low-frequency planes + small noise (neighbouring pixels share table buckets). flat: one grey "
                         "value (every lane reads the same bucket: the LDS gathers at their conflict-free floor; lab)")
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step regions; 0 = auto (>= 5, >= 150 ms in total, <= 40)")
    ap.add_argument("--share", type=int, default=0,
                    help="4k-batch8 only: frames per step of THIS rank (default 8 / world size); lets one GPU "
                         "play a rank of a larger job")
    ap.add_argument("--graph", action="store_true", help="record the K steps into one HIP graph and replay it")
    ap.add_argument("--streams", type=int, default=0,
                    help="4k-batch8 only: issue consecutive steps round-robin on this many HIP streams (one stream "
                         "per in-flight frame); 0 = auto (2: measured best for every share, profiles/r02_batch8_streams*.txt)")
    ap.add_argument("--placement-tries", type=int, default=6,
                    help="bt709hip_ring_create's `tries`: candidates per slab of the ring, the fastest-streaming pairing kept (untimed "
                         "set-up); 1 = first allocation only")
    ap.add_argument("--hunt-max-gb", type=float, default=0.0,
                    help="bt709hip_ring_options.max_bytes of the placement hunt in GB (device memory the hunt may hold at once, ring "
                         "included); 0 = the library's default: "
                         "twice the ring, i.e. the incumbent pair + one candidate pair (the input candidates are capped by the budget too)")
    ap.add_argument("--hunt-max-ms", type=int, default=0, help="bt709hip_ring_options.max_ms: wall-clock budget of the hunt; 0 = none")
    ap.add_argument("--hunt-frugal", action="store_true", help="bt709hip_ring_options.frugal: the incumbent pair + one candidate pair only, whatever --hunt-max-gb says (the default since round 6)")
    ap.add_argument("--launcher", default="processes", choices=["processes", "threads"],
                    help="--gpus N > 1: `processes` = one process per GPU (the driver's torch.distributed.run line, or this script "
                         "starting its own ranks); `threads` = ONE process driving N GPUs through bt709hip_ringset_* (a ring per "
                         "device, one launch per device per step issued from one thread): the reference's one-process shape")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="BT709HIP_OPT_COALESCE: gather this many one-frame submits into one launch (4k-batch8; 0 = off)")
    ap.add_argument("--stream-priorities", default="", metavar="P1,P2,...",
                    help="4k-batch8: scheduling priority of the 2nd, 3rd, ... stream (0 normal, -1 higher, 1 lower); lab knob")
    ap.add_argument("--no-smooth-leg", action="store_true", help="skip the extra smooth-content measurement (N=1, 4k)")
    ap.add_argument("--decoder-option", action="append", default=[], metavar="ID=VALUE",
                    help="bt709hip_decoder_set_option(ID, VALUE) on the bench decoder (tuning sweeps)")
    ap.add_argument("--library", default=None,
                    help="load this build of libbt709hip.so (python -m metalbt709decoder_amd.build --variant ...) for A/B runs")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="let ranks wrap onto fewer GPUs than ranks (functional runs of the N > 1 path on a smaller box); the line then says "
                         "\"shared_devices\": true and n_gpus counts the DISTINCT devices.  Without it such a job is refused")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: sleep instead of decoding (control-flow test)")
    return ap.parse_args(argv)


INFINITY_CACHE_BYTES = 256 << 20  # MI355X memory-side cache


def geometry(workload, ring_arg, max_batch, per_launch_arg=0, world=1, share=0):
    """Per-GPU plan of one step: ring size, frames per launch, launches, byte counts.
    4k-batch8: a step is ONE launch of this rank's share of the 8 frames (8 / world, or --share);
    consecutive steps walk the ring so that the working set stays far beyond the Infinity Cache."""
    W, H, half, ring_default, per_launch = WORKLOADS[workload]
    batch8 = workload == "4k-batch8"
    if batch8:
        per_launch = share or max(1, BATCH8_FRAMES // world)
    per_launch = per_launch_arg or per_launch
    ring = ring_arg or ring_default
    per_launch = max(1, min(per_launch, ring, max_batch))
    ring -= ring % per_launch
    OW, OH = (W // 2, H // 2) if half else (W, H)
    return {
        "W": W, "H": H, "OW": OW, "OH": OH, "half": half, "ring": ring, "per_launch": per_launch,
        "batch8": batch8,
        # frames a step decodes on this rank / launches a step issues
        "frames_per_step": per_launch if batch8 else ring,
        "launches": 1 if batch8 else ring // per_launch,
        "y_bytes": W * H, "c_bytes": W * (H // 2), "o_bytes": OW * OH * 4,
        # algorithmic bytes: 1.5 B read per source pixel + 4 B written per output pixel
        "bytes_per_frame": W * H * 3 // 2 + OW * OH * 4,
        # the ring's INPUT against the 256 MB memory-side cache: a ring whose input fits it measures the cache, not HBM (two
        # side benches of this repo did for two rounds); main() refuses such a ring for a bench record
        "ring_input_over_cache": ring * (W * H * 3 // 2) / float(INFINITY_CACHE_BYTES),
    }


class GpuRunner:
    """Owns the per-rank context, decoder and resident ring; launches through the C ABI."""

    def __init__(self, args, g, rank, local_rank):
        import numpy as np
        import metalbt709decoder_amd as mb
        from metalbt709decoder_amd import _capi
        from metalbt709decoder_amd._capi import Frame, Surface, RingPlacement
        self.RingPlacement = RingPlacement
        self.np, self._capi, self.g, self.args, self.rank = np, _capi, g, args, rank
        gamma = GAMMAS[args.gamma]
        if args.library:
            _capi.load(os.path.abspath(args.library))
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
        # which physical device this rank drives: the line reports one record per rank (config.devices) and counts the DISTINCT ones
        self.identity = {"rank": rank, "ordinal": info.device_ordinal, "pci_bus_id": info.pci_bus_id.decode(), "uuid": info.uuid.decode()}
        self.arch = info.arch.decode()
        self.device = info.name.decode() or self.arch  # some ROCm builds leave the marketing name empty
        self.props = {"compute_units": info.compute_units, "memory_clock_khz": info.memory_clock_khz,
                      "memory_bus_width_bits": info.memory_bus_width_bits, "clock_khz": info.clock_khz}
        self.dec = mb.MetalBT709Decoder()
        self.dec.metalRenderContext = self.ctx
        self.dec.gamma = gamma
        for kv in args.decoder_option:
            k, v = kv.split("=")
            self.dec.setOption(int(k), int(v))
        assert self.dec.setupMetal(), self.dec.lastStatus
        if args.coalesce:
            self.dec.setOption(_capi.OPT_COALESCE, args.coalesce)

        lib, h = self.lib, self.h
        self.ev0, self.ev1, self.ev_fork = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _capi.check(lib.bt709hip_event_create(h, C.byref(self.ev0)))
        _capi.check(lib.bt709hip_event_create(h, C.byref(self.ev1)))
        _capi.check(lib.bt709hip_event_create(h, C.byref(self.ev_fork)))
        self.Frame, self.Surface = Frame, Surface
        self.stream = None    # launch stream: the context's default, or a created one when recording a graph
        self.extra_streams, self.join_events = [], []
        self.host_frames = {}
        self.scratch = C.c_void_p()  # sentinel dispatches (tools/pmc_summary.py finds the timed regions in a kernel trace by them)
        _capi.check(lib.bt709hip_malloc(h, 128 << 10, C.byref(self.scratch)))
        # THE RING IS THE PRODUCT'S: bt709hip_ring_create (include/bt709hip.h, csrc/bt709_ring.cpp) allocates the two slabs and,
        # with tries > 1, hunts for a fast-streaming input x output pairing with the decoder's own launch as the probe --
        # rounds 2-3 did that inside this file.  Two rings are made: `first` with tries = 1 = this process's FIRST ALLOCATION
        # (what a caller of plain bt709hip_malloc gets; measured after the headline and reported as
        # roofline.first_allocation_frac), then the hunted one the headline runs on.  Ranks that share a device (functional
        # runs on a smaller box) would hunt over each other's memory: one ring, no hunt, there.
        shared = int(os.environ.get("WORLD_SIZE", "1")) > ndev
        tries = 1 if shared else max(1, args.placement_tries)
        self.hunt_s = 0.0
        self.rings = {}
        self.rings["first"] = self.make_ring(1)
        self.use_ring("first")
        self.sample_frames = sample_frames(g["ring"])
        self.fill_ring(args.content)
        if tries > 1:
            self.rings["hunted"] = self.make_ring(tries)
            self.copy_ring_input("first", "hunted")
            self.use_ring("hunted")
        self.placement = self.placement_report()
        self.lanes, self.identities, self.placements = 1, [self.identity], [self.placement]
        self.pos = 0          # 4k-batch8: ring position of the next step
        self.graph = None
        self.turn = 0
        if args.graph:
            s = C.c_void_p()
            _capi.check(lib.bt709hip_stream_create(h, C.byref(s)), "stream create")
            self.stream = s.value
        # One HIP stream per in-flight frame (north-star): a step of 1-8 frames is a 8-63 us kernel, and on ONE
        # stream every launch boundary costs ~3.5 us of idle GPU; consecutive steps touch different frames, so
        # they may overlap on two streams (round 2: 1 frame per step 717 -> 935 Gpixel/s, 8 per step 1041 ->
        # 1064) -- on THREE when a step is a single frame (round 3: 1016; four or eight are worse again).  The 32-frame launches of the other workloads want ONE
        # stream (two interleave two DRAM address streams: -4 %).
        # --graph --streams N records the fork / join pattern itself: N parallel branches in ONE graph.
        # Round 3 (tools/ab_batch8.sh, same call): one frame per step 1 / 2 / 3 / 4 / 5 / 6 / 8 streams = 726 / 936 / 1016 / 895 / 979 / 970 /
        # 888 Gpixel/s; two frames per step 833 / 1077 / 1055 / 972; recorded graphs with 1-4 parallel branches 773 / 818 / 838 / 938.
        nstreams = args.streams or ((3 if g["per_launch"] == 1 else 2) if g["batch8"] and not args.graph else 1)
        self.nstreams = nstreams if g["batch8"] else 1
        prios = [int(v) for v in args.stream_priorities.split(",") if v.strip()]
        for i in range(self.nstreams - 1):
            s, e = C.c_void_p(), C.c_void_p()
            if i < len(prios):
                _capi.check(lib.bt709hip_stream_create_with_priority(h, prios[i], C.byref(s)), "stream create")
            else:
                _capi.check(lib.bt709hip_stream_create(h, C.byref(s)), "stream create")
            _capi.check(lib.bt709hip_event_create(h, C.byref(e)), "event create")
            self.extra_streams.append(s.value)
            self.join_events.append(e)
        # Untimed pre-warm: the device sits in a low-power state between jobs and needs
        # ~20-50 launches (tens of ms) before its clocks settle (measured: 308 -> 248 us per
        # launch, tools/launchprobe.py).  Done here, before the W warmup steps, so that a
        # small --warmup still times the settled kernel.
        t_end = time.perf_counter() + float(os.environ.get("BT709_BENCH_PREWARM_S", "0.4"))
        while time.perf_counter() < t_end:
            self.step()
            self.sync()

    def make_ring(self, tries):
        g = self.g
        r = C.c_void_p()
        t0 = time.perf_counter()
        opt = ring_options(self._capi, self.args)
        self._capi.check(self.lib.bt709hip_ring_create_ex(self.dec._handle, g["W"], g["H"], g["ring"], 1 if g["half"] else 0, tries,
                                                          C.byref(opt), C.byref(r)), "bt709hip_ring_create_ex")
        if tries > 1:
            self.hunt_s += time.perf_counter() - t0
        return r

    def use_ring(self, name):
        """Frame / surface descriptors of ring `name` as the arrays the launches take."""
        g, ring = self.g, self.rings[name]
        self.ring_name = name
        if getattr(self, "graph", None) is not None:  # a recorded graph holds the OTHER ring's pointers
            self.lib.bt709hip_graph_destroy(self.h, self.graph[1])
            self.graph = None
        self.frames = (self.Frame * g["ring"])()
        self.surfs = (self.Surface * g["ring"])()
        for i in range(g["ring"]):
            self._capi.check(self.lib.bt709hip_ring_frame(ring, i, C.byref(self.frames[i]), None, C.byref(self.surfs[i])))
        self.in_stride = self.frames[1].y - self.frames[0].y if g["ring"] > 1 else g["y_bytes"] + g["c_bytes"]
        self.out_stride = self.surfs[1].bgra - self.surfs[0].bgra if g["ring"] > 1 else g["o_bytes"]
        self.d_in, self.d_out = C.c_void_p(self.frames[0].y), C.c_void_p(self.surfs[0].bgra)

    def copy_ring_input(self, src, dst):
        """Device-to-device: the frames uploaded into ring `src` also become ring `dst`'s (same layout)."""
        f0, f1 = self.Frame(), self.Frame()
        self._capi.check(self.lib.bt709hip_ring_frame(self.rings[src], 0, C.byref(f0), None, None))
        self._capi.check(self.lib.bt709hip_ring_frame(self.rings[dst], 0, C.byref(f1), None, None))
        self._capi.check(self.lib.bt709hip_copy_probe(self.h, f1.y, f0.y, self.in_stride * self.g["ring"], None), "ring copy")
        self._capi.check(self.lib.bt709hip_stream_synchronize(self.h, None))

    def placement_report(self):
        """config.placement: what bt709hip_ring_create did for the ring the headline runs on (untimed set-up)."""
        return placement_dict(self._capi, self.lib, self.h, self.rings[self.ring_name], sorted(self.rings))

    def fill_ring(self, content):
        """Uploads (outside every timed region) the ring's frames: seeded PRNG bytes or smooth planes -- or a clip's (--y4m)."""
        if getattr(self.args, "clip", None):
            self.host_frames.update(upload_clip(self.np, self._capi, self.lib, self.h, self.d_in.value, self.in_stride, self.g, self.args.clip,
                                                self.sample_frames))
            return
        self.host_frames.update(upload_frames(self.np, self._capi, self.lib, self.h, self.d_in.value, self.in_stride, self.g, content,
                                              1000 * self.rank, self.sample_frames))

    def launch(self, first, n, stream=None):
        stream = stream if stream is not None else self.stream
        fp = C.cast(C.byref(self.frames, first * C.sizeof(self.Frame)), C.POINTER(self.Frame))
        sp = C.cast(C.byref(self.surfs, first * C.sizeof(self.Surface)), C.POINTER(self.Surface))
        if self.g["half"]:
            rc = self.lib.bt709hip_decode_half_batch(self.dec._handle, n, fp, None, sp, stream, 0)
        else:
            rc = self.lib.bt709hip_decode_batch(self.dec._handle, n, fp, None, sp, stream, 0)
        if rc != 0:
            raise self._capi.Bt709Error(rc, "decode")

    def step(self):
        g, n = self.g, self.g["per_launch"]
        if g["batch8"]:  # one launch of this rank's share, walking the ring (and the streams)
            lane = self.turn % self.nstreams
            self.launch(self.pos, n, self.extra_streams[lane - 1] if lane else None)
            self.turn += 1
            self.pos = (self.pos + n) % g["ring"]
            return
        for j in range(g["launches"]):
            self.launch(j * n, n)

    def run_steps(self, k):
        """K steps: issued one by one, or (--graph) recorded once and replayed with one launch."""
        if not self.args.graph:
            for _ in range(k):
                self.step()
            return
        if self.graph is None or self.graph[0] != k:
            if self.graph is not None:
                self.lib.bt709hip_graph_destroy(self.h, self.graph[1])
            g = C.c_void_p()
            self._capi.check(self.lib.bt709hip_graph_begin_capture(self.h, self.stream), "begin capture")
            if self.extra_streams:  # fork: the other streams join the recording behind an event of the origin stream
                self._capi.check(self.lib.bt709hip_event_record(self.h, self.ev_fork, self.stream))
                for s in self.extra_streams:
                    self._capi.check(self.lib.bt709hip_stream_wait_event(self.h, s, self.ev_fork))
            for _ in range(k):
                self.step()
            for s, e in zip(self.extra_streams, self.join_events):  # join: every branch ends in the origin stream
                self._capi.check(self.lib.bt709hip_event_record(self.h, e, s))
                self._capi.check(self.lib.bt709hip_stream_wait_event(self.h, self.stream, e))
            self._capi.check(self.lib.bt709hip_graph_end_capture(self.h, self.stream, C.byref(g)), "end capture")
            self.graph = (k, g)
        self._capi.check(self.lib.bt709hip_graph_launch(self.h, self.graph[1], self.stream), "graph launch")

    def sync(self):
        for s in self.extra_streams:
            self._capi.check(self.lib.bt709hip_stream_synchronize(self.h, s), "sync")
        self._capi.check(self.lib.bt709hip_stream_synchronize(self.h, self.stream), "sync")

    def mark(self, which):
        """Event on the launch stream; with several streams the closing event first joins the others
        (and the opening one is recorded with every stream idle: timed_region syncs before it)."""
        if which:
            for s, e in zip(self.extra_streams, self.join_events):
                self._capi.check(self.lib.bt709hip_event_record(self.h, e, s))
                self._capi.check(self.lib.bt709hip_stream_wait_event(self.h, self.stream, e))
        self._capi.check(self.lib.bt709hip_event_record(self.h, self.ev1 if which else self.ev0, self.stream))

    def event_ms(self):
        ms = C.c_float()
        self._capi.check(self.lib.bt709hip_event_elapsed_ms(self.h, self.ev0, self.ev1, C.byref(ms)))
        self.lane_ms = [ms.value]
        return ms.value

    def spot_checks(self, gamma):
        """[(verdict, frames checked)] per lane (this runner has one)."""
        verdict = self.spot_check(gamma)
        return [(verdict, getattr(self, "spot_frames", []))]

    def kernel_name(self):
        return self.lib.bt709hip_last_kernel_name().decode()

    def copy_ceiling(self, launches=24):
        """Same process, same slabs: a 16-byte-per-lane non-temporal copy of the lower half of the output
        slab onto its upper half (about the bytes of one decode launch), HIP-event timed per launch,
        median.  Returns GB/s of read + written bytes.  Overwrites decoded frames: call it last."""
        half = (self.out_stride * self.g["ring"] // 2) // 4096 * 4096
        src, dst = self.d_out.value, self.d_out.value + half
        times = []
        for i in range(launches + 4):
            self.mark(0)
            self._capi.check(self.lib.bt709hip_copy_probe(self.h, dst, src, half, self.stream), "copy probe")
            self.mark(1)
            self.sync()
            if i >= 4:
                times.append(self.event_ms())
        times.sort()
        return 2 * half / (times[len(times) // 2] / 1e3) / 1e9

    def sentinel(self, closing):
        """A tiny copy dispatch (512 lanes opening, 1024 closing) outside the timed bracket: tools/pmc_summary.py averages only the
        decode dispatches between an opening and a closing sentinel of a rocprofv3 kernel trace."""
        self._capi.check(self.lib.bt709hip_copy_probe(self.h, self.scratch.value + (64 << 10), self.scratch.value,
                                                      (32 << 10) if closing else (4 << 10), self.stream), "sentinel")

    def spot_check(self, gamma):
        """Untimed parity tripwire over the launch the headline times: 16 output rows at the top, middle and bottom of one frame
        out of EACH of the 8 XCD bands of the launch (ring frames 0, 37, 70, 103, 136, 169, 202, 255 of 256: the banded map gives
        band b the frames [b F/8, (b+1) F/8), csrc/bt709_kernels.hip) -- the last rows of the last frame included -- against the
        oracle.  The ring is decoded once more first, so the bytes compared are the ones the timed launch shape writes."""
        np, g = self.np, self.g
        rows = 16
        self.run_steps(1)
        if g["batch8"]:  # a step is a share of the ring there: decode every sampled frame's launch
            for i in self.sample_frames:
                self.launch(i - i % g["per_launch"], g["per_launch"])
        self.sync()
        verdict, self.spot_frames = compare_with_oracle(np, self._capi, self.lib, self.h, self.stream, g, gamma, self.sample_frames,
                                                        self.host_frames, lambda i: (self.surfs[i].bgra, self.surfs[i].stride), rows)
        return verdict


def upload_frames(np, _capi, lib, ctx, d_in, in_stride, g, content, seed_offset, keep):
    """The ring's frames into device memory at d_in + i * in_stride (blocking, untimed): seeded PRNG bytes (0x709 + i +
    seed_offset), smooth video-like planes or flat grey.  Returns {i: host bytes} for the frames in `keep` (the spot check's)."""
    kept, base_smooth = {}, None
    for i in range(g["ring"]):
        rng = np.random.default_rng(0x709 + i + seed_offset)
        if content == "random":  # full byte range: exercises saturation
            buf = rng.integers(0, 256, (1, g["y_bytes"] + g["c_bytes"]), dtype=np.uint8)
        elif content == "flat":
            buf = np.full((1, g["y_bytes"] + g["c_bytes"]), 128, np.uint8)
        else:
            if base_smooth is None:
                base_smooth = smooth_frame(np, rng, g, 0)
            # one synthesised frame, shifted by a different amount per ring entry (cheap, still distinct)
            y, c = split_planes(base_smooth, g)
            buf = np.concatenate([np.roll(y, (2 * i, 4 * i), (0, 1)).reshape(-1),
                                  np.roll(c, (i, 4 * i), (0, 1)).reshape(-1)]).reshape(1, -1)
        _capi.check(lib.bt709hip_upload(ctx, d_in + i * in_stride, buf.shape[1], buf.ctypes.data, buf.shape[1], buf.shape[1], 1, None), "upload")
        _capi.check(lib.bt709hip_stream_synchronize(ctx, None))
        if i in keep:
            kept[i] = buf.reshape(-1)
    return kept


def load_clip(path, max_frames):
    """A YUV4MPEG2 4:2:0 clip -> {"W", "H", "frames": [(y, u, v)], "sha256", "fps"} (host memory; at most max_frames of it)."""
    import hashlib
    from metalbt709decoder_amd.y4m import Y4MReader
    sha = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            sha.update(block)
    with Y4MReader(path) as r:
        frames = []
        for yuv in r:
            frames.append(yuv)
            if len(frames) >= max_frames:
                break
        if not frames:
            sys.exit("--y4m %s: no frames" % path)
        return {"W": r.width, "H": r.height, "frames": frames, "sha256": sha.hexdigest(), "fps": r.fps, "path": path}


def clip_ring_frames(W, H):
    """Ring length of a clip workload: the headline ring's pixel count (256 x 4K) in frames of this size, a multiple of 8."""
    return max(8, (256 * 3840 * 2160 // (W * H)) // 8 * 8)


def upload_clip(np, _capi, lib, ctx, d_in, in_stride, g, clip, keep):
    """The clip's frames into ring slots 0 .. n-1 -- Y uploaded in place, the planar U and V uploaded to a staging buffer and
    interleaved into the slot's CbCr plane ON THE DEVICE (bt709hip_interleave_cbcr: the reference's file layout -> NV12) -- then
    slots n .. ring-1 as device-to-device copies of slot i mod n.  Untimed.  Returns {i: host NV12 bytes} for the slots in `keep`."""
    W, H = g["W"], g["H"]
    cw, ch = W // 2, H // 2
    n = min(len(clip["frames"]), g["ring"])
    stage = C.c_void_p()
    _capi.check(lib.bt709hip_malloc(ctx, 2 * cw * ch + 256, C.byref(stage)), "staging")
    for i in range(n):
        y, u, v = (np.ascontiguousarray(p) for p in clip["frames"][i])
        slot = d_in + i * in_stride
        _capi.check(lib.bt709hip_upload(ctx, slot, W, y.ctypes.data, W, W, H, None), "upload Y")
        _capi.check(lib.bt709hip_upload(ctx, stage.value, cw, u.ctypes.data, cw, cw, ch, None), "upload U")
        _capi.check(lib.bt709hip_upload(ctx, stage.value + cw * ch, cw, v.ctypes.data, cw, cw, ch, None), "upload V")
        _capi.check(lib.bt709hip_interleave_cbcr(ctx, stage.value, cw, stage.value + cw * ch, cw, slot + g["y_bytes"], W, cw, ch, None, 1), "interleave")
    for i in range(n, g["ring"]):
        _capi.check(lib.bt709hip_copy_probe(ctx, d_in + i * in_stride, d_in + (i % n) * in_stride, in_stride, None), "ring copy")
    _capi.check(lib.bt709hip_stream_synchronize(ctx, None))
    _capi.check(lib.bt709hip_free(ctx, stage))
    kept = {}
    for i in keep:
        y, u, v = clip["frames"][i % n]
        c = np.empty((ch, W), np.uint8)
        c[:, 0::2], c[:, 1::2] = u, v
        kept[i] = np.concatenate([np.ascontiguousarray(y).reshape(-1), c.reshape(-1)])
    return kept


def compare_with_oracle(np, _capi, lib, ctx, stream, g, gamma, picks, host_frames, surface_of, rows=16):
    """`rows` output rows at the top, middle and bottom of every picked ring frame, downloaded and byte-compared with the CPU
    oracle's decode of the frame's host bytes.  -> ("ok" | "MISMATCH ...", frames checked).  Checker code: untimed."""
    import oracle_lib
    o = oracle_lib.Oracle()
    checked = []
    for i in picks:
        y, c = split_planes(host_frames[i], g)
        ptr, stride = surface_of(i)
        for r0 in (0, (g["OH"] // 2) // 4 * 4, g["OH"] - rows):
            got = np.empty((rows, g["OW"] * 4), np.uint8)
            _capi.check(lib.bt709hip_download(ctx, got.ctypes.data, got.shape[1], ptr + r0 * stride, stride, got.shape[1], rows, stream))
            _capi.check(lib.bt709hip_stream_synchronize(ctx, stream))
            if g["half"]:
                want = o.decode_nv12_half(gamma, y[2 * r0:2 * (r0 + rows)], c[r0:r0 + rows])
            else:
                want = o.decode_nv12(gamma, y, c, rows=(r0, r0 + rows))[r0:r0 + rows]
            if not np.array_equal(got, want):
                return "MISMATCH in ring frame %d at output row %d" % (i, r0), checked
        checked.append(i)
    return "ok", checked


def ring_options(_capi, args):
    return _capi.RingOptions(int(args.hunt_max_gb * 1e9), int(args.hunt_max_ms), int(bool(args.hunt_frugal)))


def placement_dict(_capi, lib, ctx, ring, rings_resident):
    """What bt709hip_ring_create[_ex] did for `ring` (untimed set-up): candidates, probes, the choice, what the hunt cost."""
    p = _capi.RingPlacement()
    _capi.check(lib.bt709hip_ring_placement_info(ring, C.byref(p)))
    free_b, total_b = C.c_size_t(), C.c_size_t()
    _capi.check(lib.bt709hip_mem_info(ctx, C.byref(free_b), C.byref(total_b)))
    kept = [k for k in p.out_kept if k >= 0]
    return {"allocator": "bt709hip_ring_create_ex (csrc/bt709_ring.cpp): candidates per slab under a byte / time budget, the decoder's own launch as the probe",
            "tries": p.tries, "candidates": [p.in_candidates, p.out_candidates], "chosen": [p.chosen_in, p.chosen_out],
            "pairings_probed": p.probes,
            "probe_GBps": {"first_pairing": round(p.first_GBps, 1), "chosen_confirmed": round(p.chosen_GBps, 1),
                           "best": round(p.best_GBps, 1), "worst": round(p.worst_GBps, 1)},
            # allocation order; `kept` = the indices (same order) that went on to the pairing probes
            "output_prescan_GBps": [round(v, 1) for v in p.out_prescan_GBps[:p.out_candidates]], "output_kept": kept,
            # the hunt's cost: duration, the most device memory it held at once (ring included), its budget, slabs freed early to
            # stay inside it, and what ended it (0 ran to its end, 1 the byte budget cut candidates, 2 the time budget)
            "hunt_ms": round(p.hunt_ms, 1), "peak_bytes": int(p.peak_bytes), "budget_bytes": int(p.budget_bytes),
            "evicted": p.evicted, "stopped_by": p.stopped_by,
            "rings_resident": rings_resident, "device_memory_free_GB": round(free_b.value / 1e9, 1),
            "device_memory_total_GB": round(total_b.value / 1e9, 1)}


def memory_plan(g, args, world=1, lanes=1):
    """What ONE rank (GPU) of this job will hold, worked out from the geometry alone (no GPU needed: --dry-run reports it too, and a
    CPU test checks the 8-rank plan against a 288 GB MI355X and a 16 GB host budget).  Device: the ring allocated first (tries = 1,
    kept for roofline.first_allocation_frac) + the placement hunt's budget for the hunted ring (bt709hip_ring_options: --hunt-max-gb,
    or the library's default of twice the ring) -- of which the ring it keeps is a part.  Host: the frames are generated ONE AT A
    TIME and uploaded (upload_frames), so a rank holds one frame being generated, its staging copy, and the `sample_frames` kept for
    the oracle spot check -- not the ring."""
    up = lambda v: (v + 255) // 256 * 256
    frame_in, frame_out = g["y_bytes"] + g["c_bytes"], g["o_bytes"]
    ring_bytes = g["ring"] * (up(frame_in) + up(frame_out))
    hunted = max(1, args.placement_tries) > 1 and ring_bytes >= (256 << 20)
    budget = int(args.hunt_max_gb * 1e9) if args.hunt_max_gb and not args.hunt_frugal else 2 * ring_bytes
    device_peak = ring_bytes + (max(budget, ring_bytes) if hunted else 0)
    kept = len(sample_frames(g["ring"]))
    host = (kept + 2) * frame_in + 3 * 16 * g["OW"] * 4
    clip = getattr(args, "clip", None)
    if clip:
        host += len(clip["frames"]) * frame_in
    ranks = world * lanes
    return {"ring_bytes": ring_bytes, "ring_in_bytes": g["ring"] * up(frame_in), "ring_out_bytes": g["ring"] * up(frame_out),
            "rings_resident_after_setup": 2 if hunted else 1, "hunt_budget_bytes": budget if hunted else 0,
            "device_peak_bytes_per_gpu": device_peak, "device_steady_bytes_per_gpu": ring_bytes * (2 if hunted else 1),
            "host_bytes_per_rank": host, "ranks": ranks, "host_bytes_all_ranks": host * ranks,
            "fits_288GB_per_gpu": device_peak <= 288 * 10**9 - (4 << 30)}


def sample_frames(ring):
    """One frame of each eighth of the ring (= each XCD band of a whole-ring launch), first and last frame included."""
    if ring < 8:
        return list(range(ring))
    per = ring // 8
    picks = {0, ring - 1}
    for b in range(1, 7):
        picks.add(b * per + (b + 4) % per)
    return sorted(picks)


class SetRunner:
    """--launcher threads: ONE process driving `lanes` GPUs through bt709hip_ringset_* -- a context, a decoder and a ring (placed
    by its own hunt) per device, one ring launch per device per step issued from this one thread, no thread per device and no
    collective.  The reference's shape (one process that drives everything, Renderer/AAPLRenderer.m:874-985) widened to a node.
    Same interface as GpuRunner; clocks, identity, placement and parity check exist per lane."""

    def __init__(self, args, g, lanes, ndev):
        import numpy as np
        import metalbt709decoder_amd as mb
        from metalbt709decoder_amd import _capi
        self.np, self._capi, self.g, self.args, self.lanes = np, _capi, g, args, lanes
        if g["batch8"] or args.graph or args.coalesce or args.streams:
            sys.exit("--launcher threads runs the whole-ring workloads (4k, 1080p, 8k-half) only")
        if args.library:
            _capi.load(os.path.abspath(args.library))
        lib = self.lib = mb.load_library()
        ordinals = [lane % ndev for lane in range(lanes)]  # beyond the device count only with --allow-shared-devices
        shared = lanes > ndev
        tries = 1 if shared else max(1, args.placement_tries)
        opt = ring_options(_capi, args)
        h = C.c_void_p()
        t0 = time.perf_counter()
        _capi.check(lib.bt709hip_ringset_create((C.c_int * lanes)(*ordinals), lanes, GAMMAS[args.gamma], 0, g["W"], g["H"], g["ring"],
                                                1 if g["half"] else 0, tries, C.byref(opt), C.byref(h)), "bt709hip_ringset_create")
        self.hunt_s = time.perf_counter() - t0
        self.set = h
        self.ctxs = [lib.bt709hip_ringset_lane_context(h, lane) for lane in range(lanes)]
        self.decs = [lib.bt709hip_ringset_lane_decoder(h, lane) for lane in range(lanes)]
        self.rings = [lib.bt709hip_ringset_lane_ring(h, lane) for lane in range(lanes)]
        self.sample_frames = sample_frames(g["ring"])
        self.identities, self.placements, self.surfs, self.events, self.scratch, self.host_frames_by_lane = [], [], [], [], [], []
        for lane in range(lanes):
            ctx = self.ctxs[lane]
            for kv in args.decoder_option:
                k, v = kv.split("=")
                _capi.check(lib.bt709hip_decoder_set_option(self.decs[lane], int(k), int(v)), "decoder option")
            info = _capi.DeviceInfo()
            _capi.check(lib.bt709hip_context_info(ctx, C.byref(info)))
            self.identities.append({"rank": lane, "ordinal": info.device_ordinal, "pci_bus_id": info.pci_bus_id.decode(), "uuid": info.uuid.decode()})
            if lane == 0:
                self.arch = info.arch.decode()
                self.device = info.name.decode() or self.arch
                self.props = {"compute_units": info.compute_units, "memory_clock_khz": info.memory_clock_khz,
                              "memory_bus_width_bits": info.memory_bus_width_bits, "clock_khz": info.clock_khz}
            self.placements.append(placement_dict(_capi, lib, ctx, self.rings[lane], ["ring set lane %d" % lane]))
            f0, f1, surfs = _capi.Frame(), _capi.Frame(), (_capi.Surface * g["ring"])()
            _capi.check(lib.bt709hip_ring_frame(self.rings[lane], 0, C.byref(f0), None, None))
            in_stride = g["y_bytes"] + g["c_bytes"]
            if g["ring"] > 1:
                _capi.check(lib.bt709hip_ring_frame(self.rings[lane], 1, C.byref(f1), None, None))
                in_stride = f1.y - f0.y
            for i in range(g["ring"]):
                _capi.check(lib.bt709hip_ring_frame(self.rings[lane], i, None, None, C.byref(surfs[i])))
            self.surfs.append(surfs)
            # every lane's ring holds frames of its own (seed 0x709 + i + 1000 lane): a lane that decoded another lane's
            # frames, or none, cannot pass its spot check
            self.host_frames_by_lane.append(upload_frames(np, _capi, lib, ctx, f0.y, in_stride, g, args.content, 1000 * lane, self.sample_frames))
            e0, e1 = C.c_void_p(), C.c_void_p()
            _capi.check(lib.bt709hip_event_create(ctx, C.byref(e0)))
            _capi.check(lib.bt709hip_event_create(ctx, C.byref(e1)))
            self.events.append((e0, e1))
            sc = C.c_void_p()
            _capi.check(lib.bt709hip_malloc(ctx, 128 << 10, C.byref(sc)))
            self.scratch.append(sc)
        self.placement, self.identity = self.placements[0], self.identities[0]
        self.host_frames = self.host_frames_by_lane[0]
        self.nstreams, self.lane_ms = 1, [0.0] * lanes
        t_end = time.perf_counter() + float(os.environ.get("BT709_BENCH_PREWARM_S", "0.4"))
        while time.perf_counter() < t_end:
            self.run_steps(1)
            self.sync()

    def run_steps(self, k):
        for _ in range(k):  # one launch per lane per step, every lane's enqueued before the next step's
            rc = self.lib.bt709hip_ringset_decode(self.set, 0, self.g["ring"], 0)
            if rc != 0:
                raise self._capi.Bt709Error(rc, "ringset decode")

    def sync(self):
        self._capi.check(self.lib.bt709hip_ringset_synchronize(self.set), "ringset synchronize")

    def mark(self, which):
        for ctx, ev in zip(self.ctxs, self.events):
            self._capi.check(self.lib.bt709hip_event_record(ctx, ev[which], None))

    def event_ms(self):
        ms = C.c_float()
        for lane, (ctx, ev) in enumerate(zip(self.ctxs, self.events)):
            self._capi.check(self.lib.bt709hip_event_elapsed_ms(ctx, ev[0], ev[1], C.byref(ms)))
            self.lane_ms[lane] = ms.value
        return max(self.lane_ms)

    def kernel_name(self):
        return self.lib.bt709hip_last_kernel_name().decode()

    def sentinel(self, closing):
        for ctx, sc in zip(self.ctxs, self.scratch):
            self._capi.check(self.lib.bt709hip_copy_probe(ctx, sc.value + (64 << 10), sc.value, (32 << 10) if closing else (4 << 10), None), "sentinel")

    def spot_checks(self, gamma):
        self.run_steps(1)
        self.sync()
        out = []
        for lane in range(self.lanes):
            surfs = self.surfs[lane]
            out.append(compare_with_oracle(self.np, self._capi, self.lib, self.ctxs[lane], None, self.g, gamma, self.sample_frames,
                                           self.host_frames_by_lane[lane], lambda i, s=surfs: (s[i].bgra, s[i].stride)))
        return out


class DryRunner:
    """CPU stand-in used only by --dry-run (tests of the N>1 control flow).  BT709_BENCH_DRY_DEVICES = how many (fake) GPUs the
    box has (default: one per rank); BT709_BENCH_DRY_MISMATCH_RANK = the rank whose parity spot check reports a mismatch;
    BT709_BENCH_DRY_SLOW_RANK = a rank that takes 1.5 x as long per step (a straggler)."""
    device, arch, props, host_frames = "dry-run", "none", {}, {}

    def __init__(self, rank=0, local_rank=0, world=1, lanes=1):
        """lanes > 1: the --launcher threads shape (one process, `lanes` devices); unit u = rank * lanes + lane."""
        self.t = [0.0, 0.0]
        self.rank, self.lanes = rank, lanes
        ndev = dry_device_count(world * lanes)
        units = [local_rank * lanes + lane for lane in range(lanes)]
        self.identities = [{"rank": rank * lanes + lane, "ordinal": u % ndev, "pci_bus_id": "0000:%02x:00.0" % (u % ndev), "uuid": "dry-%d" % (u % ndev)}
                           for lane, u in enumerate(units)]
        self.identity = self.identities[0]
        self.placement = {"chosen": [0, 0], "probe_GBps": {"first_pairing": 0.0, "chosen_confirmed": 0.0}}
        self.placements = [self.placement] * lanes
        slow = os.environ.get("BT709_BENCH_DRY_SLOW_RANK")
        self.lane_factor = [1.5 if slow == str(i["rank"]) else 1.0 for i in self.identities]
        self.step_s = 0.002 * max(self.lane_factor)
        self.spot_frames = []

    def spot_checks(self, gamma):
        bad = os.environ.get("BT709_BENCH_DRY_MISMATCH_RANK")
        return [("MISMATCH in ring frame 0 at output row 0 (forced: BT709_BENCH_DRY_MISMATCH_RANK)" if bad == str(i["rank"]) else "ok", [])
                for i in self.identities]

    def run_steps(self, k):
        time.sleep(self.step_s * k)

    def sync(self):
        pass

    def mark(self, which):
        self.t[which] = time.perf_counter()

    def event_ms(self):
        ms = (self.t[1] - self.t[0]) * 1e3
        self.lane_ms = [ms * f / max(self.lane_factor) for f in self.lane_factor]
        return ms

    def kernel_name(self):
        return "dry-run"

    def sentinel(self, closing):
        pass


def dry_device_count(world):
    return max(1, int(os.environ.get("BT709_BENCH_DRY_DEVICES", str(world))))


def visible_devices(args, world):
    """GPUs this box shows (no device is initialised by the count); --dry-run: BT709_BENCH_DRY_DEVICES."""
    if args.dry_run:
        return dry_device_count(world)
    import metalbt709decoder_amd as mb
    from metalbt709decoder_amd import _capi
    if args.library:
        _capi.load(os.path.abspath(args.library))
    return mb.load_library().bt709hip_device_count()


def self_launch(n, argv, timeout_s=None):
    """`python3 bench.py --gpus N` from ONE plain command (the reference is one process that drives everything,
    Renderer/AAPLRenderer.m:874-985; its N-GPU counterpart must start from one command too).  Called when WORLD_SIZE is unset
    and N > 1, BEFORE the product library, torch or any GPU has been touched by this process: it starts N fresh child
    processes of this same script -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set exactly
    as torch.distributed.run sets them -- relays rank 0's stdout (the single JSON line) and every rank's stderr, and returns
    non-zero if any rank fails (the others are then ended by their exact PIDs).  No os.exec*: a child is a child."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   GROUP_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BT709_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # rank 0 owns this process's stdout (ONE JSON line); the other ranks print nothing there, whatever they do print
        # is kept apart on stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))

    def relay():  # rank 0's JSON line(s) to stdout, anything else it prints there to stderr
        for raw in procs[0].stdout:
            line = raw.decode("utf-8", "replace")
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    import threading
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    deadline = None if not timeout_s else time.time() + timeout_s
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d; ending the other ranks\n" % (r, code))
            if rc != 0 or (deadline and time.time() > deadline):
                rc = rc or 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:  # exact PIDs, never a pattern
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        pump.join(timeout=5)
    return rc


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


def timed_region(runner, steps, barrier, headline=False):
    """One measurement of EXACTLY `steps` steps: barrier + sync, the steps, sync + barrier.
    Returns (host seconds, HIP-event milliseconds).  headline: the region counts towards `value`; it is marked for a
    profiler by two sentinel dispatches OUTSIDE both clocks (before the opening sync, after the closing one)."""
    if headline:
        runner.sentinel(False)
    runner.sync()
    barrier()
    t0 = time.perf_counter()
    runner.mark(0)
    runner.run_steps(steps)
    runner.mark(1)
    runner.sync()
    t1 = time.perf_counter()
    if headline:
        runner.sentinel(True)
        runner.sync()
    barrier()
    return t1 - t0, runner.event_ms()


def main(argv=None):
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    lanes = 1  # devices THIS process drives: > 1 only with --launcher threads
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.launcher == "threads":
        lanes = args.gpus  # ONE process, N GPUs (bt709hip_ringset_*): no ranks are started
    elif "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python3 bench.py --gpus N`: this process becomes the launcher of N ranks and never touches a GPU
        # a rank that hangs (a wedged GPU) must not hang the launcher for ever: one hour covers any run of this script
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv),
                             timeout_s=float(os.environ.get("BT709_BENCH_LAUNCH_TIMEOUT_S", "3600"))))
    if world != args.gpus and lanes == 1:
        args.gpus = world  # under torch.distributed.run the launcher's world size is authoritative
    args.clip = None
    if args.y4m:  # the workload IS the clip: geometry from its header, ring = the headline ring's pixel count in frames of that size
        if lanes > 1:
            sys.exit("--y4m runs with --launcher processes")
        from metalbt709decoder_amd.y4m import Y4MReader
        with Y4MReader(args.y4m) as hdr:
            cw_, ch_ = hdr.width, hdr.height
        ring_ = args.ring or clip_ring_frames(cw_, ch_)
        args.clip = load_clip(args.y4m, ring_)
        WORKLOADS["y4m"] = (cw_, ch_, False, ring_, ring_)
        args.workload = "y4m"
    units = world * lanes  # GPUs the whole job asks for

    dist = None
    if world > 1:
        # Control plane only (barrier + max of two scalars): gloo on CPU tensors.  The data path
        # has no exchange step -- frames are independent -- so no RCCL collective exists.
        # torch is imported BEFORE the product library so the process holds one HIP runtime.
        import torch  # noqa: F401
        import torch.distributed as dist
        # gloo announces its connections on the C++ stdout ("[Gloo] Rank 0 is connected to ..."): this process's stdout
        # carries ONE JSON line and nothing else, so file descriptor 1 points at stderr while the group is formed
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    g = geometry(args.workload, args.ring, 65535, args.frames_per_launch, units, args.share)
    if g["ring_input_over_cache"] < 2.0 and not args.dry_run:
        sys.exit("--ring %d: the ring's input (%.0f MB) must be at least twice the 256 MB Infinity Cache, or the run measures the cache"
                 % (g["ring"], g["ring_input_over_cache"] * 256))
    # One rank per GPU is the contract.  Ranks that wrap onto fewer devices measure something else (several processes
    # sharing one GPU): refused before anything is allocated unless the caller asks for exactly that.
    ndev = visible_devices(args, units)
    if units > max(ndev, 0) and ndev > 0 and not args.allow_shared_devices:
        if rank == 0:
            sys.stderr.write("bench.py: %d ranks but %d visible GPU(s): one rank per GPU is the contract; pass --allow-shared-devices "
                             "for a functional run of the N > 1 path on a smaller box (the line is then labelled shared_devices)\n" % (units, ndev))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(2)
    if args.dry_run:
        runner = DryRunner(rank, local_rank, world, lanes)
    elif lanes > 1:
        if ndev <= 0:
            sys.exit("no HIP device: the product has no CPU fallback")
        runner = SetRunner(args, g, lanes, ndev)
    else:
        runner = GpuRunner(args, g, rank, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    def gather(obj):
        """Every rank's `obj`, in rank order, on every rank (control plane: gloo, pickled CPU objects)."""
        if dist is None:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    def max_over_ranks(values):
        if dist is None:
            return values
        import torch
        t = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    runner.run_steps(args.warmup)
    runner.sync()
    kernel_name = runner.kernel_name()  # of the step's launch: read here, before any sentinel / copy dispatch replaces it
    if not args.dry_run and not kernel_name.startswith("decode_nv12"):
        sys.exit("bench.py: the step launched %r, not a decode kernel" % kernel_name)
    # The contract's region: EXACTLY K steps between barrier + sync on both sides.  It is timed first and reported
    # (`k_step_region_ms`); with the driver's K = 20 it is ~10 ms of GPU work, too short to quote alone (SURVEY 8(d)
    # asks for >= 100 ms per timing), so the figure that is reported as `value` comes from regions of m * K steps,
    # m the smallest integer that makes a region >= 100 ms (every rank must agree: MAX over ranks), bracketed the
    # same way; `ms_per_step` = median region / (m * K).
    own = []  # this process's own (host seconds, [event ms per lane]) of every region that counts: per_rank below

    def region(steps):
        mine = list(timed_region(runner, steps, barrier, headline=True))
        own.append((steps, mine[0], list(runner.lane_ms)))
        return max_over_ranks(mine)
    first = region(args.steps)
    stretch = int(max(1, min(4096, -(-0.100 // max(first[0], 1e-6)))))
    stretch = int(max_over_ranks([float(stretch)])[0])
    region_steps = args.steps * stretch
    samples = [first] if stretch == 1 else []
    repeats = args.repeats
    if repeats <= 0:
        repeats = int(max(5, min(40, -(-0.150 // (first[0] * stretch)))))
        repeats = int(max_over_ranks([float(repeats)])[0])
    while len(samples) < repeats:
        samples.append(region(region_steps))
    samples.sort()
    elapsed, ev_ms = samples[len(samples) // 2]  # the median region (by host time) and ITS event time
    fastest, slowest = samples[0][0], samples[-1][0]

    # whole job per step: every rank decodes frames_per_step frames (4k-batch8: the ranks' shares add up to 8)
    out_px_per_step = units * g["frames_per_step"] * g["OW"] * g["OH"]
    to_value = lambda seconds: region_steps * out_px_per_step / seconds / 1e9
    bytes_per_launch = g["bytes_per_frame"] * g["per_launch"]
    avg_launch_s = (ev_ms / 1e3) / (region_steps * g["launches"])
    achieved = bytes_per_launch / avg_launch_s / 1e9
    read_gbps = (g["W"] * g["H"] * 3 // 2) * g["per_launch"] / avg_launch_s / 1e9

    # Which devices ran, and how each of them did: the line's `value` is the whole job over the SLOWEST unit (MAX over ranks and
    # lanes), so a straggler GPU -- every ring has a placement of its own -- must be tellable from a launcher problem.  One record
    # per rank (--launcher threads: per lane of the one process): its own median region (same regions as the headline's, its own
    # clocks), its device and its ring's placement.
    mine = sorted((host_s, lane_ms) for steps, host_s, lane_ms in own if steps == region_steps)
    my_host_s, my_lane_ms = mine[len(mine) // 2]
    # Parity tripwire on EVERY unit, each over its OWN ring (the frames differ: seed 0x709 + i + 1000 unit); a mismatch on any of
    # them nulls `value` and fails the whole job.  Untimed.
    checks = runner.spot_checks(GAMMAS[args.gamma])
    my_records = []
    for lane in range(lanes):
        launch_s = (my_lane_ms[lane] / 1e3) / (region_steps * g["launches"])
        pl = runner.placements[lane] or {}
        ident = dict(runner.identities[lane])
        ident["rank"] = rank * lanes + lane
        my_records.append(dict(ident,
                               # a process's host clock per step; the lanes of one process share it, their own GPU-side time is avg_launch_us
                               ms_per_step=round((my_host_s if lanes == 1 else my_lane_ms[lane] / 1e3) / region_steps * 1e3, 5),
                               avg_launch_us=round(launch_s * 1e6, 3), frac=round(bytes_per_launch / launch_s / 1e9 / HBM_PEAK_GBPS, 4),
                               placement={"chosen": pl.get("chosen"), "first_GBps": (pl.get("probe_GBps") or {}).get("first_pairing"),
                                          "chosen_GBps": (pl.get("probe_GBps") or {}).get("chosen_confirmed"),
                                          "hunt_ms": pl.get("hunt_ms"), "peak_bytes": pl.get("peak_bytes")},
                               parity_spot_check=checks[lane][0], parity_spot_frames=checks[lane][1]))
    per_rank = [r for records in gather(my_records) for r in records]
    failed = any(r["parity_spot_check"] != "ok" for r in per_rank)
    devices = sorted({(r["pci_bus_id"], r["uuid"]) for r in per_rank})
    n_distinct = len(devices)

    if g["batch8"]:
        step_text = ("one step = %d x 4K frames over the whole job = ONE launch of %d frame(s) per GPU, frame i -> "
                     "GPU i mod %d, walking the ring%s" % (units * g["per_launch"], g["per_launch"], units,
                                                            "; K steps replayed from one HIP graph" if args.graph else ""))
    else:
        step_text = "one step = the whole ring = %d launches x %d frames" % (g["launches"], g["per_launch"])
    result = {
        "metric": "Gpixel/s, %s NV12->sRGB BGRA decode (output pixels)" % args.workload,
        "value": round(to_value(elapsed), 3),
        "unit": "Gpixel/s",
        "n_gpus": n_distinct,  # DISTINCT physical devices (by PCI bus id); == ranks unless shared_devices
        "ranks": units,  # processes x the devices each drives (--launcher threads: 1 x N)
        "shared_devices": n_distinct < units,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / region_steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong" if g["batch8"] and not args.share else "weak",
        "vs_baseline": None,
        "dtype": "f32",  # fp32 arithmetic on u8 samples, exact-table transfer, u8 out
        "data": "y4m" if args.clip else "synthetic",
        "repeats": len(samples),  # value / ms_per_step = the MEDIAN of this many regions of `region_steps` steps
        "region_steps": region_steps,  # = steps x the smallest integer that makes a timed region >= 100 ms
        "k_step_region_ms": round(first[0] * 1e3, 4),  # the region of EXACTLY `steps` steps, timed first
        "value_min": round(to_value(slowest), 3),
        "value_max": round(to_value(fastest), 3),
        "config": {
            "workload": "%dx%d NV12 BT.709 -> %dx%d BGRA8 sRGB, gamma=%s%s; per GPU a ring of %d %s resident in HBM; %s"
                        % (g["W"], g["H"], g["OW"], g["OH"], args.gamma, ", fused 2:1 rescale" if g["half"] else "", g["ring"],
                           ("frames = the %d frames of the YUV4MPEG2 C420jpeg clip %s (sha256 %s) repeated, planar chroma interleaved on the device"
                            % (len(args.clip["frames"]), os.path.basename(args.clip["path"]), args.clip["sha256"])) if args.clip else
                           "distinct frames (%s, seed 0x709+i)" % {"random": "uniform random bytes", "smooth": "smooth video-like planes", "flat": "flat grey"}[args.content],
                           step_text),
            "frames_per_step_per_gpu": g["frames_per_step"],
            "streams": getattr(runner, "nstreams", 1),
            "coalesce": args.coalesce,
            "sharding": "independent frames per GPU, no collective",
            "launcher": ("self (bench.py started its %d ranks)" % world if os.environ.get("BT709_BENCH_SELF_LAUNCHED")
                         else "torch.distributed.run" if world > 1
                         else "threads (ONE process drives %d GPUs through bt709hip_ringset_*: one launch per device per step from one thread)" % lanes
                         if lanes > 1 else "single process"),
            "memory_plan": memory_plan(g, args, world, lanes),
            "placement": getattr(runner, "placement", None),  # rank 0's ring: allocated `tries` times, the fastest-streaming pairing kept (untimed set-up)
            "device": runner.device,
            "arch": runner.arch,
            "device_props": runner.props,
            # one record per rank, rank order: which physical device it drove
            "devices": [{"rank": r["rank"], "ordinal": r["ordinal"], "pci_bus_id": r["pci_bus_id"], "uuid": r["uuid"]} for r in per_rank],
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "kernel": kernel_name,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_us": round(avg_launch_s * 1e6, 3),
            "read_GBps": round(read_gbps, 1),
        },
        # every rank's own clocks over the headline's regions (value comes from the MAX over ranks), device, placement, spot check
        "per_rank": [{k: r[k] for k in ("rank", "ordinal", "pci_bus_id", "ms_per_step", "avg_launch_us", "frac", "placement",
                                        "parity_spot_check")} for r in per_rank],
        "parity_spot_check": "ok" if not failed else "; ".join("rank %d: %s" % (r["rank"], r["parity_spot_check"]) for r in per_rank
                                                                if r["parity_spot_check"] != "ok"),
        "parity_spot_check_ranks": len(per_rank),  # every rank checks its own ring
        "parity_spot_frames": per_rank[0]["parity_spot_frames"],  # one ring frame per XCD band, 48 rows each; the same picks on every rank
    }
    result["roofline"].update(load_traffic(args.workload, g))

    if rank == 0 and not args.dry_run and not failed and units == 1:
        result["roofline"].update(first_allocation_leg(runner, args, g, barrier, region_steps, achieved))
        if args.workload in ("4k", "8k-half") and args.content == "random" and not args.no_smooth_leg:
            result["roofline"]["smooth_content"] = smooth_leg(runner, args, g, barrier, region_steps)
        copy_gbps = runner.copy_ceiling()
        result["roofline"]["same_run_copy_GBps"] = round(copy_gbps, 1)
        result["roofline"]["frac_of_same_run_copy"] = round(achieved / copy_gbps, 4)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(runner.host_frames[0], g, GAMMAS[args.gamma], args.cpu_seconds,
                                                  own="frame 0 of the --y4m clip" if args.clip else None)
    if failed:
        result["value"] = None  # a wrong-output kernel on ANY rank yields no benchmark record
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(1)


def first_allocation_leg(runner, args, g, barrier, region_steps, hunted_gbps):
    """The same launches, the same frames, on the ring this process allocated FIRST (bt709hip_ring_create with tries = 1 = two
    plain allocations, what bt709hip_malloc gives a caller): regions bracketed like the headline's, median of 5.  Reported
    beside roofline.frac, which comes from the ring the product's placement hunt chose -- the review's request: both numbers
    from one process in one line."""
    if "hunted" not in runner.rings:  # --placement-tries 1, or ranks sharing a device: the headline IS the first allocation
        return {"first_allocation_frac": round(hunted_gbps / HBM_PEAK_GBPS, 4), "first_allocation_GBps": round(hunted_gbps, 1),
                "first_allocation_note": "no hunt in this run: the headline ran on the first allocation"}
    runner.use_ring("first")
    runner.run_steps(max(3, args.warmup // 2))
    regions = sorted(timed_region(runner, region_steps, barrier) for _ in range(5))
    _, ev_ms = regions[len(regions) // 2]
    avg_launch_s = (ev_ms / 1e3) / (region_steps * g["launches"])
    gbps = g["bytes_per_frame"] * g["per_launch"] / avg_launch_s / 1e9
    runner.use_ring("hunted")
    return {"first_allocation_frac": round(gbps / HBM_PEAK_GBPS, 4), "first_allocation_GBps": round(gbps, 1),
            "first_allocation_avg_launch_us": round(avg_launch_s * 1e6, 3),
            "first_allocation_note": "same process, same frames, the ring allocated first with tries = 1 (plain allocations)"}


def smooth_leg(runner, args, g, barrier, region_steps):
    """The same launches on video-like content (neighbouring pixels share table buckets: fewer LDS bank
    conflicts).  Reported beside the headline, never as the headline: random bytes are the worst case."""
    runner.fill_ring("smooth")
    runner.run_steps(max(3, args.warmup // 2))
    regions = sorted(timed_region(runner, region_steps, barrier) for _ in range(5))
    _, ev_ms = regions[len(regions) // 2]
    avg_launch_s = (ev_ms / 1e3) / (region_steps * g["launches"])
    gbps = g["bytes_per_frame"] * g["per_launch"] / avg_launch_s / 1e9
    runner.fill_ring("random")  # frame 0 is the cpu_baseline sample again
    return {"achieved": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4), "avg_launch_us": round(avg_launch_s * 1e6, 3)}


def load_traffic(workload, g):
    """HBM bytes per launch from the PMC passes.  PMC counters cannot be collected inside a timed run,
    so this is a REPLAY of profiles/pmc_traffic.json (written by tools/pmc_summary.py from separate
    rocprofv3 --pmc passes on the builder's GPU lease) and is labelled as such; it is dropped (null)
    when no pass exists for this workload or the pass profiled another launch size."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]
    except Exception:
        return {"traffic": None}
    algorithmic = g["bytes_per_frame"] * g["per_launch"]
    if not 0.9 < rec["hbm_bytes_per_launch"] / algorithmic < 1.5:
        return {"traffic": None, "traffic_source": "profiles/pmc_traffic.json has no pass for this launch size"}
    return {"traffic": rec["hbm_bytes_per_launch"],
            "traffic_source": "replayed from profiles/pmc_traffic.json: separate rocprofv3 --pmc passes "
                              "(FETCH_SIZE x2 per the gfx950 note, WRITE_SIZE) on the builder's lease, round %s; "
                              "not measured in this run" % rec.get("round", "?")}


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


def cpu_baseline(frame0, g, gamma, target_seconds, own=None):
    """The CPU path beside the GPU, as SURVEY 8(d) defines it: WHOLE seeded frames through the reference's own per-pixel function
    (oracle/_ref, kind "reference"; the CPU oracle, kind "port", when that library is absent or for the fused 2:1 path) -- one
    1920x1080 frame and one 3840x2160 frame on ONE thread (the reference is single-threaded scalar, BGRAToBT709Converter.m:146-198),
    median of 3 passes each, then the workload's own frame row-partitioned over all host cores for the rest of the budget.
    `value` = that all-core figure.  Checker code, timed only: this is the one place bench.py touches oracle/."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import oracle_lib
    half = g["half"]
    kind, impl = "port", oracle_lib.Oracle()
    if not half:
        try:
            impl, kind = oracle_lib.Reference(), "reference"
        except Exception:
            pass

    def seeded(W, H):  # the ring's frame 0 of that size: PRNG bytes, seed 0x709
        if (W, H) == (g["W"], g["H"]):
            buf = frame0
        else:
            buf = np.random.default_rng(0x709).integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
        return buf[:W * H].reshape(H, W), buf[W * H:].reshape(H // 2, W)

    def decode_rows(y, c, r0, r1, out, via_half):
        """Source rows [r0, r1) (even bounds) of one frame; out = the whole frame's output."""
        if via_half:
            out[r0 // 2:r1 // 2] = impl.decode_nv12_half(gamma, y[r0:r1], c[r0 // 2:r1 // 2])
        else:
            impl.decode_nv12(gamma, y, c, rows=(r0, r1), out=out)

    def one_pass(W, H, threads, via_half, pool=None):
        y, c = seeded(W, H)
        out = np.zeros((H // 2, W // 2 * 4) if via_half else (H, W * 4), np.uint8)
        step = -(-H // threads + 3) // 4 * 4  # bands of whole 4-row groups (2:1: whole output row pairs)
        bands = [(r, min(r + step, H)) for r in range(0, H, step)]
        t0 = time.perf_counter()
        if threads == 1:
            decode_rows(y, c, 0, H, out, via_half)
        else:
            list(pool.map(lambda b: decode_rows(y, c, b[0], b[1], out, via_half), bands))
        return time.perf_counter() - t0

    cores = usable_cores()
    t_start = time.perf_counter()
    frames = []
    sizes = [(1920, 1080, False), (3840, 2160, False)]
    if (g["W"], g["H"], half) not in sizes:
        sizes.append((g["W"], g["H"], half))
    for W, H, via_half in sizes:
        first = one_pass(W, H, 1, via_half)  # also pages the buffers in; counted
        times = [first] + [one_pass(W, H, 1, via_half) for _ in range(2 if first < 3.0 else 0)]
        times.sort()
        frames.append({"size": "%dx%d%s" % (W, H, " (fused 2:1)" if via_half else ""), "threads": 1, "samples": len(times),
                       "Mpx_per_s": round(W * H / times[len(times) // 2] / 1e6, 3), "seconds": round(times[len(times) // 2], 4)})
    # all cores, the workload's own frame: as many whole-frame passes as the remaining budget holds (at least 5)
    W, H = g["W"], g["H"]
    passes = []
    with ThreadPoolExecutor(cores) as pool:
        one_pass(W, H, cores, half, pool)  # threads started, pages touched
        deadline = t_start + target_seconds
        while len(passes) < 5 or time.perf_counter() < deadline:
            passes.append(one_pass(W, H, cores, half, pool))
            if len(passes) >= 200:
                break
    passes.sort()
    med = passes[len(passes) // 2]
    frames.append({"size": "%dx%d%s" % (W, H, " (fused 2:1)" if half else ""), "threads": cores, "samples": len(passes),
                   "Mpx_per_s": round(W * H / med / 1e6, 3), "seconds": round(med, 4)})
    own_single = [f for f in frames if f["threads"] == 1 and f["size"].startswith("%dx%d" % (W, H))][0]
    to_out = 0.25 if half else 1.0  # the metric counts OUTPUT pixels
    return {
        "value": round(to_out * W * H / med / 1e9, 5),
        "unit": "Gpixel/s",
        "cores": cores,
        "kind": kind,
        "sample": "whole seeded frames (PRNG bytes, seed 0x709" + ("; %dx%d: %s" % (W, H, own) if own else "") + "): %s on 1 thread (median of `samples` passes, see frames[]); then the %dx%d "
                  "frame row-partitioned over %d threads, median of %d whole-frame passes (%.1f s of CPU work in all)"
                  % (" and ".join(f["size"] for f in frames if f["threads"] == 1), W, H, cores, len(passes), time.perf_counter() - t_start),
        "frames": frames,
        "single_thread_value": round(to_out * own_single["Mpx_per_s"] / 1e3, 6),
        "single_thread_samples": own_single["samples"],
    }


if __name__ == "__main__":
    main()
