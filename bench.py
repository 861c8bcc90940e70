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
per-pixel function (oracle/_ref, kind "reference") or, when that library is absent, the CPU oracle (kind "port") on a
bounded sample of the same frames over the host cores.

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
    ap.add_argument("--coalesce", type=int, default=0,
                    help="BT709HIP_OPT_COALESCE: gather this many one-frame submits into one launch (4k-batch8; 0 = off)")
    ap.add_argument("--stream-priorities", default="", metavar="P1,P2,...",
                    help="4k-batch8: scheduling priority of the 2nd, 3rd, ... stream (0 normal, -1 higher, 1 lower); lab knob")
    ap.add_argument("--no-smooth-leg", action="store_true", help="skip the extra smooth-content measurement (N=1, 4k)")
    ap.add_argument("--decoder-option", action="append", default=[], metavar="ID=VALUE",
                    help="bt709hip_decoder_set_option(ID, VALUE) on the bench decoder (tuning sweeps)")
    ap.add_argument("--library", default=None,
                    help="load this build of libbt709hip.so (python -m metalbt709decoder_amd.build --variant ...) for A/B runs")
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
        self._capi.check(self.lib.bt709hip_ring_create(self.dec._handle, g["W"], g["H"], g["ring"], 1 if g["half"] else 0, tries,
                                                       C.byref(r)), "bt709hip_ring_create")
        return r

    def use_ring(self, name):
        """Frame / surface descriptors of ring `name` as the arrays the launches take."""
        g, ring = self.g, self.rings[name]
        self.ring_name = name
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
        p = self.RingPlacement()
        self._capi.check(self.lib.bt709hip_ring_placement_info(self.rings[self.ring_name], C.byref(p)))
        free_b, total_b = C.c_size_t(), C.c_size_t()
        self._capi.check(self.lib.bt709hip_mem_info(self.h, C.byref(free_b), C.byref(total_b)))
        kept = [k for k in p.out_kept if k >= 0]
        return {"allocator": "bt709hip_ring_create (csrc/bt709_ring.cpp): candidates per slab, the decoder's own launch as the probe",
                "tries": p.tries, "candidates": [p.in_candidates, p.out_candidates], "chosen": [p.chosen_in, p.chosen_out],
                "pairings_probed": p.probes,
                "probe_GBps": {"first_pairing": round(p.first_GBps, 1), "chosen_confirmed": round(p.chosen_GBps, 1),
                               "best": round(p.best_GBps, 1), "worst": round(p.worst_GBps, 1)},
                # allocation order; `kept` = the indices (same order) that went on to the pairing probes
                "output_prescan_GBps": [round(v, 1) for v in p.out_prescan_GBps[:p.out_candidates]], "output_kept": kept,
                "rings_resident": sorted(self.rings), "device_memory_free_GB": round(free_b.value / 1e9, 1),
                "device_memory_total_GB": round(total_b.value / 1e9, 1)}

    def fill_ring(self, content):
        """Uploads (outside every timed region) the ring's frames: seeded PRNG bytes or smooth planes."""
        np, g, lib, h = self.np, self.g, self.lib, self.h
        base_smooth = None
        for i in range(g["ring"]):
            rng = np.random.default_rng(0x709 + i + 1000 * self.rank)
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
            self._capi.check(lib.bt709hip_upload(h, self.d_in.value + i * self.in_stride, buf.shape[1], buf.ctypes.data,
                                                 buf.shape[1], buf.shape[1], 1, None), "upload")
            self._capi.check(lib.bt709hip_stream_synchronize(h, None))
            if i in self.sample_frames:
                self.host_frames[i] = buf.reshape(-1)

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
        return ms.value

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
        import oracle_lib
        np, g = self.np, self.g
        rows = 16
        self.run_steps(1)
        if g["batch8"]:  # a step is a share of the ring there: decode every sampled frame's launch
            for i in self.sample_frames:
                self.launch(i - i % g["per_launch"], g["per_launch"])
        self.sync()
        o = oracle_lib.Oracle()
        checked = []
        for i in self.sample_frames:
            y, c = split_planes(self.host_frames[i], g)
            for r0 in (0, (g["OH"] // 2) // 4 * 4, g["OH"] - rows):
                got = np.empty((rows, g["OW"] * 4), np.uint8)
                self._capi.check(self.lib.bt709hip_download(self.h, got.ctypes.data, got.shape[1],
                                                            self.surfs[i].bgra + r0 * self.surfs[i].stride,
                                                            self.surfs[i].stride, got.shape[1], rows, self.stream))
                self.sync()
                if g["half"]:
                    want = o.decode_nv12_half(gamma, y[2 * r0:2 * (r0 + rows)], c[r0:r0 + rows])
                else:
                    want = o.decode_nv12(gamma, y, c, rows=(r0, r0 + rows))[r0:r0 + rows]
                if not np.array_equal(got, want):
                    return "MISMATCH in ring frame %d at output row %d" % (i, r0)
            checked.append(i)
        self.spot_frames = checked
        return "ok"


def sample_frames(ring):
    """One frame of each eighth of the ring (= each XCD band of a whole-ring launch), first and last frame included."""
    if ring < 8:
        return list(range(ring))
    per = ring // 8
    picks = {0, ring - 1}
    for b in range(1, 7):
        picks.add(b * per + (b + 4) % per)
    return sorted(picks)


class DryRunner:
    """CPU stand-in used only by --dry-run (tests of the N>1 control flow)."""
    device, arch, props, host_frames = "dry-run", "none", {}, {}

    def __init__(self):
        self.t = [0.0, 0.0]

    def run_steps(self, k):
        time.sleep(0.002 * k)

    def sync(self):
        pass

    def mark(self, which):
        self.t[which] = time.perf_counter()

    def event_ms(self):
        return (self.t[1] - self.t[0]) * 1e3

    def kernel_name(self):
        return "dry-run"

    def sentinel(self, closing):
        pass


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
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python3 bench.py --gpus N`: this process becomes the launcher of N ranks and never touches a GPU
        # a rank that hangs (a wedged GPU) must not hang the launcher for ever: one hour covers any run of this script
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv),
                             timeout_s=float(os.environ.get("BT709_BENCH_LAUNCH_TIMEOUT_S", "3600"))))
    if world != args.gpus:
        args.gpus = world  # under torch.distributed.run the launcher's world size is authoritative

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

    g = geometry(args.workload, args.ring, 65535, args.frames_per_launch, world, args.share)
    if g["ring_input_over_cache"] < 2.0 and not args.dry_run:
        sys.exit("--ring %d: the ring's input (%.0f MB) must be at least twice the 256 MB Infinity Cache, or the run measures the cache"
                 % (g["ring"], g["ring_input_over_cache"] * 256))
    runner = DryRunner() if args.dry_run else GpuRunner(args, g, rank, local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

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
    first = max_over_ranks(list(timed_region(runner, args.steps, barrier, headline=True)))
    stretch = int(max(1, min(4096, -(-0.100 // max(first[0], 1e-6)))))
    stretch = int(max_over_ranks([float(stretch)])[0])
    region_steps = args.steps * stretch
    samples = [first] if stretch == 1 else []
    repeats = args.repeats
    if repeats <= 0:
        repeats = int(max(5, min(40, -(-0.150 // (first[0] * stretch)))))
        repeats = int(max_over_ranks([float(repeats)])[0])
    while len(samples) < repeats:
        samples.append(max_over_ranks(list(timed_region(runner, region_steps, barrier, headline=True))))
    samples.sort()
    elapsed, ev_ms = samples[len(samples) // 2]  # the median region (by host time) and ITS event time
    fastest, slowest = samples[0][0], samples[-1][0]

    # whole job per step: every rank decodes frames_per_step frames (4k-batch8: the ranks' shares add up to 8)
    out_px_per_step = world * g["frames_per_step"] * g["OW"] * g["OH"]
    to_value = lambda seconds: region_steps * out_px_per_step / seconds / 1e9
    bytes_per_launch = g["bytes_per_frame"] * g["per_launch"]
    avg_launch_s = (ev_ms / 1e3) / (region_steps * g["launches"])
    achieved = bytes_per_launch / avg_launch_s / 1e9
    read_gbps = (g["W"] * g["H"] * 3 // 2) * g["per_launch"] / avg_launch_s / 1e9

    if g["batch8"]:
        step_text = ("one step = %d x 4K frames over the whole job = ONE launch of %d frame(s) per GPU, frame i -> "
                     "GPU i mod %d, walking the ring%s" % (world * g["per_launch"], g["per_launch"], world,
                                                            "; K steps replayed from one HIP graph" if args.graph else ""))
    else:
        step_text = "one step = the whole ring = %d launches x %d frames" % (g["launches"], g["per_launch"])
    result = {
        "metric": "Gpixel/s, %s NV12->sRGB BGRA decode (output pixels)" % args.workload,
        "value": round(to_value(elapsed), 3),
        "unit": "Gpixel/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / region_steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong" if g["batch8"] and not args.share else "weak",
        "vs_baseline": None,
        "dtype": "f32",  # fp32 arithmetic on u8 samples, exact-table transfer, u8 out
        "data": "synthetic",
        "repeats": len(samples),  # value / ms_per_step = the MEDIAN of this many regions of `region_steps` steps
        "region_steps": region_steps,  # = steps x the smallest integer that makes a timed region >= 100 ms
        "k_step_region_ms": round(first[0] * 1e3, 4),  # the region of EXACTLY `steps` steps, timed first
        "value_min": round(to_value(slowest), 3),
        "value_max": round(to_value(fastest), 3),
        "config": {
            "workload": "%dx%d NV12 BT.709 -> %dx%d BGRA8 sRGB, gamma=%s%s; per GPU a ring of %d distinct frames "
                        "(%s, seed 0x709+i) resident in HBM; %s"
                        % (g["W"], g["H"], g["OW"], g["OH"], args.gamma, ", fused 2:1 rescale" if g["half"] else "",
                           g["ring"], {"random": "uniform random bytes", "smooth": "smooth video-like planes", "flat": "flat grey"}[args.content],
                           step_text),
            "frames_per_step_per_gpu": g["frames_per_step"],
            "streams": getattr(runner, "nstreams", 1),
            "coalesce": args.coalesce,
            "sharding": "independent frames per GPU, no collective",
            "launcher": ("self (bench.py started its %d ranks)" % world if os.environ.get("BT709_BENCH_SELF_LAUNCHED")
                         else "torch.distributed.run" if world > 1 else "single process"),
            "placement": getattr(runner, "placement", None),  # ring allocated `tries` times, the fastest-streaming one kept (untimed set-up)
            "device": runner.device,
            "arch": runner.arch,
            "device_props": runner.props,
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
    }
    result["roofline"].update(load_traffic(args.workload, g))

    failed = False
    if rank == 0 and not args.dry_run:
        # parity tripwire on every run (rank 0 of a multi-GPU job too); a mismatch fails the run
        result["parity_spot_check"] = runner.spot_check(GAMMAS[args.gamma])
        failed = result["parity_spot_check"] != "ok"
        result["parity_spot_frames"] = getattr(runner, "spot_frames", [])  # one ring frame per XCD band, 48 rows each
        if world == 1:
            result["roofline"].update(first_allocation_leg(runner, args, g, barrier, region_steps, achieved))
            if args.workload in ("4k", "8k-half") and args.content == "random" and not args.no_smooth_leg:
                result["roofline"]["smooth_content"] = smooth_leg(runner, args, g, barrier, region_steps)
            copy_gbps = runner.copy_ceiling()
            result["roofline"]["same_run_copy_GBps"] = round(copy_gbps, 1)
            result["roofline"]["frac_of_same_run_copy"] = round(achieved / copy_gbps, 4)
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(runner.host_frames[0], g, GAMMAS[args.gamma], args.cpu_seconds)
        if failed:
            result["value"] = None  # a wrong-output kernel yields no benchmark record
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
