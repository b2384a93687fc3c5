#!/usr/bin/env python3
"""bench.py — Mpaths/s and HDR frames/s of the path-tracing reference pass on MI355X.

A step = one pass of the hot path over one batch: the 1920x1080 S-cornell frame accumulated to
8 spp with maxBounces 4 (BASELINE.json configs[1]), inputs resident in HBM.  With
N > 1 GPUs the image is cut into interleaved 16-pixel stripes, one stripe set per rank, and the
per-rank RGBA32F tiles are gathered to rank 0 over RCCL inside the timed region (strong scaling:
the total work is fixed).  Steps are enqueued back to back with up to three frames in flight
(PROSPER_PT_RENDER_PIPELINED, prosper's frames-in-flight idea; --in-order for A/B): every step's kernels,
its accumulate and its gather complete inside the timed region, which is bracketed by barrier + synchronize.
Before the W warm-up steps the device is woken with PREHEAT_STEPS untimed steps (clock ramp, see below).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from prosper_amd import capi, scenes, structs as S, tiling  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
STRIPE_WIDTH = tiling.STRIPE_WIDTH

CONFIGS = {
    # name: (scene builder, width, height, spp, maxBounces, ibl)
    "c2": ("S-cornell 1920x1080 8spp maxBounces 4", lambda: scenes.cornell(), 1920, 1080, 8, 4, False),
    "c3": ("S-sponza-class 1920x1080 8spp maxBounces 4 IBL", lambda: scenes.sponza_class(), 1920, 1080, 8, 4, True),
    "c4": ("S-sponza-class + 1024 lights + foliage 1920x1080 8spp maxBounces 4 IBL",
           lambda: scenes.sponza_class(lights=True, foliage=True), 1920, 1080, 8, 4, True),
    # BASELINE.json configs[4]: meant for 8 GPUs (a rank then renders 1/8 of the stripes); runs on one as well
    "c5": ("S-sponza-class 3840x2160 64spp maxBounces 4 IBL", lambda: scenes.sponza_class(), 3840, 2160, 64, 4, True),
    "c1": ("S-cornell 256x256 1spp maxBounces 1", lambda: scenes.cornell(), 256, 256, 1, 1, False),
}


def algorithmic_bytes(c, stats):
    """B_alg of SURVEY §8d from the deterministic work counters of one launch."""
    tri_long = c["triangleTests"] - c["shortIndexTriangleTests"]
    hits = c["closestHits"] + c["anyHitCalls"]
    short_share = (c["shortIndexHits"] / hits) if hits else 0.0
    chit = c["closestHits"] * (314.0 * short_share + 320.0 * (1.0 - short_share))
    ahit = c["anyHitCalls"] * (138.0 * short_share + 144.0 * (1.0 - short_share))
    return (c["nodeVisits"] * float(stats.nodeBytes) + c["shortIndexTriangleTests"] * 30.0 + tri_long * 36.0 + chit +
            ahit + c["lightSamples"] * 36.0 + c["spotLightSamples"] * 52.0 + c["skyLookups"] * 32.0 +
            c["pixelsWritten"] * 16.0 + c["historyReads"] * 16.0)


PREHEAT_STEPS = 24


def measured_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the
    gfx950 guide prescribes, + WRITE_SIZE), or None if that config/kernel has not been measured."""
    path = os.path.join(ROOT, "profiles", "r01_%s_traffic.json" % config)
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def measured_issue(config, kernel):
    """Vector-ALU issue-slot occupancy and lane utilisation of `kernel` from the committed PMC pass
    (profiles/r01_valu_issue.json): what actually bounds the traversal kernels."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_valu_issue.json")) as f:
            k = json.load(f)["configs"][config][kernel]
        return {"valu_busy": k["valu_busy"], "lane_util": k["lane_util"], "source": "profiles/r01_valu_issue.json"}
    except (OSError, KeyError, ValueError):
        return None


def make_pc(focal, frame_index, max_bounces, ibl, skip_history):
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT
    if ibl:
        flags |= S.PC_FLAG_IBL
    if skip_history:
        flags |= S.PC_FLAG_SKIP_HISTORY
    return S.ReferencePC(0, flags, frame_index, 1e-5, 1.0, focal, 3, max_bounces)


def host_cores():
    """Hardware threads this process may actually use: min(affinity, cgroup CPU quota)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(world, cam, focal, width, height, spp, max_bounces, ibl, budget_s=15.0):
    """The oracle (a scalar port) timed on this box's host cores on a bounded sample."""
    from oracle import binding as oracle
    osc = oracle.OracleScene(world, brute_force=False)
    cores = host_cores()
    img = None
    frames = 0
    t0 = time.perf_counter()
    while frames < spp:
        pc = make_pc(focal, frames + 1, max_bounces, ibl, frames == 0)
        img, _ = osc.render(pc, cam, width, height, history=img, threads=cores)
        frames += 1
        elapsed = time.perf_counter() - t0
        if elapsed + elapsed / frames > budget_s:
            break
    elapsed = time.perf_counter() - t0
    osc.close()
    return {
        "value": width * height * frames / elapsed / 1e6,
        "unit": "Mpaths/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d of %d spp of the same %dx%d frame (%.1f s, OpenMP over rows)" % (frames, spp, width, height, elapsed),
    }


def bench_tone_map(ctx, torch, width, height, stream, repeats=50):
    """The step after the path (not part of `value`): blit + tone_map.comp in one kernel over
    the frame just rendered; 16 B read + 4 B written per pixel, the 442 KB LUT stays in L2."""
    import numpy as np
    from prosper_amd import dds
    g = np.linspace(0.0, 1.0, 48)
    b, gg, r = np.meshgrid(g, g, g, indexing="ij")
    ctx.set_tone_map_lut(dds.encode_r9g9b9e5(np.stack([r, gg, b], axis=-1)))  # identity-like LUT: timing only
    out = torch.empty((height, width), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.tone_map(1.0, 1.0, device_ptr=out.data_ptr(), to_host=False, stream=stream)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(repeats):
        ctx.tone_map(1.0, 1.0, device_ptr=out.data_ptr(), to_host=False, stream=stream)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / repeats
    nbytes = 20.0 * width * height
    return {"kernel": "tone_map_kernel", "ms_per_frame": ms, "algorithmic_bytes": nbytes,
            "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def bench_restir_di(ctx, torch, cam, focal, width, height, stream, repeats=30):
    """A second client of the traversal (not part of `value`): the ReSTIR-DI trace pass over a
    G-buffer made from the product's own debug views of the primary hits, one random light per pixel."""
    import numpy as np

    def view(name):
        pc = S.ReferencePC(S.DrawType[name], S.PC_FLAG_SKIP_HISTORY, 1, 1e-5, 1.0, focal, 3, 1)
        ctx.render(pc, cam, width, height, stream=stream)
        return ctx.read_hdr(stream)[..., :3].astype(np.float64)
    pos, raw_n, alb = view("Position"), view("ShadingNormal"), view("Albedo")
    rough, metal = view("Roughness")[..., 0], view("Metallic")[..., 0]
    hit = raw_n.sum(axis=-1) > 0.0
    n = np.where(hit[..., None], raw_n * 2.0 - 1.0, np.array([0.0, 0.0, 1.0]))
    n = n / np.abs(n).sum(axis=-1, keepdims=True)           # gbuffer.frag signedOctEncode
    ey = n[..., 1] * 0.5 + 0.5
    enc = np.stack([n[..., 0] * 0.5 + ey, n[..., 0] * -0.5 + ey, np.clip(n[..., 2] * 1e30, 0.0, 1.0)], axis=-1)
    c2c = np.frombuffer(bytes(cam.cameraToClip), np.float32).reshape(4, 4).T.astype(np.float64)
    w2c = np.frombuffer(bytes(cam.worldToCamera), np.float32).reshape(4, 4).T.astype(np.float64)
    clip = np.concatenate([pos, np.ones(pos.shape[:2] + (1,))], axis=-1) @ (c2c @ w2c).T
    depth = np.where(hit, clip[..., 2] / np.where(clip[..., 3] == 0, 1.0, clip[..., 3]), 0.0).astype(np.float32)
    ar = np.concatenate([alb, np.maximum(rough, 0.05)[..., None]], axis=-1).astype(np.float32)
    nm = np.stack([enc[..., 0], enc[..., 1], metal, enc[..., 2]], axis=-1).astype(np.float32)
    rng = np.random.default_rng(1)
    world = ctx._world
    lights = 1 + world.point_lights.count + world.spot_lights.count
    idx = rng.integers(0, lights, size=(height, width)).astype(np.int32)
    res = np.stack([idx.view(np.float32), np.ones((height, width), np.float32)], axis=-1)
    t = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (ar, nm, depth, res)]
    pc = S.RestirTracePC(0, 1, 3)
    for _ in range(3):
        ctx.restir_di_trace_device(pc, cam, width, height, *[x.data_ptr() for x in t], stream=stream)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(repeats):
        ctx.restir_di_trace_device(pc, cam, width, height, *[x.data_ptr() for x in t], stream=stream)
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / repeats
    return {"kernel": "restir_di_trace_kernel", "ms_per_frame": ms, "Mpixels_per_s": width * height / ms / 1e3,
            "gbuffer_bytes_per_pixel": 60, "lights": int(lights)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--megakernel", action="store_true", help="use the one-lane-per-pixel kernel (A/B)")
    ap.add_argument("--persistent", action="store_true", help="use the persistent path-regeneration kernel (A/B)")
    ap.add_argument("--single-chain", action="store_true",
                    help="one chain of launches on one stream instead of two overlapping half-batches (A/B)")
    ap.add_argument("--in-order", action="store_true",
                    help="no frames in flight: every step starts after the previous one has finished (A/B)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world_size, args.gpus))
    # PROSPER_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend, to rehearse the N > 1 flow
    # (stripes, gather, de-interleave, max-over-ranks timing) on a one-GPU box; never a measurement.
    rehearse = os.environ.get("PROSPER_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))

    workload, builder, width, height, spp, max_bounces, ibl = CONFIGS[args.config]
    if not tiling.check_divisible(width, world_size):
        raise SystemExit("%d stripes do not divide over %d ranks" % (width // STRIPE_WIDTH, world_size))
    world = builder()

    camera = Camera.from_world(world, width, height)
    cam, focal = camera.update_buffer()

    ctx = capi.Context(device=local_rank, flags=S.CREATE_MEGAKERNEL if args.megakernel else (
        S.CREATE_PERSISTENT if args.persistent else (S.CREATE_SINGLE_CHAIN if args.single_chain else 0)))
    ctx.upload_scene(world)
    stats = ctx.scene_stats()
    tile = tiling.tile_for_rank(rank, world_size)
    local_w = tiling.local_width(width, rank, world_size)
    # Two HDR tiles, used alternately: while the tile of step i is gathered (RCCL's own stream), step i + 1
    # already renders into the other one.  N = 1 only ever uses the first.
    tiles = [torch.zeros((height, local_w, 4), dtype=torch.float32, device="cuda") for _ in range(2 if world_size > 1 else 1)]
    hdr = tiles[0]
    ctx.set_output_buffer(hdr.data_ptr(), hdr.numel() * 4)
    stream = torch.cuda.current_stream().cuda_stream
    recv = [[torch.empty_like(hdr) for _ in range(world_size)] if (world_size > 1 and rank == 0) else None for _ in tiles]
    pending = [None for _ in tiles]  # in-flight gather of each tile
    overlap = [True]
    full = None

    render_flags = 0 if (args.in_order or args.single_chain or args.megakernel or args.persistent) else S.RENDER_PIPELINED

    def gather_sync(t):
        """Rehearsal path (gloo, through host memory)."""
        torch.cuda.synchronize()
        host = t.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        return tiling.deinterleave([p.cuda() for p in parts], width) if rank == 0 else None

    def finish_gather(b):
        """Waits (stream-side) for tile b's gather and de-interleaves it on rank 0."""
        nonlocal full
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
            if rank == 0:
                full = tiling.deinterleave(recv[b], width)

    def render_and_gather(i, record=None):
        """Step i: render the rank's stripes into tile i % 2, start its gather (the one data-path collective:
        per-rank RGBA32F stripes to rank 0, RCCL over xGMI), then complete the previous step's."""
        nonlocal full
        b = i % len(tiles)
        finish_gather(b)  # the gather that last read this tile (two steps ago)
        ctx.set_output_buffer(tiles[b].data_ptr(), tiles[b].numel() * 4)
        pc = make_pc(focal, 1, max_bounces, ibl, True)
        if record:
            record[0].record()
        # frames in flight (as the reference keeps frames in flight): the path stages of steps i + 1 and i + 2 overlap
        # step i; the accumulate kernel and the gather stay in stream order
        ctx.render(pc, cam, width, height, tile=tile, frames=spp, stream=stream, flags=render_flags)
        if record:
            record[1].record()
        if world_size == 1:
            full = tiles[b]
        elif rehearse:
            full = gather_sync(tiles[b])
        elif overlap[0]:
            try:
                pending[b] = dist.gather(tiles[b], recv[b], dst=0, async_op=True)
            except (RuntimeError, TypeError, ValueError) as e:  # a backend without async gather: plain gather
                if step_index > 0:
                    raise
                print("bench: async gather unavailable (%s); gathering synchronously" % e, file=sys.stderr)
                overlap[0] = False
                dist.gather(tiles[b], recv[b], dst=0)
                if rank == 0:
                    full = tiling.deinterleave(recv[b], width)
            else:
                finish_gather(1 - b)
        else:
            dist.gather(tiles[b], recv[b], dst=0)
            if rank == 0:
                full = tiling.deinterleave(recv[b], width)

    def drain():
        for b in range(len(tiles)):
            finish_gather((b + 1) % len(tiles))

    step_index = 0

    def step(record=None):
        nonlocal step_index
        render_and_gather(step_index, record)
        step_index += 1

    # deterministic work counters of one launch (outside the timed region)
    ctx.reset_counters(stream)
    pc = make_pc(focal, 1, max_bounces, ibl, True)
    ctx.render(pc, cam, width, height, tile=tile, frames=spp, flags=S.RENDER_COUNT_WORK, stream=stream)
    counters = ctx.counters(stream).as_dict()
    bytes_per_launch = algorithmic_bytes(counters, stats)
    stage_bytes = [algorithmic_bytes(ctx.stage_counters(i, stream).as_dict(), stats) for i in range(4)]
    # Per-launch hipEvents (on the launch streams) cost ~35 us per render - 1 % of a full frame, 5 % of a rank's
    # share at N = 8 - and only one render's are kept by the library: they are switched on for the warm-up
    # (same code path exercised) and for ONE timed step (`timed_step` below), whose per-launch durations the roofline
    # object uses; the step time itself comes from the two events around every step.
    ctx.set_kernel_timing(True)

    # Device wake-up: the clocks of an idle MI355X take ~40 ms of load to ramp (the first ten 2.5 ms steps after idle
    # run 5-10 % slower, in every pipeline mode).  A renderer is past that after its first frames; so that the W
    # warm-up steps and the K timed steps measure the steady state whatever W the caller picks, up to PREHEAT_STEPS steps
    # run first (fewer for the very large configurations; the same count on every rank: they gather).  Untimed, reported as `preheat_steps`; the timed region
    # is untouched.
    preheat_steps = max(2, min(PREHEAT_STEPS, int(round(4e8 / (width * height * spp / world_size)))))
    for _ in range(preheat_steps):
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # per-launch events on ONE step: with frames in flight the third-last (two steps follow it, so its launches share
    # the GPU like any steady-state step's; the last steps drain the pipeline and run faster), else the last
    timed_step = args.steps - 3 if (render_flags and args.steps >= 4) else args.steps - 1
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctx.set_kernel_timing(i == timed_step)
        # kernel-only time: events on the stream the render kernels are launched on
        step((starts[i], stops[i]))
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, stops)) / max(1, args.steps)
    if os.environ.get("PROSPER_BENCH_STEP_TIMES") and rank == 0:
        print("step ms: " + " ".join("%.3f" % (starts[i].elapsed_time(starts[i + 1])) for i in range(args.steps - 1)), file=sys.stderr)
    # per-kernel split of the LAST timed step, from the hipEvents recorded around every launch
    _, per_kernel = ctx.last_render_timing()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        paths_per_step = width * height * spp
        ms_per_step = elapsed * 1e3 / args.steps
        pass_achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        kernels = {}
        for i, (name, (sum_ms, launches)) in enumerate(per_kernel.items()):
            if launches:
                per_launch_ms = sum_ms / launches
                per_launch_bytes = stage_bytes[i] / launches
                kernels[name] = {
                    "launches_per_step": launches, "ms_per_launch": per_launch_ms,
                    "algorithmic_bytes_per_launch": per_launch_bytes,
                    "achieved_GBps": per_launch_bytes / (per_launch_ms * 1e-3) / 1e9,
                }
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
        achieved = kernels[dominant]["achieved_GBps"]
        # with two concurrent launch chains the per-launch durations overlap in time: the share of the step's
        # device time a launch accounts for is its duration scaled by (device time per step) / (sum of durations)
        summed_ms = sum(k["ms_per_launch"] * k["launches_per_step"] for k in kernels.values())
        exclusive_scale = min(1.0, kernel_ms / summed_ms) if summed_ms > 0 else 1.0
        result = {
            "metric": "Mpaths/s",
            "value": paths_per_step * args.steps / elapsed / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "frames_per_s": 1e3 / ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "width": width,
                "height": height,
                "spp": spp,
                "max_bounces": max_bounces,
                "triangles": int(stats.triangleCount),
                "bvh_nodes": int(stats.nodeCount),
                "parallelism": "image stripes x%d%s" % (world_size, " + RCCL gather" if world_size > 1 else ""),
                "pipeline": "megakernel" if args.megakernel else ("persistent" if args.persistent else (
                    "wavefront, 1 launch chain" if args.single_chain else (
                        "wavefront, 3 frames in flight (one launch chain each)" if render_flags
                        else "wavefront, 2 concurrent launch chains"))),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the PMC passes were taken on whole-batch launches of a whole frame (N = 1; the pipelined default and
                # --single-chain launch that shape): no figure for half-batch launches or for a rank's share
                "traffic": measured_traffic(args.config, dominant) if world_size == 1 and (
                    render_flags or args.single_chain) else None,
                "issue": measured_issue(args.config, dominant),
                "algorithmic_bytes_per_launch": kernels[dominant]["algorithmic_bytes_per_launch"],
                "kernel_ms": kernels[dominant]["ms_per_launch"],
                "kernel_ms_source": "hipEvents around every launch of timed step %d of %d" % (timed_step + 1, args.steps),
                "launches_per_step": kernels[dominant]["launches_per_step"],
                # the default pipeline runs two half-batches as two chains of launches on two streams: a launch's
                # duration (hipEvents, = rocprofv3's) includes the time it shares the GPU with the other chain's
                # launch, so sum(kernel time) > wall time and `frac` is a per-launch, not a whole-GPU, figure
                "concurrent_chains": 1 if (args.single_chain or args.megakernel or args.persistent) else 2,
                "frames_in_flight": 3 if render_flags else 1,
                "preheat_steps": preheat_steps,
                "kernel_ms_exclusive": kernels[dominant]["ms_per_launch"] * exclusive_scale,
                "frac_exclusive": achieved / exclusive_scale / HBM_PEAK_GBS,
            },
            "kernels": kernels,
            "whole_pass": {
                "algorithmic_bytes_per_step": bytes_per_launch,
                "device_ms_per_step": kernel_ms,
                "achieved_GBps": pass_achieved,
                "frac_of_hbm_peak": pass_achieved / HBM_PEAK_GBS,
                "bytes_per_path": bytes_per_launch / max(1, counters["paths"]),
            },
            "counters": counters,
            "mean_radiance": float(full[..., :3].mean().item()),
        }
        if world_size == 1:
            result["tone_map"] = bench_tone_map(ctx, torch, width, height, stream)
            result["restir_di_trace"] = bench_restir_di(ctx, torch, cam, focal, width, height, stream)
        if world_size == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(world, cam, focal, width, height, spp, max_bounces, ibl)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))

    ctx.set_output_buffer(0, 0)
    ctx.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
