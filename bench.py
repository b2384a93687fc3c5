#!/usr/bin/env python3
"""bench.py — Mpaths/s and HDR frames/s of the path-tracing reference pass on MI355X.

A step = one pass of the hot path over one batch: the 1920x1080 S-cornell frame accumulated to 8 spp with maxBounces 4
(BASELINE.json configs[1]), inputs resident in HBM.  With N > 1 GPUs the image is cut into interleaved
16-pixel stripes, one stripe set per rank, and the per-rank RGBA32F tiles are gathered to rank 0 inside the timed
region by the PRODUCT: prosper_pt_gather_tiles = ncclGather over RCCL/xGMI + a HIP de-interleave kernel on the root
(strong scaling: the total work is fixed).  torch.distributed only carries the rendezvous (communicator id, barrier,
max-over-ranks of the wall time).  Steps are enqueued back to back with up to three frames in flight
(PROSPER_PT_RENDER_PIPELINED, prosper's frames-in-flight idea; --in-order for A/B): every step's kernels, its
accumulate and its gather complete inside the timed region, which is bracketed by barrier + synchronize.

The LAST stdout line is ONE JSON object of under 4 KB (tests/test_bench_contract.py): the contract fields, a numbers-only
`roofline`, `cpu_baseline`, and per sub-configuration (`configs`: C1, C3, C4, FlightHelmet, the C5 rank share) ms/step,
Mpaths/s, the dominant kernel's fractions and the CPU tracer's rate.  Everything else - per-kernel tables, counters, notes,
the full roofline objects - goes to bench_detail.json (next to this file, and under gpurun_out/).  What the detail file
carries besides the contract fields:
  roofline   the roof that binds the dominant kernel: VALU issue.  achieved = wave-instructions/s of that kernel
             (SQ_INSTS_VALU per launch, measured IN THIS RUN by rocprofv3 child passes over the same launch shape, /
             the kernel's exclusive share of the step time); peak = the guide's 1024 SIMDs x 2.4 GHz / 2 cycles, the
             chip's measured v_fma_f32 rate (profiles/r02_valu_calibration.json) beside it; frac = the USEFUL fraction =
             achieved / peak x lane utilisation; hbm = measured (PMC) and algorithmic bytes over that time
             against 8 TB/s, with cache_served where the byte model prices bytes the caches serve; work_normalised =
             issue slots per node visit / triangle test / any-hit call against their static cost (profiles/r04_unit_costs.json)
  single_sample_frames, rank_share   one render per 1-spp frame as RtReference::record makes them; one rank's stripes of
             the frame of an 8-rank job, alone on this GPU
  configs    C3 (+ the C5 rank share), C4 and the reference's bundled FlightHelmet, each with ms/step, Mpaths/s, the same
             roofline object, per-kernel table, counters, upload and BVH build time
  cpu_baseline  the oracle (a scalar port) on this box's host cores, bounded sample

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: starts the N ranks itself as a child torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402

from prosper_amd import capi, scenes, structs as S, tiling  # noqa: E402
from prosper_amd.rt_reference import Camera  # noqa: E402

import pmc_tools  # noqa: E402  (scripts/: measurement tooling)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
STRIPE_WIDTH = tiling.STRIPE_WIDTH
PREHEAT_STEPS = 24

CONFIGS = {
    # name: (workload, scene builder, width, height, spp, maxBounces, ibl)
    "c2": ("S-cornell 1920x1080 8spp maxBounces 4", lambda: scenes.cornell(), 1920, 1080, 8, 4, False),
    "c3": ("S-sponza-class 1920x1080 8spp maxBounces 4 IBL", lambda: scenes.sponza_class(), 1920, 1080, 8, 4, True),
    "c4": ("S-sponza-class + 1024 lights + foliage 1920x1080 8spp maxBounces 4 IBL",
           lambda: scenes.sponza_class(lights=True, foliage=True), 1920, 1080, 8, 4, True),
    # BASELINE.json configs[4]: meant for 8 GPUs (a rank then renders 1/8 of the stripes); runs on one as well
    "c5": ("S-sponza-class 3840x2160 64spp maxBounces 4 IBL", lambda: scenes.sponza_class(), 3840, 2160, 64, 4, True),
    "c1": ("S-cornell 256x256 1spp maxBounces 1", lambda: scenes.cornell(), 256, 256, 1, 1, False),
    # the reference's one bundled asset (src/main.cpp:32-33), from the packed fixture (tests/golden/flight_helmet.npz)
    "helmet": ("FlightHelmet 1920x1080 8spp maxBounces 4 IBL", lambda: helmet_world(), 1920, 1080, 8, 4, True),
    # the same with every texture at the 2048 x 2048 texels the asset ships with (replicated fixture texels: the footprint
    # of prosper's own load, 252 MB at level 0)
    "helmet2k": ("FlightHelmet, 2048^2 textures, 1920x1080 8spp maxBounces 4 IBL", lambda: helmet_world(2048), 1920, 1080, 8, 4, True),
}


def helmet_world(texture_size=None):
    from prosper_amd import flight_helmet
    return flight_helmet.load_fixture(texture_size=texture_size)


def algorithmic_bytes(c, stats):
    """B_alg of SURVEY 8d from the deterministic work counters of one launch."""
    tri_long = c["triangleTests"] - c["shortIndexTriangleTests"]
    hits = c["closestHits"] + c["anyHitCalls"]
    short_share = (c["shortIndexHits"] / hits) if hits else 0.0
    chit = c["closestHits"] * (314.0 * short_share + 320.0 * (1.0 - short_share))
    ahit = c["anyHitCalls"] * (138.0 * short_share + 144.0 * (1.0 - short_share))
    return (c["nodeVisits"] * float(stats.nodeBytes) + c["shortIndexTriangleTests"] * 30.0 + tri_long * 36.0 + chit +
            ahit + c["lightSamples"] * 36.0 + c["spotLightSamples"] * 52.0 + c["skyLookups"] * 32.0 +
            c["pixelsWritten"] * 16.0 + c["historyReads"] * 16.0)


VALU_PEAK_GUIDE = 1024 * 2.4e9 / 2.0  # wave-instructions/s: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU op (MI355X_MICROARCH.md)
VALU_PEAK_GUIDE_SOURCE = "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md, cycle constants)"


def valu_peak_calibrated():
    """The chip's MEASURED issue rate of independent v_fma_f32 (wave-instructions/s, best over the occupancies the
    calibration ran): profiles/r02_valu_calibration.json, scripts/valu_calibration.hip.  Reported beside the guide's
    figure, never instead of it."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_valu_calibration.json")) as f:
            rows = json.load(f)["results"]
        best = max(r["wave_insts_per_s_chip"] for r in rows if r["inst"] in ("fma_f32", "v_fma_f32"))
        return best, "profiles/r02_valu_calibration.json (v_fma_f32, best occupancy; the clock drops to 2.0-2.3 GHz under that load)"
    except (OSError, KeyError, ValueError):
        return None, "no calibration file"


def unit_costs():
    """Static VALU instructions per unit of work (scripts/unit_costs.py over scripts/unit_costs.hip ->
    profiles/r04_unit_costs.json): the yardstick of the work-normalised figures."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_unit_costs.json")) as f:
            d = json.load(f)
        return {k: v["valu"] for k, v in d["units"].items()}, "profiles/r04_unit_costs.json (kernel sources sha16 %s)" % d.get("kernel_source_sha16")
    except (OSError, KeyError, ValueError):
        return None, "profiles/r04_unit_costs.json is missing"


def priced_work(c, unit):
    """Lane-instructions the traversal work counters of one kernel stage are worth at the static unit costs: what the
    stage would issue if every lane of every wave instruction did useful work and no scheduling code existed.  The unit
    costs count both sides of each branch, so this is an upper estimate of the work (the efficiency built on it is
    optimistic); being work / issued slots it cannot rise by issuing more instructions."""
    finishes = c["closestHits"] + c["anyHitCalls"]  # triangles a ray went through (at least these ran the long half of the test)
    return (c["paths"] * unit["camera_ray"] + c["nodeVisits"] * unit["node_visit_closest"] +
            c["triangleTests"] * unit["triangle_edge_functions"] + finishes * unit["triangle_finish"] +
            c["anyHitCalls"] * unit["any_hit_settle"] + c.get("anyHitTexelFetches", 0) * unit["any_hit_exact"])


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


def cpu_baseline(world, cam, focal, width, height, spp, max_bounces, ibl, budget_s=15.0, tile=None):
    """The oracle (a scalar port) timed on this box's host cores on a bounded sample: accumulated frames of the same
    workload until the budget is spent (at least one).  `tile` = (rank, ranks): only that rank's stripes."""
    from oracle import binding as oracle
    t_build = time.perf_counter()
    osc = oracle.OracleScene(world, brute_force=False)
    build_s = time.perf_counter() - t_build
    cores = host_cores()
    img = None
    frames = 0
    local_w = width if tile is None else tiling.local_width(width, tile[0], tile[1])
    kw = {} if tile is None else {"tile": tiling.tile_for_rank(tile[0], tile[1])}
    t0 = time.perf_counter()
    while frames < spp:
        pc = make_pc(focal, frames + 1, max_bounces, ibl, frames == 0)
        img, _ = osc.render(pc, cam, width, height, history=img, threads=cores, **kw)
        frames += 1
        elapsed = time.perf_counter() - t0
        if elapsed + elapsed / frames > budget_s:
            break
    elapsed = time.perf_counter() - t0
    osc.close()
    return {
        "value": local_w * height * frames / elapsed / 1e6,
        "unit": "Mpaths/s",
        "cores": cores,
        "kind": "port",
        "sample_frames": frames,
        "sample_seconds": elapsed,
        "hierarchy_build_seconds": build_s,
        "sample": "%d of %d spp of the same %dx%d frame%s (%.1f s; 16x16-pixel tiles from a shared counter, one worker per "
                  "hardware thread, -O3 x86-64-v3, fp contraction off, the oracle's own BVH2)" % (
                      frames, spp, width, height, "" if tile is None else ", the stripes of rank %d of %d" % tile, elapsed),
    }


def rounded(x, digits=5):
    """Floats to `digits` significant digits (the slim line), everything else as it is."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: rounded(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rounded(v, digits) for v in x]
    return x


def slim_roofline(r):
    """The numbers of a roofline object (roofline_object): no notes, no sources."""
    if not r:
        return None
    hbm = r.get("hbm") or {}
    wn = r.get("work_normalised") or {}
    out = {k: r.get(k) for k in ("bound", "kernel", "unit", "peak", "peak_calibrated", "kernel_ms", "kernel_ms_alone",
                                 "launches_per_step", "share_of_step", "achieved", "issue_frac", "lane_util", "frac", "traffic")}
    out["efficiency"] = wn.get("efficiency")
    out["pmc"] = "live" if str(r.get("pmc_source", "")).startswith("rocprofv3") else ("committed" if str(r.get("pmc_source", "")).startswith("profiles/") else "none")
    out["hbm"] = {k: hbm.get(k) for k in ("peak_GBps", "algorithmic_bytes_per_launch", "algorithmic_frac", "measured_bytes_per_launch",
                                          "measured_frac", "measured_over_algorithmic", "cache_served")}
    ws = r.get("whole_step") or {}
    out["whole_step"] = {k: ws.get(k) for k in ("issue_frac", "useful_frac", "hbm_measured_frac")}
    return out


def slim_config(c):
    """One sub-configuration of the slim line."""
    if "error" in c:
        return {"error": c["error"][:120]}
    r = c.get("roofline") or {}
    hbm = r.get("hbm") or {}
    out = {"ms_per_step": c.get("ms_per_step"), "Mpaths_per_s": c.get("Mpaths_per_s"), "kernel": r.get("kernel"),
           "kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"), "issue_frac": r.get("issue_frac"), "lane_util": r.get("lane_util"),
           "efficiency": (r.get("work_normalised") or {}).get("efficiency"),
           "hbm_measured_frac": hbm.get("measured_frac"), "traffic": r.get("traffic"),
           "cpu_Mpaths_per_s": (c.get("cpu_baseline") or {}).get("value")}
    if "nearest_hbm_roof" in c:
        out["nearest_hbm_roof"] = {"kernel": c["nearest_hbm_roof"]["kernel"], "frac": c["nearest_hbm_roof"]["frac_of_peak"]}
    if "extrapolated" in c:
        out["extrapolated"] = c["extrapolated"]
    return {k: v for k, v in out.items() if v is not None}


DETAIL_NAME = "bench_detail.json"
SLIM_LIMIT = 4096


def slim_line(result, detail_path=None):
    """The ONE stdout line (under SLIM_LIMIT bytes): contract fields + numbers; the rest lives in bench_detail.json."""
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "frames_per_s",
                                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = result["config"]
    line["config"] = {k: cfg.get(k) for k in ("workload", "width", "height", "spp", "max_bounces", "triangles", "bvh_nodes",
                                              "parallelism", "pipeline")}
    line["roofline"] = slim_roofline(result.get("roofline"))
    if "cpu_baseline" in result:
        cb = result["cpu_baseline"]
        sample = ("%d of %d spp of the same frame, %.1f s" % (cb["sample_frames"], cfg["spp"], cb["sample_seconds"])
                  if "sample_frames" in cb else str(cb.get("sample", ""))[:60])
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": sample}
        line["gpu_over_cpu"] = result.get("gpu_over_cpu")
    for k in ("ranks_seen", "gather_ms", "mean_radiance"):
        if k in result:
            line[k] = result[k]
    for k in ("single_sample_frames", "rank_share"):
        if k in result and "ms_per_frame" in result[k]:
            line[k] = {"ms": result[k]["ms_per_frame"]}
        elif k in result and "ms_per_step" in result[k]:
            line[k] = {"ms": result[k]["ms_per_step"], "bound_on_scaling_efficiency": result[k].get("bound_on_scaling_efficiency")}
    if result.get("configs"):
        line["configs"] = {name: slim_config(c) for name, c in result["configs"].items()}
    line["detail"] = detail_path or DETAIL_NAME
    line = rounded(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= SLIM_LIMIT:  # never again a line the driver cannot parse: drop the optional parts, largest first
        for k in ("configs", "single_sample_frames", "rank_share"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < SLIM_LIMIT:
                break
    return text


def write_detail(result, path=None):
    """The full result: to `path` when given (tests), else bench_detail.json next to this file and a copy under gpurun_out/
    (the scratch directory a GPU box's run sends back)."""
    targets = [path] if path else [os.path.join(ROOT, DETAIL_NAME), os.path.join(ROOT, "gpurun_out", DETAIL_NAME)]
    for t in targets:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(t)), exist_ok=True)
            with open(t, "w") as f:
                json.dump(result, f, indent=1)
                f.write("\n")
        except OSError as e:
            print("bench: could not write %s (%s)" % (t, e), file=sys.stderr)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process - which has not touched the GPU and never will -
    starts `python -m torch.distributed.run --nproc-per-node N bench.py <the same arguments>` as a CHILD, relays its
    stdout (rank 0's one line) and exits with its code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last = None
    for raw in child.stdout:
        text = raw.strip()
        if text.startswith("{") and text.endswith("}"):
            last = text
        elif text:
            print(text, file=sys.stderr)  # (launcher chatter must not share stdout with the one line)
    rc = child.wait()
    if last is not None:
        print(last)
    sys.stdout.flush()
    raise SystemExit(rc if rc != 0 else (0 if last is not None else 1))


def bench_tone_map(ctx, torch, width, height, stream, repeats=50):
    """The step after the path (not part of `value`): blit + tone_map.comp in one kernel over
    the frame just rendered; 16 B read + 4 B written per pixel, the 442 KB LUT stays in L2."""
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


def bench_restir_di(ctx, torch, world, cam, focal, width, height, stream, repeats=30):
    """A second client of the traversal (not part of `value`): the ReSTIR-DI trace pass over a
    G-buffer made from the product's own debug views of the primary hits, one random light per pixel."""

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


def kernel_table(per_kernel, stage_bytes, ms_per_step, pmc, alone=None, stage_counters=None):
    """Per kernel: launches, the raw per-launch duration inside the pipelined step (hipEvents; launches of the frames in
    flight overlap and queue behind one another, so these sum to 2-3x the step), the duration of the same launch ALONE on
    the GPU (one chain, in order: `alone`), and its EXCLUSIVE share of the step = alone x (ms_per_step / sum of the alone
    durations of the step's launches): the shares add up to the step time.  (Without `alone`: raw x ms_per_step / sum
    raw.)  Then algorithmic bytes and - when PMC figures for this launch shape exist - VALU wave-instructions, lane
    utilisation and HBM bytes, priced against the exclusive time; and the work-normalised figures (issue slots per unit
    of work against the static cost of that work)."""
    unit, _ = unit_costs()
    basis = {}
    for name, (sum_ms, launches) in per_kernel.items():
        if launches:
            basis[name] = (alone[name] if alone and name in alone else sum_ms / launches)
    total = sum(basis[name] * per_kernel[name][1] for name in basis)
    scale = min(1.0, ms_per_step / total) if total > 0 else 1.0
    kernels = {}
    for i, (name, (sum_ms, launches)) in enumerate(per_kernel.items()):
        if not launches:
            continue
        k = {"launches_per_step": launches, "ms_per_launch_raw": sum_ms / launches,
             "ms_per_launch_alone": alone.get(name) if alone else None, "ms_per_launch": basis[name] * scale,
             "algorithmic_bytes_per_launch": stage_bytes[i] / launches}
        p = (pmc or {}).get("kernels", {}).get(name)
        if p:
            t = k["ms_per_launch"] * 1e-3
            k.update({
                "valu_insts_per_launch": p["valu_insts_per_launch"],
                "valu_wave_insts_per_s": p["valu_insts_per_launch"] / t,
                # SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) reads 1.036 for the fully converged wf_accumulate: the two
                # counters do not weigh every instruction alike.  Clamped; the raw ratio is kept beside it.
                "lane_util": min(1.0, p["lane_util"]),
                "lane_util_raw": p["lane_util"],
                "hbm_bytes_per_launch": p["hbm_bytes_per_launch"],
                "hbm_GBps": p["hbm_bytes_per_launch"] / t / 1e9,
                "traffic_over_algorithmic": p["hbm_bytes_per_launch"] / k["algorithmic_bytes_per_launch"]
                if k["algorithmic_bytes_per_launch"] else None,
            })
            c = stage_counters[i] if stage_counters else None
            if unit and c:
                slots = 64.0 * p["valu_insts_per_launch"] * launches  # lane issue slots of the stage per step
                if name == "wf_shade":
                    # (no static yardstick: the straight-line count of a hit's shading holds every texture-addressing,
                    # light-type and lobe branch at once, three times what a lane executes)
                    work, units_done, what = None, c["closestHits"], "closest hit"
                elif name == "wf_accumulate":
                    work, units_done, what = None, c["pixelsWritten"], "texel-frame"
                else:
                    work, units_done, what = priced_work(c, unit), c["nodeVisits"] + c["triangleTests"] + c["anyHitCalls"], "node visit + triangle test + any-hit call"
                k["work_normalised"] = {
                    "unit": what,
                    "units_per_step": units_done,
                    "issue_slots_per_unit": slots / units_done if units_done else None,
                    "active_lane_insts_per_unit": slots * min(1.0, p["lane_util"]) / units_done if units_done else None,
                    "static_lane_insts_of_the_work": work,
                    "efficiency": (work / slots) if (work is not None and slots) else None,
                }
        kernels[name] = k
    return kernels, scale


def time_alone(alone_ctx, world, cam, pc, width, height, spp, stream, torch):
    """Per-launch durations with every launch alone on the GPU: the same workload on a one-chain, in-order context."""
    alone_ctx.upload_scene(world)
    buf = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
    alone_ctx.set_output_buffer(buf.data_ptr(), buf.numel() * 4)
    alone_ctx.set_kernel_timing(False)
    for _ in range(3):
        alone_ctx.render(pc, cam, width, height, frames=spp, stream=stream)
    acc = {}
    repeats = 3
    alone_ctx.set_kernel_timing(True)
    for _ in range(repeats):
        alone_ctx.render(pc, cam, width, height, frames=spp, stream=stream)
        _, per = alone_ctx.last_render_timing()
        for name, (sum_ms, launches) in per.items():
            if launches:
                acc[name] = acc.get(name, 0.0) + sum_ms / launches / repeats
    alone_ctx.set_kernel_timing(False)
    torch.cuda.synchronize()
    alone_ctx.set_output_buffer(0, 0)
    return acc


def roofline_object(kernels, pmc_source, ms_per_step, timed_note, step_algorithmic_bytes):
    """The roof that binds the dominant kernel is vector-ALU issue, so `bound` is "valu": `achieved` =
    wave-instructions/s of that kernel, `peak` = the guide's figure (the calibrated, lower one beside it), `frac` = the
    USEFUL fraction = achieved / peak x lane utilisation: issue slots whose lanes did something.  The contract's HBM
    figures sit in `hbm`: measured (PMC) and algorithmic (SURVEY 8d byte model) bytes over the kernel's time against
    8 TB/s; `cache_served` says when the byte model prices bytes that LDS / L2 / the Infinity Cache serve (its rate would
    exceed the HBM peak at step level), i.e. when the algorithmic fraction is not a statement about HBM."""
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
    d = kernels[dominant]
    cal, cal_source = valu_peak_calibrated()
    t = d["ms_per_launch"] * 1e-3
    step_alg_GBps = step_algorithmic_bytes / (ms_per_step * 1e-3) / 1e9
    r = {
        "bound": "valu",
        "kernel": dominant,
        "unit": "Gwave-inst/s",
        "peak": VALU_PEAK_GUIDE / 1e9,
        "peak_source": VALU_PEAK_GUIDE_SOURCE,
        "peak_calibrated": cal / 1e9 if cal else None,
        "peak_calibrated_source": cal_source,
        "kernel_ms": d["ms_per_launch"],
        "kernel_ms_alone": d["ms_per_launch_alone"],
        "kernel_ms_raw": d["ms_per_launch_raw"],
        "kernel_ms_source": timed_note,
        "launches_per_step": d["launches_per_step"],
        "share_of_step": d["ms_per_launch"] * d["launches_per_step"] / ms_per_step,
        "pmc_source": pmc_source,
    }
    hbm = {"peak_GBps": HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
           "algorithmic_GBps": d["algorithmic_bytes_per_launch"] / t / 1e9,
           "algorithmic_frac": d["algorithmic_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS,
           "step_algorithmic_bytes": step_algorithmic_bytes,
           "step_algorithmic_GBps": step_alg_GBps,
           "cache_served": bool(step_alg_GBps > HBM_PEAK_GBS or d["algorithmic_bytes_per_launch"] / t / 1e9 > HBM_PEAK_GBS),
           "note": "algorithmic = SURVEY 8d unit costs x the work counters, zero reuse assumed; measured = 2 x FETCH_SIZE + "
                   "WRITE_SIZE (PMC, gfx950 correction); both over the kernel's exclusive time"}
    if "valu_wave_insts_per_s" in d:
        total_insts = sum(k["valu_insts_per_launch"] * k["launches_per_step"] for k in kernels.values() if "valu_insts_per_launch" in k)
        total_hbm = sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in kernels.values() if "hbm_bytes_per_launch" in k)
        issue = d["valu_wave_insts_per_s"] / VALU_PEAK_GUIDE
        hbm.update({"measured_bytes_per_launch": d["hbm_bytes_per_launch"], "measured_GBps": d["hbm_GBps"],
                    "measured_frac": d["hbm_GBps"] / HBM_PEAK_GBS, "measured_over_algorithmic": d["traffic_over_algorithmic"]})
        r.update({
            "achieved": d["valu_wave_insts_per_s"] / 1e9,
            "issue_frac": issue,
            "issue_frac_of_calibrated_peak": d["valu_wave_insts_per_s"] / cal if cal else None,
            "lane_util": d["lane_util"],
            "frac": issue * d["lane_util"],
            "frac_is": "useful fraction = achieved / peak x lane_util (issue slots whose lanes worked); against the calibrated peak: "
                       + ("%.3f" % (d["valu_wave_insts_per_s"] / cal * d["lane_util"]) if cal else "n/a"),
            "work_normalised": d.get("work_normalised"),
            "traffic": d["hbm_bytes_per_launch"],
            "hbm": hbm,
            # the whole step against the same roofs: every kernel's instructions and bytes over the step time
            "whole_step": {
                "valu_Gwave_insts_per_s": total_insts / (ms_per_step * 1e-3) / 1e9,
                "issue_frac": total_insts / (ms_per_step * 1e-3) / VALU_PEAK_GUIDE,
                "issue_frac_of_calibrated_peak": total_insts / (ms_per_step * 1e-3) / cal if cal else None,
                "useful_frac": sum(k["valu_insts_per_launch"] * k["launches_per_step"] * k["lane_util"] for k in kernels.values()
                                   if "valu_insts_per_launch" in k) / (ms_per_step * 1e-3) / VALU_PEAK_GUIDE,
                "hbm_bytes": total_hbm,
                "hbm_measured_frac": total_hbm / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hbm_algorithmic_frac": step_alg_GBps / HBM_PEAK_GBS,
            },
        })
    else:
        r.update({"achieved": None, "frac": None, "lane_util": None, "traffic": None, "hbm": hbm})
    return r


def time_single_sample_frames(ctx, torch, cam, focal, width, height, max_bounces, ibl, stream, frames=96):
    """One prosper_pt_render per accumulated frame, as RtReference::record makes them (RtReference.cpp:161-383: one path
    per pixel per frame, the frame index advancing, history read back every frame), three frames in flight: what an
    interactive prosper session sees, beside the batched 8-spp steps of `value`."""
    def run(n, first):
        for f in range(n):
            pc = make_pc(focal, 1 + ((first + f) % 4000), max_bounces, ibl, first + f == 0)
            ctx.render(pc, cam, width, height, frames=1, stream=stream, flags=S.RENDER_PIPELINED)
    run(24, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(frames, 24)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    return {"frames": frames, "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "Mpaths_per_s": width * height / ms / 1e3,
            "what": "one render call per 1-spp frame (frame index advancing, history accumulated), 3 frames in flight"}


def time_rank_share(ctx, torch, cam, focal, width, height, spp, max_bounces, ibl, stream, ranks, rank, steps, full_ms=None):
    """What ONE rank of an N-rank job would do per step - its stripes of the frame, no gather - timed on this GPU: the
    per-rank side of the strong-scaling figure while no multi-GPU node is at hand."""
    tile = tiling.tile_for_rank(rank, ranks)
    lw = tiling.local_width(width, rank, ranks)
    buf = torch.zeros((height, lw, 4), dtype=torch.float32, device="cuda")
    ctx.set_output_buffer(buf.data_ptr(), buf.numel() * 4)
    pc = make_pc(focal, 1, max_bounces, ibl, True)
    for _ in range(max(3, steps // 2)):  # at least one render per workspace slot: their first use allocates
        ctx.render(pc, cam, width, height, tile=tile, frames=spp, stream=stream, flags=S.RENDER_PIPELINED)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.render(pc, cam, width, height, tile=tile, frames=spp, stream=stream, flags=S.RENDER_PIPELINED)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    ctx.set_output_buffer(0, 0)
    out = {"ranks": ranks, "rank": rank, "local_width": lw, "steps": steps, "ms_per_step": ms,
           "Mpaths_per_s_of_the_share": lw * height * spp / ms / 1e3,
           "what": "rank %d of %d: its 16-pixel stripes of the %dx%d frame at %d spp, 3 frames in flight, no gather" % (rank, ranks, width, height, spp)}
    if full_ms:
        out["ideal_ms"] = full_ms / ranks
        out["bound_on_scaling_efficiency"] = (full_ms / ranks) / ms
    return out


def time_config(ctx, alone_ctx, torch, name, world, width, height, spp, max_bounces, ibl, steps, warmup, stream, render_flags, pmc,
                cpu_budget_s=0.0):
    """A sub-configuration on one GPU (C3, C4): upload, one counted render, preheat, `steps` timed pipelined steps with
    per-launch events on one of them."""
    t0 = time.perf_counter()
    ctx.upload_scene(world)
    upload_wall = time.perf_counter() - t0
    stats = ctx.scene_stats()
    cam, focal = Camera.from_world(world, width, height).update_buffer()
    pc = make_pc(focal, 1, max_bounces, ibl, True)
    hdr = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
    ctx.set_output_buffer(hdr.data_ptr(), hdr.numel() * 4)
    ctx.reset_counters(stream)
    ctx.render(pc, cam, width, height, frames=spp, flags=S.RENDER_COUNT_WORK, stream=stream)
    counters = ctx.counters(stream).as_dict()
    stage_counters = [ctx.stage_counters(i, stream).as_dict() for i in range(4)]
    stage_bytes = [algorithmic_bytes(c, stats) for c in stage_counters]
    ctx.set_kernel_timing(False)
    preheat = max(2, min(PREHEAT_STEPS, int(round(4e8 / (width * height * spp)))))
    for _ in range(preheat + warmup):
        ctx.render(pc, cam, width, height, frames=spp, stream=stream, flags=render_flags)
    torch.cuda.synchronize()
    timed_step = steps - 3 if (render_flags and steps >= 4) else steps - 1
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.set_kernel_timing(i == timed_step)
        ctx.render(pc, cam, width, height, frames=spp, stream=stream, flags=render_flags)
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) * 1e3 / steps
    _, per_kernel = ctx.last_render_timing()
    ctx.set_kernel_timing(False)
    alone = time_alone(alone_ctx, world, cam, pc, width, height, spp, stream, torch) if alone_ctx is not None else None
    kernels, _ = kernel_table(per_kernel, stage_bytes, ms_per_step, pmc, alone, stage_counters)
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
    out = {
        "ms_per_step": ms_per_step, "Mpaths_per_s": width * height * spp / ms_per_step / 1e3, "steps": steps,
        "triangles": int(stats.triangleCount), "bvh_nodes": int(stats.nodeCount),
        "upload_ms": stats.uploadSeconds * 1e3, "bvh_build_ms": stats.bvhBuildSeconds * 1e3,
        "texture_upload_ms": stats.textureSeconds * 1e3, "upload_wall_ms": upload_wall * 1e3,
        "scene_device_MB": stats.deviceBytes / 1e6,
        "alpha_triangles": int(stats.alphaTriangleCount), "alpha_bound_bytes": int(stats.alphaBoundBytes),
        "counters": counters,
        "dominant_kernel": dominant, "kernels": kernels,
        "algorithmic_bytes_per_step": algorithmic_bytes(counters, stats),
        "pmc_source": pmc["source"] if pmc else None,
        "mean_radiance": float(hdr[..., :3].mean().item()),
    }
    if pmc and "hbm_bytes_per_launch" in kernels[dominant]:
        out["measured_traffic_ratio"] = {k: v.get("traffic_over_algorithmic") for k, v in kernels.items()}
        out["hbm_bytes_per_step"] = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in kernels.values() if "hbm_bytes_per_launch" in v)
        out["lane_util"] = {k: v.get("lane_util") for k, v in kernels.items()}
    # the same roofline object as the headline's, for this configuration's dominant kernel, and the kernel nearest
    # the HBM roof beside it (C3's wf_shade is the one kernel of the pass that waits for memory)
    out["roofline"] = roofline_object(
        kernels, out["pmc_source"], ms_per_step,
        "each launch alone on the GPU x ms_per_step / sum of the alone durations (as in the headline)",
        out["algorithmic_bytes_per_step"])
    hbm = {k: v["hbm_GBps"] for k, v in kernels.items() if "hbm_GBps" in v and k != "wf_accumulate"}
    if hbm:
        k = max(hbm, key=hbm.get)
        out["nearest_hbm_roof"] = {"kernel": k, "GBps": hbm[k], "frac_of_peak": hbm[k] / HBM_PEAK_GBS,
                                   "traffic_over_algorithmic": kernels[k].get("traffic_over_algorithmic")}
    ctx.set_output_buffer(0, 0)
    if cpu_budget_s:
        # the CPU tracer beside every configuration - a 1-spp sample of the same frame on the host cores
        try:
            out["cpu_baseline"] = cpu_baseline(world, cam, focal, width, height, spp, max_bounces, ibl, budget_s=cpu_budget_s)
            out["gpu_over_cpu"] = out["Mpaths_per_s"] / out["cpu_baseline"]["value"]
        except Exception as e:
            out["cpu_baseline_error"] = str(e)[:300]
    if name == "c3":
        # BASELINE C5 (this scene, 3840x2160, 64 spp, 8 GPUs) as one of its ranks sees it
        try:
            w5, h5 = 3840, 2160
            cam5, focal5 = Camera.from_world(world, w5, h5).update_buffer()
            share = time_rank_share(ctx, torch, cam5, focal5, w5, h5, 64, max_bounces, ibl, stream, 8, 3, 3)
            if cpu_budget_s:
                # the CPU tracer on the same rank's stripes, 1 of the 64 spp (a 64-spp sample would take minutes)
                share["cpu_baseline"] = cpu_baseline(world, cam5, focal5, w5, h5, 64, max_bounces, ibl, budget_s=cpu_budget_s, tile=(3, 8))
            out["c5_rank_share"] = share
        except Exception as e:
            out["c5_rank_share"] = {"error": str(e)[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 child passes (roofline then quotes the "
                    "committed profile if it matches the kernel sources, else null)")
    ap.add_argument("--pmc-budget", type=float, default=240.0, help="seconds all rocprofv3 child passes together may take")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU tracing for the headline's cpu_baseline")
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json here and under gpurun_out/)")
    ap.add_argument("--no-subconfigs", action="store_true", help="skip the C3 / C4 / FlightHelmet sub-objects")
    ap.add_argument("--no-extras", action="store_true", help="skip the tone-map and ReSTIR-DI legs")
    ap.add_argument("--subconfigs", default="c1,c3,c4,helmet,helmet2k")
    ap.add_argument("--megakernel", action="store_true", help="use the one-lane-per-pixel kernel (A/B)")
    ap.add_argument("--persistent", action="store_true", help="use the persistent path-regeneration kernel (A/B)")
    ap.add_argument("--single-chain", action="store_true",
                    help="one chain of launches on one stream instead of two overlapping half-batches (A/B)")
    ap.add_argument("--in-order", action="store_true",
                    help="no frames in flight: every step starts after the previous one has finished (A/B)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the ranks as a child process.  Nothing above this line has touched the
        # GPU (the library is loaded lazily, torch is not imported yet), and this process never does.
        width = CONFIGS[args.config][2]
        if not tiling.check_divisible(width, args.gpus):
            raise SystemExit("--gpus %d: %d stripes of %d pixels do not divide over the ranks" % (args.gpus, width // STRIPE_WIDTH, STRIPE_WIDTH))
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world_size, args.gpus))
    wavefront = not (args.megakernel or args.persistent)

    # ---- PMC passes of the headline configuration, as child processes, BEFORE this process touches the GPU ----
    sub_names = [c for c in args.subconfigs.split(",") if c] if (args.config == "c2" and not args.no_subconfigs) else []
    pmc_by_config = {}
    pmc_gave_up = False
    pmc_deadline = time.perf_counter() + args.pmc_budget  # all profiler passes together: bounded, whatever a pass does
    if world_size == 1 and wavefront:
        for name in [args.config] + sub_names:
            if name not in ("c2", "c3", "c4", "c5", "helmet"):  # (helmet2k: timed, not profiled - the kernels are helmet's)
                continue
            got = None
            left = pmc_deadline - time.perf_counter()
            if not args.no_pmc and not pmc_gave_up and left > 20.0:
                try:
                    t0 = time.perf_counter()
                    got = pmc_tools.collect(name, timeout=max(20.0, min(120.0, left / 3.0)))
                    got["source"] = "rocprofv3 --pmc child passes of this run (%.0f s; scripts/pmc_tools.py)" % (time.perf_counter() - t0)
                except Exception as e:  # no profiler, a refused counter, a timeout: the line says so instead of failing
                    print("bench: live PMC passes unavailable for %s (%s)" % (name, str(e)[:300]), file=sys.stderr)
                    # one failure is the box's answer: no further passes are tried (a hanging profiler must not cost the
                    # run a timeout per pass and configuration); the committed profiles serve if they match the sources
                    pmc_gave_up = True
            if got is None:
                got = pmc_tools.load_committed(name)
            pmc_by_config[name] = got
    pmc = pmc_by_config.get(args.config)
    pmc_source = pmc["source"] if pmc else "none (no live pass, no committed profile for these kernel sources)"

    import torch
    import torch.distributed as dist

    # PROSPER_BENCH_REHEARSE=1: every rank on GPU 0, tiles exchanged through host memory over gloo, de-interleaved by
    # the product's kernel: rehearses the N > 1 flow on a one-GPU box (RCCL refuses two ranks on one device); never a
    # measurement.
    rehearse = os.environ.get("PROSPER_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (communicator id, barriers, max-over-ranks): CPU tensors over gloo.  The data plane - the
        # gather of the HDR tiles - is the library's own RCCL communicator (prosper_pt_comm_init).
        # (gloo announces its connections on STDOUT; this program's stdout carries exactly one JSON line)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    workload, builder, width, height, spp, max_bounces, ibl = CONFIGS[args.config]
    if world_size > 1 and not tiling.check_divisible(width, world_size):
        # equal tiles -> the one ncclGather the design names; the grouped send/recv path for uneven stripe counts exists
        # in the library but has never run on more than one GPU, so the bench does not time it
        raise SystemExit("--gpus %d: %d stripes of %d pixels do not divide over the ranks" % (world_size, width // STRIPE_WIDTH, STRIPE_WIDTH))
    world = builder()
    camera = Camera.from_world(world, width, height)
    cam, focal = camera.update_buffer()

    ctx = capi.Context(device=local_rank, flags=S.CREATE_MEGAKERNEL if args.megakernel else (
        S.CREATE_PERSISTENT if args.persistent else (S.CREATE_SINGLE_CHAIN if args.single_chain else 0)))
    t0 = time.perf_counter()
    ctx.upload_scene(world)
    upload_wall = time.perf_counter() - t0
    stats = ctx.scene_stats()
    tile = tiling.tile_for_rank(rank, world_size)
    local_w = tiling.local_width(width, rank, world_size)
    hdr = torch.zeros((height, local_w, 4), dtype=torch.float32, device="cuda")
    ctx.set_output_buffer(hdr.data_ptr(), hdr.numel() * 4)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda") if (rank == 0 and world_size > 1) else None

    gather_mode = "none"
    fallback_group, fallback_recv = None, None
    if world_size > 1 and not rehearse:
        ids = [None]
        ok = 1
        try:
            if rank == 0:
                ids[0] = capi.Context.comm_unique_id()
        except capi.ProsperPtError as e:
            print("bench: prosper_pt_comm_get_unique_id failed (%s)" % e, file=sys.stderr)
        dist.broadcast_object_list(ids, src=0)
        try:
            if ids[0] is None:
                raise capi.ProsperPtError(-6, "rank 0 could not create a communicator id")
            ctx.comm_init(ids[0], rank, world_size)  # ncclCommInitRank: collective over the ranks
        except capi.ProsperPtError as e:
            print("bench: rank %d: prosper_pt_comm_init failed (%s)" % (rank, e), file=sys.stderr)
            ok = 0
        agreed = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 1:
            gather_mode = "prosper_pt_gather_tiles: ncclGather over RCCL + HIP de-interleave on rank 0, on the context's comm stream"
        else:
            # the library's own communicator is not available on this box: torch's RCCL process group carries the tiles,
            # the product's kernel still de-interleaves them (said in config.parallelism)
            ctx.comm_destroy()
            fallback_group = dist.new_group(backend="nccl")
            if rank == 0:
                fallback_recv = [torch.empty_like(hdr) for _ in range(world_size)]
            gather_mode = "FALLBACK: torch.distributed RCCL gather + prosper_pt_deinterleave_tiles (prosper_pt_comm_init failed, see stderr)"
    elif world_size > 1:
        gather_mode = "REHEARSAL: gloo through host memory + prosper_pt_deinterleave_tiles"

    render_flags = 0 if (args.in_order or args.single_chain or not wavefront) else S.RENDER_PIPELINED

    def gather(step_tile):
        """The one data-path collective of step i, enqueued behind its render."""
        if world_size == 1:
            return
        if fallback_group is not None:
            dist.gather(step_tile, fallback_recv, dst=0, group=fallback_group)
            if rank == 0:
                staging = torch.cat([p.reshape(-1) for p in fallback_recv])
                ctx.deinterleave_tiles(staging.data_ptr(), world_size, STRIPE_WIDTH, width, height, full.data_ptr(), stream=stream)
            return
        if not rehearse:
            ctx.gather_tiles(root=0, device_ptr=full.data_ptr() if rank == 0 else None,
                             byte_size=full.numel() * 4 if rank == 0 else 0, stream=stream)
            return
        torch.cuda.synchronize()
        host = step_tile.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        if rank == 0:
            staging = torch.cat([p.reshape(-1) for p in parts]).cuda()
            ctx.deinterleave_tiles(staging.data_ptr(), world_size, STRIPE_WIDTH, width, height, full.data_ptr(), stream=stream)
            torch.cuda.synchronize()

    def step(record=None):
        pc = make_pc(focal, 1, max_bounces, ibl, True)
        if record:
            record[0].record()
        # frames in flight: the path stages of steps i + 1 and i + 2 overlap step i and its gather; the accumulate
        # kernel - the one writer of the tile - waits for the gather of the previous step inside the library
        ctx.render(pc, cam, width, height, tile=tile, frames=spp, stream=stream, flags=render_flags)
        if record:
            record[1].record()
        gather(hdr)

    def drain():
        if world_size > 1 and not rehearse and fallback_group is None:
            ctx.gather_wait(stream)

    # deterministic work counters of one launch (outside the timed region)
    ctx.reset_counters(stream)
    pc = make_pc(focal, 1, max_bounces, ibl, True)
    ctx.render(pc, cam, width, height, tile=tile, frames=spp, flags=S.RENDER_COUNT_WORK, stream=stream)
    counters = ctx.counters(stream).as_dict()
    bytes_per_step = algorithmic_bytes(counters, stats)
    stage_counters = [ctx.stage_counters(i, stream).as_dict() for i in range(4)]
    stage_bytes = [algorithmic_bytes(c, stats) for c in stage_counters]
    ctx.set_kernel_timing(True)

    # Device wake-up: the clocks of an idle MI355X take ~40 ms of load to ramp (the first ten 2.5 ms steps after idle
    # run 5-10 % slower, in every pipeline mode).  Untimed, reported as `preheat_steps`; the timed region is untouched.
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
        step((starts[i], stops[i]))
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("PROSPER_BENCH_STEP_TIMES") and rank == 0:
        print("step ms: " + " ".join("%.3f" % (starts[i].elapsed_time(starts[i + 1])) for i in range(args.steps - 1)), file=sys.stderr)
    _, per_kernel = ctx.last_render_timing()
    ctx.set_kernel_timing(False)

    t = torch.tensor([elapsed], dtype=torch.float64)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # what the communicator itself says about the job (ncclCommCount) and the device time of the last gather + de-interleave
    ranks_seen, ranks_seen_by, gather_ms = world_size, "torch.distributed (gloo)", None
    if world_size > 1 and not rehearse and fallback_group is None:
        info = ctx.comm_info(stream)
        ranks_seen, ranks_seen_by, gather_ms = int(info.ranks), "ncclCommCount of the library's communicator", float(info.lastGatherMs)
    elif world_size > 1:
        ranks_seen = dist.get_world_size()

    alone_ctx = None
    if rank == 0:
        if world_size > 1:
            torch.cuda.synchronize()
        image = full if world_size > 1 else hdr
        paths_per_step = width * height * spp
        ms_per_step = elapsed * 1e3 / args.steps
        alone = None
        if world_size == 1 and wavefront and render_flags:
            alone_ctx = capi.Context(device=local_rank, flags=S.CREATE_SINGLE_CHAIN)
            alone = time_alone(alone_ctx, world, cam, make_pc(focal, 1, max_bounces, ibl, True), width, height, spp, stream, torch)
        kernels, scale = kernel_table(per_kernel, stage_bytes, ms_per_step, pmc if world_size == 1 else None, alone,
                                      stage_counters if world_size == 1 else None)
        if alone:
            timed_note = ("each launch ALONE on the GPU (one chain, in order, hipEvents, mean of 3 renders) x %.3f = ms_per_step / "
                          "sum of the alone durations of a step's launches: with three frames in flight the launches overlap, the "
                          "exclusive shares add up to the step" % scale)
        else:
            timed_note = ("hipEvents around every launch of timed step %d of %d, scaled by %.3f = ms_per_step / sum of the raw "
                          "durations: the exclusive shares add up to the step" % (timed_step + 1, args.steps, scale))
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
                "parallelism": "image stripes x%d%s" % (world_size, "; " + gather_mode if world_size > 1 else ""),
                "pipeline": "megakernel" if args.megakernel else ("persistent" if args.persistent else (
                    "wavefront, 1 launch chain" if args.single_chain else (
                        "wavefront, 3 frames in flight (one launch chain each)" if render_flags
                        else "wavefront, 2 concurrent launch chains"))),
                "preheat_steps": preheat_steps,
            },
            "roofline": roofline_object(kernels, pmc_source, ms_per_step, timed_note, bytes_per_step),
            "kernels": kernels,
            "scene": {"upload_ms": stats.uploadSeconds * 1e3, "bvh_build_ms": stats.bvhBuildSeconds * 1e3,
                      "texture_upload_ms": stats.textureSeconds * 1e3, "upload_wall_ms": upload_wall * 1e3,
                      "device_MB": stats.deviceBytes / 1e6, "variant_flags": int(stats.variantFlags)},
            "whole_pass": {"algorithmic_bytes_per_step": bytes_per_step, "bytes_per_path": bytes_per_step / max(1, counters["paths"])},
            "counters": counters,
            "mean_radiance": float(image[..., :3].mean().item()),
        }
        if world_size == 1 and wavefront:
            if not args.no_extras:
                result["single_sample_frames"] = time_single_sample_frames(ctx, torch, cam, focal, width, height, max_bounces, ibl, stream)
                result["rank_share"] = time_rank_share(ctx, torch, cam, focal, width, height, spp, max_bounces, ibl, stream, 8, 3, 30, ms_per_step)
                ctx.set_output_buffer(hdr.data_ptr(), hdr.numel() * 4)
                ctx.render(make_pc(focal, 1, max_bounces, ibl, True), cam, width, height, frames=spp, stream=stream)
                result["tone_map"] = bench_tone_map(ctx, torch, width, height, stream)
                result["restir_di_trace"] = bench_restir_di(ctx, torch, world, cam, focal, width, height, stream)
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(world, cam, focal, width, height, spp, max_bounces, ibl, budget_s=args.cpu_budget)
                result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
            if sub_names:
                result["configs"] = {}
                for name in sub_names:
                    w2, b2, ww, hh, s2, mb2, ibl2 = CONFIGS[name]
                    try:
                        # 30 steps (1 s for C4): a timed region between two synchronisations also holds the pipeline's ramp-up and
                        # drain - about a third of a step with three frames in flight, i.e. +3 % over 10 steps, +1 % over 30
                        # CPU tracer beside it: ONE accumulated frame of the same workload (4-9 s on 16 cores for the
                        # S-sponza-class scenes); helmet2k shares helmet's (same geometry, same rays: the texels differ in size only)
                        cpu_s = 0.0 if (args.no_cpu_baseline or name == "helmet2k") else 0.5
                        sub = time_config(ctx, alone_ctx, torch, name, b2(), ww, hh, s2, mb2, ibl2, max(30, args.steps), 2, stream, render_flags,
                                          pmc_by_config.get(name), cpu_budget_s=cpu_s)
                        sub["workload"] = w2
                        if name == "helmet2k" and "cpu_baseline" in result["configs"].get("helmet", {}):
                            sub["cpu_baseline"] = dict(result["configs"]["helmet"]["cpu_baseline"], sample="configs.helmet's sample (same geometry and rays)")
                        result["configs"][name] = sub
                        share = sub.pop("c5_rank_share", None)
                        if share is not None:
                            # BASELINE C5 needs 8 GPUs: what ONE of its ranks does, timed on this GPU, and the job's rate if
                            # the eight shares ran side by side with a free gather - an extrapolation, labelled as one
                            c5 = dict(share)
                            if "ms_per_step" in share:
                                c5["Mpaths_per_s"] = share["Mpaths_per_s_of_the_share"]
                                c5["extrapolated"] = "1 of 8 rank shares on 1 GPU; x8 = %.0f Mpaths/s if the ranks scaled perfectly" % (8.0 * share["Mpaths_per_s_of_the_share"])
                            c5["workload"] = CONFIGS["c5"][0] + ", rank 3 of 8"
                            result["configs"]["c5_rank_share"] = c5
                    except Exception as e:  # a sub-configuration must never cost the headline line
                        result["configs"][name] = {"error": str(e)[:300]}
        if world_size > 1:
            result["ranks_seen"] = ranks_seen
            result["ranks_seen_by"] = ranks_seen_by
            result["gather_ms"] = gather_ms
        write_detail(result, args.detail)
        print(slim_line(result, args.detail))

    ctx.set_output_buffer(0, 0)
    ctx.close()
    if alone_ctx is not None:
        alone_ctx.close()
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
