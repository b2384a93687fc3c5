#!/usr/bin/env python3
"""Randomised parity sweep: random camera poses, push constants, extents, frame counts and rank tiles over the
synthetic scenes, HIP path (through the C-ABI) against the CPU oracle, every pixel bit for bit.

    python scripts/parity_fuzz.py [--cases 60] [--seed 1] [--scenes cornell,sponza,foliage,wall,zoo,alpha,helmet]

Prints one line per case and a summary; exit code 1 if any pixel differs.  (tests/test_gpu_parity.py holds the
fixed cases; this is the wide net: profiles/r01_parity_fuzz.txt keeps a run.)
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def run(cases, seed, scene_names, log=print):
    """-> (cases with differing pixels, pixels compared)"""
    from conftest import same_bits
    from oracle import binding as O
    from prosper_amd import capi, scenes, structs as S, tiling

    builders = {
        "cornell": lambda: scenes.cornell(with_skybox=True),
        "sponza": lambda: scenes.sponza_class(texture_size=64, sky_size=32, detail=0.35),
        "foliage": lambda: scenes.sponza_class(lights=True, foliage=True, texture_size=64, sky_size=32, detail=0.35),
        "wall": scenes.texture_wall,
        "zoo": scenes.transform_zoo,
        "alpha": scenes.alpha_wall,     # MASK / BLEND quads: every wrap mode, filter, factor, cutoff (alpha bounds)
        "helmet": lambda: __import__("prosper_amd.flight_helmet", fromlist=["x"]).load_fixture(),
    }
    names = list(scene_names)
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0)
    worlds, oracles = {}, {}
    bad_cases = 0
    total_px = 0
    t_start = time.time()
    for case in range(cases):
        name = names[case % len(names)]
        if name not in worlds:
            worlds[name] = builders[name]()
            oracles[name] = O.OracleScene(worlds[name], brute_force=name in ("cornell", "wall", "zoo", "alpha"))
        world = worlds[name]
        # a quarter of the textured cases: the scene is uploaded with placeholder texels and placeholder materials
        # (tests/test_adoption.py streamed_state: what prosper shows while it loads) and the images / materials are ADOPTED
        # in two or three steps (prosper_pt_update_textures / _materials, round 4), a render in flight between the steps
        streamed_note = ""
        if len(world.textures) > 2 and rng.random() < 0.25:
            from test_adoption import streamed_state
            images = len(world.textures) - 1
            cuts = sorted(set(int(x) for x in rng.integers(0, images + 1, size=int(rng.integers(1, 3))))) + [images]
            ctx.upload_scene(streamed_state(world, 0))
            warm_cam, warm_fl = O.camera_uniforms(world.camera["eye"], world.camera["target"], world.camera["up"], world.camera["fov"],
                                                  world.camera["zN"], world.camera["zF"], 64, 48)
            done = 0
            for cut in cuts:
                if cut > done:
                    ctx.update_textures(world.textures[done + 1:cut + 1], done + 1)
                    done = cut
                ctx.update_materials(streamed_state(world, done).materials, 0)
                if cut != images:  # a frame of the half-loaded scene stays in flight while the next images arrive
                    ctx.render(S.ReferencePC(0, S.PC_FLAG_SKIP_HISTORY, 1, 1e-5, 1.0, warm_fl, 3, 2), warm_cam, 64, 48, flags=S.RENDER_PIPELINED)
            streamed_note = " streamed %s" % "+".join(str(c) for c in cuts)
        elif len(world.metadatas) >= 3 and rng.random() < 0.3:
            # a third of the others: the MESHES arrive after the upload (prosper_pt_update_meshes: a worker thread builds the
            # new geometry while frames stay in flight; model instances show once all their sub-meshes are there)
            meshes = len(world.metadatas)
            order = [int(x) for x in rng.permutation(meshes)]
            cuts = sorted(set(int(x) for x in rng.integers(0, meshes + 1, size=int(rng.integers(1, 4))))) + [meshes]
            ctx.upload_scene(world.with_meshes_loaded(order[:cuts[0]]))
            warm_cam, warm_fl = O.camera_uniforms(world.camera["eye"], world.camera["target"], world.camera["up"], world.camera["fov"],
                                                  world.camera["zN"], world.camera["zF"], 64, 48)
            for a, b in zip(cuts, cuts[1:]):
                ctx.render(S.ReferencePC(0, S.PC_FLAG_SKIP_HISTORY, 1, 1e-5, 1.0, warm_fl, 3, 2), warm_cam, 64, 48, flags=S.RENDER_PIPELINED)
                if b > a:
                    ctx.update_meshes(world, order[a:b], wait=bool(rng.random() < 0.5))
            ctx.finish_mesh_updates()
            streamed_note = " meshes %s" % "+".join(str(c) for c in cuts)
        else:
            ctx.upload_scene(world)
        # a third of the cases on instanced scenes: some instances moved by prosper_pt_update_transforms (the GPU refit)
        # after the upload, against the oracle's own scene built at the moved pose
        oracle_scene = oracles[name]
        moved_note = ""
        if len(world.model_instances) >= 3 and rng.random() < 0.33:
            import copy
            from prosper_amd.world import rotate_y, rotate_x, translate
            moved = copy.copy(world)
            moved.model_instances = list(world.model_instances)
            moved._frozen = None
            size = float(np.linalg.norm(np.array(world.camera["eye"]) - np.array(world.camera["target"])))
            picks = rng.choice(len(world.model_instances), size=int(rng.integers(1, 4)), replace=False)
            for k in picks:
                model, m = moved.model_instances[int(k)]
                delta = translate(tuple(rng.normal(0, 0.05 * size, 3))) @ rotate_y(rng.uniform(-0.8, 0.8)) @ rotate_x(rng.uniform(-0.3, 0.3))
                moved.model_instances[int(k)] = (model, delta @ m)
            ctx.update_transforms(moved)
            oracle_scene = O.OracleScene(moved, brute_force=name in ("cornell", "wall", "zoo", "alpha"))
            moved_note = " moved %d" % len(picks)
        base = world.camera
        eye0, tgt0 = np.array(base["eye"], np.float64), np.array(base["target"], np.float64)
        dist = np.linalg.norm(eye0 - tgt0)
        # a pose around the scene's own camera: orbit, dolly, roll, sometimes looking away or from far outside
        yaw, pitch = rng.uniform(-math.pi, math.pi), rng.uniform(-0.6, 0.6)
        r = dist * float(np.exp(rng.uniform(-1.2, 0.8))) * (30.0 if rng.random() < 0.08 else 1.0)
        eye = tgt0 + r * np.array([math.cos(pitch) * math.sin(yaw), math.sin(pitch), math.cos(pitch) * math.cos(yaw)])
        target = tgt0 + rng.normal(0, 0.15 * dist, 3)
        roll = rng.uniform(-0.5, 0.5)
        up = (math.sin(roll), math.cos(roll), 0.0)
        fov = math.radians(rng.uniform(8.0, 110.0))
        w = int(rng.choice([33, 64, 97, 128, 160, 256]))
        h = int(rng.choice([17, 48, 64, 96, 135]))
        ranks = int(rng.choice([1, 1, 1, 2, 4])) if w % 16 == 0 else 1
        tile = None
        if ranks > 1 and tiling.check_divisible(w, ranks):
            tile = tiling.tile_for_rank(int(rng.integers(0, ranks)), ranks)
        cam, fl = O.camera_uniforms(tuple(eye), tuple(target), up, fov, base["zN"], base["zF"], w, h)
        draw = int(rng.choice([0] * 6 + list(range(1, 11))))
        flags = 0
        flags |= S.PC_FLAG_ACCUMULATE if rng.random() < 0.85 else 0
        flags |= S.PC_FLAG_IBL if rng.random() < 0.6 else 0
        flags |= S.PC_FLAG_DEPTH_OF_FIELD if rng.random() < 0.3 else 0
        flags |= S.PC_FLAG_CLAMP_INDIRECT if rng.random() < 0.7 else 0
        bounces = int(rng.integers(0, 7))
        roulette = int(rng.integers(0, 5))
        frames = int(rng.choice([1, 1, 2, 3, 5]))
        first = int(rng.integers(1, 4000))
        aperture = float(rng.choice([1e-5, 0.02, 0.2]))
        focus = float(rng.uniform(0.3, 2.0) * dist)
        want = None
        pcs = []
        for f in range(frames):
            fl_flags = flags | (S.PC_FLAG_SKIP_HISTORY if f == 0 else 0)
            pcs.append(S.ReferencePC(draw, fl_flags, (first + f - 1) % 4096 + 0, aperture, focus, fl, roulette, bounces))
        # GPU: either one batched call or frame by frame (both are the reference's accumulation)
        batched = frames > 1 and rng.random() < 0.5 and all(p.frameIndex == (pcs[0].frameIndex + i) % 4096 for i, p in enumerate(pcs))
        pipelined = S.RENDER_PIPELINED if rng.random() < 0.5 else 0  # frames in flight: same pixels
        if batched:
            ctx.render(pcs[0], cam, w, h, frames=frames, tile=tile, flags=pipelined)
        else:
            for p in pcs:
                ctx.render(p, cam, w, h, tile=tile, flags=pipelined)
        got = ctx.read_hdr()
        for p in pcs:
            want, _ = oracle_scene.render(p, cam, w, h, history=want)
        if oracle_scene is not oracles[name]:
            oracle_scene.close()
        if tile is not None:  # this rank's stripes, in ascending order
            cols = [x for x in range(w) if (x // tile.stripeWidth) % tile.stripeCount == tile.stripeIndex]
            want = want[:, cols]
        ok = same_bits(got, want).all(axis=2)
        nbad = int((~ok).sum())
        bad_cases += nbad > 0
        total_px += ok.size
        nan = int(np.isnan(got[..., :3]).any(axis=2).sum())
        log("case %3d %-8s %3dx%-3d draw %2d flags %02x bounces %d rr %d frames %d%s%s  eye-dist %.2g fov %3.0f  -> %s (%d NaN px)" % (
            case, name, w, h, draw, flags, bounces, roulette, frames, (" batched" if batched else "") + (" pipelined" if pipelined else "") + moved_note + streamed_note,
            " tile %d/%d" % (tile.stripeIndex, tile.stripeCount) if tile is not None else "", r, math.degrees(fov),
            "ok" if nbad == 0 else "%d PIXELS DIFFER" % nbad, nan))
    ctx.close()
    log("%d cases, %d pixels, %d cases with differences, %.0f s" % (cases, total_px, bad_cases, time.time() - t_start))
    return bad_cases, total_px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scenes", default="cornell,sponza,foliage,wall,zoo,alpha,helmet")
    ap.add_argument("--quiet", action="store_true", help="print only the cases that are not a plain ok, a progress line every 100 cases, and the summary")
    args = ap.parse_args()
    seen = [0]

    def log(*a):
        text = " ".join(str(x) for x in a)
        if args.quiet and text.startswith("case "):
            seen[0] += 1
            if text.rstrip().endswith("-> ok (0 NaN px)"):
                if seen[0] % 100 == 0:
                    print("... %d cases" % seen[0], flush=True)
                return
        print(text, flush=True)
    print("# scripts/parity_fuzz.py --cases %d --seed %d --scenes %s" % (args.cases, args.seed, args.scenes), flush=True)
    bad, _ = run(args.cases, args.seed, [n for n in args.scenes.split(",") if n], log=log)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
