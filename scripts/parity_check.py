#!/usr/bin/env python3
"""GPU vs oracle on a full-size scene at reduced resolution; prints differing pixels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prosper_amd import capi, scenes, structs as S
from oracle import binding as oracle
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
w, h = 480, 270
world = scenes.sponza_class(texture_size=128, lights=(cfg == "c4"), foliage=(cfg == "c4")) if cfg != "c2" else scenes.cornell()
c = world.camera
cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
ctx = capi.Context(0)
ctx.upload_scene(world)
osc = oracle.OracleScene(world)
want = None
for f in (1, 2):
    flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL | (S.PC_FLAG_SKIP_HISTORY if f == 1 else 0)
    pc = S.ReferencePC(0, flags, f, 1e-5, 1.0, fl, 3, 4)
    ctx.render(pc, cam, w, h)
    want, _ = osc.render(pc, cam, w, h, history=want)
got = ctx.read_hdr()
same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(axis=2)
print(cfg, "pixels differing:", int((~same).sum()), "of", same.size, "max abs diff", float(np.nanmax(np.abs(got - want))))
for y, x in np.argwhere(~same)[:5]:
    print("  px", x, y, got[y, x], want[y, x])
