#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prosper_amd import capi, scenes, structs as S
from oracle import binding as oracle
w, h = 480, 270
world = scenes.sponza_class(texture_size=128)
c = world.camera
cam, fl = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
ctx = capi.Context(0)
ctx.upload_scene(world)
osc = oracle.OracleScene(world)
def diff(pc, tag):
    ctx.render(pc, cam, w, h)
    got = ctx.read_hdr()
    want, _ = osc.render(pc, cam, w, h)
    same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(axis=2)
    print(tag, "differing:", int((~same).sum()), [(int(x), int(y)) for y, x in np.argwhere(~same)[:8]])
for f in (1, 2):
    for mb in (1, 2, 3, 4):
        for ibl in (0, 1):
            flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | (S.PC_FLAG_IBL if ibl else 0) | S.PC_FLAG_SKIP_HISTORY
            diff(S.ReferencePC(0, flags, f, 1e-5, 1.0, fl, 3, mb), "frame %d maxBounces %d ibl %d" % (f, mb, ibl))
diff(S.ReferencePC(1, S.PC_FLAG_SKIP_HISTORY, 1, 1e-5, 1.0, fl, 3, 1), "PrimitiveID f1")
