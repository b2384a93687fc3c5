#!/usr/bin/env python3
"""Throughput of back-to-back renders, in order vs PROSPER_PT_RENDER_PIPELINED (two frames in flight), at the full
frame and at the per-rank shares of an N-rank job: python scripts/pipelined_bench.py c2 [c3 c4]."""
import sys, time; sys.path.insert(0,'.')
import ctypes as C
from prosper_amd import capi, scenes, structs as S, tiling
from prosper_amd.rt_reference import Camera
hip=C.CDLL("libamdhip64.so")
cfgs={"c2":(scenes.cornell,False),"c3":(lambda: scenes.sponza_class(),True),"c4":(lambda: scenes.sponza_class(lights=True,foliage=True),True)}
import os
SPP=int(os.environ.get("SPP","8"))
for name in sys.argv[1:] or ["c2"]:
  b,ibl=cfgs[name]; world=b(); w,h=1920,1080
  cam,focal=Camera.from_world(world,w,h).update_buffer()
  flags=S.PC_FLAG_ACCUMULATE|S.PC_FLAG_CLAMP_INDIRECT|S.PC_FLAG_SKIP_HISTORY|(S.PC_FLAG_IBL if ibl else 0)
  pc=S.ReferencePC(0,flags,1,1e-5,1.0,focal,3,4)
  ctx=capi.Context(0); ctx.upload_scene(world)
  for ranks in ((8,4,2,1) if name=="c2" else (1,)):
    tile=tiling.tile_for_rank(0,ranks) if ranks>1 else None
    for rf,label in ((0,"in order"),(S.RENDER_PIPELINED,"pipelined")):
        for it in range(4): ctx.render(pc,cam,w,h,frames=SPP,tile=tile,flags=rf)
        hip.hipDeviceSynchronize()
        K=30 if name=="c2" else 6; t=time.perf_counter()
        for it in range(K): ctx.render(pc,cam,w,h,frames=SPP,tile=tile,flags=rf)
        hip.hipDeviceSynchronize()
        dt=(time.perf_counter()-t)/K*1e3
        print("%s ranks %d %-9s: %.3f ms/frame  %.0f Mpaths/s"%(name,ranks,label,dt,w*h*SPP/ranks/dt/1e3),flush=True)
  ctx.close()
