import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from prosper_amd import capi, scenes, structs as S
from prosper_amd.rt_reference import Camera
from conftest import default_pc, same_bits
w,h=1920,1080
world=scenes.cornell()
cam,fl=Camera.from_world(world,w,h).update_buffer()
ctx=capi.Context(0); ctx.upload_scene(world); ctx.set_kernel_timing(True)
pc=default_pc(S,fl,max_bounces=4)
ctx.render(pc,cam,w,h,frames=40)
tot,per=ctx.last_render_timing(); a=ctx.read_hdr()
print("40 spp chunked: total %.2f ms"%tot, {k:(round(v[0],2),v[1]) for k,v in per.items()})
for k in range(5):
    p=default_pc(S,fl,frame_index=1+8*k,max_bounces=4,skip_history=(k==0))
    ctx.render(p,cam,w,h,frames=8)
b=ctx.read_hdr()
print("equal:", bool(same_bits(a,b).all()), "alpha", float(a[...,3].min()), float(a[...,3].max()))
