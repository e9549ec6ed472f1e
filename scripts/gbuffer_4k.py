import sys, torch
sys.path.insert(0,'.')
from radish_pt_amd import api
import bench
for scene,W,H in (("teasets_1m",3840,2160),("teapots_lights",1920,1080)):
    sd=bench.make_scene(scene); cam=bench.make_camera(scene,W,H)
    ctx=api.Context(0); ctx.upload_scene(sd); ctx.set_camera(cam)
    gb=api.GBuffer(); gb.create(W,H,0)
    ts=[]
    for r in range(6):
        ctx.gbuffer_render(gb.c_struct(cam_fallback=cam),0); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
    print(scene, W, H, "gbuffer ms", min(ts[1:]))
    ctx.close()
