import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from radish_pt_amd import api, scenes
from helpers import random_rays
sd = scenes.cornell()
ctx = api.Context(0); ctx.upload_scene(sd)
rays = random_rays(20000, 5)
d = torch.from_numpy(rays).cuda(); hits = torch.zeros(len(rays),4,dtype=torch.int32,device='cuda')
# per-ray visits via counters: find the longest ray by bisection on batches is slow; use the known long ray
long = np.array([[8.7514436e-01, 8.9130765e-01, 1.8423585e-02, -3.5287568e-01, 1.0000000e-07, -1.0994885e-01]], np.float32)
import os
reg = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "long_rays_cornell.npy"))
for name, r in (("literal-path ray (tiny dir component)", long), ("regular long ray x1", reg[-1:]), ("regular long rays x8", reg),
                ("regular long ray x64 copies", np.repeat(reg[-1:], 64, 0)),
                ("1 long + 63 instant misses (full wave, 1 walker)", np.concatenate([reg[-1:], np.tile(np.array([[50, 50, 50, 0.6, 0.64, 0.48]], np.float32), (63, 1))])),
                ("3 long + 61 instant misses", np.concatenate([reg[-3:], np.tile(np.array([[50, 50, 50, 0.6, 0.64, 0.48]], np.float32), (61, 1))])),
                ("20000 random", rays)):
    dr = torch.from_numpy(np.ascontiguousarray(r)).cuda(); h = torch.zeros(len(r),4,dtype=torch.int32,device='cuda')
    ctx.counters_reset(); ctx.trace_closest(dr, h, api.RDH_PT_COUNT); c = ctx.counters()
    ts=[]
    for _ in range(5):
        ctx.trace_closest(dr, h, 0); ctx.synchronize(); ts.append(ctx.last_kernel_ms())
    ms = min(ts); steps = c['nodeVisits']/len(r)
    if 'instant' in name: steps = c['nodeVisits'] / (1 if name.startswith('1') else 3)
    print(f"{name}: {len(r)} rays, {steps:.0f} visits/ray, kernel {ms*1e3:.1f} us -> {ms*1e6/ max(steps,1):.0f} ns per mean-ray step")
