#!/usr/bin/env python3
"""k_walk_persistent alone (for rocprofv3 --pmc passes): 8 M incoherent class-0 rays on one scene, three launches.
usage: walker_only.py [cornell|teapots]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from radish_pt_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
N = 8 << 20
rng = np.random.default_rng(5)
sd = scenes.cornell() if name == "cornell" else scenes.teapots()
lo, hi = sd.vertices.min(0), sd.vertices.max(0)
o = rng.uniform(lo, hi, (N, 3)).astype(np.float32)
d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = np.where(np.abs(d) < 1e-3, 1e-3, d); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = torch.from_numpy(np.concatenate([o, d.astype(np.float32)], 1).astype(np.float32)).cuda()
hits = torch.zeros(N, 4, dtype=torch.int32, device="cuda")
ctx = api.Context(0); ctx.upload_scene(sd)
for r in range(3):
    ctx.trace_closest(rays, hits, api.RDH_PT_PERSISTENT); ctx.synchronize()
print(name, "walker ms", ctx.last_kernel_ms())
