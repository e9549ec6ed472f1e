#!/usr/bin/env python3
"""Packet walk statistics (CPU, numpy; what was measured BEFORE traverse.h's packetWalk was written): for random 8x8 pixel blocks of a
scene's camera, how many nodes does a WAVE visit when its 64 lanes walk the threaded order together (the wave stands at node
n = the smallest node any lane wants next; only the lanes that want n test it) against the lanes' own visit counts.
usage: python scripts/packet_walk_model.py [teapots|cornell|teasets_1m] [blocks] [jitter]
teapots, 300 blocks: 107 wave steps per block for 89 visits per ray (5 707 lane visits per block): the union of 64 neighbouring walks is
1.2 walks.  (The shadow segments of ReSTIR's pass 1 are another matter: scripts/packet_walk_model_shadow.py — 1 280 wave steps per
block for 114 visits per segment, 3.6 orderings per block: not walked as packets.)"""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
f32 = np.float32
name = sys.argv[1] if len(sys.argv) > 1 else 'teapots'
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 200
jitter = len(sys.argv) > 3 and sys.argv[3] == 'jitter'
W, H = 1920, 1080
sd = bench.make_scene(name)
cam = bench.make_camera(name, W, H)
boxes = np.asarray(sd.boxes, f32).reshape(-1, 6)
verts = np.asarray(sd.vertices, f32).reshape(-1, 3, 3)
nodes = sd.nodes  # list of six structured arrays
print(name, 'tris', len(verts), 'nodes per ordering', len(nodes[0]))
pos = np.asarray(cam['position'], np.float64); right = np.asarray(cam['right'], np.float64); up = np.asarray(cam['up'], np.float64); view = np.asarray(cam['view'], np.float64)
tanf = float(cam['tanFovY']); aspect = W / H
def ray_dirs(xs, ys, rng):
    jx = rng.random(xs.shape) if jitter else 0.5
    jy = rng.random(xs.shape) if jitter else 0.5
    u = 1 - ((xs + jx) / W) * 2; v = 1 - ((ys + jy) / H) * 2
    fx = u * aspect * tanf; fy = v * tanf
    d = fx[:, None] * right + fy[:, None] * up + view
    return d / np.linalg.norm(d, axis=1)[:, None]
def ordering(d):
    x, y, z = -d[:, 0], -d[:, 1], -d[:, 2]
    ax, ay, az = abs(x), abs(y), abs(z)
    o = np.where(ax > ay, np.where(ax > az, np.where(x > 0, 0, 1), np.where(z > 0, 4, 5)), np.where(ay > az, np.where(y > 0, 2, 3), np.where(z > 0, 4, 5)))
    return o
def packet(o, d, nd):
    prim = nd['primitiveId']; box = nd['boundingBoxId']; nxt = nd['nextNodeIfMiss']; end = len(nd)
    L = len(d); inv = 1.0 / d
    p = np.zeros(L, np.int64); tmax = np.full(L, np.inf)
    n = 0; wave_steps = 0; lane_visits = 0; wave_tris = 0; lane_tris = 0
    while n < end:
        act = p == n
        wave_steps += 1; lane_visits += int(act.sum())
        b = boxes[box[n]]
        t1 = (b[:3] - o) * inv; t2 = (b[3:] - o) * inv
        tn = np.minimum(t1, t2).max(1); tf = np.maximum(t1, t2).min(1)
        hit = act & (tf >= 0) & (tf >= tn) & (tn < tmax)
        if prim[n] >= 0 and hit.any():
            wave_tris += 1; lane_tris += int(hit.sum())
            v0, v1, v2 = verts[prim[n]].astype(np.float64)
            e1 = v1 - v0; e2 = v2 - v0
            pv = np.cross(d, e2); det = pv @ e1
            with np.errstate(all='ignore'):
                tv = o - v0; u = (tv * pv).sum(1) / det; qv = np.cross(tv, e1); v = (d * qv).sum(1) / det; t = (qv @ e2) / det
            ok = hit & (abs(det) > 1e-12) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < tmax)
            tmax = np.where(ok, t, tmax)
        p = np.where(act, np.where(hit, n + 1, nxt[n]), p)
        n = n + 1 if hit.any() else int(nxt[n])
    return wave_steps, lane_visits, wave_tris, lane_tris
rng = np.random.default_rng(5)
tot = np.zeros(4); t0 = time.time(); norders = []
bx = rng.integers(0, W // 8, nblocks); by = rng.integers(0, H // 8, nblocks)
for k in range(nblocks):
    xs = (bx[k] * 8 + np.arange(64) % 8).astype(np.float64); ys = (by[k] * 8 + np.arange(64) // 8).astype(np.float64)
    d = ray_dirs(xs, ys, rng); o = np.tile(pos, (64, 1))
    od = ordering(d); us = np.unique(od); norders.append(len(us))
    for q in us:
        m = od == q
        tot += packet(o[m], d[m], nodes[q])
print('blocks', nblocks, 'jitter', jitter, 'time %.1fs' % (time.time() - t0))
print('wave node steps per block %.1f ; lane visits per block %.1f (per ray %.1f) ; ratio lane/wave %.1f' % (tot[0] / nblocks, tot[1] / nblocks, tot[1] / nblocks / 64, tot[1] / tot[0]))
print('wave leaf tests per block %.1f ; lane tri tests per block %.1f' % (tot[2] / nblocks, tot[3] / nblocks))
print('orderings per block: mean %.2f max %d' % (np.mean(norders), max(norders)))
