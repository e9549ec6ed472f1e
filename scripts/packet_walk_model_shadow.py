#!/usr/bin/env python3
"""As scripts/packet_walk_model.py, for the any-hit walk of ReSTIR's shadow segments (primary hit point of each pixel of a block -> a random
point on a random emissive triangle): the segments of a block fan out over the light grid, 3.6 orderings per block, and a wave would
visit 1 280 nodes for 114 per segment.  Measured on the CPU so that the kernel did not have to be written to find out."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
name='teapots_lights'; W,H=1920,1080
sd = bench.make_scene(name); cam = bench.make_camera(name, W, H)
boxes = np.asarray(sd.boxes, np.float32).reshape(-1, 6); verts = np.asarray(sd.vertices, np.float32).reshape(-1, 3, 3); nodes = sd.nodes
pos = np.asarray(cam['position'], np.float64); right = np.asarray(cam['right'], np.float64); up = np.asarray(cam['up'], np.float64); view = np.asarray(cam['view'], np.float64)
tanf = float(cam['tanFovY']); aspect = W / H
lights = np.asarray(sd.light_prim_ids)
print('lights', len(lights))
def ordering(d):
    x, y, z = -d[:, 0], -d[:, 1], -d[:, 2]
    ax, ay, az = abs(x), abs(y), abs(z)
    return np.where(ax > ay, np.where(ax > az, np.where(x > 0, 0, 1), np.where(z > 0, 4, 5)), np.where(ay > az, np.where(y > 0, 2, 3), np.where(z > 0, 4, 5)))
def tri_hit(o, d, prim_id, tmax, mask):
    v0, v1, v2 = verts[prim_id].astype(np.float64)
    e1 = v1 - v0; e2 = v2 - v0
    pv = np.cross(d, e2); det = pv @ e1
    with np.errstate(all='ignore'):
        tv = o - v0; u = (tv * pv).sum(1) / det; qv = np.cross(tv, e1); v = (d * qv).sum(1) / det; t = (qv @ e2) / det
    return mask & (abs(det) > 1e-12) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < tmax), t
def packet(o, d, nd, tmax, any_hit):
    prim = nd['primitiveId']; box = nd['boundingBoxId']; nxt = nd['nextNodeIfMiss']; end = len(nd)
    L = len(d); inv = 1.0 / d
    p = np.zeros(L, np.int64); tmax = tmax.copy(); found = np.zeros(L, bool)
    n = 0; wave_steps = 0; lane_visits = 0
    while n < end:
        act = p == n
        if not act.any():
            n = int(p.min()); continue
        wave_steps += 1; lane_visits += int(act.sum())
        b = boxes[box[n]]
        t1 = (b[:3] - o) * inv; t2 = (b[3:] - o) * inv
        tn = np.minimum(t1, t2).max(1); tf = np.maximum(t1, t2).min(1)
        hit = act & (tf >= 0) & (tf >= tn) & (tn < tmax)
        if prim[n] >= 0 and hit.any():
            ok, t = tri_hit(o, d, prim[n], tmax, hit)
            if any_hit:
                found |= ok
            else:
                tmax = np.where(ok, t, tmax)
        p = np.where(act, np.where(hit, n + 1, nxt[n]), p)
        if any_hit: p = np.where(found, end, p)
        n = n + 1 if (hit & ~found).any() else int(min(nxt[n], p.min()))
    return wave_steps, lane_visits, tmax, found
rng = np.random.default_rng(7); nblocks = 150
tot = np.zeros(2); tot_any = np.zeros(2); nord = []
for k in range(nblocks):
    bx = rng.integers(0, W // 8); by = rng.integers(0, H // 8)
    xs = (bx * 8 + np.arange(64) % 8).astype(np.float64); ys = (by * 8 + np.arange(64) // 8).astype(np.float64)
    u = 1 - ((xs + rng.random(64)) / W) * 2; v = 1 - ((ys + rng.random(64)) / H) * 2
    d = (u * aspect * tanf)[:, None] * right + (v * tanf)[:, None] * up + view; d /= np.linalg.norm(d, axis=1)[:, None]
    o = np.tile(pos, (64, 1)); od = ordering(d); tm = np.full(64, np.inf)
    for q in np.unique(od):
        m = od == q
        ws, lv, t, _ = packet(o[m], d[m], nodes[q], tm[m], False); tm[m] = t; tot += (ws, lv)
    hitm = np.isfinite(tm)
    if not hitm.any(): continue
    P = o[hitm] + d[hitm] * tm[hitm][:, None]
    lp = lights[rng.integers(0, len(lights), hitm.sum())]
    bc = rng.random((hitm.sum(), 2)); s = np.sqrt(bc[:, 0]); b0 = 1 - s; b1 = bc[:, 1] * s
    T = verts[lp, 0] * b0[:, None] + verts[lp, 1] * b1[:, None] + verts[lp, 2] * (1 - b0 - b1)[:, None]
    sd_ = T - P; dist = np.linalg.norm(sd_, axis=1); sd_ /= dist[:, None]
    so = P + sd_ * 1e-4; od2 = ordering(sd_); nord.append(len(np.unique(od2)))
    for q in np.unique(od2):
        m = od2 == q
        ws, lv, _, _ = packet(so[m], sd_[m], nodes[q], dist[m] - 2e-4, True); tot_any += (ws, lv)
print('closest: wave steps/block %.1f lane visits/block %.1f' % (tot[0]/nblocks, tot[1]/nblocks))
print('any-hit: wave steps/block %.1f lane visits/block %.1f (per ray %.1f) orderings/block %.2f' % (tot_any[0]/nblocks, tot_any[1]/nblocks, tot_any[1]/nblocks/64, np.mean(nord)))
