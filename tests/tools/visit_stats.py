"""Where do the box tests of a frame go?  (analysis; uses the CPU oracle's visit-histogram hook)

For BASELINE config 2 (or --scene teapots) at a pixel stride: per-ray visit-count distribution, and how much of all
visits the K hottest nodes of each of the six orderings would absorb (the case for an LDS-resident hot set)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from radish_pt_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--stride", type=int, default=7)
ap.add_argument("--depth", type=int, default=8)
args = ap.parse_args()
W, H = 1920, 1080
sd = scenes.cornell() if args.scene == "cornell" else scenes.teapots()
cam = scenes.cornell_camera(W, H) if args.scene == "cornell" else scenes.teapots_camera(W, H)
o = pyoracle.OracleScene(sd)
l = pyoracle.lib()
l.orc_debug_visit_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
l.orc_debug_visit_hist.restype = None
n = sd.bvh_size
hist = np.zeros(6 * n, np.uint32)
rl = np.zeros(128, np.uint64)
l.orc_debug_visit_hist(o.h, hist.ctypes.data, rl.ctypes.data)
d = np.zeros((W * H, 3), np.float32)
i = np.zeros((W * H, 3), np.float32)
o.path_trace(cam, d, i, 0, 3, args.depth, pix=(0, W * H, args.stride))
l.orc_debug_visit_hist(o.h, None, None)
st = o.stats()
print(st)
total = hist.sum()
print(f"nodes {n}, total visits {total}, per ray {total / (st['closestRays'] + st['anyRays']):.1f}")
for kind, name in ((0, "closest"), (1, "any")):
    cnt, sm = rl[kind * 32:kind * 32 + 32], rl[64 + kind * 32:64 + kind * 32 + 32]
    print(f"{name}: rays {cnt.sum()}, visits {sm.sum()}")
    for b in range(32):
        if cnt[b]:
            lo = 0 if b == 0 else 1 << (b - 1)
            print(f"   visits in [{lo:6d},{(1 << b):6d}): rays {cnt[b]:9d} ({100 * cnt[b] / cnt.sum():5.1f} %)  visits {sm[b]:11d} ({100 * sm[b] / max(sm.sum(), 1):5.1f} % of all)")
h6 = hist.reshape(6, n)
print("per-ordering share of visits:", np.round(h6.sum(1) / total, 3))
for K in (128, 256, 512, 1024, 2048, 4096):
    cov = sum(np.sort(h6[k])[::-1][:K].sum() for k in range(6))
    print(f"hottest {K:5d} nodes per ordering ({6 * K * 32 // 1024:4d} KB): {100 * cov / total:5.1f} % of visits")
# the same if the hot set is chosen by tree position (first-K in breadth-first order = nodes with the largest subtrees)
nodes0 = sd.nodes[0]


def parents_area(nodes, boxes):
    """priority of a node = surface area of its PARENT's box (a node is visited iff its parent's box was hit)."""
    prim, boxid, nxt = nodes["primitiveId"], nodes["boundingBoxId"], nodes["nextNodeIfMiss"]
    ext = boxes[:, 3:6] - boxes[:, 0:3]
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
    pri = np.zeros(len(nodes))
    pri[0] = np.inf
    for i in range(len(nodes)):
        if prim[i] < 0:
            c1 = i + 1
            c2 = nxt[c1]
            pri[c1] = pri[c2] = area[boxid[i]]
    return pri


boxes = np.asarray(sd.boxes).reshape(-1, 6)
print("node dtype:", sd.nodes[0].dtype)
for K in (64, 128, 256, 512):
    cov = 0
    for k in range(6):
        pri = parents_area(sd.nodes[k], boxes)
        hot = np.argsort(-pri, kind="stable")[:K]
        cov += h6[k][hot].sum()
    print(f"parent-area heuristic, {K:4d} nodes per ordering: {100 * cov / total:5.1f} % of visits")
for K in (32, 64, 128, 256, 512, 1024):
    cov = sum(h6[k][:K].sum() for k in range(6))
    print(f"first {K:5d} records of each ordering (depth-first prefix): {100 * cov / total:5.1f} % of visits")
