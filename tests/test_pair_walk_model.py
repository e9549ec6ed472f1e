"""The sibling-pair walk (radish_pt_amd/csrc/device/traverse.h: pairStart / pairPopOne / pairStep) as an executable
model on the CPU, independent of the oracle and of the device code: for random rays over a small BVH built by the host library it
must make the decisions of the reference's threaded walk (DevScene::intersect / testOcclusion, /root/reference/src/scene.h:262-334)
— the same hit record, the same number of box visits and triangle tests — in all six orderings, for closest-hit and any-hit walks.

What the model states, and the GPU tests then check bit for bit on the real kernels:
  * the six orderings of MTBVHNode are six pre-orders of ONE binary tree that differ only in which child comes first;
  * a walk that enters a node may test BOTH children at once: the near child's test is final; the far child's is provisional — a box
    that fails under the closest distance of now fails under the (smaller) one of later, and a box that passes is re-checked
    against the closest distance of the moment the walk reaches it;
  * counting the far child when it is POPPED (pushing failed far children too, with distance +inf) reproduces the reference's visit
    count even for an any-hit walk that ends early.
"""
import numpy as np
import pytest

f32 = np.float32
INF = f32(np.inf)
FLT_MAX = f32(3.402823466e38)


def _box_test(lo, hi, o, inv):
    """aabbFast: the last branch of AABB::intersect (bvh.h:125-154) for rays of the common class; float32 throughout."""
    t1 = (lo - o) * inv
    t2 = (hi - o) * inv
    n = np.minimum(t1, t2)
    f = np.maximum(t1, t2)
    d = f - n
    yz, zx, xy = f[2] - n[1], f[0] - n[2], f[1] - n[0]
    overlap = (d[1] + d[2] > yz) and (d[2] + d[0] > zx) and (d[0] + d[1] > xy)
    t_min = max(max(n[0], n[1]), n[2])
    t_max = min(min(f[0], f[1]), f[2])
    return bool(overlap and t_max >= 0 and t_max >= t_min), f32(t_min)


def _tri_test(o, d, v0, v1, v2):
    """intersectTriangle (intersections.h:20-68): two-sided Moeller-Trumbore; returns (hit, dist)."""
    e01, e02 = v1 - v0, v2 - v0
    pvec = np.cross(d, e02).astype(f32)
    det = f32(np.dot(e01, pvec))
    if abs(det) < f32(1.1920928955078125e-7):
        return False, f32(0)
    v0o = o - v0
    if det < 0:
        det, v0o = -det, -v0o
    bx = f32(np.dot(v0o, pvec))
    if bx < 0 or bx > det:
        return False, f32(0)
    qvec = np.cross(v0o, e01).astype(f32)
    by = f32(np.dot(d, qvec))
    if by < 0 or bx + by > det:
        return False, f32(0)
    dist = f32(np.dot(e02, qvec)) * (f32(1) / det)
    return bool(dist > 0), f32(dist)


def _ordering(d):
    """getMTBVHId(-dir) (scene.h:114-129)."""
    x, y, z = -d
    ax, ay, az = abs(x), abs(y), abs(z)
    if ax > ay:
        if ax > az:
            return 0 if x > 0 else 1
        return 4 if z > 0 else 5
    if ay > az:
        return 2 if y > 0 else 3
    return 4 if z > 0 else 5


def _threaded(nodes, boxes, verts, o, d, inv, tmax, any_hit):
    """The reference's loop: node -> box test -> (leaf: triangle test) -> node + 1 / nextNodeIfMiss."""
    prim_of, box_of, nxt = nodes["primitiveId"], nodes["boundingBoxId"], nodes["nextNodeIfMiss"]
    end, node, hit, n_nodes, n_tris = len(nodes), 0, -1, 0, 0
    while node != end:
        n_nodes += 1
        b = boxes[box_of[node]]
        ok, t = _box_test(b[:3], b[3:], o, inv)
        if ok and t < tmax:
            p = prim_of[node]
            if p >= 0:
                n_tris += 1
                th, dist = _tri_test(o, d, *verts[3 * p:3 * p + 3])
                if th and dist < tmax:
                    if any_hit:
                        return 1, tmax, n_nodes, n_tris
                    hit, tmax = p, dist
            node += 1
        else:
            node = nxt[node]
    return hit, tmax, n_nodes, n_tris


def _build_pairs(all_nodes):
    """buildSharedTree (radish_hip.hip): the tree the six arrays describe, as sibling pairs with per-ordering 'child 1 first' bits."""
    n0 = all_nodes[0]
    S = len(n0)
    children = {}  # box id of an inner node -> (box id of ordering 0's first child, of its second child)
    for p in range(S):
        if n0["primitiveId"][p] < 0:
            first = p + 1
            second = n0["nextNodeIfMiss"][first]
            children[int(n0["boundingBoxId"][p])] = (int(n0["boundingBoxId"][first]), int(n0["boundingBoxId"][second]))
    leaf_prim = {int(n0["boundingBoxId"][p]): int(n0["primitiveId"][p]) for p in range(S) if n0["primitiveId"][p] >= 0}
    bits = {b: 0 for b in children}
    for k in range(1, 6):
        nk = all_nodes[k]
        for p in range(S):
            if nk["primitiveId"][p] < 0:
                b = int(nk["boundingBoxId"][p])
                first = int(nk["boundingBoxId"][p + 1])
                second = int(nk["boundingBoxId"][nk["nextNodeIfMiss"][p + 1]])
                assert {first, second} == set(children[b]), "the orderings are not pre-orders of one tree"
                if first == children[b][1]:
                    bits[b] |= 1 << k
    return int(n0["boundingBoxId"][0]), children, leaf_prim, bits


def _pairs(root, children, leaf_prim, bits, boxes, verts, o, d, inv, tmax, any_hit, ordering, count_exact=True):
    """pairStart / pairPopOne / pairStep.  Stack entries: (box id of the far child, its provisional boundDist)."""
    n_nodes, n_tris, hit = 0, 0, -1

    def leaf(b, tmax, hit, n_tris):
        n_tris += 1
        p = leaf_prim[b]
        th, dist = _tri_test(o, d, *verts[3 * p:3 * p + 3])
        if th and dist < tmax:
            return True, (tmax if any_hit else dist), p, n_tris  # an any-hit walk ends here: its bound is the segment's length
        return False, tmax, hit, n_tris

    stack, cur = [], None
    n_nodes += 1  # the root: a single box every walk tests first
    ok, t = _box_test(boxes[root][:3], boxes[root][3:], o, inv)
    if ok and t < tmax:
        if root in leaf_prim:
            acc, tmax, hit, n_tris = leaf(root, tmax, hit, n_tris)
            if acc and any_hit:
                return 1, tmax, n_nodes, n_tris
        else:
            cur = root
    while cur is not None or stack:
        if cur is None:  # pop: the far child the sequential walk reaches next
            b, dist = stack.pop()
            n_nodes += 1
            if not (dist < tmax):
                continue
            if b in leaf_prim:
                acc, tmax, hit, n_tris = leaf(b, tmax, hit, n_tris)
                if acc and any_hit:
                    return 1, tmax, n_nodes, n_tris
                continue
            cur = b
        c0, c1 = children[cur]
        near, far = (c1, c0) if (bits[cur] >> ordering) & 1 else (c0, c1)
        okn, tn = _box_test(boxes[near][:3], boxes[near][3:], o, inv)
        okf, tf = _box_test(boxes[far][:3], boxes[far][3:], o, inv)
        hn, hf = okn and tn < tmax, okf and tf < tmax
        n_nodes += 1  # the near child, now; the far one when it is popped
        if hf or count_exact:
            stack.append((far, tf if hf else INF))
        cur = None
        if hn:
            if near in leaf_prim:
                acc, tmax, hit, n_tris = leaf(near, tmax, hit, n_tris)
                if acc and any_hit:
                    return 1, tmax, n_nodes, n_tris
            else:
                cur = near
    return hit, tmax, n_nodes, n_tris


@pytest.mark.parametrize("any_hit", [False, True])
def test_pair_walk_makes_the_threaded_walks_decisions(any_hit):
    from radish_pt_amd import scenes

    sd = scenes.tiny(n_tris=40, seed=3)
    boxes = np.asarray(sd.boxes, dtype=f32)
    verts = np.asarray(sd.vertices, dtype=f32)
    root, children, leaf_prim, bits = _build_pairs(sd.nodes)
    # the orderings come in opposite pairs (bvh.cpp:171-180): 2k visits one child first, 2k + 1 the other
    for b, m in bits.items():
        for k in (0, 2, 4):
            assert ((m >> k) & 1) != ((m >> (k + 1)) & 1)
    rng = np.random.default_rng(7)
    seen, hits = set(), 0
    for _ in range(240):
        o = rng.uniform(-1.8, 1.8, 3).astype(f32)
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(f32)
        if (np.abs(d) < 1e-3).any() or (np.abs(d) > 1 - 1e-3).any():
            continue  # keep to the common ray class (the others are traced whole over the threaded arrays on the device too)
        inv = (f32(1) / d).astype(f32)
        k = _ordering(d)
        seen.add(k)
        tmax = f32(rng.uniform(0.5, 3.0)) if any_hit else FLT_MAX
        ref = _threaded(sd.nodes[k], boxes, verts, o, d, inv, tmax, any_hit)
        got = _pairs(root, children, leaf_prim, bits, boxes, verts, o, d, inv, tmax, any_hit, k)
        assert got == ref, (k, ref, got)
        lean = _pairs(root, children, leaf_prim, bits, boxes, verts, o, d, inv, tmax, any_hit, k, count_exact=False)
        assert lean[:2] == ref[:2] and lean[3] == ref[3]  # failed far children not pushed: same hit, same triangle tests
        hits += ref[0] != -1 if not any_hit else ref[0] == 1
    assert seen == set(range(6)) and hits > 20
