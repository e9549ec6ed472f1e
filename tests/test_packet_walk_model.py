"""The packet walk (radish_pt_amd/csrc/device/traverse.h: packetWalk) as an executable model on the CPU, independent of the
oracle and of the device code.  A wave walks the threaded array ONE node at a time for all its lanes: it stands at node n, the lanes
whose own walk is at n (p == n) test its box against THEIR ray and THEIR closest distance and move on (p = n + 1 on a hit — a leaf's
triangle is tested at once — p = nextNodeIfMiss[n] on a miss), and the wave goes to n + 1 if any lane hit, else to nextNodeIfMiss[n],
WITHOUT looking at the lanes that wait further on.

What the model states, and the GPU tests then check bit for bit on the real kernels:
  * the wave's next node computed that way is always the smallest node any lane wants next (subtrees nest in a pre-order, so a lane
    that waits at n skipped there from an ancestor-or-self m of n and nextNodeIfMiss[m] >= nextNodeIfMiss[n]);
  * some lane is active at every node the wave visits (no wasted visits);
  * every lane makes exactly the visits and triangle tests of the reference's own loop (DevScene::intersect,
    /root/reference/src/scene.h:262-301) for its ray, in the same order, with the same hit — also for rays that have nothing to do with
    each other (a packet of incoherent rays is slow, not wrong).
"""
import numpy as np
import pytest

from test_pair_walk_model import FLT_MAX, _box_test, _ordering, _tri_test, f32


def _sequential(nodes, boxes, verts, o, d, inv):
    """The reference's loop for one ray; returns (hit prim, distance, visited nodes, tested triangles)."""
    prim_of, box_of, nxt = nodes["primitiveId"], nodes["boundingBoxId"], nodes["nextNodeIfMiss"]
    end, node, hit, tmax, visited, tested = len(nodes), 0, -1, FLT_MAX, [], []
    while node != end:
        visited.append(node)
        b = boxes[box_of[node]]
        ok, t = _box_test(b[:3], b[3:], o, inv)
        if ok and t < tmax:
            p = prim_of[node]
            if p >= 0:
                tested.append(int(p))
                th, dist = _tri_test(o, d, *verts[3 * p:3 * p + 3])
                if th and dist < tmax:
                    hit, tmax = int(p), dist
            node += 1
        else:
            node = int(nxt[node])
    return hit, tmax, visited, tested


def _packet(nodes, boxes, verts, rays, budget=10 ** 9):
    """packetWalk for the lanes `rays` = [(o, d, inv)]; returns per lane (hit, distance, visited, tested) and the wave's visit count.
    After `budget` visits as a packet the lanes that are not done go on each on its own, from its own node with its own closest distance
    (the device does the same after kPacketBudget visits: a block whose rays have parted must not keep one wave for the whole launch)."""
    prim_of, box_of, nxt = nodes["primitiveId"], nodes["boundingBoxId"], nodes["nextNodeIfMiss"]
    end = len(nodes)
    L = len(rays)
    p = [0] * L
    tmax = [FLT_MAX] * L
    hit = [-1] * L
    visited = [[] for _ in range(L)]
    tested = [[] for _ in range(L)]
    n, wave_visits = 0, 0
    while n != end:
        assert n == min(p), "the wave stands at the smallest node any lane wants"  # the invariant the device code relies on
        act = [i for i in range(L) if p[i] == n]
        assert act, "a node nobody wants"
        wave_visits += 1
        b = boxes[box_of[n]]
        any_hit = False
        for i in act:
            o, d, inv = rays[i]
            visited[i].append(n)
            ok, t = _box_test(b[:3], b[3:], o, inv)
            if ok and t < tmax[i]:
                any_hit = True
                q = prim_of[n]
                if q >= 0:
                    tested[i].append(int(q))
                    th, dist = _tri_test(o, d, *verts[3 * q:3 * q + 3])
                    if th and dist < tmax[i]:
                        hit[i], tmax[i] = int(q), dist
                p[i] = n + 1
            else:
                p[i] = int(nxt[n])
        n = n + 1 if any_hit else int(nxt[n])  # no reduction over the lanes that wait
        if wave_visits == budget:
            break
    for i in range(L):  # what is left of every lane's walk, lane by lane: the state is (node, closest distance, hit) and nothing else
        o, d, inv = rays[i]
        while p[i] != end:
            m = p[i]
            visited[i].append(m)
            b = boxes[box_of[m]]
            ok, t = _box_test(b[:3], b[3:], o, inv)
            if ok and t < tmax[i]:
                q = prim_of[m]
                if q >= 0:
                    tested[i].append(int(q))
                    th, dist = _tri_test(o, d, *verts[3 * q:3 * q + 3])
                    if th and dist < tmax[i]:
                        hit[i], tmax[i] = int(q), dist
                p[i] = m + 1
            else:
                p[i] = int(nxt[m])
    assert all(x == end for x in p)
    return [(hit[i], tmax[i], visited[i], tested[i]) for i in range(L)], wave_visits


@pytest.mark.parametrize("budget", [10 ** 9, 7])
@pytest.mark.parametrize("coherent", [True, False])
def test_packet_walk_makes_every_lanes_own_walk(coherent, budget):
    from radish_pt_amd import scenes

    sd = scenes.tiny(n_tris=48, seed=5)
    boxes = np.asarray(sd.boxes, dtype=f32)
    verts = np.asarray(sd.vertices, dtype=f32)
    rng = np.random.default_rng(11)
    packets, hits, lane_visits, wave_visits = 0, 0, 0, 0
    for _ in range(12):
        eye = rng.uniform(-2.5, 2.5, 3).astype(f32)
        rays = []
        while len(rays) < 64:
            if coherent:  # a pinhole block: one origin, directions within a degree of each other
                if not rays:
                    axis = rng.normal(size=3)
                    axis /= np.linalg.norm(axis)
                d = axis + rng.normal(size=3) * 0.01
                o = eye
            else:
                d = rng.normal(size=3)
                o = rng.uniform(-2.0, 2.0, 3).astype(f32)
            d = (d / np.linalg.norm(d)).astype(f32)
            if (np.abs(d) < 1e-3).any() or (np.abs(d) > 1 - 1e-3).any():
                continue  # the common ray class only (the others never take part in a packet on the device either)
            rays.append((o, d, (f32(1) / d).astype(f32)))
        by_order = {}
        for r in rays:
            by_order.setdefault(_ordering(r[1]), []).append(r)
        for k, lanes in by_order.items():  # packetWalkAll: ordering by ordering
            got, wv = _packet(sd.nodes[k], boxes, verts, lanes, budget)
            for (o, d, inv), g in zip(lanes, got):
                ref = _sequential(sd.nodes[k], boxes, verts, o, d, inv)
                assert g == ref
                hits += ref[0] != -1
                lane_visits += len(ref[2])
            packets += 1
            wave_visits += wv
    assert hits > 30 and packets >= 12
    if coherent and budget > 1000:  # neighbouring rays: the union of 64 walks is not much more than one walk
        assert wave_visits * 8 < lane_visits
