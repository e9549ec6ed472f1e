"""Procedural stand-in scenes for the BASELINE configs, in the reference's DevScene layout.

The reference's `scenes/` directory (scene.txt, OBJ models, textures, sobol_10k_200.bin) is git-ignored and absent
(SURVEY.md F4), so every config runs on a synthetic scene of the same triangle count and material mix, built here
from closed-form geometry.  The arrays produced are exactly what `Scene::buildDevData` + `DevScene::create` would
hand to the kernels (`/root/reference/src/scene.cpp:190-249,461-551`): an un-indexed world-space triangle soup,
per-vertex normals/texcoords, per-triangle material ids, the Material table, AABBs + six threaded BVH arrays,
the emissive-triangle list with its alias table, and the Sobol table.
"""
import colorsys
import os

import numpy as np

from . import hostlib
from . import layouts as L

_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")
_DIRS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sobol_dirs_200x32.u32")


def sobol_table():
    """uint32[10000][200] Joe–Kuo Sobol' points, unscrambled (stand-in for the absent sobol_10k_200.bin,
    `/root/reference/src/scene.cpp:543-548`), generated from the committed direction numbers
    (`data/sobol_dirs_200x32.u32`: v[dim][bit], 25.6 KB, taken from scipy's bundled Joe–Kuo table) by the Gray-code
    recurrence x[i] = x[i-1] XOR v[:, index of the lowest zero bit of i-1].  The 8 MB table itself is a run-time cache, not
    a tracked file; tests/test_golden.py checks the first points against scipy and the published Joe–Kuo values."""
    path = os.path.join(_CACHE, "sobol_10k_200.bin")
    if os.path.exists(path):
        t = np.fromfile(path, dtype="<u4")
        if t.size == L.SOBOL_NUM * L.SOBOL_DIM:
            return t.reshape(L.SOBOL_NUM, L.SOBOL_DIM)
    v = np.fromfile(_DIRS, dtype="<u4").reshape(L.SOBOL_DIM, 32)
    t = np.zeros((L.SOBOL_NUM, L.SOBOL_DIM), dtype="<u4")
    cur = np.zeros(L.SOBOL_DIM, dtype="<u4")
    for i in range(1, L.SOBOL_NUM):
        c = (~(i - 1) & i).bit_length() - 1
        cur = cur ^ v[:, c]
        t[i] = cur
    try:
        os.makedirs(_CACHE, exist_ok=True)
        t.tofile(path)
    except OSError:
        pass
    return t


class SceneData:
    """Host arrays of one scene (all numpy, reference layouts)."""

    def __init__(self, name, vertices, normals, texcoords, material_ids, materials, textures=(), env_map_tex_id=-1):
        """textures: list of float32 [h, w, 3] arrays (the reference's `Image` pixels, row-major); material map ids index
        into it (-1 none, -2 procedural base colour).  env_map_tex_id: index of the environment map or -1."""
        self.name = name
        self.textures = [np.ascontiguousarray(t, dtype=np.float32) for t in textures]
        self.env_map_tex_id = int(env_map_tex_id)
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.texcoords = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        self.material_ids = np.ascontiguousarray(material_ids, dtype=np.int32)
        self.materials = np.ascontiguousarray(materials, dtype=L.MATERIAL_DTYPE)
        self.num_prims = self.vertices.shape[0] // 3
        assert self.normals.shape[0] == self.num_prims * 3 and self.texcoords.shape[0] == self.num_prims * 3
        assert self.material_ids.shape[0] == self.num_prims
        self.boxes, self.nodes = hostlib.build_bvh(self.vertices)
        self.bvh_size = self.boxes.shape[0]
        self.light_prim_ids, self.light_unit_radiance, self.light_power = hostlib.build_light_list(
            self.vertices, self.material_ids, self.materials
        )
        self.num_lights = len(self.light_prim_ids)
        power = self.light_power
        if self.env_map_tex_id >= 0:  # Scene::createLightSampler: the env map is the LAST light entry (src/scene.cpp:146-164)
            env = self.textures[self.env_map_tex_id]
            self.env_map_sampler, env_sum = hostlib.build_envmap_sampler(env, env.shape[1], env.shape[0])
            power = np.concatenate([power, np.array([env_sum], np.float32)]).astype(np.float32)
        else:
            self.env_map_sampler = np.zeros(0, dtype=L.BINOMIAL_DTYPE)
        if len(power):
            self.light_sampler, s = hostlib.build_alias_table(power)
            self.sum_light_power_inv = np.float32(1.0) / np.float32(s)  # src/scene.cpp:527
        else:
            self.light_sampler = np.zeros(0, dtype=L.BINOMIAL_DTYPE)
            self.sum_light_power_inv = np.float32(0)
        self.sobol = sobol_table()

    def nbytes(self):
        return sum(
            a.nbytes
            for a in [self.vertices, self.normals, self.texcoords, self.material_ids, self.boxes] + list(self.nodes)
        )


# --------------------------------------------------------------------------------------------------
# geometry helpers (float32 throughout)
# --------------------------------------------------------------------------------------------------
def _quad(p0, p1, p2, p3, normal):
    """Two triangles (p0,p1,p2),(p0,p2,p3); all six vertex normals = `normal`."""
    p = np.array([p0, p1, p2, p0, p2, p3], dtype=np.float32)
    n = np.tile(np.asarray(normal, dtype=np.float32), (6, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 0], [1, 1], [0, 1]], dtype=np.float32)
    return p, n, uv


def _revolved(segments, bands, radius, height, seed):
    """'Teapot' stand-in: a closed-ish surface of revolution about +y with `segments` x `bands` x 2 triangles.
    The profile is a squat body with a neck and a lid knob; the pole rings keep a 0.002 radius so that no
    triangle is degenerate."""
    u = np.linspace(0.0, 1.0, bands + 1, dtype=np.float64)
    body = np.sin(np.pi * np.clip(u / 0.78, 0, 1)) ** 0.55 * (1.0 - 0.18 * u)
    knob = 0.22 * np.sin(np.pi * np.clip((u - 0.78) / 0.22, 0, 1)) ** 0.8
    r = radius * (np.where(u < 0.78, body, 0.0) + np.where(u >= 0.78, knob, 0.0) + 0.12 * np.exp(-((u - 0.78) / 0.05) ** 2))
    r = np.maximum(r, 0.002)
    y = height * (u + 0.03 * np.sin(2 * np.pi * u * (1 + seed % 3)))
    y = np.maximum.accumulate(y)
    dr = np.gradient(r, u)
    dy = np.gradient(y, u)
    phi = np.linspace(0.0, 2 * np.pi, segments, endpoint=False, dtype=np.float64)
    cp, sp = np.cos(phi), np.sin(phi)
    # ring points [bands+1, segments, 3] and outward normals
    P = np.stack([r[:, None] * cp[None, :], np.repeat(y[:, None], segments, 1), r[:, None] * sp[None, :]], -1)
    N = np.stack([dy[:, None] * cp[None, :], np.repeat(-dr[:, None], segments, 1), dy[:, None] * sp[None, :]], -1)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-12)
    UV = np.stack(np.meshgrid(phi / (2 * np.pi), u, indexing="xy"), -1)  # [bands+1, segments, 2]
    j0 = np.arange(segments)
    j1 = (j0 + 1) % segments
    tris_p, tris_n, tris_uv = [], [], []
    for i in range(bands):
        a, b, c, d = (i, j0), (i, j1), (i + 1, j1), (i + 1, j0)
        for tri in ((a, c, b), (a, d, c)):  # counter-clockwise seen from outside
            tris_p.append(np.stack([P[t[0], t[1]] for t in tri], 1))
            tris_n.append(np.stack([N[t[0], t[1]] for t in tri], 1))
            tris_uv.append(np.stack([UV[t[0], t[1]] for t in tri], 1))
    p = np.concatenate(tris_p, 0).reshape(-1, 3).astype(np.float32)
    n = np.concatenate(tris_n, 0).reshape(-1, 3).astype(np.float32)
    uv = np.concatenate(tris_uv, 0).reshape(-1, 2).astype(np.float32)
    return p, n, uv


class _Builder:
    def __init__(self):
        self.p, self.n, self.uv, self.ids, self.mats = [], [], [], [], []

    def material(self, m):
        self.mats.append(m)
        return len(self.mats) - 1

    def add(self, p, n, uv, mat_id, translate=(0, 0, 0)):
        p = (p + np.asarray(translate, dtype=np.float32)).astype(np.float32)
        self.p.append(p)
        self.n.append(n)
        self.uv.append(uv)
        self.ids.append(np.full(p.shape[0] // 3, mat_id, dtype=np.int32))

    def room(self, x0, x1, y0, y1, z0, z1, white, red, green):
        """Five inward-facing walls, open toward +z (the camera side)."""
        self.add(*_quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), (0, 1, 0)), white)  # floor
        self.add(*_quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0)), white)  # ceiling
        self.add(*_quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1)), white)  # back
        self.add(*_quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (1, 0, 0)), red)  # left
        self.add(*_quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0)), green)  # right

    def ceiling_light(self, cx, cz, half, y, mat):
        """Quad whose GEOMETRIC normal (winding) faces -y, as `sampleDirectLight`'s single-sided test needs
        (`/root/reference/src/scene.h:444-448`).  Its VERTEX normals face +y: the emitter-hit test of
        `singleKernelPT` (`src/pathtrace.cu:252-256`) keeps a BSDF-sampled hit only when dot(shading normal,
        ray direction) >= 0, so this choice makes both tests select the same (lower) face."""
        self.add(
            *_quad((cx - half, y, cz - half), (cx + half, y, cz - half), (cx + half, y, cz + half),
                   (cx - half, y, cz + half), (0, 1, 0)),
            mat,
        )

    def finish(self, name, textures=(), env_map_tex_id=-1):
        return SceneData(
            name, np.concatenate(self.p), np.concatenate(self.n), np.concatenate(self.uv), np.concatenate(self.ids),
            np.array(self.mats, dtype=L.MATERIAL_DTYPE), textures, env_map_tex_id,
        )


def cornell(segments=64, bands=48):
    """S1 'Cornell' (BASELINE configs 1-2): 2x2x2 box open to the camera, 0.5x0.5 ceiling light (17,12,4),
    three revolved stand-ins (Lambertian .9 / metallic 1 roughness .3 / dielectric 1.5).
    10 + 2 + 3*segments*bands*2 triangles (18 444 at the defaults)."""
    b = _Builder()
    white = b.material(L.make_material(L.LAMBERTIAN, (0.73, 0.73, 0.73)))
    red = b.material(L.make_material(L.LAMBERTIAN, (0.63, 0.065, 0.05)))
    green = b.material(L.make_material(L.LAMBERTIAN, (0.14, 0.45, 0.091)))
    light = b.material(L.make_material(L.LIGHT, (17.0, 12.0, 4.0)))
    diffuse = b.material(L.make_material(L.LAMBERTIAN, (0.9, 0.9, 0.9)))
    metal = b.material(L.make_material(L.METALLIC_WORKFLOW, (0.95, 0.8, 0.45), metallic=1.0, roughness=0.3))
    glass = b.material(L.make_material(L.DIELECTRIC, (1.0, 1.0, 1.0), ior=1.5))
    b.room(-1, 1, 0, 2, -1, 1, white, red, green)
    b.ceiling_light(0.0, 0.0, 0.25, 1.999, light)
    for k, (mat, pos) in enumerate([(diffuse, (-0.5, 0.0, -0.35)), (metal, (0.45, 0.0, -0.1)), (glass, (-0.05, 0.0, 0.5))]):
        p, n, uv = _revolved(segments, bands, 0.3, 0.55, k)
        b.add(p, n, uv, mat, pos)
    return b.finish("cornell")


def cornell_camera(width, height):
    return hostlib.make_camera(width, height, eye=(0.0, 1.0, 4.2), rotation=(-90.0, 0.0, 0.0), fovy=20.0)


def _hsv_light(i):
    h = (L_hash(i) % 1024) / 1024.0
    r, g, b = colorsys.hsv_to_rgb(h, 0.8, 1.0)
    return (5.0 * r, 5.0 * g, 5.0 * b)


def L_hash(a):
    """utilhash (`/root/reference/src/mathUtil.h:199-207`) on Python ints, used only to colour the light grid."""
    M = 0xFFFFFFFF
    a = ((a + 0x7ED55D16) + (a << 12)) & M
    a = ((a ^ 0xC761C23C) ^ (a >> 19)) & M
    a = ((a + 0x165667B1) + (a << 5)) & M
    a = ((a + 0xD3A2646C) ^ (a << 9)) & M
    a = ((a + 0xFD7046C5) + (a << 3)) & M
    a = ((a ^ 0xB55A4F09) ^ (a >> 16)) & M
    return a


def teapots(segments=64, bands=49, grid=4, emissive_grid=None):
    """S3 'Teapots' (BASELINE config 3): grid x grid stand-ins of segments*bands*2 triangles (6 272 at the defaults;
    16 of them + room 10 + light 2 = 100 364), materials cycled Lambertian / metallic (roughness 0.1-0.6) /
    dielectric.  emissive_grid=(16, 32) adds 512 small emissive quads = 1 024 light triangles (config 4, S4)."""
    b = _Builder()
    white = b.material(L.make_material(L.LAMBERTIAN, (0.73, 0.73, 0.73)))
    red = b.material(L.make_material(L.LAMBERTIAN, (0.63, 0.065, 0.05)))
    green = b.material(L.make_material(L.LAMBERTIAN, (0.14, 0.45, 0.091)))
    light = b.material(L.make_material(L.LIGHT, (17.0, 12.0, 4.0)))
    b.room(-4, 4, 0, 3, -4, 4, white, red, green)
    b.ceiling_light(0.0, 0.0, 0.75, 2.999, light)
    k = 0
    span = 6.0
    for gz in range(grid):
        for gx in range(grid):
            kind = k % 3
            if kind == 0:
                hue = (k * 0.17) % 1.0
                col = colorsys.hsv_to_rgb(hue, 0.55, 0.85)
                mat = b.material(L.make_material(L.LAMBERTIAN, col))
            elif kind == 1:
                rough = 0.1 + 0.5 * ((k // 3) % 6) / 5.0
                mat = b.material(L.make_material(L.METALLIC_WORKFLOW, (0.9, 0.75, 0.5), metallic=1.0, roughness=rough))
            else:
                mat = b.material(L.make_material(L.DIELECTRIC, (1.0, 1.0, 1.0), ior=1.5))
            p, n, uv = _revolved(segments, bands, 0.55, 0.9, k)
            x = -span / 2 + span * (gx + 0.5) / grid
            z = -span / 2 + span * (gz + 0.5) / grid
            b.add(p, n, uv, mat, (x, 0.0, z))
            k += 1
    name = "teapots"
    if emissive_grid:
        rows, cols = emissive_grid
        half = 0.04
        for i in range(rows * cols):
            r, c = divmod(i, cols)
            mat = b.material(L.make_material(L.LIGHT, _hsv_light(i)))
            cx = -3.6 + 7.2 * (c + 0.5) / cols
            cz = -3.6 + 7.2 * (r + 0.5) / rows
            b.ceiling_light(cx, cz, half, 2.9 - 0.002 * (i % 7), mat)
        name = "teapots_lights"
    return b.finish(name)


def teapots_camera(width, height):
    return hostlib.make_camera(width, height, eye=(0.3, 1.9, 7.4), rotation=(-91.5, -11.0, 0.0), fovy=19.0)


def _checker(w, h, a, b, cells=8):
    y, x = np.mgrid[0:h, 0:w]
    m = (((x * cells) // w + (y * cells) // h) % 2).astype(np.float32)[..., None]
    return (np.asarray(a, np.float32) * (1 - m) + np.asarray(b, np.float32) * m).astype(np.float32)


def _sky(w, h):
    """Small synthetic HDR environment: gradient sky + a bright sun lobe + dim ground (stand-in for an .hdr file)."""
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    V, U = np.meshgrid(v, u, indexing="ij")
    sky = np.stack([0.25 + 0.3 * (1 - V), 0.4 + 0.35 * (1 - V), 0.7 + 0.3 * (1 - V)], -1) * (V < 0.5)[..., None]
    ground = np.stack([0.12 + 0 * V, 0.1 + 0 * V, 0.08 + 0 * V], -1) * (V >= 0.5)[..., None]
    sun = 40.0 * np.exp(-(((U - 0.3) / 0.03) ** 2 + ((V - 0.22) / 0.04) ** 2))
    return (sky + ground + sun[..., None] * np.array([1.0, 0.9, 0.7])).astype(np.float32)


def cornell_textured(segments=16, bands=12, env=True):
    """Cornell stand-in that exercises every branch of getTexturedMaterialAndSurface (`src/scene.h:88-112`) and the
    environment map (`:374-414`): checker base-colour texture on the floor, procedural base colour on the back wall,
    metallic + roughness + normal maps on the metal object, env map visible through the open front."""
    rng = np.random.default_rng(3)
    tex = [
        _checker(32, 32, (0.8, 0.8, 0.8), (0.2, 0.25, 0.6)),                                  # 0 base colour
        _checker(16, 16, (0.9, 0.9, 0.9), (0.3, 0.3, 0.3), cells=4),                          # 1 metallic (.r)
        (0.25 + 0.5 * rng.random((8, 8, 1))).repeat(3, axis=2).astype(np.float32),            # 2 roughness (.r)
        (np.array([0.5, 0.5, 1.0]) + 0.25 * (rng.random((16, 16, 3)) - 0.5)).astype(np.float32),  # 3 normal map
    ]
    env_id = -1
    if env:
        tex.append(_sky(64, 32))
        env_id = 4
    b = _Builder()

    def mat(**kw):
        maps = {k: kw.pop(k) for k in list(kw) if k.endswith("MapId")}
        m = L.make_material(**kw)
        for k, v in maps.items():
            m[k] = v
        return b.material(m)

    white = mat(type=L.LAMBERTIAN, baseColor=(0.73, 0.73, 0.73))
    floor = mat(type=L.LAMBERTIAN, baseColor=(1, 1, 1), baseColorMapId=0)
    back = mat(type=L.LAMBERTIAN, baseColor=(1, 1, 1), baseColorMapId=-2)
    red = mat(type=L.LAMBERTIAN, baseColor=(0.63, 0.065, 0.05))
    green = mat(type=L.LAMBERTIAN, baseColor=(0.14, 0.45, 0.091))
    light = mat(type=L.LIGHT, baseColor=(17.0, 12.0, 4.0))
    metal = mat(type=L.METALLIC_WORKFLOW, baseColor=(0.95, 0.8, 0.45), metallic=1.0, roughness=0.3, metallicMapId=1,
                roughnessMapId=2, normalMapId=3)
    glass = mat(type=L.DIELECTRIC, baseColor=(1.0, 1.0, 1.0), ior=1.5)
    x0, x1, y0, y1, z0, z1 = -1, 1, 0, 2, -1, 1
    b.add(*_quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), (0, 1, 0)), floor)
    b.add(*_quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0)), white)
    b.add(*_quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1)), back)
    b.add(*_quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (1, 0, 0)), red)
    b.add(*_quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0)), green)
    b.ceiling_light(0.0, 0.0, 0.25, 1.999, light)
    for k, (m_, pos) in enumerate([(metal, (0.45, 0.0, -0.1)), (glass, (-0.35, 0.0, 0.3))]):
        p, n, uv = _revolved(segments, bands, 0.3, 0.55, k)
        b.add(p, n, uv, m_, pos)
    return b.finish("cornell_textured", tex, env_id)


def tiny(n_tris=24, seed=1):
    """A handful of random triangles + one light quad: small enough for brute-force cross-checks."""
    rng = np.random.default_rng(seed)
    b = _Builder()
    mats = [
        b.material(L.make_material(L.LAMBERTIAN, (0.8, 0.6, 0.4))),
        b.material(L.make_material(L.METALLIC_WORKFLOW, (0.9, 0.9, 0.9), metallic=0.7, roughness=0.4)),
        b.material(L.make_material(L.DIELECTRIC, (1.0, 1.0, 1.0), ior=1.33)),
    ]
    light = b.material(L.make_material(L.LIGHT, (8.0, 8.0, 8.0)))
    for i in range(n_tris):
        c = rng.uniform(-1, 1, 3)
        p = (c + rng.uniform(-0.6, 0.6, (3, 3))).astype(np.float32)
        nrm = np.cross(p[1] - p[0], p[2] - p[0])
        nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
        b.add(p, np.tile(nrm, (3, 1)), np.zeros((3, 2), np.float32), mats[i % 3])
    b.ceiling_light(0.0, 0.0, 0.5, 1.8, light)
    return b.finish("tiny")


def load_scene_file(path):
    """A Radish scene text file (+ the OBJ meshes and textures it names) → (SceneData, camera, settings dict).
    `Scene::Scene` + `Scene::buildDevData` (/root/reference/src/scene.cpp:108-141,190-249): parsing, mesh/texture loading
    and flattening happen in libradish_host.so (rdh_scene_parse); SceneData then builds the BVH, the light list and the
    alias tables exactly as for the procedural scenes."""
    p = hostlib.parse_scene(path)
    sd = SceneData(os.path.basename(path), p["vertices"], p["normals"], p["texcoords"], p["material_ids"], p["materials"],
                   textures=p["textures"], env_map_tex_id=p["env_map_tex_id"])
    settings = {"trace_depth": p["trace_depth"], "iterations": p["iterations"], "image_name": p["image_name"]}
    return sd, p["camera"], settings
