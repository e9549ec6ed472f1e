"""Scene text format, OBJ and texture loading (the step before the hot path).

The reference ships no scene files (its `scenes/` directory is git-ignored), so these tests write small
scenes in the grammar its parser accepts (`/root/reference/src/scene.cpp:108-141,256-459`) and check what
libradish_host.so's rdh_scene_parse returns against values derived independently here (numpy float64 transforms, Pillow
decodes, hand-written RGBE).  Parity with the reference's own loader is unpinned (glm / stb_image are not vendored)."""
import os
import struct

import numpy as np
import pytest

from radish_pt_amd import hostlib, layouts as L, scenes

CUBE_QUADS = """# a unit cube as six quads, with normals and uvs
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 -1
vn 0 0 1
vn -1 0 0
vn 1 0 0
vn 0 -1 0
vn 0 1 0
g cube
f 1/1/1 4/4/1 3/3/1 2/2/1
f 5/1/2 6/2/2 7/3/2 8/4/2
f 1/1/3 5/2/3 8/3/3 4/4/3
f 2/1/4 3/2/4 7/3/4 6/4/4
f 1/1/5 2/2/5 6/3/5 5/4/5
f 4/1/6 8/2/6 7/3/6 3/4/6
"""

PLANE = """v -2 0 -2
v 2 0 -2
v 2 0 2
v -2 0 2
v 0 0 3
f 1 4 3
f -5 -3 -4
f 1 2 3 4 5
"""


def write_hdr(path, img):
    """Flat (non-RLE) Radiance RGBE, top row first."""
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            for x in range(w):
                r, g, b = (float(v) for v in img[y, x])
                m = max(r, g, b)
                if m < 1e-32:
                    f.write(bytes([0, 0, 0, 0]))
                else:
                    e = int(np.floor(np.log2(m))) + 1
                    s = 256.0 / (2.0 ** e)
                    f.write(bytes([int(r * s), int(g * s), int(b * s), e + 128]))


def rgbe_round(img):
    out = np.zeros_like(img, dtype=np.float32)
    for idx in np.ndindex(img.shape[:2]):
        r, g, b = (float(v) for v in img[idx])
        m = max(r, g, b)
        if m >= 1e-32:
            e = int(np.floor(np.log2(m))) + 1
            s = 256.0 / (2.0 ** e)
            out[idx] = [np.float32(int(c * s)) * np.float32(2.0 ** (e - 8)) for c in (r, g, b)]
    return out


@pytest.fixture(scope="module")
def scene_dir(tmp_path_factory):
    from PIL import Image

    d = tmp_path_factory.mktemp("radish_scene")
    (d / "cube.obj").write_text(CUBE_QUADS)
    (d / "plane.obj").write_text(PLANE)
    rng = np.random.default_rng(3)
    albedo = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    Image.fromarray(albedo).save(d / "albedo.png")
    grey16 = rng.integers(0, 65536, (4, 3), dtype=np.uint16)
    Image.fromarray(grey16).save(d / "rough16.png")
    pal = Image.fromarray(rng.integers(0, 256, (6, 6, 3), dtype=np.uint8)).convert("P", palette=Image.ADAPTIVE, colors=16)
    pal.save(d / "normal_pal.png")
    with open(d / "metal.ppm", "wb") as f:
        metal = rng.integers(0, 256, (2, 2, 3), dtype=np.uint8)
        f.write(b"P6\n# comment\n2 2\n255\n" + metal.tobytes())
    Image.fromarray(rng.integers(0, 256, (4, 4, 3), dtype=np.uint8)).save(d / "photo.jpg", quality=95)
    env = (rng.random((4, 8, 3)) * 3.0).astype(np.float32)
    env[0, 0] = 0.0
    write_hdr(d / "sky.hdr", env)
    scene = """Material floor
Type Lambertian
BaseColor albedo.png
Metallic 0
Roughness rough16.png
Ior 1.5
NormalMap normal_pal.png

Material steel
Type MetallicWorkflow
BaseColor 0.9 0.8 0.7
Metallic metal.ppm
Roughness 0.25
Ior 1.5
NormalMap Null

Material glass
Type Dielectric
BaseColor Procedural
Metallic 0
Roughness 0
Ior 1.33
NormalMap Null

Material lamp
Type Light
BaseColor 10 9 8
Metallic 0
Roughness 1
Ior 1
NormalMap Null

Material poster
Type Lambertian
BaseColor photo.jpg
Metallic 0
Roughness 1
Ior 1.5
NormalMap Null

Object 0
plane.obj
Material floor

Object 1
cube.obj
Material steel
Translate 0.5 0 -1
Rotate 30 45 60
Scale 0.5 1 2

Object 2
cube.obj
Material glass
Scale 0.25 0.25 0.25
Translate -1 0 0

Object 3
cube.obj
Material lamp
Translate 0 3 0
Scale 1 0.01 1

Object 4
missing.obj
Material steel
Translate 9 9 9

Object 5
cube.obj
Material Null

Camera
Resolution 40 30
FovY 19.5
LensRadius 0.0
FocalDist 5
ApertureMask Null
Sample 32
Depth 6
File test_out
Eye 0.5 2 6
Rotation -92 -14 0
Up 0 1 0

EnvMap sky.hdr
"""
    (d / "scene.txt").write_text(scene)
    meta = {"albedo": albedo, "grey16": grey16, "metal": metal, "env": env}
    return d, meta


def test_parse_materials_textures_camera(scene_dir):
    from PIL import Image

    d, meta = scene_dir
    p = hostlib.parse_scene(str(d / "scene.txt"))
    m = p["materials"]
    assert len(m) == 6  # five named + the `Material Null` default
    assert [int(t) for t in m["type"]] == [L.LAMBERTIAN, L.METALLIC_WORKFLOW, L.DIELECTRIC, L.LIGHT, L.LAMBERTIAN, L.LAMBERTIAN]
    assert np.allclose(m["baseColor"][1], [0.9, 0.8, 0.7]) and np.allclose(m["baseColor"][3], [10, 9, 8])
    assert m["baseColorMapId"][2] == -2 and m["ior"][2] == np.float32(1.33) and m["roughness"][1] == np.float32(0.25)
    # texture ids in order of first use: albedo 0, rough16 1, normal_pal 2, metal 3, photo 4, sky 5
    assert (m["baseColorMapId"][0], m["roughnessMapId"][0], m["normalMapId"][0], m["metallicMapId"][1]) == (0, 1, 2, 3)
    assert m["baseColorMapId"][4] == 4 and p["env_map_tex_id"] == 5 and len(p["textures"]) == 6
    d6 = m[5]  # Material() defaults (src/material.h:276-286)
    assert np.allclose(d6["baseColor"], 0.9) and d6["roughness"] == 1 and d6["ior"] == 1.5 and d6["baseColorMapId"] == -1
    # decoders: LDR v -> v/255, textures flipped vertically, 16-bit -> high byte, the env map NOT flipped
    want = (meta["albedo"][::-1].astype(np.float32)) / np.float32(255)
    assert np.array_equal(p["textures"][0], want)
    g = ((meta["grey16"] >> 8).astype(np.float32) / np.float32(255))[::-1]
    assert np.array_equal(p["textures"][1], np.repeat(g[..., None], 3, -1))
    palref = np.asarray(Image.open(d / "normal_pal.png").convert("RGB"), np.float32)[::-1] / np.float32(255)
    assert np.array_equal(p["textures"][2], palref)
    assert np.array_equal(p["textures"][3], meta["metal"][::-1].astype(np.float32) / np.float32(255))
    jpg = np.asarray(Image.open(d / "photo.jpg").convert("RGB"), np.float32)[::-1] / np.float32(255)
    assert np.array_equal(p["textures"][4], jpg)  # through the Pillow callback
    assert np.array_equal(p["textures"][5], rgbe_round(meta["env"]))
    cam = p["camera"]
    ref = hostlib.make_camera(40, 30, eye=(0.5, 2, 6), rotation=(-92, -14, 0), fovy=19.5, lens_radius=0.0, focal_dist=5.0)
    for f in ("resolution", "position", "rotation", "view", "right", "fov", "rotationMatInv", "lensRadius", "focalDist", "tanFovY"):
        assert np.array_equal(cam[f], ref[f]), f
    assert (p["trace_depth"], p["iterations"], p["image_name"]) == (6, 32, "test_out")


def test_parse_geometry(scene_dir):
    d, _ = scene_dir
    p = hostlib.parse_scene(str(d / "scene.txt"))
    v, n, uv, ids = p["vertices"], p["normals"], p["texcoords"], p["material_ids"]
    # plane.obj: 1 triangle + 1 triangle by negative indices + a pentagon fanned into 3; then 4 cubes of 12 (missing.obj skipped)
    assert len(ids) == 5 + 4 * 12 and len(v) == 3 * len(ids)
    assert ids.tolist() == [0] * 5 + [1] * 12 + [2] * 12 + [3] * 12 + [5] * 12
    plane = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2], [0, 0, 3]], np.float32)
    assert np.array_equal(v[0:3], plane[[0, 3, 2]])
    assert np.array_equal(v[3:6], plane[[0, 2, 1]])  # f -5 -3 -4
    assert np.array_equal(v[6:15], plane[[0, 1, 2, 0, 2, 3, 0, 3, 4]])
    assert np.allclose(n[0:6], [0, 1, 0])  # no `vn`: the face geometric normal (both wind counter-clockwise seen from +y)
    assert np.array_equal(uv[0:15], np.zeros((15, 2), np.float32))  # no `vt` at all in that file
    # cube 1: Translate(0.5,0,-1) * Rx(30) * Ry(45) * Rz(60) * Scale(0.5,1,2), float64 reference
    def rot(axis, deg):
        a = np.radians(deg)
        c, s = np.cos(a), np.sin(a)
        return {0: np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 1: np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
                2: np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]
    A = rot(0, 30) @ rot(1, 45) @ rot(2, 60) @ np.diag([0.5, 1.0, 2.0])
    t = np.array([0.5, 0, -1])
    cube = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float64)
    first = cube[[0, 3, 2, 1]]  # f 1 4 3 2: diagonals equal -> "else" branch: [0,1,3], [1,2,3]
    want = (first[[0, 1, 3, 1, 2, 3]] @ A.T) + t
    assert np.allclose(v[15:21], want, atol=2e-6)
    nrm = np.linalg.inv(A).T @ np.array([0, 0, -1.0])
    assert np.allclose(n[15], nrm / np.linalg.norm(nrm), atol=2e-6)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(uv[15:21], np.array([[0, 0], [0, 1], [1, 0], [0, 1], [1, 1], [1, 0]], np.float32))
    # cube 2: later transform lines override earlier ones, order in the file is irrelevant (scene.cpp:294-305)
    assert np.allclose(v[15 + 36:15 + 72].min(0), [-1, 0, 0]) and np.allclose(v[15 + 36:15 + 72].max(0), [-0.75, 0.25, 0.25])
    # cube 5 (`Material Null`): identity transform defined here (the reference leaves it uninitialised)
    assert np.allclose(v[-36:].min(0), 0) and np.allclose(v[-36:].max(0), 1)


def test_load_scene_file_builds_scene_data(scene_dir):
    d, _ = scene_dir
    sd, cam, settings = scenes.load_scene_file(str(d / "scene.txt"))
    assert sd.num_prims == 53 and sd.bvh_size == 105 and settings["trace_depth"] == 6
    assert sd.num_lights == 12 and len(sd.light_sampler) == 13  # 12 emissive triangles + the env map as the last entry
    assert sd.env_map_tex_id == 5 and len(sd.env_map_sampler) == 4 * 8
    assert cam["resolution"].tolist() == [40, 30]


def test_parse_errors(tmp_path):
    with pytest.raises(RuntimeError, match="Error reading from file"):
        hostlib.parse_scene(str(tmp_path / "nope.txt"))
    (tmp_path / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s1.txt").write_text("Object 0\na.obj\nMaterial ghost\n\n")
    with pytest.raises(RuntimeError, match="Material ghost not found"):
        hostlib.parse_scene(str(tmp_path / "s1.txt"))
    (tmp_path / "s2.txt").write_text("Material m\nType Lambertian\nBaseColor 1 1 1\nMetallic 0\nRoughness 1\nIor 1\nNormalMap Null\n\n")
    with pytest.raises(RuntimeError, match="No mesh data loaded"):
        hostlib.parse_scene(str(tmp_path / "s2.txt"))
    (tmp_path / "s3.txt").write_text("Material m\nType Lambertian\nBaseColor missing.png\nMetallic 0\nRoughness 1\nIor 1\nNormalMap Null\n\n")
    with pytest.raises(RuntimeError, match="missing.png"):
        hostlib.parse_scene(str(tmp_path / "s3.txt"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    (tmp_path / "s4.txt").write_text("Object 0\nbad.obj\nMaterial Null\n\n")
    with pytest.raises(RuntimeError, match="references vertex"):
        hostlib.parse_scene(str(tmp_path / "s4.txt"))


def _png(width, height, depth, ctype, rows, interlace=0):
    """A PNG file from raw scanlines (filter byte 0 prepended to each), any header values — including invalid ones."""
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    return (bytes([137, 80, 78, 71, 13, 10, 26, 10]) + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, interlace))
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_malformed_png_headers_are_rejected(tmp_path):
    """IHDR bit depths the PNG specification does not allow (depth 0 gave a zero stride and an integer division by
    zero in the host process, depths 3 / 5 / 6 / 7 were decoded silently), 16-bit palette indices, sub-byte depths on colour
    types that forbid them, and absurd sizes — each must come back as an error through the C ABI, never as a crash."""
    scene = "Material m\nType Lambertian\nBaseColor {}\nMetallic 0\nRoughness 1\nIor 1\nNormalMap Null\n\n" \
            "Object 0\na.obj\nMaterial m\n\nCamera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\n" \
            "Sample 1\nDepth 2\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"
    (tmp_path / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    bad = {
        "depth0.png": _png(2, 2, 0, 0, [[0], [0]]),
        "depth3.png": _png(2, 2, 3, 0, [[0], [0]]),
        "depth7.png": _png(2, 2, 7, 0, [[0, 0], [0, 0]]),
        "depth32.png": _png(2, 2, 32, 0, [[0] * 8, [0] * 8]),
        "pal16.png": _png(2, 2, 16, 3, [[0] * 4, [0] * 4]),
        "rgb4.png": _png(2, 2, 4, 2, [[0] * 3, [0] * 3]),
        "huge.png": _png(60000, 60000, 8, 0, [[0]]),
        "interlaced.png": _png(2, 2, 8, 0, [[0, 0], [0, 0]], interlace=1),
    }
    for name, blob in bad.items():
        (tmp_path / name).write_bytes(blob)
        (tmp_path / "s.txt").write_text(scene.format(name))
        with pytest.raises(RuntimeError):
            hostlib.parse_scene(str(tmp_path / "s.txt"))
    # and the valid small-depth forms still decode: 1-bit grey, 2 x 2, rows 0b10 / 0b01 -> white black / black white
    (tmp_path / "ok1.png").write_bytes(_png(2, 2, 1, 0, [[0b10000000], [0b01000000]]))
    (tmp_path / "s.txt").write_text(scene.format("ok1.png"))
    ps = hostlib.parse_scene(str(tmp_path / "s.txt"))
    tex = ps["textures"][0] if isinstance(ps, dict) else ps.textures[0]
    assert np.asarray(tex).reshape(2, 2, 3)[..., 0].max() > 0


def test_hdr_rle_and_pfm(tmp_path):
    rng = np.random.default_rng(5)
    w, h = 16, 3
    rgbe = rng.integers(1, 200, (h, w, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(120, 136, (h, w))
    rgbe[1, 4:12] = rgbe[1, 4]  # a run
    with open(tmp_path / "rle.hdr", "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            f.write(bytes([2, 2, 0, w]))
            for k in range(4):
                row = rgbe[y, :, k]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and row[x + run] == row[x] and run < 127:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, row[x]]))
                        x += run
                    else:
                        lit = min(w - x, 5)
                        f.write(bytes([lit]) + row[x:x + lit].tobytes())
                        x += lit
    pfm = rng.random((2, 3, 3)).astype("<f4")
    with open(tmp_path / "t.pfm", "wb") as f:
        f.write(b"PF\n3 2\n-1.0\n" + pfm[::-1].tobytes())  # PFM stores the bottom row first
    (tmp_path / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "s.txt").write_text("Material m\nType Lambertian\nBaseColor t.pfm\nMetallic 0\nRoughness 1\nIor 1\nNormalMap Null\n\n"
                                    "Object 0\na.obj\nMaterial m\n\nEnvMap rle.hdr\n")
    p = hostlib.parse_scene(str(tmp_path / "s.txt"))
    want = rgbe[..., :3].astype(np.float32) * np.exp2(rgbe[..., 3:4].astype(np.float32) - 136)
    assert np.array_equal(p["textures"][1], want)
    assert np.array_equal(p["textures"][0], pfm[::-1])  # decoded top row first, then flipped like every non-env texture
