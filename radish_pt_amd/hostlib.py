"""ctypes binding of libradish_host.so (include/radish_host.h): BVH / alias table / light list / camera."""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
# RADISH_HOST_LIB selects another build of the same library (tests/tools/sanitize_cpu.sh: the ASan/UBSan build)
HOST_LIB_PATH = os.environ.get("RADISH_HOST_LIB") or os.path.join(_HERE, "csrc", "libradish_host.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(
                f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first"
            )
        l = C.CDLL(HOST_LIB_PATH)
        l.rdh_build_bvh.restype = C.c_int32
        l.rdh_build_bvh.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        l.rdh_build_alias_table.restype = C.c_int32
        l.rdh_build_alias_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_float)]
        l.rdh_build_light_list.restype = C.c_int32
        l.rdh_build_light_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 3
        l.rdh_build_envmap_pdf.restype = C.c_int32
        l.rdh_build_envmap_pdf.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        l.rdh_camera_update.restype = None
        l.rdh_camera_update.argtypes = [C.c_void_p]
        _lib = l
    return _lib


def build_bvh(vertices):
    """vertices: float32 [3N,3] triangle soup → (boxes float32 [2N-1,6], nodes [6] of MTBVH_NODE_DTYPE[2N-1])."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    n = v.shape[0] // 3
    size = 2 * n - 1
    boxes = np.empty((size, 6), dtype=np.float32)
    nodes = [np.empty(size, dtype=L.MTBVH_NODE_DTYPE) for _ in range(6)]
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in nodes])
    rc = lib().rdh_build_bvh(v.ctypes.data, n, boxes.ctypes.data, ptrs)
    if rc != size:
        raise RuntimeError(f"rdh_build_bvh failed: {rc}")
    return boxes, nodes


def build_alias_table(values):
    vals = np.ascontiguousarray(values, dtype=np.float32)
    table = np.zeros(len(vals), dtype=L.BINOMIAL_DTYPE)
    s = C.c_float(0)
    rc = lib().rdh_build_alias_table(vals.ctypes.data, len(vals), table.ctypes.data, C.byref(s))
    if rc != 0:
        raise RuntimeError(f"rdh_build_alias_table failed: {rc}")
    return table, np.float32(s.value)


def build_light_list(vertices, material_ids, materials):
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    n = v.shape[0] // 3
    ids = np.ascontiguousarray(material_ids, dtype=np.int32)
    mats = np.ascontiguousarray(materials, dtype=L.MATERIAL_DTYPE)
    prim = np.empty(n, dtype=np.int32)
    rad = np.empty((n, 3), dtype=np.float32)
    power = np.empty(n, dtype=np.float32)
    cnt = lib().rdh_build_light_list(
        v.ctypes.data, ids.ctypes.data, n, mats.ctypes.data, len(mats), prim.ctypes.data, rad.ctypes.data,
        power.ctypes.data,
    )
    if cnt < 0:
        raise RuntimeError(f"rdh_build_light_list failed: {cnt}")
    return prim[:cnt].copy(), rad[:cnt].copy(), power[:cnt].copy()


def build_envmap_sampler(texels, width, height):
    """Alias table over the env map's pixels + its total power (Scene::createLightSampler, src/scene.cpp:146-157)."""
    t = np.ascontiguousarray(texels, dtype=np.float32).reshape(-1, 3)
    pdf = np.empty(width * height, dtype=np.float32)
    rc = lib().rdh_build_envmap_pdf(t.ctypes.data, width, height, pdf.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"rdh_build_envmap_pdf failed: {rc}")
    return build_alias_table(pdf)


def make_camera(width, height, eye, rotation, fovy, lens_radius=0.0, focal_dist=1.0):
    """Camera as Scene::loadCamera + Camera::update would leave it (src/scene.cpp:319-392)."""
    cam = np.zeros((), dtype=L.CAMERA_DTYPE)
    cam["resolution"] = (width, height)
    cam["position"] = eye
    cam["rotation"] = rotation
    cam["fov"] = (0.0, fovy)
    cam["lensRadius"] = lens_radius
    cam["focalDist"] = focal_dist
    buf = np.frombuffer(cam.tobytes(), dtype=np.uint8).copy()
    lib().rdh_camera_update(buf.ctypes.data)
    return np.frombuffer(buf.tobytes(), dtype=L.CAMERA_DTYPE)[0].copy()


# ----------------------------------------------------------------------------------------------------------------------
# Scene files (include/radish_host.h: rdh_scene_parse)
# ----------------------------------------------------------------------------------------------------------------------
class _HostTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.POINTER(C.c_float))]


class _ParsedScene(C.Structure):
    _fields_ = [
        ("numPrims", C.c_int32), ("vertices", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
        ("texcoords", C.POINTER(C.c_float)), ("materialIds", C.POINTER(C.c_int32)),
        ("numMaterials", C.c_int32), ("materials", C.c_void_p),
        ("numTextures", C.c_int32), ("textures", C.POINTER(_HostTexture)),
        ("envMapTexId", C.c_int32), ("apertureMaskTexId", C.c_int32), ("hasCamera", C.c_int32),
        ("camera", C.c_uint8 * 196), ("traceDepth", C.c_int32), ("iterations", C.c_int32),
        ("imageName", C.c_char * 256), ("opaque", C.c_void_p),
    ]


_DECODE_FN = C.CFUNCTYPE(C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32),
                         C.POINTER(C.c_int32), C.c_void_p)


def _pillow_decode(path, flip_y, rgb_out, w_out, h_out, _user):
    """Decoder for formats the C++ loader does not read itself (JPEG, BMP, TGA, ...): stbi_loadf semantics — 3 channels,
    LDR sample v → v/255 (stbi_ldr_to_hdr_gamma(1.f), src/scene.cpp:109)."""
    try:
        from PIL import Image

        im = np.asarray(Image.open(path.decode()).convert("RGB"), dtype=np.float32) / np.float32(255.0)
        if flip_y:
            im = im[::-1]
        im = np.ascontiguousarray(im)
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        buf = libc.malloc(im.nbytes)
        C.memmove(buf, im.ctypes.data, im.nbytes)
        rgb_out[0] = C.cast(buf, C.POINTER(C.c_float))
        w_out[0], h_out[0] = im.shape[1], im.shape[0]
        return 0
    except Exception:
        return 1


def parse_scene(path):
    """rdh_scene_parse → dict of numpy copies: vertices/normals/texcoords/material_ids/materials/textures (list of [h,w,3]),
    env_map_tex_id, camera (CAMERA_DTYPE scalar or None), trace_depth, iterations, image_name."""
    l = lib()
    l.rdh_scene_parse.restype = C.c_int32
    l.rdh_scene_parse.argtypes = [C.c_char_p, _DECODE_FN, C.c_void_p, C.POINTER(C.POINTER(_ParsedScene)), C.c_char_p, C.c_int32]
    l.rdh_scene_parse_free.restype = None
    l.rdh_scene_parse_free.argtypes = [C.POINTER(_ParsedScene)]
    out = C.POINTER(_ParsedScene)()
    err = C.create_string_buffer(1024)
    cb = _DECODE_FN(_pillow_decode)
    rc = l.rdh_scene_parse(os.fsencode(path), cb, None, C.byref(out), err, len(err))
    if rc != 0:
        raise RuntimeError(f"rdh_scene_parse({path}): {err.value.decode(errors='replace')} (code {rc})")
    try:
        s = out.contents
        n = s.numPrims
        res = {
            "vertices": np.ctypeslib.as_array(s.vertices, (3 * n, 3)).copy(),
            "normals": np.ctypeslib.as_array(s.normals, (3 * n, 3)).copy(),
            "texcoords": np.ctypeslib.as_array(s.texcoords, (3 * n, 2)).copy(),
            "material_ids": np.ctypeslib.as_array(s.materialIds, (n,)).copy(),
            "materials": np.frombuffer(C.string_at(s.materials, 44 * s.numMaterials), dtype=L.MATERIAL_DTYPE).copy(),
            "textures": [np.ctypeslib.as_array(s.textures[i].data, (s.textures[i].height, s.textures[i].width, 3)).copy()
                         for i in range(s.numTextures)],
            "env_map_tex_id": int(s.envMapTexId),
            "aperture_mask_tex_id": int(s.apertureMaskTexId),
            "camera": np.frombuffer(bytes(s.camera), dtype=L.CAMERA_DTYPE)[0].copy() if s.hasCamera else None,
            "trace_depth": int(s.traceDepth), "iterations": int(s.iterations),
            "image_name": s.imageName.decode(errors="replace"),
        }
    finally:
        l.rdh_scene_parse_free(out)
    return res
