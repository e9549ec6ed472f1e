"""ctypes binding of libradish_host.so (include/radish_host.h): BVH / alias table / light list / camera."""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "csrc", "libradish_host.so")

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
