"""Binary layouts shared with the reference: the "keep sceneStructs.h" contract.

Every dtype is packed, little-endian and size-checked against the reference's sizeof
(`/root/reference/src/sceneStructs.h`, `material.h`, `bvh.h`, `sampler.h`, `restir.h`).
"""
import numpy as np

# src/sceneStructs.h:118-130 — 196 bytes
CAMERA_DTYPE = np.dtype(
    [
        ("resolution", "<i4", (2,)),
        ("position", "<f4", (3,)),
        ("rotation", "<f4", (3,)),
        ("view", "<f4", (3,)),
        ("up", "<f4", (3,)),
        ("right", "<f4", (3,)),
        ("fov", "<f4", (2,)),
        ("pixelLength", "<f4", (2,)),
        ("rotationMatInv", "<f4", (9,)),
        ("viewProjection", "<f4", (16,)),
        ("lensRadius", "<f4"),
        ("focalDist", "<f4"),
        ("tanFovY", "<f4"),
    ]
)
assert CAMERA_DTYPE.itemsize == 196

# src/material.h:276-286 — 44 bytes
MATERIAL_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("baseColor", "<f4", (3,)),
        ("metallic", "<f4"),
        ("roughness", "<f4"),
        ("ior", "<f4"),
        ("baseColorMapId", "<i4"),
        ("normalMapId", "<i4"),
        ("metallicMapId", "<i4"),
        ("roughnessMapId", "<i4"),
    ]
)
assert MATERIAL_DTYPE.itemsize == 44

# Material::Type (src/material.h:129)
LAMBERTIAN, METALLIC_WORKFLOW, DIELECTRIC, DISNEY, LIGHT = 0, 1, 2, 3, 4

# src/bvh.h:167-169 — 12 bytes
MTBVH_NODE_DTYPE = np.dtype([("primitiveId", "<i4"), ("boundingBoxId", "<i4"), ("nextNodeIfMiss", "<i4")])
assert MTBVH_NODE_DTYPE.itemsize == 12

# src/sampler.h:66-69 — 8 bytes
BINOMIAL_DTYPE = np.dtype([("prob", "<f4"), ("failId", "<i4")])
assert BINOMIAL_DTYPE.itemsize == 8

# src/restir.h:88-99 — 36 bytes
RESERVOIR_DTYPE = np.dtype(
    [("Li", "<f4", (3,)), ("wi", "<f4", (3,)), ("dist", "<f4"), ("numSamples", "<i4"), ("weight", "<f4")]
)
assert RESERVOIR_DTYPE.itemsize == 36

HIT_DTYPE = np.dtype([("primId", "<i4"), ("u", "<f4"), ("v", "<f4"), ("t", "<f4")])
assert HIT_DTYPE.itemsize == 16

SOBOL_NUM = 10000  # SobolSampleNum, src/sampler.h:12
SOBOL_DIM = 200  # SobolSampleDim, src/sampler.h:13


def make_material(type=LAMBERTIAN, baseColor=(0.9, 0.9, 0.9), metallic=0.0, roughness=1.0, ior=1.5):
    """Material with the reference's defaults (src/material.h:276-286); no textures."""
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["type"] = type
    m["baseColor"] = baseColor
    m["metallic"] = metallic
    m["roughness"] = roughness
    m["ior"] = ior
    m["baseColorMapId"] = m["normalMapId"] = m["metallicMapId"] = m["roughnessMapId"] = -1
    return m
