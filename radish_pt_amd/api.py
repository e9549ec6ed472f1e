"""Host-side mirror of the reference's render API over libradish_hip.so (include/radish_hip.h).

Same names, argument meaning and side effects as the reference's C++ free functions and globals
(`/root/reference/src/pathtrace.h:18-23`, `restir.h:103-106`, `gBuffer.h:24-27`, `common.h:50-72`):

    Settings.traceDepth, Settings.reservoirReuse, State.looper, State.scene
    pathTraceInit() / pathTraceFree()
    pathTrace(directIllum, indirectIllum, iter)          # advances State.looper mod 10000
    pathTraceDirect(directIllum, iter)
    ReSTIRInit() / ReSTIRFree() / ReSTIRDirect(directIllum, iter, gBuffer)
    GBuffer.create(w, h) / destroy() / render(devScene, cam) / update(cam)
    DevScene.create(scene) / destroy()

Images are CUDA (HIP) float32 torch tensors of shape [H*W, 3] — the reference's `glm::vec3*` device buffers.  PyTorch
is plumbing here (device memory, streams, torch.distributed); every computation happens in the HIP library.  There is
no CPU path: importing works without a GPU, calling anything raises.
"""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
# RADISH_HIP_LIB selects an alternative build of the same library (tuning experiments); never a different backend.
HIP_LIB_PATH = os.environ.get("RADISH_HIP_LIB") or os.path.join(_HERE, "csrc", "libradish_hip.so")

RDH_PT_MEGAKERNEL, RDH_PT_WAVEFRONT, RDH_PT_SORT_MATERIAL, RDH_PT_COUNT, RDH_PT_PROFILE = 0, 1, 2, 4, 8
RDH_PT_PERSISTENT = 16
RDH_PT_NO_SCHEDULE = 32
RDH_PT_ONE_LANE_PER_PIXEL = 64
RDH_PT_MEGA_GBUFFER = RDH_PT_ONE_LANE_PER_PIXEL
RDH_PT_NO_DEFER = 128
RDH_PT_WG_PER_RAY = 256
RDH_PT_PARTITION_GBUFFER = 512
RDH_PT_RESTIR_FUSED = 1024
RDH_PT_WF_SUBFRAMES = 2048
RDH_PT_WF_SMALL_LISTS = 4096
RDH_PT_NO_PAIRS = 16384
RDH_PT_NO_PACKETS = 8192
RDH_PT_PAIRS = 32768
RDH_PT_AUTO = 65536
SOBOL_SAMPLE_NUM = 10000  # SobolSampleNum, src/sampler.h:12

# Every symbol include/radish_hip.h declares (tests check that the library exports all of them).
EXPORTS = [
    "rdh_create", "rdh_destroy", "rdh_last_error", "rdh_set_stream", "rdh_synchronize", "rdh_scene_upload",
    "rdh_scene_free", "rdh_set_camera", "rdh_set_partition", "rdh_tiles_per_rank", "rdh_untile", "rdh_path_trace",
    "rdh_path_trace_direct", "rdh_gbuffer_render", "rdh_restir_init", "rdh_restir_free", "rdh_restir_direct",
    "rdh_restir_read", "rdh_restir_read_scratch", "rdh_copy_image_to_pbo", "rdh_denoise_eaw", "rdh_denoise_svgf", "rdh_denoise_modulate",
    "rdh_denoise_add", "rdh_denoise_temporal_accumulate", "rdh_denoise_estimate_variance", "rdh_denoise_filter_variance", "rdh_restir_exchange_pack", "rdh_restir_exchange_unpack", "rdh_trace_closest", "rdh_trace_occluded", "rdh_counters_reset", "rdh_counters_read",
    "rdh_last_kernel_ms", "rdh_profile_reset", "rdh_profile_read", "rdh_debug_persist_stamps", "rdh_debug_persist_phases",
    "rdh_gbuffer_exchange_pack", "rdh_gbuffer_exchange_unpack", "rdh_comm_unique_id", "rdh_comm_init", "rdh_comm_destroy",
    "rdh_dump_rays", "rdh_set_occupancy_share", "rdh_allgather_tiles", "rdh_path_trace_gathered", "rdh_restir_exchange", "rdh_restir_direct_gathered", "rdh_gbuffer_exchange",
    "rdh_comm_init_all", "rdh_path_trace_gathered_all", "rdh_gbuffer_exchange_all", "rdh_restir_direct_gathered_all",
    "rdh_comm_set_overlap", "rdh_comm_join",
]


class SceneDescC(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p), ("boundingBoxes", C.c_void_p),
        ("bvhNodes", C.c_void_p * 6), ("bvhSize", C.c_int32), ("numPrims", C.c_int32),
        ("materialIds", C.c_void_p), ("materials", C.c_void_p), ("numMaterials", C.c_int32),
        ("numLights", C.c_int32), ("lightPrimIds", C.c_void_p), ("lightUnitRadiance", C.c_void_p),
        ("sumLightPowerInv", C.c_float), ("lightSamplerLength", C.c_int32), ("lightSampler", C.c_void_p),
        ("sampleSequence", C.c_void_p),
        ("numTextures", C.c_int32), ("textures", C.c_void_p), ("envMapTexId", C.c_int32),
        ("envMapSamplerLength", C.c_int32), ("envMapSampler", C.c_void_p),
    ]


class TextureC(C.Structure):  # rdh_texture
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.c_void_p)]


class GBufferC(C.Structure):  # == the reference's GBuffer, 272 bytes
    _fields_ = [
        ("albedo", C.c_void_p), ("normal", C.c_void_p * 2), ("motion", C.c_void_p), ("depth", C.c_void_p * 2),
        ("primId", C.c_void_p * 2), ("frameIdx", C.c_int32), ("lastCam", C.c_uint8 * 196), ("width", C.c_int32),
        ("height", C.c_int32),
    ]


assert C.sizeof(GBufferC) == 272


class CountersC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closestRays", "anyRays", "nodeVisits", "triTests", "closestHits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RestirParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("reuseMask", "risCount", "numSpatial", "temporalClamp", "faithfulRIS")]


_lib = None


def lib():
    """Load libradish_hip.so; raises if it has not been built — there is no fallback implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(
                f"{HIP_LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "radish_pt_amd has no CPU or PyTorch fallback for the render path."
            )
        l = C.CDLL(HIP_LIB_PATH)
        vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
        sig = {
            "rdh_create": ([C.POINTER(vp), i32], i32),
            "rdh_destroy": ([vp], None),
            "rdh_last_error": ([vp], C.c_char_p),
            "rdh_set_stream": ([vp, vp], i32),
            "rdh_synchronize": ([vp], i32),
            "rdh_scene_upload": ([vp, C.POINTER(SceneDescC)], i32),
            "rdh_scene_free": ([vp], i32),
            "rdh_set_camera": ([vp, vp], i32),
            "rdh_set_partition": ([vp, i32, i32, i32], i32),
            "rdh_tiles_per_rank": ([vp], i32),
            "rdh_untile": ([vp, vp, vp], i32),
            "rdh_path_trace": ([vp, vp, vp, i32, i32, i32, u32], i32),
            "rdh_path_trace_direct": ([vp, vp, i32, i32, u32], i32),
            "rdh_gbuffer_render": ([vp, C.POINTER(GBufferC), u32], i32),
            "rdh_restir_init": ([vp], i32),
            "rdh_restir_free": ([vp], i32),
            "rdh_restir_direct": ([vp, vp, i32, i32, C.POINTER(GBufferC), C.POINTER(RestirParamsC), u32], i32),
            "rdh_restir_read": ([vp, i32, vp], i32),
            "rdh_restir_read_scratch": ([vp, i32, vp, i64], i64),
            "rdh_copy_image_to_pbo": ([vp, vp, vp, i32, i32, i32, i32, C.c_float], i32),
            "rdh_denoise_eaw": ([vp, vp, vp, C.POINTER(GBufferC), vp] + [C.c_float] * 3 + [i32], i32),
            "rdh_denoise_svgf": ([vp, vp, vp, vp, vp, vp, C.POINTER(GBufferC), vp] + [C.c_float] * 3 + [i32], i32),
            "rdh_denoise_modulate": ([vp, vp, C.POINTER(GBufferC)], i32),
            "rdh_denoise_add": ([vp, vp, vp, vp, i32, i32], i32),
            "rdh_denoise_temporal_accumulate": ([vp, vp, vp, vp, vp, vp, C.POINTER(GBufferC), i32], i32),
            "rdh_denoise_estimate_variance": ([vp, vp, vp, i32, i32], i32),
            "rdh_denoise_filter_variance": ([vp, vp, vp, i32, i32], i32),
            "rdh_restir_exchange_pack": ([vp, vp], i32),
            "rdh_restir_exchange_unpack": ([vp, vp], i32),
            "rdh_trace_closest": ([vp, vp, i64, vp, u32], i32),
            "rdh_trace_occluded": ([vp, vp, i64, vp, u32], i32),
            "rdh_counters_reset": ([vp], i32),
            "rdh_counters_read": ([vp, C.POINTER(CountersC)], i32),
            "rdh_last_kernel_ms": ([vp, C.POINTER(C.c_float)], i32),
            "rdh_profile_reset": ([vp], i32),
            "rdh_profile_read": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)], i32),
            "rdh_debug_persist_stamps": ([vp, vp], i32),
            "rdh_debug_persist_phases": ([vp, vp], i32),
            "rdh_gbuffer_exchange_pack": ([vp, C.POINTER(GBufferC), vp], i32),
            "rdh_gbuffer_exchange_unpack": ([vp, C.POINTER(GBufferC), vp], i32),
            "rdh_comm_unique_id": ([vp], i32),
            "rdh_comm_init": ([vp, vp, i32, i32], i32),
            "rdh_comm_destroy": ([vp], i32),
            "rdh_comm_init_all": ([C.POINTER(vp), i32], i32),
            "rdh_path_trace_gathered_all": ([C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp), i32, i32, i32, u32], i32),
            "rdh_gbuffer_exchange_all": ([C.POINTER(vp), i32, C.POINTER(GBufferC)], i32),
            "rdh_restir_direct_gathered_all": ([C.POINTER(vp), i32, C.POINTER(vp), i32, i32, C.POINTER(GBufferC), C.POINTER(RestirParamsC), u32], i32),
            "rdh_comm_set_overlap": ([vp, i32], i32),
            "rdh_comm_join": ([vp], i32),
            "rdh_allgather_tiles": ([vp, vp, vp], i32),
            "rdh_path_trace_gathered": ([vp, vp, vp, i32, i32, i32, u32], i32),
            "rdh_restir_exchange": ([vp], i32),
            "rdh_restir_direct_gathered": ([vp, vp, i32, i32, C.POINTER(GBufferC), C.POINTER(RestirParamsC), u32], i32),
            "rdh_gbuffer_exchange": ([vp, C.POINTER(GBufferC)], i32),
            "rdh_dump_rays": ([vp, i32, i32, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)], i32),
            "rdh_set_occupancy_share": ([vp, i32], i32),
        }
        for name, (args, res) in sig.items():
            fn = getattr(l, name)
            fn.argtypes, fn.restype = args, res
        _lib = l
    return _lib


class RadishError(RuntimeError):
    pass


def _torch():
    import torch

    return torch


class Context:
    """One rdh_ctx (one device, one stream)."""

    def __init__(self, device=0, use_torch_stream=True):
        torch = _torch()  # imported first so that the HIP runtime torch ships is the one the library binds to
        if not torch.cuda.is_available():
            raise RadishError("radish_pt_amd needs a HIP device; there is no CPU fallback")
        h = C.c_void_p()
        rc = lib().rdh_create(C.byref(h), device)
        if rc != 0:
            raise RadishError(f"rdh_create failed with code {rc} (no usable HIP device?)")
        self.h = h
        self.device = device
        self.rank, self.world, self.tile = 0, 1, 64
        self.width = self.height = 0
        if use_torch_stream:
            self.check(lib().rdh_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))

    # ---- argument checks: the C ABI takes raw device pointers, so a wrongly shaped tensor would be a silent out-of-bounds
    # device write; every image-like argument is checked here (dtype, device, contiguity, minimum element count) ----
    def _ptr(self, t, min_elems, what, dtype=None):
        torch = _torch()
        dtype = dtype or torch.float32
        if not isinstance(t, torch.Tensor):
            raise RadishError(f"{what}: expected a torch tensor, got {type(t).__name__}")
        if not t.is_cuda or t.device.index != self.device:
            raise RadishError(f"{what}: tensor is on {t.device}, the context is on cuda:{self.device}")
        if t.dtype != dtype:
            raise RadishError(f"{what}: dtype {t.dtype}, expected {dtype}")
        if not t.is_contiguous():
            raise RadishError(f"{what}: tensor is not contiguous")
        if t.numel() < min_elems:
            raise RadishError(f"{what}: {t.numel()} elements, the call writes/reads {min_elems}")
        return t.data_ptr()

    def _image_elems(self, gathered=False):
        """floats of one vec3 image argument: the frame, or this rank's packed tiles on a partition"""
        if getattr(self, "world", 1) > 1 and not gathered:
            return self.tiles_per_rank() * self.tile * self.tile * 3
        return self.width * self.height * 3

    def check(self, rc):
        if rc != 0:
            msg = lib().rdh_last_error(self.h)
            raise RadishError(f"libradish_hip error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            lib().rdh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene / camera ----
    def upload_scene(self, sd):
        d = SceneDescC()
        d.vertices, d.normals, d.texcoords = sd.vertices.ctypes.data, sd.normals.ctypes.data, sd.texcoords.ctypes.data
        d.boundingBoxes = sd.boxes.ctypes.data
        d.bvhNodes = (C.c_void_p * 6)(*[a.ctypes.data for a in sd.nodes])
        d.bvhSize, d.numPrims = sd.bvh_size, sd.num_prims
        d.materialIds, d.materials, d.numMaterials = sd.material_ids.ctypes.data, sd.materials.ctypes.data, len(sd.materials)
        d.numLights = sd.num_lights
        d.lightPrimIds = sd.light_prim_ids.ctypes.data
        d.lightUnitRadiance = sd.light_unit_radiance.ctypes.data
        d.sumLightPowerInv = float(sd.sum_light_power_inv)
        d.lightSamplerLength = len(sd.light_sampler)
        d.lightSampler = sd.light_sampler.ctypes.data
        d.sampleSequence = sd.sobol.ctypes.data
        texs = getattr(sd, "textures", [])
        tex_c = (TextureC * max(len(texs), 1))()
        for i, t in enumerate(texs):
            tex_c[i].width, tex_c[i].height, tex_c[i].data = t.shape[1], t.shape[0], t.ctypes.data
        d.numTextures = len(texs)
        d.textures = C.cast(tex_c, C.c_void_p)
        d.envMapTexId = getattr(sd, "env_map_tex_id", -1)
        env = getattr(sd, "env_map_sampler", None)
        d.envMapSamplerLength = 0 if env is None else len(env)
        d.envMapSampler = None if env is None or len(env) == 0 else env.ctypes.data
        self.check(lib().rdh_scene_upload(self.h, C.byref(d)))

    def set_camera(self, cam):
        buf = np.frombuffer(np.asarray(cam, dtype=L.CAMERA_DTYPE).tobytes(), np.uint8).copy()
        self.check(lib().rdh_set_camera(self.h, buf.ctypes.data))
        self.width, self.height = (int(v) for v in cam["resolution"])

    def set_partition(self, rank, world, tile=64):
        self.check(lib().rdh_set_partition(self.h, rank, world, tile))
        self.rank, self.world, self.tile = rank, world, tile

    def tiles_per_rank(self):
        n = lib().rdh_tiles_per_rank(self.h)
        if n < 0:
            self.check(n)
        return n

    def untile(self, gathered, frame):
        shard = self.tiles_per_rank() * self.tile * self.tile * 3
        self.check(lib().rdh_untile(self.h, self._ptr(gathered, shard * self.world, "untile: gathered"),
                                    self._ptr(frame, self.width * self.height * 3, "untile: frame")))

    # ---- hot path ----
    def path_trace(self, direct, indirect, iter, looper, max_depth, flags=RDH_PT_PERSISTENT):
        n = self._image_elems() if self.width else 0
        self.check(lib().rdh_path_trace(self.h, self._ptr(direct, n, "pathTrace: directIllum"),
                                        self._ptr(indirect, n, "pathTrace: indirectIllum"), iter, looper, max_depth, flags))

    def path_trace_direct(self, direct, iter, looper, flags=0):
        n = self._image_elems() if self.width else 0
        self.check(lib().rdh_path_trace_direct(self.h, self._ptr(direct, n, "pathTraceDirect: directIllum"), iter, looper, flags))

    def gbuffer_render(self, gb_c, flags=0):
        self.check(lib().rdh_gbuffer_render(self.h, C.byref(gb_c), flags))

    def restir_init(self):
        self.check(lib().rdh_restir_init(self.h))

    def restir_free(self):
        self.check(lib().rdh_restir_free(self.h))

    def restir_direct(self, direct, iter, looper, gb_c, reuse_mask, ris_count=32, num_spatial=5, temporal_clamp=20,
                      faithful_ris=1, flags=0):
        p = RestirParamsC(reuse_mask, ris_count, num_spatial, temporal_clamp, faithful_ris)
        n = self._image_elems() if self.width else 0
        self.check(lib().rdh_restir_direct(self.h, self._ptr(direct, n, "ReSTIRDirect: directIllum"), iter, looper, C.byref(gb_c),
                                           C.byref(p), flags))

    def copy_image_to_pbo(self, pbo, image, width, height, kind=0, tone_mapping=0, scale=1.0):
        torch = _torch()
        n = width * height
        if pbo.numel() * pbo.element_size() < 4 * n or not pbo.is_cuda or not pbo.is_contiguous():
            raise RadishError("copyImageToPBO: PBO must be a contiguous device buffer of 4 bytes per pixel")
        want = {0: (3 * n, torch.float32), 1: (2 * n, torch.float32), 2: (n, torch.float32), 3: (n, torch.int32)}.get(kind)
        if want is None:
            raise RadishError(f"copyImageToPBO: kind {kind}")
        self.check(lib().rdh_copy_image_to_pbo(self.h, pbo.data_ptr(), self._ptr(image, want[0], "copyImageToPBO: image", want[1]),
                                               width, height, kind, tone_mapping, scale))

    # ---- denoisers (src/denoiser.cu) ----
    @staticmethod
    def _cam_buf(cam):
        return np.frombuffer(np.asarray(cam, dtype=L.CAMERA_DTYPE).tobytes(), np.uint8).copy()

    def denoise_eaw(self, color_out, color_in, gb_c, cam, sig_lumin, sig_normal, sig_depth, level):
        cb = self._cam_buf(cam)
        self.check(lib().rdh_denoise_eaw(self.h, color_out.data_ptr(), color_in.data_ptr(), C.byref(gb_c), cb.ctypes.data,
                                         sig_lumin, sig_normal, sig_depth, level))

    def denoise_svgf(self, color_out, color_in, var_out, var_in, var_filtered, gb_c, cam, sig_lumin, sig_normal, sig_depth, level):
        cb = self._cam_buf(cam)
        self.check(lib().rdh_denoise_svgf(self.h, color_out.data_ptr(), color_in.data_ptr(), var_out.data_ptr(), var_in.data_ptr(),
                                          var_filtered.data_ptr(), C.byref(gb_c), cb.ctypes.data, sig_lumin, sig_normal,
                                          sig_depth, level))

    def denoise_modulate(self, image, gb_c):
        self.check(lib().rdh_denoise_modulate(self.h, image.data_ptr(), C.byref(gb_c)))

    def denoise_add(self, out, in1, in2, width, height):
        self.check(lib().rdh_denoise_add(self.h, out.data_ptr(), in1.data_ptr(), in2.data_ptr(), width, height))

    def denoise_temporal_accumulate(self, color_out, color_in_accum, moment_out, moment_in_accum, color_in, gb_c, first):
        self.check(lib().rdh_denoise_temporal_accumulate(self.h, color_out.data_ptr(), color_in_accum.data_ptr(),
                                                         moment_out.data_ptr(), moment_in_accum.data_ptr(),
                                                         color_in.data_ptr(), C.byref(gb_c), 1 if first else 0))

    def denoise_estimate_variance(self, variance, moment, width, height):
        self.check(lib().rdh_denoise_estimate_variance(self.h, variance.data_ptr(), moment.data_ptr(), width, height))

    def denoise_filter_variance(self, var_out, var_in, width, height):
        self.check(lib().rdh_denoise_filter_variance(self.h, var_out.data_ptr(), var_in.data_ptr(), width, height))

    def restir_exchange_pack(self, packed):
        shard = self.tiles_per_rank() * self.tile * self.tile * 9
        self.check(lib().rdh_restir_exchange_pack(self.h, self._ptr(packed, shard, "restir_exchange_pack")))

    def restir_exchange_unpack(self, gathered):
        shard = self.tiles_per_rank() * self.tile * self.tile * 9
        self.check(lib().rdh_restir_exchange_unpack(self.h, self._ptr(gathered, shard * self.world, "restir_exchange_unpack")))

    def gbuffer_exchange_pack(self, gb_c, packed):
        shard = self.tiles_per_rank() * self.tile * self.tile * 9
        self.check(lib().rdh_gbuffer_exchange_pack(self.h, C.byref(gb_c), self._ptr(packed, shard, "gbuffer_exchange_pack")))

    def gbuffer_exchange_unpack(self, gb_c, gathered):
        shard = self.tiles_per_rank() * self.tile * self.tile * 9
        self.check(lib().rdh_gbuffer_exchange_unpack(self.h, C.byref(gb_c), self._ptr(gathered, shard * self.world, "gbuffer_exchange_unpack")))

    # ---- collectives inside the library (RCCL, bound at run time) ----
    @staticmethod
    def comm_unique_id():
        """128 bytes (ncclUniqueId): made on rank 0, handed to every rank's comm_init."""
        buf = (C.c_uint8 * 128)()
        rc = lib().rdh_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise RadishError(f"rdh_comm_unique_id failed with code {rc} (RCCL not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128)(*unique_id)
        self.check(lib().rdh_comm_init(self.h, C.cast(buf, C.c_void_p), rank, world))
        self.rank, self.world = rank, world

    def comm_destroy(self):
        self.check(lib().rdh_comm_destroy(self.h))

    # ---- ONE process, n GPUs: the reference's host model (one frame loop, main.cpp:163-202) ----
    @staticmethod
    def comm_init_all(ctxs):
        """ctxs[i] (one Context per DEVICE) becomes rank i of len(ctxs): rdh_comm_init_all (grouped ncclCommInitRank)."""
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        rc = lib().rdh_comm_init_all(arr, n)
        if rc != 0:
            ctxs[0].check(rc)
        for i, c in enumerate(ctxs):
            c.rank, c.world = i, n

    @staticmethod
    def path_trace_gathered_all(ctxs, direct_frames, indirect_frames, iter, looper, max_depth, flags=RDH_PT_PERSISTENT):
        """pathTrace on every context of this process (rdh_path_trace_gathered_all): whole-frame images in and out per device."""
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        px = ctxs[0].width * ctxs[0].height * 3
        d = (C.c_void_p * n)(*[c._ptr(t, px, "path_trace_gathered_all: direct") for c, t in zip(ctxs, direct_frames)])
        i = (C.c_void_p * n)(*[c._ptr(t, px, "path_trace_gathered_all: indirect") for c, t in zip(ctxs, indirect_frames)])
        rc = lib().rdh_path_trace_gathered_all(arr, n, d, i, iter, looper, max_depth, flags)
        if rc != 0:
            msgs = [lib().rdh_last_error(c.h) for c in ctxs]
            raise RadishError(f"libradish_hip error {rc}: " + "; ".join(f"ctx {k}: {m.decode()}" for k, m in enumerate(msgs) if m))

    def allgather_tiles(self, packed, frame):
        shard = self.tiles_per_rank() * self.tile * self.tile * 3
        self.check(lib().rdh_allgather_tiles(self.h, self._ptr(packed, shard, "allgather_tiles: packed"),
                                             self._ptr(frame, self.width * self.height * 3, "allgather_tiles: frame")))

    def path_trace_gathered(self, direct_frame, indirect_frame, iter, looper, max_depth, flags=RDH_PT_PERSISTENT):
        n = self.width * self.height * 3
        self.check(lib().rdh_path_trace_gathered(self.h, self._ptr(direct_frame, n, "path_trace_gathered: direct"),
                                                 self._ptr(indirect_frame, n, "path_trace_gathered: indirect"), iter, looper,
                                                 max_depth, flags))

    def restir_exchange(self):
        self.check(lib().rdh_restir_exchange(self.h))

    def comm_set_overlap(self, enable):
        """ReSTIR's exchanges on a communication stream beside rendering (default on) or on the render stream (off)."""
        self.check(lib().rdh_comm_set_overlap(self.h, 1 if enable else 0))

    def comm_join(self):
        """The render stream waits for the exchanges in flight (before anything else reads G-buffer planes they complete)."""
        self.check(lib().rdh_comm_join(self.h))

    @staticmethod
    def _all_error(ctxs, rc):
        msgs = [lib().rdh_last_error(c.h) for c in ctxs]
        raise RadishError(f"libradish_hip error {rc}: " + "; ".join(f"ctx {k}: {m.decode()}" for k, m in enumerate(msgs) if m))

    @staticmethod
    def gbuffer_exchange_all(ctxs, gb_cs):
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        gbs = (GBufferC * n)(*gb_cs)
        rc = lib().rdh_gbuffer_exchange_all(arr, n, gbs)
        if rc != 0:
            Context._all_error(ctxs, rc)

    @staticmethod
    def restir_direct_gathered_all(ctxs, direct_frames, iter, looper, gb_cs, reuse_mask, ris_count=32, num_spatial=5, temporal_clamp=20,
                                   faithful_ris=1, flags=0):
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        px = ctxs[0].width * ctxs[0].height * 3
        d = (C.c_void_p * n)(*[c._ptr(t, px, "restir_direct_gathered_all") for c, t in zip(ctxs, direct_frames)])
        gbs = (GBufferC * n)(*gb_cs)
        p = RestirParamsC(reuse_mask, ris_count, num_spatial, temporal_clamp, faithful_ris)
        rc = lib().rdh_restir_direct_gathered_all(arr, n, d, iter, looper, gbs, C.byref(p), flags)
        if rc != 0:
            Context._all_error(ctxs, rc)

    def restir_direct_gathered(self, direct_frame, iter, looper, gb_c, reuse_mask, ris_count=32, num_spatial=5, temporal_clamp=20,
                               faithful_ris=1, flags=0):
        p = RestirParamsC(reuse_mask, ris_count, num_spatial, temporal_clamp, faithful_ris)
        self.check(lib().rdh_restir_direct_gathered(self.h, self._ptr(direct_frame, self.width * self.height * 3, "restir_direct_gathered"),
                                                    iter, looper, C.byref(gb_c), C.byref(p), flags))

    def gbuffer_exchange(self, gb_c):
        self.check(lib().rdh_gbuffer_exchange(self.h, C.byref(gb_c)))

    def restir_read(self, which):
        out = np.zeros(self.width * self.height, dtype=L.RESERVOIR_DTYPE)
        self.check(lib().rdh_restir_read(self.h, which, out.ctypes.data))
        return out

    def restir_read_scratch(self, which):
        """What the last split pass 1 left per slot: 0 primary rays [slots, 6], 1 shadow segments [slots, 6], 2 the set-aside
        lists as int32 {count[4], primarySlots[256], shadowSlots[256]} (rdh_restir_read_scratch)."""
        nbytes = int(lib().rdh_restir_read_scratch(self.h, which, None, 0))
        if nbytes < 0:
            self.check(nbytes)
        out = np.zeros(nbytes // 4, np.int32 if which == 2 else np.float32)
        got = int(lib().rdh_restir_read_scratch(self.h, which, out.ctypes.data, nbytes))
        if got < 0:
            self.check(got)
        return out if which == 2 else out.reshape(-1, 6)

    def trace_closest(self, rays, hits, flags=RDH_PT_PERSISTENT):  # flags without RDH_PT_PERSISTENT: one lane per ray
        n = rays.numel() // 6  # n == 0 still reaches the library (it returns at once)
        self.check(lib().rdh_trace_closest(self.h, self._ptr(rays, 6 * n, "trace_closest: rays"), n,
                                           self._ptr(hits, 4 * n, "trace_closest: hits", _torch().int32), flags))

    def trace_occluded(self, segments, out, flags=RDH_PT_PERSISTENT):
        n = segments.numel() // 6
        self.check(lib().rdh_trace_occluded(self.h, self._ptr(segments, 6 * n, "trace_occluded: segments"), n,
                                            self._ptr(out, n, "trace_occluded: out", _torch().int32), flags))

    def dump_rays(self, looper, max_depth):
        """The frame's own ray lists (see rdh_dump_rays): (closest rays float32 [n,6], occlusion segments float32 [m,6]) on the device."""
        torch = _torch()
        nc, na = C.c_int64(0), C.c_int64(0)
        self.check(lib().rdh_dump_rays(self.h, looper, max_depth, None, 0, None, 0, C.byref(nc), C.byref(na)))
        dev = torch.device("cuda", self.device)
        closest = torch.zeros(max(nc.value, 1), 6, device=dev)
        segs = torch.zeros(max(na.value, 1), 6, device=dev)
        n2, a2 = C.c_int64(0), C.c_int64(0)
        self.check(lib().rdh_dump_rays(self.h, looper, max_depth, closest.data_ptr(), nc.value, segs.data_ptr(), na.value,
                                       C.byref(n2), C.byref(a2)))
        assert (n2.value, a2.value) == (nc.value, na.value)
        return closest[:nc.value], segs[:na.value]

    def set_occupancy_share(self, share):
        self.check(lib().rdh_set_occupancy_share(self.h, share))

    def counters_reset(self):
        self.check(lib().rdh_counters_reset(self.h))

    def counters(self):
        c = CountersC()
        self.check(lib().rdh_counters_read(self.h, C.byref(c)))
        return c.as_dict()

    def synchronize(self):
        self.check(lib().rdh_synchronize(self.h))

    def profile_reset(self):
        self.check(lib().rdh_profile_reset(self.h))

    def profile_read(self):
        """(total ms, launches) of the traversal-kernel launches made with RDH_PT_PROFILE since the last reset."""
        ms, n = C.c_double(0), C.c_int64(0)
        self.check(lib().rdh_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_persist_stamps(self):
        out = np.zeros((5, 4096), dtype=np.uint64)
        self.check(lib().rdh_debug_persist_stamps(self.h, out.ctypes.data))
        return out

    def debug_persist_phases(self):
        out = np.zeros(16, dtype=np.uint64)
        self.check(lib().rdh_debug_persist_phases(self.h, out.ctypes.data))
        return out

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self.check(lib().rdh_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


# ======================================================================================================================
# The reference's globals and free functions
# ======================================================================================================================
class ReservoirReuse:  # src/common.h:41-48
    NONE, Temporal, Spatial, TemporalSpatial = 0, 1, 2, 3


class Settings:  # src/common.h:50-66, defaults src/common.cpp:3-15
    traceDepth = 0
    useReservoir = True
    reservoirReuse = ReservoirReuse.Temporal
    accumulate = False
    # knobs that exist only here (not in the reference)
    ptFlags = RDH_PT_PERSISTENT  # the fastest pathTrace kernel (RDH_PT_MEGAKERNEL = the reference's one-lane-per-pixel structure)
    restirNumSpatial = 5      # src/restir.cu:87
    restirRISCount = 32       # RESERVOIR_SIZE, src/restir.h:9
    restirTemporalClamp = 20  # src/restir.cu:168
    restirFaithfulRIS = 1     # src/restir.h:21


class State:  # src/common.h:68-72
    camChanged = True
    looper = 0
    scene = None  # a `Scene` below


class DevScene:
    """`DevScene::create/destroy` (src/scene.h:74-75): owns the device copy of a scene (inside an rdh_ctx)."""

    def __init__(self):
        self.ctx = None

    def create(self, scene_data, device=0):
        self.ctx = Context(device)
        self.ctx.upload_scene(scene_data)

    def destroy(self):
        if self.ctx:
            self.ctx.close()
            self.ctx = None


class Scene:
    """The slice of the reference's `Scene` the render API reads: `camera` and `devScene` (src/scene.h:520-577)."""

    def __init__(self, scene_data, camera, device=0):
        self.data = scene_data
        self.camera = camera
        self.devScene = DevScene()
        self.devScene.create(scene_data, device)

    def clear(self):  # Scene::clear (src/scene.cpp:251-254)
        self.devScene.destroy()


def _ctx():
    if State.scene is None or State.scene.devScene.ctx is None:
        raise RadishError("State.scene is not set")
    ctx = State.scene.devScene.ctx
    ctx.set_camera(State.scene.camera)
    return ctx


def _advance_looper():
    State.looper = (State.looper + 1) % SOBOL_SAMPLE_NUM  # src/pathtrace.cu:380-381


def pathTraceInit():  # src/pathtrace.cu:28
    _ctx().synchronize()


def pathTraceFree():  # src/pathtrace.cu:30
    pass


def pathTrace(directIllum, indirectIllum, iter):
    """src/pathtrace.cu:351-385.  Blocking, like the reference (checkCUDAError = device sync)."""
    ctx = _ctx()
    ctx.path_trace(directIllum, indirectIllum, iter, State.looper, Settings.traceDepth, Settings.ptFlags)
    ctx.synchronize()
    _advance_looper()


def pathTraceDirect(directIllum, iter):
    """src/pathtrace.cu:387-407."""
    ctx = _ctx()
    ctx.path_trace_direct(directIllum, iter, State.looper, Settings.ptFlags & RDH_PT_COUNT)
    ctx.synchronize()
    _advance_looper()


class ToneMapping:  # src/common.h:23-25
    NONE, Filmic, ACES = 0, 1, 2


def copyImageToPBO(devPBO, devImage, width, height, toneMapping=ToneMapping.NONE, scale=1.0):
    """copyImageToPBO's four overloads (src/pathtrace.h:25-29, src/pathtrace.cu:120-147), selected the way C++ overload
    resolution does — by the image's element type: float32 [n,3] → vec3 (tone-mapped), float32 [n,2] → vec2, float32 [n] →
    float, int32 [n] → pixel indices.  devPBO: uint8 [n,4]."""
    torch = _torch()
    n = width * height
    if devImage.dtype == torch.int32:
        kind = 3
    elif devImage.dtype == torch.float32:
        kind = {3 * n: 0, 2 * n: 1, n: 2}.get(devImage.numel())
    else:
        kind = None
    if kind is None or devPBO.numel() * devPBO.element_size() != 4 * n:
        raise RadishError("copyImageToPBO: image/PBO shape does not match width*height")
    _ctx().copy_image_to_pbo(devPBO, devImage, width, height, kind, toneMapping, scale)


def ReSTIRInit():  # src/restir.cu:235-245
    _ctx().restir_init()


def ReSTIRFree():  # src/restir.cu:247-251
    if State.scene is not None and State.scene.devScene.ctx is not None:
        State.scene.devScene.ctx.restir_free()


def ReSTIRDirect(directIllum, iter, gBuffer):
    """src/restir.cu:205-233 (swap of the reservoir buffers and the first-frame flag happen inside the library)."""
    ctx = _ctx()
    ctx.restir_direct(directIllum, iter, State.looper, gBuffer.c_struct(), Settings.reservoirReuse,
                      Settings.restirRISCount, Settings.restirNumSpatial, Settings.restirTemporalClamp,
                      Settings.restirFaithfulRIS, Settings.ptFlags & RDH_PT_COUNT)
    ctx.synchronize()
    _advance_looper()


class GBuffer:
    """src/gBuffer.h:15-58.  Planes are torch CUDA tensors; `c_struct()` is the 272-byte reference layout."""

    def __init__(self):
        self.width = self.height = 0
        self.frameIdx = 0
        self.lastCam = None

    def create(self, width, height, device=0):  # src/denoiser.cu:329-344
        torch = _torch()
        n = width * height
        dev = torch.device("cuda", device)
        self.width, self.height = width, height
        self.albedo = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        self.normal = [torch.zeros(n, 3, dtype=torch.float32, device=dev) for _ in range(2)]
        self.motion = torch.zeros(n, dtype=torch.int32, device=dev)
        self.depth = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(2)]
        self.primId = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
        self.frameIdx = 0
        self.lastCam = None

    def destroy(self):  # src/denoiser.cu:346-359
        for name in ("albedo", "normal", "motion", "depth", "primId"):
            if hasattr(self, name):
                delattr(self, name)

    def c_struct(self, cam_fallback=None):
        g = GBufferC()
        g.albedo = self.albedo.data_ptr()
        g.normal = (C.c_void_p * 2)(*[t.data_ptr() for t in self.normal])
        g.motion = self.motion.data_ptr()
        g.depth = (C.c_void_p * 2)(*[t.data_ptr() for t in self.depth])
        g.primId = (C.c_void_p * 2)(*[t.data_ptr() for t in self.primId])
        g.frameIdx = self.frameIdx
        last = self.lastCam if self.lastCam is not None else cam_fallback
        if last is not None:
            raw = np.asarray(last, dtype=L.CAMERA_DTYPE).tobytes()
            g.lastCam = (C.c_uint8 * 196)(*raw)
        g.width, g.height = self.width, self.height
        return g

    def render(self, devScene, cam):
        """GBuffer::render (src/gBuffer.cu:83-103).  On the very first frame the reference's lastCam is
        uninitialised memory; here it is defined to be `cam`."""
        ctx = devScene.ctx
        ctx.set_camera(cam)
        ctx.gbuffer_render(self.c_struct(cam_fallback=cam), Settings.ptFlags & RDH_PT_COUNT)
        ctx.synchronize()

    def update(self, cam):  # src/gBuffer.cu:78-81
        self.lastCam = np.asarray(cam, dtype=L.CAMERA_DTYPE).copy()
        self.frameIdx ^= 1


# ======================================================================================================================
# Denoisers: the reference's filter classes (src/denoiser.h:16-77, src/denoiser.cu:388-558) over the rdh_denoise_* entries
# ======================================================================================================================
def modulateAlbedo(devImage, gBuffer):  # src/denoiser.cu:363-371
    _ctx().denoise_modulate(devImage, gBuffer.c_struct())


def addImage(out, in1, in2=None, width=None, height=None):
    """addImage(devImage, in, w, h) and addImage(out, in1, in2, w, h) (src/denoiser.cu:373-386)."""
    if in2 is None or isinstance(in2, int):  # two-image overload: addImage(devImage, in, width, height)
        width, height = (in2, width) if isinstance(in2, int) else (width, height)
        in1, in2 = out, in1
    _ctx().denoise_add(out, in1, in2, width, height)


class EAWaveletFilter:  # src/denoiser.h:16-36
    def __init__(self, width=0, height=0, sigLumin=0.0, sigNormal=0.0, sigDepth=0.0):
        self.width, self.height = width, height
        self.sigLumin, self.sigNormal, self.sigDepth = sigLumin, sigNormal, sigDepth

    def filter(self, colorOut, colorIn, gBuffer, cam, level, varianceOut=None, varianceIn=None, filteredVar=None):
        ctx = _ctx()
        if varianceOut is None:  # src/denoiser.cu:388-397
            ctx.denoise_eaw(colorOut, colorIn, gBuffer.c_struct(), cam, self.sigLumin, self.sigNormal, self.sigDepth, level)
        else:  # src/denoiser.cu:399-409
            ctx.denoise_svgf(colorOut, colorIn, varianceOut, varianceIn, filteredVar, gBuffer.c_struct(), cam, self.sigLumin,
                             self.sigNormal, self.sigDepth, level)


class LeveledEAWFilter:  # src/denoiser.h:38-49, src/denoiser.cu:411-434
    def create(self, width, height, level, device=0):
        torch = _torch()
        self.level = level
        self.waveletFilter = EAWaveletFilter(width, height, 64.0, 0.2, 1.0)
        self.tmpImg = torch.zeros(width * height, 3, device=torch.device("cuda", device))

    def destroy(self):
        self.tmpImg = None

    def filter(self, colorOut, colorIn, gBuffer, cam):
        """Five à-trous passes (levels 0..4, as in the reference, whatever `level` says).  `colorOut` is `glm::vec3 *&` in the
        reference — the result ends up in what was tmpImg; returns the tensor that holds it."""
        f = self.waveletFilter
        f.filter(colorOut, colorIn, gBuffer, cam, 0)
        for lv in (1, 2, 3, 4):
            f.filter(self.tmpImg, colorOut, gBuffer, cam, lv)
            colorOut, self.tmpImg = self.tmpImg, colorOut
        return colorOut


class SpatioTemporalFilter:  # src/denoiser.h:51-77, src/denoiser.cu:436-560
    def create(self, width, height, level, device=0):
        torch = _torch()
        dev = torch.device("cuda", device)
        self.level = level
        self.accumColor = [torch.zeros(width * height, 3, device=dev) for _ in range(2)]
        self.accumMoment = [torch.zeros(width * height, 3, device=dev) for _ in range(2)]
        self.variance = torch.zeros(width * height, device=dev)
        self.waveletFilter = EAWaveletFilter(width, height, 4.0, 128.0, 1.0)
        self.tmpColor = torch.zeros(width * height, 3, device=dev)
        self.tmpVar = torch.zeros(width * height, device=dev)
        self.filteredVar = torch.zeros(width * height, device=dev)
        self.firstTime = True
        self.frameIdx = 0

    def destroy(self):
        self.accumColor = self.accumMoment = self.variance = self.tmpColor = self.tmpVar = self.filteredVar = None

    def temporalAccumulate(self, colorIn, gBuffer):  # :461-485
        f = self.frameIdx
        _ctx().denoise_temporal_accumulate(self.accumColor[f], self.accumColor[f ^ 1], self.accumMoment[f], self.accumMoment[f ^ 1],
                                           colorIn, gBuffer.c_struct(), self.firstTime)
        self.firstTime = False

    def estimateVariance(self):  # :487-509
        w = self.waveletFilter
        _ctx().denoise_estimate_variance(self.variance, self.accumMoment[self.frameIdx], w.width, w.height)

    def filterVariance(self):  # :511-523
        w = self.waveletFilter
        _ctx().denoise_filter_variance(self.filteredVar, self.variance, w.width, w.height)

    def filter(self, colorOut, colorIn, gBuffer, cam):
        """src/denoiser.cu:525-558, swap for swap.  Returns the tensor that holds the result (`colorOut` is `glm::vec3 *&`)."""
        f, w = self.frameIdx, self.waveletFilter
        self.temporalAccumulate(colorIn, gBuffer)
        self.estimateVariance()
        self.filterVariance()
        w.filter(colorOut, self.accumColor[f], gBuffer, cam, 0, self.tmpVar, self.variance, self.filteredVar)
        colorOut, self.accumColor[f] = self.accumColor[f], colorOut
        self.tmpVar, self.variance = self.variance, self.tmpVar
        self.filterVariance()
        w.filter(colorOut, self.accumColor[f], gBuffer, cam, 1, self.tmpVar, self.variance, self.filteredVar)
        self.tmpVar, self.variance = self.variance, self.tmpVar
        for lv in (2, 3, 4):
            self.filterVariance()
            w.filter(self.tmpColor, colorOut, gBuffer, cam, lv, self.tmpVar, self.variance, self.filteredVar)
            self.tmpColor, colorOut = colorOut, self.tmpColor
            self.tmpVar, self.variance = self.variance, self.tmpVar
        return colorOut

    def nextFrame(self):  # :560
        self.frameIdx ^= 1
