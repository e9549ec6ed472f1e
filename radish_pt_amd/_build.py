"""In-tree builds of the product libraries: libradish_host.so (g++) and libradish_hip.so (hipcc, gfx950).

Everything lands next to its sources so the `.so` files travel with the gpurun snapshot; nothing is installed.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "radish_pt_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HOST_SOURCES = [os.path.join(CSRC, "host", "scene_build.cpp"), os.path.join(CSRC, "host", "scene_load.cpp")]
HIP_SOURCES = [os.path.join(CSRC, "radish_hip.hip")]
HIP_DEPS = [
    os.path.join(CSRC, "device", f)
    for f in ("rmath.h", "layouts.h", "traverse.h", "bsdf.h", "lights.h", "kernels_pt.h", "kernels_restir.h", "kernels_wave.h", "kernels_persist.h", "kernels_walk.h", "wg_trace.h", "kernels_display.h", "kernels_denoise.h")
] + [os.path.join(ROOT, "include", "radish_hip.h")]

# -ffp-contract=off: the HIP path must execute the same IEEE-754 operation sequence as the CPU checker
# (DESIGN.md "Numerics contract"); no fast-math anywhere.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_host(force=False):
    out = os.path.join(CSRC, "libradish_host.so")
    deps = HOST_SOURCES + [os.path.join(ROOT, "include", "radish_host.h")]
    if force or _newer(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-o", out]
             + HOST_SOURCES + ["-lz"])
    return out


def build_hip(force=False):
    out = os.path.join(CSRC, "libradish_hip.so")
    if force or _newer(out, HIP_SOURCES + HIP_DEPS):
        _run([HIPCC] + HIP_FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", out] + HIP_SOURCES
             + ["-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_variant(name, defines):
    """Tuning builds: csrc/variants/libradish_hip_<name>.so compiled with extra -D flags (select with RADISH_HIP_LIB)."""
    d = os.path.join(CSRC, "variants")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, f"libradish_hip_{name}.so")
    _run([HIPCC] + HIP_FLAGS + [f"-D{x}" for x in defines] + ["-I", os.path.join(ROOT, "include"), "-o", out]
         + HIP_SOURCES + ["-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_all(force=False):
    build_host(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
