"""The reference-signature side of the boundary, radish_pt_amd/csrc/radish_shim.hpp, must at least compile: -fsyntax-only,
-Wall -Werror, against tests/shim/shim_syntax_check.cpp's minimal declarations of the reference types it touches (glm and the
reference's own headers are not in the build image), in its single-GPU, its RADISH_SHIM_MULTI_GPU (one process per GPU) and its
RADISH_SHIM_ONE_PROCESS_GPUS (one process, n GPUs) form."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.parametrize("multi_gpu", [False, True, "one_process"])
def test_shim_compiles(multi_gpu):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not found")
    cmd = [HIPCC, "-fsyntax-only", "-std=c++17", "-Wall", "-Werror", "-Wno-unused-function", "-x", "c++",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "radish_pt_amd", "csrc"), "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__=1"]
    if multi_gpu == "one_process":
        cmd.append("-DSHIM_CHECK_ONE_PROCESS")
    elif multi_gpu:
        cmd.append("-DSHIM_CHECK_MULTI_GPU")
    cmd.append(os.path.join(ROOT, "tests", "shim", "shim_syntax_check.cpp"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
