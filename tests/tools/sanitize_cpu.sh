#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code (the scene loader parses files it did not
# write): builds libradish_host.so and liboracle.so with -fsanitize=address,undefined into /tmp and runs (1) the CPU test
# files that exercise them under LD_PRELOAD=libasan, (2) a native driver over valid, malformed and truncated scene files
# (the loader's error paths throw C++ exceptions inside the library, which the preloaded-ASan Python process cannot host).
# GPU sanitizers are not available on the pool; the HIP library is not part of this run.
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/radish_san; rm -rf $OUT; mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ $FLAGS -shared -o $OUT/libradish_host.so radish_pt_amd/csrc/host/scene_build.cpp radish_pt_amd/csrc/host/scene_load.cpp -lz
g++ $FLAGS -shared -o $OUT/liboracle.so oracle/oracle.cpp
g++ $FLAGS -o $OUT/driver tests/tools/sanitize_driver.cpp radish_pt_amd/csrc/host/scene_build.cpp radish_pt_amd/csrc/host/scene_load.cpp -lz
ASAN=$(g++ -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 RADISH_HOST_LIB=$OUT/libradish_host.so RADISH_ORACLE_LIB=$OUT/liboracle.so \
  python -m pytest tests/test_scene_loader.py tests/test_host_scene.py tests/test_oracle_kat.py tests/test_golden.py -x -q -m "not gpu" \
  -k "not parse_errors" -p no:cacheprovider
# native driver: a scene written by the test suite's generator, then damaged copies of it
python - "$OUT" <<'PY'
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from PIL import Image
from test_scene_loader import CUBE_QUADS, PLANE, write_hdr
d = sys.argv[1] + "/scene"; os.makedirs(d, exist_ok=True)
open(d + "/cube.obj", "w").write(CUBE_QUADS); open(d + "/plane.obj", "w").write(PLANE)
rng = np.random.default_rng(1)
Image.fromarray(rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)).save(d + "/a.png")
write_hdr(d + "/sky.hdr", rng.random((4, 8, 3)).astype(np.float32))
good = ("Material m\nType MetallicWorkflow\nBaseColor a.png\nMetallic 0.5\nRoughness 0.5\nIor 1.5\nNormalMap Null\n\n"
        "Object 0\ncube.obj\nMaterial m\nRotate 10 20 30\n\nObject 1\nplane.obj\nMaterial Null\n\n"
        "Camera\nResolution 8 6\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 2\nFile f\nEye 0 1 4\n\nEnvMap sky.hdr\n")
open(d + "/good.txt", "w").write(good)
for k, cut in enumerate((10, 60, 130, 200, len(good) - 5)):
    open(d + f"/cut{k}.txt", "w").write(good[:cut])
open(d + "/badmat.txt", "w").write("Object 0\ncube.obj\nMaterial nope\n\n")
open(d + "/badobj.txt", "w").write("Object 0\nbad.obj\nMaterial Null\n\n"); open(d + "/bad.obj", "w").write("v 0 0 0\nf 1 2 9\nf -7 1 1\n")
png = open(d + "/a.png", "rb").read()
open(d + "/trunc.png", "wb").write(png[: len(png) // 2])
open(d + "/badpng.txt", "w").write(good.replace("a.png", "trunc.png"))
hdr = open(d + "/sky.hdr", "rb").read()
open(d + "/trunc.hdr", "wb").write(hdr[: len(hdr) - 40])
open(d + "/badhdr.txt", "w").write(good.replace("sky.hdr", "trunc.hdr"))
PY
(cd $OUT/scene && ../driver good.txt cut0.txt cut1.txt cut2.txt cut3.txt cut4.txt badmat.txt badobj.txt badpng.txt badhdr.txt missing.txt)
echo "sanitizers: clean"
