"""Register budgets of the hot kernels (CPU test: hipcc cross-compiles gfx950 and reports per-kernel resources).

The persistent kernels are launched with exactly as many waves as stay resident, so their speed hangs on occupancy steps of the
512-entry VGPR file: 168 registers for three waves per SIMD (k_pt_persistent, k_wf_shade), 128 for four (RIS, G-buffer), 72 for
seven (the walkers).  Round 2 crossed these lines by accident several times (163 -> 169 for an unrelated refactoring, 72 -> 73 in
k_wf_trace: -14 % waves); this test says so at once instead of a benchmark saying so later."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _resources():
    from radish_pt_amd import _build

    flags = [f for f in _build.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC] + flags + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "radish_pt_amd", "csrc", "radish_hip.hip"),
                                 "-o", os.path.join(tmp, "x.o"), "-Rpass-analysis=kernel-resource-usage"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", out.stderr)[1:]:
        name = block.split("\n")[0].split(" ")[0].strip()
        get = lambda key: int(re.search(key + r": (\d+)", block).group(1))
        res[name] = {"vgpr": get("VGPRs"), "scratch": get(r"ScratchSize \[bytes/lane\]"), "occupancy": get(r"Occupancy \[waves/SIMD\]"),
                     "lds": get(r"LDS Size \[bytes/block\]")}
    return res


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_hot_kernels_stay_under_their_occupancy_steps():
    res = _resources()

    def one(prefix):
        hits = {k: v for k, v in res.items() if k.startswith(prefix)}
        assert hits, f"no kernel matches {prefix}"
        return hits

    # (mangled-name prefix, VGPR bound, waves per SIMD it must keep, bytes of scratch allowed); COUNT variants (tests only) are not
    # bound.  Template arguments: <COUNT, PAIRS> for k_pt_persistent / k_wf_trace, <COUNT, ANY, DEFER> for the walkers.  The
    # sibling-pair variants (round 3) hold a 64-byte record in registers: one occupancy step below the threaded ones in the
    # wavefront trace kernel and the closest-hit walker, and k_pt_persistent<.., true> is held to three waves by its launch bound
    # at the price of four spilled dwords.
    budgets = [
        ("_ZN2rd15k_pt_persistentILb0ELb0E", 168, 3, 0),
        ("_ZN2rd15k_pt_persistentILb0ELb1E", 168, 3, 16),
        ("_ZN2rd10k_wf_shadeE", 168, 3, 0),
        ("_ZN2rd10k_wf_traceILb0ELb0E", 72, 7, 0),
        ("_ZN2rd10k_wf_traceILb0ELb1E", 80, 6, 0),
        ("_ZN2rd17k_walk_persistentILb0E", 72, 7, 0),
        ("_ZN2rd11k_walk_pairILb0ELb1E", 72, 7, 0),
        ("_ZN2rd11k_walk_pairILb0ELb0E", 80, 6, 0),
        ("_ZN2rd12k_restir_risILb1E", 128, 4, 0),
        ("_ZN2rd20k_gbuffer_persistentILb0E", 128, 4, 0),
        ("_ZN2rd13k_walk_packetILb0E", 64, 8, 0),      # packet walks: the walk itself needs few registers; many waves hide the
        ("_ZN2rd16k_gbuffer_packetILb0E", 72, 7, 0),   # scalar-load latency of a wave's one-node-at-a-time chain
    ]
    for prefix, vgprs, waves, scratch in budgets:
        for name, r in one(prefix).items():
            assert r["vgpr"] <= vgprs and r["occupancy"] >= waves, (name, r, f"budget {vgprs} VGPRs / {waves} waves per SIMD")
            assert r["scratch"] <= scratch, (name, r, "spills to scratch")
