"""Shared fixtures.  `-m "not gpu"` runs the oracle / host-logic / ABI-export tests on CPU; `-m gpu` runs the parity
tests proper, which call the HIP path through the C ABI and check it against the oracle."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build what is missing (seconds); the HIP library is only (re)built when hipcc is present.
    from radish_pt_amd import _build

    import __graft_entry__

    _build.build_host()
    __graft_entry__.build_oracle()
    if not os.path.exists(os.path.join(ROOT, "radish_pt_amd", "csrc", "libradish_hip.so")):
        _build.build_hip()


@pytest.fixture(scope="session")
def cornell_small():
    from radish_pt_amd import scenes

    return scenes.cornell(segments=16, bands=12)


@pytest.fixture(scope="session")
def cornell_full():
    from radish_pt_amd import scenes

    return scenes.cornell()


@pytest.fixture(scope="session")
def tiny_scene():
    from radish_pt_amd import scenes

    return scenes.tiny()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible — there is no CPU fallback to run instead")
    from radish_pt_amd import api

    ctx = api.Context(0)
    yield ctx
    ctx.close()
