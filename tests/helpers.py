import numpy as np


def random_rays(n, seed, lo=(-1.2, -0.2, -1.2), hi=(1.2, 2.2, 4.5)):
    """Origins in a box around the Cornell scene, unit directions; a few axis-parallel and tiny-component
    directions so every branch of AABB::intersect is exercised."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    k = max(n // 16, 1)
    for axis in range(3):  # exactly axis-parallel
        idx = rng.choice(n, k, replace=False)
        d[idx] = 0
        d[idx, axis] = rng.choice([-1.0, 1.0], k)
    idx = rng.choice(n, k, replace=False)  # one tiny component
    d[idx, rng.integers(0, 3, k)] = 1e-7
    idx = rng.choice(n, k, replace=False)  # one exactly-zero component
    d[idx, rng.integers(0, 3, k)] = 0.0
    return np.concatenate([o, d], axis=1).astype(np.float32)


def random_segments(n, seed, lo=(-1.0, 0.0, -1.0), hi=(1.0, 2.0, 1.0)):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    y = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    return np.concatenate([x, y], axis=1).astype(np.float32)


def bits(a):
    """View float arrays as uint32 so comparisons are bit-exact (NaN == NaN, +0 != -0)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba != bb)
        first = tuple(bad[0])
        raise AssertionError(
            f"{what}: {len(bad)} of {ba.size} floats differ bitwise; first at {first}: "
            f"{np.asarray(a, np.float32)[first]!r} vs {np.asarray(b, np.float32)[first]!r}"
        )
