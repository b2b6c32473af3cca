"""Platform-independent deterministic pseudo-random tensors (oracle / test infrastructure).

Values come from splitmix64 applied to (crc32(name), element index), so the
same (name, shape) gives bit-identical numbers on every machine and library
version.  That lets golden fixtures store only *outputs*; inputs and parameters
are regenerated here.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def bits(name, n):
    seed = np.uint64(zlib.crc32(name.encode()))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(_splitmix64(idx + (seed << np.uint64(32))))


def uniform(name, shape, lo=0.0, hi=1.0, dtype=np.float32):
    """U[lo, hi) with 24 random bits (exactly representable in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(name, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def randint(name, shape, lo, hi):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (bits(name, n) >> np.uint64(33)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape)
