"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

Deterministic stand-in for the network: a hash of the int8 feature planes
drives float32 priors and a float32 value.  The same arithmetic exists in C
(`oz_fake_eval` in az_oracle.c); both are exact (integer hash, then IEEE
float32 operations in a fixed order), so the reference's Python search, the C
oracle and the HIP engine can all be fed identical "network outputs".
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


_coef_cache = {}


def _coefs(n):
    if n not in _coef_cache:
        with np.errstate(over="ignore"):
            j = np.arange(1, n + 1, dtype=np.uint64) * _G
        _coef_cache[n] = _mix64(j)
    return _coef_cache[n]


def position_hash(planes):
    """planes: int8 array (..., F, H, W) -> uint64 hash per leading index."""
    p = np.ascontiguousarray(planes).astype(np.int8)
    lead = p.shape[:-3]
    flat = p.reshape(lead + (-1,)).view(np.uint8).astype(np.uint64)
    with np.errstate(over="ignore"):
        return (flat * _coefs(flat.shape[-1])).sum(axis=-1, dtype=np.uint64)


def fake_eval_planes(planes, O, salt=0):
    """planes int8 (B, F, H, W) or (F, H, W) -> (priors float32 (B,O), values float32 (B,)).
    salt != 0 gives a second, different "network" (two-evaluator games)."""
    single = planes.ndim == 3
    if single:
        planes = planes[None]
    h = position_hash(planes)  # (B,)
    if salt:
        h = h ^ _mix64(np.uint64(salt))
    with np.errstate(over="ignore"):
        i = np.arange(1, O + 1, dtype=np.uint64) * _G
        z = _mix64(h[:, None] + i[None, :])
    k = (z >> np.uint64(40)).astype(np.int64) + 1
    a = k.astype(np.float32) * np.float32(1.0 / 16777216.0)
    p = a * a
    p = p * p
    scale = np.float32(4.0 / float(O))
    pri = (p * scale).astype(np.float32)
    zv = _mix64(h ^ np.uint64(0xABCDEF0123456789))
    v = ((zv >> np.uint64(40)).astype(np.float64) - 8388608.0).astype(np.float32) * np.float32(1.0 / 16777216.0)
    if single:
        return pri[0], v[0]
    return pri, v
