"""
numpy restatement of the perf-mode noise stream (TEST INFRASTRUCTURE ONLY):
Philox4x32-10 (Salmon et al., SC'11; constants as in Random123) keyed by the 64-bit
seed, counter = (sequence lo, sequence hi, position | (feature // 4) << 24, t),
then Box-Muller on word pairs.  Mirrors csrc/rowwise.hip: philox_normal().
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _M, p1 & _M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _M, p0 & _M
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M
    return c0, c1, c2, c3


def philox_normal(seed: int, t: int, seq_offset: int, B: int, L: int, F: int) -> np.ndarray:
    """[B, L, F] float32 N(0,1) draws of step t for global sequences seq_offset..seq_offset+B-1."""
    seq, pos = np.meshgrid(np.arange(B, dtype=np.uint64) + np.uint64(seq_offset), np.arange(L, dtype=np.uint64), indexing="ij")
    seq, pos = seq.ravel(), pos.ravel()
    out = np.zeros((B * L, F), dtype=np.float32)
    scale = np.float32(2.3283064365386963e-10)
    for f in range(F):
        o = philox4x32_10(seq & _M, seq >> np.uint64(32), pos | np.uint64((f >> 2) << 24), np.full(B * L, t, dtype=np.uint64),
                          seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        pair = (f >> 1) & 1
        u1 = (o[2 * pair].astype(np.float32) + np.float32(0.5)) * scale
        u2 = (o[2 * pair + 1].astype(np.float32) + np.float32(0.5)) * scale
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        th = np.float32(6.283185307179586) * u2
        out[:, f] = rad * np.sin(th) if f & 1 else rad * np.cos(th)
    return out.reshape(B, L, F)
