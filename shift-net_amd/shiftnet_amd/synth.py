"""Synthetic GoPro/DAVIS-shaped clips (no datasets ship with the reference, no network here).

Integer-only generators, so the bytes are identical on every machine (checked
by crc32 in the golden fixtures): a textured "sharp" clip that translates one
pixel per frame, a temporally box-blurred copy of it as the deblur input, and
an additive-noise copy as the denoise input (SURVEY.md §8d "Synthetic inputs").
"""
from __future__ import annotations

import zlib
from typing import Tuple

import numpy as np


def _texture(yy: np.ndarray, xx: np.ndarray, c: int, seed: int) -> np.ndarray:
    """uint8 texture on integer coordinate grids (wraps smoothly, has edges and gradients)."""
    a = (xx * 5 + yy * 3 + c * 37 + seed * 11) & 255
    b = ((xx >> 2) * (yy >> 3) + c * 19) & 127
    e = (((xx >> 4) ^ (yy >> 4)) & 1) * 64              # checker edges
    tri = np.abs(((xx + 2 * yy + seed) & 127) - 64)       # triangle wave 0..64
    v = (a >> 2) + (b >> 1) + e + tri
    return np.clip(v, 0, 255).astype(np.uint8)


def sharp_clip(t: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """[t, h, w, 3] uint8; frame k is the base texture translated by k pixels in x and k//2 in y."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    out = np.empty((t, h, w, 3), np.uint8)
    for k in range(t):
        for c in range(3):
            out[k, :, :, c] = _texture(yy + k // 2, xx + k, c, seed)
    return out


def blurred_clip(t: int, h: int, w: int, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(blur, sharp): blur[k] = round(mean(sharp[k-2..k+2])) with the sharp clip generated 2 frames wider."""
    s = sharp_clip(t + 4, h, w, seed).astype(np.int32)
    acc = s[0:t] + s[1:t + 1] + s[2:t + 2] + s[3:t + 3] + s[4:t + 4]
    blur = ((acc * 2 + 5) // 10).astype(np.uint8)
    return blur, s[2:t + 2].astype(np.uint8)


def noise_i16(t: int, h: int, w: int, sigma: int, seed: int = 0) -> np.ndarray:
    """Integer approx-Gaussian noise with std ~= sigma (sum of 4 LCG uniforms), [t,h,w,3] int16."""
    n = t * h * w * 3
    idx = np.arange(n, dtype=np.uint64)
    acc = np.zeros(n, np.int64)
    for j in range(4):
        z = (idx + np.uint64(seed * 7919 + j * 104729 + 1)) * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)
        z ^= z >> np.uint64(29)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(32)
        acc += (z & np.uint64(0xFFFF)).astype(np.int64) - 32768
    # sum of 4 U(-32768,32767): std = 65536*sqrt(4/12) = 37837.2 ; scale to sigma
    return ((acc * sigma) // 37837).astype(np.int16).reshape(t, h, w, 3)


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def unit_noise(shape, seed: int = 0) -> np.ndarray:
    """float32 ~N(0,1) of the given shape, bit-identical everywhere (integer noise / 256)."""
    n = int(np.prod(shape))
    t = (n + 2) // 3
    v = noise_i16(t, 1, 1, 256, seed).reshape(-1)[:n].astype(np.float32) / 256.0
    return v.reshape(shape)
