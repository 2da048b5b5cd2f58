"""Deterministic synthetic tensors shared by the golden generator and the tests.

Pure integer arithmetic (a 32-bit mix of the element index and a seed), so the same values come
out on any machine, any numpy/torch version, CPU or GPU box.  Values are uniform in [-amp, amp)
on a 2^-7 grid scaled by ``amp`` when amp is a power of two, hence exactly representable in bf16.
"""
from __future__ import annotations

import numpy as np
import torch


def mix32(idx: np.ndarray, seed: int) -> np.ndarray:
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed * 0x85EBCA6B + 0x1234567)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def synth_f32(shape, seed: int, amp: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    h = mix32(np.arange(n, dtype=np.uint64), seed)
    # 8 random bits -> k/128 - 1 in [-1, 1): exact in bf16 (8 significant bits)
    v = (h >> np.uint32(24)).astype(np.float32) / np.float32(128.0) - np.float32(1.0)
    return (v * np.float32(amp)).reshape(shape)


def synth(shape, seed: int, amp: float = 1.0, dtype=torch.bfloat16) -> torch.Tensor:
    return torch.from_numpy(synth_f32(shape, seed, amp)).to(dtype)


def synth_ints(n: int, seed: int, lo: int, hi: int) -> list:
    """n integers in [lo, hi]."""
    h = mix32(np.arange(n, dtype=np.uint64), seed ^ 0x5151)
    return [int(lo + int(x) % (hi - lo + 1)) for x in h]


def bits16(t: torch.Tensor) -> np.ndarray:
    """bf16/f16 tensor -> uint16 bit patterns (numpy has no bf16)."""
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def from_bits16(a: np.ndarray, dtype=torch.bfloat16) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(dtype)


def fnv64(a: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum of an integer array (for pool-content fixtures)."""
    x = np.ascontiguousarray(a).reshape(-1).view(np.uint8).astype(np.uint64)
    idx = np.arange(x.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        w = (idx * np.uint64(0x100000001B3) + np.uint64(0xCBF29CE484222325))
        return int(np.bitwise_xor.reduce((x + np.uint64(1)) * w) ^ np.uint64(x.size))
