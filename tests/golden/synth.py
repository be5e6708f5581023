"""Deterministic synthetic inputs shared by make_golden.py (which feeds them to the reference)
and by the tests (which feed the same arrays to the oracle and to the HIP path).

Everything comes from numpy's legacy ``RandomState`` (MT19937 + polar normal), whose stream
is frozen across numpy versions, computed in float64 and rounded once -- so the build
container and the GPU box regenerate bit-identical inputs and the big tensors need not
be stored in the fixtures.  Large outputs are stored as *fingerprints* (a seeded sample of
entries plus sums) to keep fixtures small.
"""
from __future__ import annotations

import zlib

import numpy as np


def _seed(tag: str) -> int:
    return zlib.crc32(tag.encode()) & 0x7FFFFFFF


def normal(tag: str, shape, scale: float = 1.0, dtype=np.float32) -> np.ndarray:
    return (np.random.RandomState(_seed(tag)).standard_normal(tuple(shape)) * scale).astype(dtype)


def uniform(tag: str, shape, dtype=np.float32) -> np.ndarray:
    return np.random.RandomState(_seed(tag)).random_sample(tuple(shape)).astype(dtype)


def integers(tag: str, shape, high: int) -> np.ndarray:
    return np.random.RandomState(_seed(tag)).randint(0, high, size=tuple(shape)).astype(np.int64)


def probs(tag: str, shape, dtype=np.float32, temperature: float = 1.0) -> np.ndarray:
    """Per-pixel simplex along axis 1 (softmax of a normal field), float64 math, one rounding."""
    z = np.random.RandomState(_seed(tag)).standard_normal(tuple(shape)) / temperature
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(dtype)


def _softmax1(z: np.ndarray) -> np.ndarray:
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def peaked_pair(tag: str, shape, dtype=np.float32, sharpness: float = 6.0, agree: float = 0.9, block: int = 4):
    """Two CORRELATED, PEAKED per-pixel simplexes (x, y) along axis 1: a trained head's regime, where the mutual information is
    O(0.1 .. 1) instead of the O(1e-3 .. 1e-6) of ``probs`` (independent near-uniform fields).  A shared class field -- piecewise
    constant over ``block`` x ``block`` pixels for 4-D shapes, so neighbouring displacements carry information too -- drives both;
    ``agree`` mixes it with independent noise.  float64 math, one rounding."""
    rs = np.random.RandomState(_seed(tag))
    shape = tuple(shape)
    if len(shape) == 4:
        n, k, h, w = shape
        low = rs.standard_normal((n, k, -(-h // block), -(-w // block)))
        shared = np.repeat(np.repeat(low, block, axis=2), block, axis=3)[:, :, :h, :w]
    else:
        shared = rs.standard_normal(shape)
    nx, ny = rs.standard_normal(shape), rs.standard_normal(shape)
    mix = np.sqrt(max(0.0, 1.0 - agree * agree))
    x = _softmax1(sharpness * (agree * shared + mix * nx))
    y = _softmax1(sharpness * (agree * shared + mix * ny))
    return x.astype(dtype), y.astype(dtype)


def mask(tag: str, shape, keep: float = 0.7) -> np.ndarray:
    return (np.random.RandomState(_seed(tag)).random_sample(tuple(shape)) < keep).astype(np.float32)


# ----------------------------------------------------------------------------- fingerprints
MAX_SAMPLE = 2048


def sample_index(numel: int, tag: str) -> np.ndarray:
    if numel <= MAX_SAMPLE:
        return np.arange(numel)
    return np.sort(np.random.RandomState(_seed("idx/" + tag)).choice(numel, MAX_SAMPLE, replace=False))


def fingerprint(arr, tag: str) -> dict:
    """{'shape', 'sample', 'sum', 'abssum'} of an array (sample = entries at sample_index)."""
    a = np.asarray(arr)
    flat = a.reshape(-1).astype(np.float64)
    return {"shape": np.asarray(a.shape, dtype=np.int64), "sample": flat[sample_index(flat.size, tag)],
            "sum": np.float64(flat.sum()), "abssum": np.float64(np.abs(flat).sum())}


def check_fingerprint(arr, fp: dict, tag: str, rtol: float, atol: float) -> None:
    """Assert ``arr`` matches a stored fingerprint; tolerances are relative to the entry scale."""
    a = np.asarray(arr)
    assert tuple(a.shape) == tuple(int(v) for v in fp["shape"]), (tag, a.shape, fp["shape"])
    flat = a.reshape(-1).astype(np.float64)
    got = flat[sample_index(flat.size, tag)]
    np.testing.assert_allclose(got, fp["sample"], rtol=rtol, atol=atol, err_msg=f"sample mismatch: {tag}")
    scale = float(fp["abssum"]) + 1e-30
    assert abs(flat.sum() - float(fp["sum"])) <= rtol * scale * 4 + atol * flat.size, \
        (tag, flat.sum(), float(fp["sum"]), scale)


def fp_pack(prefix: str, fp: dict) -> dict:
    return {f"{prefix}#{k}": v for k, v in fp.items()}


def fp_unpack(npz, prefix: str) -> dict:
    return {k: npz[f"{prefix}#{k}"] for k in ("shape", "sample", "sum", "abssum")}


# ----------------------------------------------------------------------------- checkpoint key trees
def tree_lines(obj, prefix: str = "") -> list:
    """Flatten a nested checkpoint into 'path :: kind' lines: tensors as dtype + shape, other leaves as their Python type name.
    Shared by make_golden.py (run on the reference's trainers) and the tests (run on this repo's)."""
    import torch
    lines = []
    if isinstance(obj, dict):
        if not obj:
            lines.append(f"{prefix} :: empty-dict")
        for k in obj:
            lines += tree_lines(obj[k], f"{prefix}/{k}" if prefix else str(k))
    elif isinstance(obj, (list, tuple)):
        if not obj:
            lines.append(f"{prefix} :: empty-{type(obj).__name__}")
        for i, v in enumerate(obj):
            lines += tree_lines(v, f"{prefix}[{i}]")
    elif torch.is_tensor(obj):
        lines.append(f"{prefix} :: tensor {str(obj.dtype).replace('torch.', '')} {list(obj.shape)}")
    else:
        lines.append(f"{prefix} :: {type(obj).__name__}")
    return lines
