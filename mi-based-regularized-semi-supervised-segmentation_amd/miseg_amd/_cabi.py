"""ctypes binding of the gfx950 C-ABI library (include/miseg_hip.h).

The prototypes are read from the header itself, so the Python side can never drift from the
declared ABI, and ``declared_symbols()`` lets the CPU test-suite check that the built library
exports every entry point.  There is NO fallback: if the library is missing or a kernel call
fails, the product raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
HEADER = os.path.join(REPO_ROOT, "include", "miseg_hip.h")
LIB_PATH = os.environ.get("MISEG_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libmiseg_hip.so")   # override: A/B runs of two builds

F32, BF16, F16 = 0, 1, 2

_CTYPES = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "void": None}


def _parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\s*\*)\s+(miseg_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _CTYPES[ret]
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[base])
        protos[name] = (restype, argtypes)
    return protos


PROTOTYPES = _parse_header()


def declared_symbols() -> List[str]:
    return sorted(PROTOTYPES)


class MisegError(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) the in-tree library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MisegError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                             f"`make -C {os.path.join(PKG_ROOT, 'csrc')}` -- there is no CPU fallback")
        # torch first: it bundles its own libamdhip64.so.7; loading ours first would bind /opt/rocm's copy of the
        # same SONAME into the process and the two runtimes then disagree about the device.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class KernelTimer:
    """Optional per-entry-point device timing with HIP events on the launching stream (bench.py roofline).
    ``work`` = algorithmic (flops, bytes) of the call, supplied by the caller of ``call``."""

    def __init__(self, only=None):
        self.records = {}  # name -> [ [start_event, end_event], flops, bytes ]
        self.only = only   # None = every tagged call; else the set of tags to time (keeps host overhead off the rest)

    def wrap(self, name, work, fn):
        if self.only is not None and name not in self.only:
            fn()
            return
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.setdefault(name, []).append((s, e, work[0], work[1]))

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = {"calls": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms), "min_ms": min(ms),
                         "flops_per_call": sum(r[2] for r in recs) / len(recs), "bytes_per_call": sum(r[3] for r in recs) / len(recs)}
        return out


TIMER = None  # set to a KernelTimer by bench.py
RECORDING = 0         # handle of the launch tape being recorded (miseg_amd.tape.StepTape), else 0
TAPE_TAGS: list = []  # while recording: (op index, tag, (flops, bytes)) of every tagged call -- bench.py times tagged ops of a replayed tape


def call(name: str, *args, work=None, tag=None) -> None:
    """Invoke an int-returning entry point and turn a negative status into an exception."""
    fn = getattr(lib(), name)
    if RECORDING and work is not None:
        TAPE_TAGS.append((int(lib().miseg_tape_len(RECORDING)), tag or name, work))
    if TIMER is not None and work is not None:
        box = []
        TIMER.wrap(tag or name, work, lambda: box.append(fn(*args)))
        rc = box[0]
    else:
        rc = fn(*args)
    if rc != 0:
        msg = lib().miseg_last_error()
        raise MisegError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")


_QUERY_CACHE: dict = {}


def query(name: str, *args) -> int:
    """A size / capability query of the library (pure functions of their arguments: cached -- a backward pass asks ~150 of them)."""
    key = (name,) + args
    v = _QUERY_CACHE.get(key)
    if v is None:
        v = getattr(lib(), name)(*args)
        if v < 0:
            raise MisegError(f"{name}{args} -> {v}: unsupported configuration")
        v = _QUERY_CACHE[key] = int(v)
    return v
