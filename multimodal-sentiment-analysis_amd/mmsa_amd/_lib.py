"""ctypes binding of libmmsa_hip.so.

Signatures are read from ``include/mmsa.h`` (the C ABI's single source of truth), so the Python side can
never drift from the header.  There is no fallback: if the shared library is missing or an entry point
returns non-zero, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmmsa_hip.so")
HEADER = os.path.normpath(os.path.join(PKG_DIR, "..", "..", "include", "mmsa.h"))

_SCALARS = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "unsigned long long": ctypes.c_ulonglong,
    "long long": ctypes.c_longlong,
}


class MmsaError(RuntimeError):
    pass


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """{name: (return type, [(type, arg name), ...])} for every MMSA_API declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"MMSA_API\s+(int|long)\s+(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        parsed = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            mm = re.match(r"^(.*?)(\w+)$", a)
            parsed.append((mm.group(1).strip(), mm.group(2)))
        out[name] = (ret, parsed)
    return out


def _ctype(t: str):
    if "*" in t:
        return ctypes.c_void_p
    t = t.replace("const ", "").strip()
    return _SCALARS[t]


_lib = None
_protos = None


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmsaError(
            f"{LIB_PATH} is missing: build it with `python -m mmsa_amd.build` (needs hipcc). "
            "There is no CPU or PyTorch fallback for the HIP kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    for name, (ret, args) in protos.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = ctypes.c_long if ret == "long" else ctypes.c_int
        fn.argtypes = [_ctype(t) for t, _ in args]
    _lib, _protos = lib, protos
    return lib


def prototypes():
    load()
    return _protos


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise MmsaError(f"{name} failed with status {rc} "
                        f"({ {-1: 'bad argument', -2: 'HIP launch error', -3: 'limit exceeded'}.get(rc, 'unknown') })")
    return rc


def query(name: str, *args):
    """Invoke a size/geometry helper (non-negative result or a negative error code)."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc < 0:
        raise MmsaError(f"{name} rejected its arguments (status {rc})")
    return rc
