"""ctypes binding of libtatt_hip.so (the C-ABI declared in include/tatt_hip.h).

The signatures are parsed from the header, so the header is the single source of truth.  There is no
fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TATT_LIB") or os.path.join(_HERE, "libtatt_hip.so")     # TATT_LIB: another build of the same library (A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tatt_hip.h")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "uint32_t": ctypes.c_uint32,
    "size_t": ctypes.c_size_t, "hipStream_t": ctypes.c_void_p, "void": None,
}


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [(argtype, argname), ...])} for every declaration in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"^\s*(const char\*|size_t|int)\s+(tatt_\w+)\s*\(([^)]*)\)\s*;", txt, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    parsed.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    parsed.append((_CT[ty.replace("const ", "").strip()], nm))
        res = {"const char*": ctypes.c_char_p, "size_t": ctypes.c_size_t, "int": ctypes.c_int}[ret]
        decls[name] = (res, parsed, ret)
    return decls


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C tatt_amd/csrc` (or __graft_entry__.build()); "
                "tatt_amd has no non-HIP fallback.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        for name, (res, args, _) in self.decls.items():
            fn = getattr(self.cdll, name)       # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = [a for a, _ in args]
            takes_stream = bool(args) and args[-1][1] == "stream"
            setattr(self, name[len("tatt_"):], self._wrap(name, fn, res, takes_stream, len(args)))

    def _wrap(self, name, fn, res, takes_stream, nargs):
        is_status = res is ctypes.c_int and name not in ("tatt_version",)

        def call(*args):
            conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
            if takes_stream and len(conv) == nargs - 1:
                conv.append(torch.cuda.current_stream().cuda_stream)
            rc = fn(*conv)
            if is_status and rc != 0:
                raise RuntimeError(f"{name} failed ({rc}): {self.cdll.tatt_last_error().decode()}")
            return rc

        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
