"""ctypes loader for libcq_halo2.so.  Fails loudly if the HIP extension is missing: there is
no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcq_halo2.so")
HEADER = os.path.join(HERE, "..", "include", "cq_halo2.h")

_lib = None


class CqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cq_halo2 error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m sha2_on_cq_halo2_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _header_text() -> str:
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)      # comments
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)        # preprocessor lines
    return src


def header_symbols():
    """Every function name declared in include/cq_halo2.h."""
    return sorted(p.name for p in header_prototypes())


class Prototype:
    """One `extern "C"` prototype of include/cq_halo2.h: C return type and parameter types as written there."""

    def __init__(self, name, ret, params):
        self.name, self.ret, self.params = name, ret, params

    def __repr__(self):
        return f"{self.ret} {self.name}({', '.join(self.params)})"


def header_prototypes():
    """Parses the header (the ABI's single source of truth): the ctypes signatures below are DERIVED from it, so the
    Python binding cannot drift from the C declarations -- and the compiler holds the definitions in csrc/capi*.hip to
    the same declarations (they include the header under `extern "C"`)."""
    src = _header_text()
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)  # struct bodies
    src = re.sub(r"typedef[^;{]*;", "", src)                                       # opaque / function-pointer typedefs
    src = src.replace('extern "C" {', "")
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(cq_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [re.sub(r"\s+", " ", p.strip()) for p in params.split(",")]
        out.append(Prototype(name, re.sub(r"\s+", " ", ret), plist))
    return out


def header_fnptr_typedefs():
    """Names of the function-pointer typedefs (callbacks crossing the ABI)."""
    return set(re.findall(r"typedef\s+[\w\s\*]+?\(\s*\*\s*(cq_\w+)\s*\)", _header_text()))


_SCALARS = {"int": C.c_int, "uint32_t": C.c_uint32, "int32_t": C.c_int32, "uint64_t": C.c_uint64, "int64_t": C.c_int64,
            "size_t": C.c_size_t, "double": C.c_double, "uint8_t": C.c_uint8}


def ctype_of_param(decl: str, fnptrs) -> object:
    """ctypes class of one C parameter declaration: every pointer, array parameter and callback is a machine address
    (c_void_p accepts ints, None, byref(), arrays and CFUNCTYPE instances); scalars keep their width."""
    if "*" in decl or "[" in decl:
        return C.c_void_p
    words = [w for w in decl.replace("const", " ").split() if w]
    ty = words[0] if len(words) == 1 else " ".join(words[:-1])  # drop the parameter name
    if ty in fnptrs:
        return C.c_void_p
    if ty not in _SCALARS:
        raise TypeError(f"cq_halo2.h: parameter type not understood: {decl!r}")
    return _SCALARS[ty]


def ctype_of_return(ret: str):
    ret = ret.replace("const", " ").strip()
    ret = re.sub(r"\s+", " ", ret).replace(" *", "*")
    if ret == "void":
        return None
    if ret == "char*":
        return C.c_char_p
    if ret.endswith("*"):
        return C.c_void_p
    if ret not in _SCALARS:
        raise TypeError(f"cq_halo2.h: return type not understood: {ret!r}")
    return _SCALARS[ret]


def signatures():
    """{name: (restype, [argtypes])} for every prototype of the header."""
    fnptrs = header_fnptr_typedefs()
    return {p.name: (ctype_of_return(p.ret), [ctype_of_param(d, fnptrs) for d in p.params]) for p in header_prototypes()}


def _declare(lib):
    for name, (restype, argtypes) in signatures().items():
        fn = getattr(lib, name)  # AttributeError: the header declares a symbol the library does not export
        fn.restype = restype
        fn.argtypes = argtypes
