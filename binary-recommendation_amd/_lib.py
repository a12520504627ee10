"""ctypes binding of libbinrec_hip.so, generated from include/binrec.h.

The prototypes are parsed out of the header so the Python side can never drift from the
C-ABI.  There is NO CPU fallback: if the library is missing or fails to load, importing the
ops raises.
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "binrec.h")
LIB_PATH = os.environ.get("BR_LIB_PATH") or os.path.join(HERE, "libbinrec_hip.so")     # BR_LIB_PATH: a diagnostic build of the same ABI (tools/diag)

_SCALARS = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32, "float": ctypes.c_float,
    "double": ctypes.c_double, "brStream": ctypes.c_void_p,
}


def parse_header(path: str = HEADER) -> dict[str, tuple[object, list[object], list[str]]]:
    """-> {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"enum\s*\{.*?\}\s*;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int64_t|int)\s+(br[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else _SCALARS[ret.strip()]
        argtypes, argnames = [], []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.search(r"([A-Za-z_][A-Za-z0-9_]*)$", a).group(1)
                ty = a[: -len(nm)].strip()
                if "*" in ty:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[ty.replace("const", "").strip()])
                argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


class BinrecError(RuntimeError):
    pass


def parse_struct(name: str, path: str = HEADER):
    """ctypes field list of `typedef struct name {...} name;` in the header (pointers -> c_void_p)."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    m = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", src, flags=re.S)
    if not m:
        raise BinrecError(f"struct {name} not found in {path}")
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        if "*" in decl:
            for nm in re.findall(r"\*\s*([A-Za-z_][A-Za-z0-9_]*)", decl):
                fields.append((nm, ctypes.c_void_p))
        else:
            ty, names = decl.split(" ", 1)
            for nm in names.split(","):
                fields.append((nm.strip(), _SCALARS[ty]))
    return fields


def parse_enums(path: str = HEADER) -> dict:
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for body in re.findall(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
        for item in body.split(","):
            if "=" in item:
                k, v = item.split("=")
                out[k.strip()] = int(v.strip(), 0)
    return out


_DEBUG_SYNC = os.environ.get("BR_DEBUG_SYNC", "0") not in ("", "0")
_lib = None
_protos = None
_probe = None   # optional callable pair (before(name, args), after(name, args)) — bench.py's HIP-event timing


def set_probe(probe):
    """probe: object with .before(name, args) / .after(name, args), or None."""
    global _probe
    _probe = probe


class _Lib:
    """Attribute access returns the ctypes entry point, wrapped so a probe can bracket launches."""

    def __init__(self, cdll, protos):
        self._cdll = cdll
        for name in protos:
            setattr(self, name, self._wrap(name, getattr(cdll, name)))

    @staticmethod
    def _wrap(name, fn):
        if _DEBUG_SYNC:
            # BR_DEBUG_SYNC=1: every entry point is named on stderr before it runs and the device is synchronised behind it, so an
            # asynchronous GPU fault (which otherwise surfaces at some later, unrelated host sync) ends the process right after the line
            # that names the launch that caused it.  Diagnostic mode: serialises the host with the GPU.
            def call_sync(*args):
                import sys
                import torch
                sys.stderr.write(f"[binrec] {name}\n"); sys.stderr.flush()
                rc = fn(*args)
                if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize()
                return rc
            call_sync.__name__ = name
            return call_sync

        def call(*args):
            p = _probe
            if p is None:
                return fn(*args)
            p.before(name, args)
            rc = fn(*args)
            p.after(name, args)
            return rc
        call.__name__ = name
        return call


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BinrecError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
    # torch first: its wheel carries its own libamdhip64.so, and the process must end up with ONE HIP runtime.  Loaded after torch, our
    # library's NEEDED libamdhip64 resolves to the copy torch already mapped; loaded before it, the system copy under /opt/rocm comes in
    # as a second runtime and every HIP call from this library fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _names) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BinrecError(f"libbinrec_hip.so does not export {name} declared in include/binrec.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = _Lib(lib, _protos)
    return _lib


def prototypes():
    load()
    return _protos


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().brGetLastError()
        raise BinrecError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
