"""ctypes loader for ``libcss_hip.so`` (the C ABI declared in ``include/css_hip.h``).

The argument/return types of every entry point are parsed from the header, so the Python
bindings cannot drift from the C declarations.  There is NO fallback: if the shared library is
missing or an entry point returns an error code, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading
from typing import Dict, List

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("CSS_HIP_LIB") or os.path.join(_HERE, "csrc", "libcss_hip.so")   # (override: A/B builds of the library)
HEADER_PATH = os.path.join(_ROOT, "include", "css_hip.h")

F32, BF16 = 0, 1

ERRORS = {-1: "CSS_ERR_ARG (bad shape / alignment / unsupported configuration)", -2: "CSS_ERR_DTYPE",
          -3: "CSS_ERR_LAUNCH (HIP launch failed)", -4: "CSS_ERR_WORKSPACE"}


class CssHipError(RuntimeError):
    pass


_CTYPE = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "unsigned long long": ctypes.c_ulonglong, "css_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, tuple]:
    """name -> (restype, [argtypes], [argnames]) for every ``CSS_API`` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"CSS_API\s+([\w\s]+?)\s+(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                typ, an = mm.group(1).strip(), mm.group(2)
                argnames.append(an)
                if "*" in typ:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPE[typ.replace("const ", "").strip()])
        out[name] = (_CTYPE[ret], argtypes, argnames)
    return out


_lib = None
_sigs = None
_lock = threading.Lock()


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> ctypes.CDLL:
    global _lib, _sigs
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise CssHipError(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C css_amd/csrc`). css_amd has no CPU / eager fallback.")
                l = ctypes.CDLL(LIB_PATH)
                sigs = parse_header()
                for name, (ret, argtypes, _) in sigs.items():
                    fn = getattr(l, name)          # AttributeError here = header/library mismatch
                    fn.restype = ret
                    fn.argtypes = argtypes
                _sigs = sigs
                _lib = l
    return _lib


def signatures() -> Dict[str, tuple]:
    lib()
    return _sigs


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise CssHipError(f"unsupported compute dtype {dt}")


def _conv_arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def call(name: str, *args):
    """Call an ``int css_*`` entry point; tensors are passed as device pointers, ``None`` as NULL."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv_arg(a) for a in args])
    if rc != 0:
        raise CssHipError(f"{name} failed: {ERRORS.get(rc, rc)}")


def query(name: str, *args):
    """Entry points that return a value (sizes, counts) rather than a status."""
    return getattr(lib(), name)(*[_conv_arg(a) for a in args])


def dev_stream(t: torch.Tensor):
    """(device index, hipStream_t) to launch on for tensor ``t``.  Must be a GPU tensor."""
    if not t.is_cuda:
        raise CssHipError("css_amd ops need tensors on an MI355X (cuda/hip device); there is no CPU path")
    d = t.device.index
    return d, torch.cuda.current_stream(t.device).cuda_stream
