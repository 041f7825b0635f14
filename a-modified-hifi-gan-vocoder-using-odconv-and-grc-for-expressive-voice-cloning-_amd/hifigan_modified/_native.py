"""ctypes binding of libmi355x_vocoder.so (C ABI declared in include/mi355x_vocoder.h).

There is NO fallback: if the shared library is missing or a kernel launch fails this module raises.
The library is built in-tree (``make -C csrc`` or ``__graft_entry__.build()``), next to this package.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MV_LIB") or os.path.join(_PKG_DIR, "libmi355x_vocoder.so")   # MV_LIB: a debug build (e.g. -DMV_OD_TIMING)
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "mi355x_vocoder.h")

MV_F32, MV_BF16, MV_F16 = 0, 1, 2
MV_F32_W16 = 3        # mv_mrf_* operand mode: fp32 storage, hi + lo f16 activations x single f16 weights
MV_F32_W16P = 4       # the same with the chain's INPUT already in pair rows (mv_odconv_cl_fwd_pair wrote it)
ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SILU = 0, 1, 2, 3

_ERR = {-1: "MV_ERR_ARG (shape/size contract violated)", -2: "MV_ERR_DTYPE", -3: "MV_ERR_UNSUPPORTED"}

_CTYPE = {"int": c_int, "long": c_long, "float": c_float, "size_t": ctypes.c_size_t}


class NativeLibraryError(RuntimeError):
    pass


def _parse_header(path):
    """{name: [ctypes argtypes]} for every `int mv_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct[^{]*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(mv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPE[base])
        protos[name] = (ret, argtypes)
    return protos


_lib = None
_protos = None


def declared_symbols():
    global _protos
    if _protos is None:
        _protos = _parse_header(HEADER_PATH)
    return _protos


def lib():
    """Load (once) and return the ctypes library with argtypes set from the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_PKG_DIR, 'csrc')}` "
            "(or __graft_entry__.build()). The MI355X vocoder path has no CPU/PyTorch fallback.")
    l = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in declared_symbols().items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name} declared in {HEADER_PATH}") from e
        fn.argtypes = argtypes
        fn.restype = {"int": c_int, "size_t": ctypes.c_size_t}.get(ret, c_char_p)
    _lib = l
    return l


def check(rc, name):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{name}: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{name}: HIP launch failed with hipError_t={rc}")


_TRACE = bool(os.environ.get("MV_TRACE"))   # debugging aid: print every entry point and synchronise after it
_CAPDBG = bool(os.environ.get("MV_CAPTURE_DEBUG"))   # debugging aid: name the first entry point after which a stream capture is dead


def call(name, *args):
    fn = getattr(lib(), name)
    if _TRACE:
        import sys
        import torch
        print(f"[mv] {name}", file=sys.stderr, flush=True)
        check(fn(*args), name)
        torch.cuda.synchronize()
        return
    check(fn(*args), name)
    if _CAPDBG:
        import sys
        import torch
        try:
            torch.cuda.is_current_stream_capturing()
        except Exception as e:      # noqa: BLE001
            print(f"[mv] stream capture invalidated at or before {name}: {e}", file=sys.stderr, flush=True)
            raise


class MrfParams(ctypes.Structure):
    """mv_mrf_params (include/mi355x_vocoder.h)."""
    _fields_ = ([(n, c_void_p * 3) for n in ("conv_w", "conv_b", "lora_A", "lora_B", "lora_scaling", "proj_w",
                                            "proj_b", "norm_w", "norm_b", "res_w", "res_b")]
                + [(n, c_void_p) for n in ("fusion_w", "fusion_b", "norm2_w", "norm2_b")])
