"""ctypes binding of libdeepsee_hip.so (the C ABI declared in include/deepsee_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
importing/using it raises.  PyTorch only supplies device memory (``tensor.data_ptr()``) and the
current HIP stream handle.
"""
import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSEE_LIB") or os.path.join(_HERE, "libdeepsee_hip.so")  # DSEE_LIB: kernel experiments only

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_MASK = 0, 1, 2, 3, 4


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "mul", "off", "kdir", "dshift", "ups", "korder")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class DseeError(RuntimeError):
    pass


_lib = None
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "deepsee_hip.h")

_CTYPES = {"int": C.c_int, "long": C.c_long, "float": C.c_float, "size_t": C.c_size_t, "uint64_t": C.c_uint64,
           "int32_t": C.c_int32, "int64_t": C.c_int64, "hipStream_t": C.c_void_p, "const char*": C.c_char_p}


def header_prototypes(path=HEADER_PATH):
    """{name: (restype, [argtypes])} parsed from the declarations of include/deepsee_hip.h, so that every call is
    checked against the header: arity, and the width of every scalar (int / long / size_t / uint64_t / float)."""
    hdr = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(dsee_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr):
        args = []
        for a in m.group(3).split(","):
            a = a.strip()
            if a == "void":
                continue
            ty = a.rsplit(None, 1)[0] if not a.endswith("*") else a      # drop the parameter name
            ty = ty.strip()
            if ty.endswith("*"):
                args.append(C.c_void_p)                                  # every pointer is a raw (device) address
            else:
                args.append(_CTYPES[ty.replace("const ", "")])
        protos[m.group(2)] = (_CTYPES[m.group(1)], args)
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DseeError("libdeepsee_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                            % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in header_prototypes().items():
            fn = getattr(_lib, name)            # AttributeError: the library lacks a symbol the header declares
            fn.restype, fn.argtypes = res, args
    return _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _conv(v):
    if isinstance(v, torch.Tensor) or v is None:
        return ptr(v)
    if isinstance(v, (C._SimpleCData,)):
        return v.value                 # explicit C.c_long(...) etc. from older call sites: the argtypes decide the width
    if isinstance(v, bool):
        return int(v)
    return v


CALLS = 0      # C-ABI entry points invoked by this process (bench.py reads the difference over one eager step)


def call(name, *args):
    """Invoke ``dsee_<name>`` on the current stream (appended as the last argument).  Arguments are converted by the
    argtypes parsed from include/deepsee_hip.h: a wrong arity or a float where the header says int raises."""
    global CALLS
    CALLS += 1
    fn = getattr(lib(), "dsee_" + name)
    if len(args) + 1 != len(fn.argtypes):
        raise DseeError("dsee_%s takes %d arguments + stream (include/deepsee_hip.h), got %d"
                        % (name, len(fn.argtypes) - 1, len(args)))
    try:
        rc = fn(*[_conv(a) for a in args], stream())
    except (C.ArgumentError, TypeError) as e:
        raise DseeError("dsee_%s: call does not match include/deepsee_hip.h: %s" % (name, e)) from None
    if rc != 0:
        raise DseeError("dsee_%s failed (%d): %s" % (name, rc, lib().dsee_last_error().decode()))


def pad4(c):
    return (c + 3) // 4 * 4


def kpad(kh, kw, cin_s):
    return (kh * kw * cin_s + 31) // 32 * 32


def wrows(cout):
    return (cout + 127) // 128 * 128


def geom_fwd(n, hi, wi, cin_s, cout_s, k, stride, pad, ups=0):
    hl, wl = hi << ups, wi << ups
    ho = (hl + 2 * pad - k) // stride + 1
    wo = (wl + 2 * pad - k) // stride + 1
    korder = 1 if (ups == 0 and cin_s % 32 == 0 and k * k <= 32) else 0
    return ConvGeom(n, hi, wi, cin_s, ho, wo, cout_s, k, k, stride, -pad, 1, 0, ups, korder)


def geom_dgrad(fwd):
    """Data-gradient geometry of a forward conv (at the logical, i.e. possibly upsampled, input resolution)."""
    stride = fwd.mul
    dshift = {1: 0, 2: 1, 4: 2}[stride]
    korder = 1 if (dshift == 0 and fwd.Cout % 32 == 0 and fwd.KH * fwd.KW <= 32) else 0
    return ConvGeom(fwd.N, fwd.Ho, fwd.Wo, fwd.Cout, fwd.Hi << fwd.ups, fwd.Wi << fwd.ups, fwd.Cin, fwd.KH, fwd.KW,
                    1, -fwd.off, -1, dshift, 0, korder)
