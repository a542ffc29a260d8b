"""ctypes binding of libdeepsee_hip.so (the C ABI declared in include/deepsee_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
importing/using it raises.  PyTorch only supplies device memory (``tensor.data_ptr()``) and the
current HIP stream handle.
"""
import ctypes as C
import os

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


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DseeError("libdeepsee_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                            % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.dsee_last_error.restype = C.c_char_p
        for fn in ("conv2d_wgrad_workspace", "conv2d_wgrad_table_workspace", "norm_workspace", "channel_dot_workspace",
                   "onehot_conv3x3_wgrad_workspace", "label_segsum_workspace", "loss_workspace",
                   "wino43_wgrad_workspace", "wino43_wgrad_table_workspace", "conv3x3_thin_wgrad_workspace"):
            getattr(_lib, "dsee_" + fn).restype = C.c_size_t
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
    if isinstance(v, float):
        return C.c_float(v)
    if isinstance(v, bool):
        return C.c_int(int(v))
    if isinstance(v, int):
        return C.c_int(v)
    return v


def call(name, *args):
    """Invoke ``dsee_<name>`` on the current stream (appended as the last argument)."""
    fn = getattr(lib(), "dsee_" + name)
    rc = fn(*[_conv(a) for a in args], stream())
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
