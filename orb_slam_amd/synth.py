"""ctypes front for libsynthframes.so — the exact-integer synthetic frame / descriptor generator
(bench and test input only; see csrc/synth_frames.c for the definition of the families)."""
import ctypes
import os
import numpy as np

NOISE, BLOCKS, FLAT, LOWTEX = 0, 1, 2, 3
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsynthframes.so")
        if not os.path.exists(path):
            raise RuntimeError("libsynthframes.so missing: run `make` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        L.synth_frames.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.synth_frames.restype = None
        L.synth_descriptors.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
        L.synth_descriptors.restype = None
        _LIB = L
    return _LIB


def frames(w, h, family, first_index, n):
    """-> uint8 array [n, h, w], frame i generated with index first_index+i."""
    out = np.empty((n, h, w), dtype=np.uint8)
    _lib().synth_frames(out.ctypes.data, w, h, family, first_index, n)
    return out


def frame(w, h, family, index):
    return frames(w, h, family, index, 1)[0]


def descriptors(n, seed):
    """-> uint8 array [n, 32] of PRNG 256-bit descriptors."""
    out = np.empty((n, 32), dtype=np.uint8)
    _lib().synth_descriptors(out.ctypes.data, n, seed)
    return out
