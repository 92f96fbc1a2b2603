"""orb_slam_amd — MI355X-native ORB front-end (ORBextractor + ORBmatcher hot path of raulmur/ORB_SLAM).

  csrc/    hand-written HIP kernels (gfx950) + the C ABI of include/orbx.h  -> liborbx.so
  cpp/     C++ shim keeping the reference's ORBextractor / ORBmatcher class surface
  capi.py  ctypes binding of the C ABI (tests, bench)
  synth.py exact-integer synthetic frame generator (tests, bench)
"""
from . import capi, synth  # noqa: F401
from .capi import ORBextractor, OrbxError, match_top2, hamming256  # noqa: F401
