"""super4pcs_amd: MI355X-native (gfx950) hot path of Super4PCS global registration.

Layout (only what the path needs):
  csrc/        hand-written HIP kernels + the C ABI (include/s4p_capi.h) + host structures
  capi.py      ctypes binding of the C ABI (plumbing)
  matcher.py   Python mirror of the reference matcher interface on top of the C++ engine
  datasets.py  synthetic cloud generators of the BASELINE.json configs
  build.py     hipcc build of lib/libsuper4pcs_amd.so for gfx950
"""
__all__ = ["capi", "datasets", "build"]
