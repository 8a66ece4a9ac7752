"""MI355X-native D-FINE hot path (train step + inference forward).

Host side mirrors the reference's `src.d_fine` construction API;
the arithmetic lives in `csrc/` as hand-written HIP for gfx950 behind a C ABI
(`include/dfine_hip.h`), bound with ctypes in `custom_d_fine_amd.hip`.
"""
__version__ = "0.1.0"
