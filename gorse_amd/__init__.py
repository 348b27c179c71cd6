"""gorse_amd -- MI355X-native CF training + exact top-k hot path for Gorse.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/gorse_hip.h),
capi.py (ctypes binding), host mirror of the reference's cf / ann interfaces, synth.py
(synthetic stand-ins for the reference's datasets).  No CPU fallback exists anywhere in
this package: the HIP library must be built and a gfx950 device must be present.
"""
__all__ = ["capi", "synth"]
