"""MI355X-native fork/join barrier engine: HIP kernels behind the C ABI of include/dj_engine.h,
plus the host mirror of the reference's request logic.  The compute path is
libdj_engine.so (gfx950); nothing in this package computes on the CPU."""
from . import capi  # noqa: F401

__all__ = ["capi", "engine", "store", "core"]
