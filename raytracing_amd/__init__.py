"""MI355X-native trace loop for the yukatan1701/RayTracing renderer.

The product is the shared library raytracing_amd/librtx_hip.so (HIP kernels for gfx950 behind the C
ABI of include/rtx.h, plus the C++ host layer of include/rt_host.h) and the `rt` executable.  This
package is the Python plumbing around it: ctypes bindings, and the torch.distributed glue that
shards a frame over ranks.  There is no CPU fallback: without the library, loading fails.
"""
from .rtx import Backend, Camera, Frame, Stats, RtxError, hip_backend, HIP_LIB_PATH, ASSETS_DIR  # noqa: F401

__all__ = ["Backend", "Camera", "Frame", "Stats", "RtxError", "hip_backend", "HIP_LIB_PATH", "ASSETS_DIR"]
