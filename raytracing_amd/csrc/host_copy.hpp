// The way of a finished frame from device memory into the caller's host buffer (rtx_render, rtx_group_render; the
// counterpart of the reference handing its pixel store to image.save_image, src/Scene.cpp:227).
//
// A copy into pageable memory goes through the runtime's own staging at about 13 GB/s -- 2 ms for a 3840x2160 frame,
// three times what the frame takes to render.  Into page-locked memory the same bytes move at the speed of the link.
// So: a destination that is already page-locked (hipHostMalloc, hipHostRegister, rtx_host_alloc, a pinned torch
// tensor) is written directly; a pageable one is page-locked here once, and the registration kept while the caller
// keeps passing the same buffer, if the caller has said that this is allowed (RTX_OPT_PIN_OUTPUT: the buffer must then
// stay allocated until another one is passed or the context is destroyed); otherwise the plain copy.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace rtx {

struct HostStage {
    bool may_pin = false;             // RTX_OPT_PIN_OUTPUT
    void *pinned = nullptr;           // the caller's buffer this object has page-locked
    size_t pinned_bytes = 0;

    void release() {
        if (pinned) (void)hipHostUnregister(pinned);
        pinned = nullptr, pinned_bytes = 0;
    }

    static bool pageLocked(const void *p) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();  // (plain host memory is an "invalid value" to this query)
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    }

    // Copies on `stream` and waits for it.  The right device is current.
    bool toHost(void *dst, const void *d_src, size_t bytes, hipStream_t stream, std::string *why) {
        hipError_t e = hipSuccess;
        if (bytes) {
            const bool covered = pinned && dst == pinned && bytes <= pinned_bytes;
            if (!covered && may_pin && !pageLocked(dst)) {
                release();
                if (hipHostRegister(dst, bytes, hipHostRegisterDefault) == hipSuccess) pinned = dst, pinned_bytes = bytes;
                else (void)hipGetLastError();  // (the plain copy still works)
            }
            e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            if (why) *why = std::string("copy of the frame to host memory: ") + hipGetErrorString(e);
            return false;
        }
        return true;
    }
};

}  // namespace rtx
