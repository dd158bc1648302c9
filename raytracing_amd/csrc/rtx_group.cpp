// rtx_group_*: one frame over several GPUs from one process, behind the C ABI (include/rtx.h).
//
// The reference spreads the rows of a frame over OpenMP threads (src/Scene.cpp:214-215, thread count from
// `-threads`, :232).  Here rank r of G renders rows r, r + G, ... on its own GPU (rtx_frame.row_first /
// row_step) into a compact device buffer; the buffers meet on rank 0's GPU in one exchange -- RCCL
// send/recv pairs inside one group call, point-to-point xGMI transfers into the root, no ring -- and a
// de-interleave kernel puts the rows in place.  One host thread drives all devices: every step is an
// asynchronous enqueue.
//
// Streams.  Every rank renders on a stream of its own; every GPU has one more stream for the exchange (its
// sends, and on the root the receives and the de-interleave).  All buffers exist twice and frames alternate
// between the two sets, so the exchange of frame k runs under the render of frame k + 1: rtx_group_render_device
// returns after the enqueue with the frame left in the root's memory, and a buffer is only written again (two
// frames later) after the events that mark its last reader.  Nothing is allocated after the first frame of a size.
//
// Built on the public entry points only (rtx_render_device, rtx_deinterleave_rows_device, rtx_get_stats).
// RCCL is loaded at run time, and only when the group spans more than one GPU: a process that already holds
// an RCCL (PyTorch ships its own) must not get a second one, and a single-GPU user needs none.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "host_copy.hpp"

namespace {

thread_local std::string g_group_err;
int groupFail(int code, const std::string &msg) {
    g_group_err = msg;
    return code;
}

// the handful of RCCL entry points the exchange needs (signatures as in rccl/rccl.h)
typedef void *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &why) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            why = "librccl lacks an expected symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclUint8 = 1;  // ncclUint8

}  // namespace

struct rtx_group {
    int n = 0;
    std::vector<int> device;          // GPU of rank r
    std::vector<rtx_ctx *> ctx;       // its context
    std::vector<hipStream_t> stream;  // the stream it renders on
    // the exchange: one RCCL rank per distinct GPU; comm_of[r] = index of rank r's GPU in comm_devices
    std::vector<int> comm_devices;
    std::vector<ncclComm_t> comms;
    std::vector<int> comm_of;
    std::vector<hipStream_t> xstream;  // per distinct GPU: the stream its sends / receives (and the root's de-interleave) go to
    bool via_rccl_always = false;      // RTX_GROUP_FORCE_RCCL: ranks that share the root's GPU send to it as well
    // Two sets of buffers, used by alternate frames.  Events per rank and set: `rendered` (the rank's rows of the
    // set's frame are complete), `sent` (they have left the rank's own buffer); per set: `placed` (the root has
    // de-interleaved the set's gathered rows -- every reader of them is through).
    int flip = 0;
    std::vector<unsigned char *> local[2];  // rank r's rows when they have to travel (on its GPU)
    std::vector<size_t> local_bytes[2];
    std::vector<hipEvent_t> rendered[2], sent[2];
    hipEvent_t placed[2] = {nullptr, nullptr};
    unsigned char *gathered[2] = {nullptr, nullptr}, *frame[2] = {nullptr, nullptr};  // on the root's GPU
    size_t gathered_bytes[2] = {0, 0}, frame_bytes[2] = {0, 0};
    rtx::HostStage stage;  // rtx_group_render's way to the caller's host buffer
};

#define GROUP_HIP(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return groupFail(RTX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

// everything the group has enqueued, on every stream
void drain(rtx_group *g) {
    for (int r = 0; r < (int)g->stream.size(); ++r) {
        (void)hipSetDevice(g->device[r]);
        if (g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
    }
    for (size_t k = 0; k < g->xstream.size(); ++k) {
        (void)hipSetDevice(g->comm_devices[k]);
        if (g->xstream[k]) (void)hipStreamSynchronize(g->xstream[k]);
    }
}

// A buffer that has to grow is replaced; work that may still use the old one is waited for first.  (Only the
// first frame of a larger size gets here.)
int growBuffer(rtx_group *g, unsigned char **p, size_t *have, size_t need) {
    if (need <= *have) return RTX_OK;
    drain(g);
    (void)hipFree(*p);
    *p = nullptr, *have = 0;
    GROUP_HIP(hipMalloc((void **)p, need));
    *have = need;
    return RTX_OK;
}

// the part of rtx_group_render_device that can fail after work has been enqueued
int enqueueFrame(rtx_group *g, const rtx_camera *cam, int width, int height, int depth, int b, bool *group_open) {
    const int G = g->n, root_dev = g->device[0];
    const size_t row_bytes = (size_t)width * 3;
    const int rows_per_rank = (height + G - 1) / G;
    const size_t block = (size_t)rows_per_rank * row_bytes;
    hipStream_t root_x = g->xstream[g->comm_of[0]];
    // ---- every rank renders its rows: in place into its block on the root's GPU when it shares that GPU,
    // into a buffer of its own otherwise.  The set's buffers were last used two frames ago. ----
    std::vector<bool> travels(G, false);
    for (int r = 0; r < G; ++r) {
        travels[r] = r != 0 && (g->device[r] != root_dev || g->via_rccl_always);
        GROUP_HIP(hipSetDevice(g->device[r]));
        unsigned char *dst = travels[r] ? g->local[b][r] : g->gathered[b] + (size_t)r * block;
        GROUP_HIP(hipStreamWaitEvent(g->stream[r], travels[r] ? g->sent[b][r] : g->placed[b], 0));
        const rtx_frame fr = {width, height, depth, r, G};
        if (int e = rtx_render_device(g->ctx[r], cam, &fr, dst, g->stream[r]))
            return groupFail(e, std::string("rank ") + std::to_string(r) + ": " + rtx_last_error());
        GROUP_HIP(hipEventRecord(g->rendered[b][r], g->stream[r]));
    }
    // ---- the exchange, on the GPUs' exchange streams: one group of send / recv pairs, every transfer straight
    // into the root ----
    bool any = false;
    for (int r = 0; r < G; ++r) {
        any = any || travels[r];
        hipStream_t reader = travels[r] ? g->xstream[g->comm_of[r]] : root_x;  // whoever touches rank r's rows next
        GROUP_HIP(hipSetDevice(travels[r] ? g->device[r] : root_dev));
        GROUP_HIP(hipStreamWaitEvent(reader, g->rendered[b][r], 0));
    }
    if (any) {
        if (int rc = g_rccl.GroupStart()) return groupFail(RTX_E_DEVICE, std::string("ncclGroupStart: ") + g_rccl.GetErrorString(rc));
        *group_open = true;
        for (int r = 1; r < G; ++r) {
            if (!travels[r]) continue;
            const rtx_frame fr = {width, height, depth, r, G};
            const size_t bytes = (size_t)rtx_frame_rows(&fr) * row_bytes;
            if (bytes == 0) continue;
            int rc = g_rccl.Send(g->local[b][r], bytes, kNcclUint8, g->comm_of[0], g->comms[g->comm_of[r]], g->xstream[g->comm_of[r]]);
            if (rc == 0) rc = g_rccl.Recv(g->gathered[b] + (size_t)r * block, bytes, kNcclUint8, g->comm_of[r], g->comms[g->comm_of[0]], root_x);
            if (rc != 0) return groupFail(RTX_E_DEVICE, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(rc));
        }
        *group_open = false;
        if (int rc = g_rccl.GroupEnd()) return groupFail(RTX_E_DEVICE, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(rc));
        for (int r = 1; r < G; ++r) {
            if (!travels[r]) continue;
            GROUP_HIP(hipSetDevice(g->device[r]));
            GROUP_HIP(hipEventRecord(g->sent[b][r], g->xstream[g->comm_of[r]]));
        }
    }
    GROUP_HIP(hipSetDevice(root_dev));
    if (G > 1)
        if (int e = rtx_deinterleave_rows_device(g->ctx[0], g->gathered[b], G, rows_per_rank, width, height, g->frame[b], root_x))
            return groupFail(e, rtx_last_error());
    GROUP_HIP(hipEventRecord(g->placed[b], root_x));
    return RTX_OK;
}

}  // namespace

extern "C" {

const char *rtx_group_last_error(void) { return g_group_err.c_str(); }

int rtx_group_create(rtx_group **out, const int *device_ids, int n_devices) {
    if (!out) return groupFail(RTX_E_INVALID, "rtx_group_create: out is null");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return groupFail(RTX_E_INVALID, "rtx_group_create: 1 to 64 ranks");
    rtx_group *g = new rtx_group();
    g->n = n_devices;
    g->via_rccl_always = getenv("RTX_GROUP_FORCE_RCCL") != nullptr;
    auto failed = [&](int code, const std::string &msg) {
        rtx_group_destroy(g);
        return groupFail(code, msg);
    };
    for (int r = 0; r < n_devices; ++r) {
        rtx_ctx *c = nullptr;
        if (int e = rtx_create(&c, device_ids[r])) return failed(e, std::string("rtx_group_create: rank ") + std::to_string(r) + ": " + rtx_last_error());
        // (every handle goes into the group before the next call can fail: rtx_group_destroy frees what is there)
        g->device.push_back(device_ids[r]);
        g->ctx.push_back(c);
        g->stream.push_back(nullptr);
        for (int b = 0; b < 2; ++b) {
            g->local[b].push_back(nullptr), g->local_bytes[b].push_back(0);
            g->rendered[b].push_back(nullptr), g->sent[b].push_back(nullptr);
        }
        if (hipSetDevice(device_ids[r]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess)
            return failed(RTX_E_DEVICE, "rtx_group_create: stream");
        for (int b = 0; b < 2; ++b)
            if (hipEventCreateWithFlags(&g->rendered[b][r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->sent[b][r], hipEventDisableTiming) != hipSuccess)
                return failed(RTX_E_DEVICE, "rtx_group_create: event");
        int k = 0;
        while (k < (int)g->comm_devices.size() && g->comm_devices[k] != device_ids[r]) ++k;
        if (k == (int)g->comm_devices.size()) {
            g->comm_devices.push_back(device_ids[r]);
            g->xstream.push_back(nullptr);
            if (hipStreamCreateWithFlags(&g->xstream[k], hipStreamNonBlocking) != hipSuccess) return failed(RTX_E_DEVICE, "rtx_group_create: stream");
        }
        g->comm_of.push_back(k);
    }
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipEventCreateWithFlags(&g->placed[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->placed[1], hipEventDisableTiming) != hipSuccess)
        return failed(RTX_E_DEVICE, "rtx_group_create: event");
    if (g->comm_devices.size() > 1 || g->via_rccl_always) {
        std::string why;
        if (!g_rccl.load(why)) return failed(RTX_E_DEVICE, "rtx_group_create: " + why);
        g->comms.resize(g->comm_devices.size());
        int rc = g_rccl.CommInitAll(g->comms.data(), (int)g->comm_devices.size(), g->comm_devices.data());
        if (rc != 0) {
            g->comms.clear();
            return failed(RTX_E_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc));
        }
    }
    *out = g;
    return RTX_OK;
}

void rtx_group_destroy(rtx_group *g) {
    if (!g) return;
    drain(g);
    for (ncclComm_t c : g->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        (void)hipSetDevice(g->device[r]);
        for (int b = 0; b < 2; ++b) {
            (void)hipFree(g->local[b][r]);
            if (g->rendered[b][r]) (void)hipEventDestroy(g->rendered[b][r]);
            if (g->sent[b][r]) (void)hipEventDestroy(g->sent[b][r]);
        }
        if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
        rtx_destroy(g->ctx[r]);
    }
    for (size_t k = 0; k < g->xstream.size(); ++k) {
        (void)hipSetDevice(g->comm_devices[k]);
        if (g->xstream[k]) (void)hipStreamDestroy(g->xstream[k]);
    }
    if (!g->device.empty()) {
        (void)hipSetDevice(g->device[0]);
        for (int b = 0; b < 2; ++b) {
            (void)hipFree(g->gathered[b]);
            (void)hipFree(g->frame[b]);
            if (g->placed[b]) (void)hipEventDestroy(g->placed[b]);
        }
        g->stage.release();
    }
    delete g;
}

int rtx_group_size(const rtx_group *g) { return g ? g->n : 0; }

rtx_ctx *rtx_group_ctx(rtx_group *g, int rank) { return g && rank >= 0 && rank < g->n ? g->ctx[rank] : nullptr; }

int rtx_group_commit(rtx_group *g) {
    if (!g) return groupFail(RTX_E_INVALID, "null group");
    for (int r = 0; r < g->n; ++r)
        if (int e = rtx_commit(g->ctx[r])) return groupFail(e, std::string("rank ") + std::to_string(r) + ": " + rtx_last_error());
    return RTX_OK;
}

int rtx_group_set_option(rtx_group *g, int option, uint32_t value) {
    if (!g) return groupFail(RTX_E_INVALID, "null group");
    for (int r = 0; r < g->n; ++r)
        if (int e = rtx_set_option(g->ctx[r], option, value)) return groupFail(e, std::string("rank ") + std::to_string(r) + ": " + rtx_last_error());
    if (option == RTX_OPT_PIN_OUTPUT) {
        g->stage.may_pin = value != 0;
        if (!g->stage.may_pin) g->stage.release();
    }
    return RTX_OK;
}

int rtx_group_render_device(rtx_group *g, const rtx_camera *cam, int width, int height, int depth, void **d_frame_out) {
    if (!g || !cam || !d_frame_out) return groupFail(RTX_E_INVALID, "null argument");
    *d_frame_out = nullptr;
    if (width <= 0 || height <= 0) return groupFail(RTX_E_INVALID, "bad frame size");
    const int G = g->n, root_dev = g->device[0];
    const size_t row_bytes = (size_t)width * 3;
    const size_t block = (size_t)((height + G - 1) / G) * row_bytes;
    const int b = g->flip;
    // ---- the set's buffers (only the first frame of a larger size allocates) ----
    GROUP_HIP(hipSetDevice(root_dev));
    if (int e = growBuffer(g, &g->gathered[b], &g->gathered_bytes[b], block * G)) return e;
    if (G > 1)
        if (int e = growBuffer(g, &g->frame[b], &g->frame_bytes[b], (size_t)height * row_bytes)) return e;
    for (int r = 1; r < G; ++r) {
        if (g->device[r] == root_dev && !g->via_rccl_always) continue;
        GROUP_HIP(hipSetDevice(g->device[r]));
        if (int e = growBuffer(g, &g->local[b][r], &g->local_bytes[b][r], block)) return e;
    }
    bool group_open = false;
    const int rc = enqueueFrame(g, cam, width, height, depth, b, &group_open);
    if (rc != RTX_OK) {
        // leave nothing half done behind: close the RCCL group call, wait for what was enqueued (the next call may
        // replace the buffers it uses)
        const std::string why = g_group_err;
        if (group_open) (void)g_rccl.GroupEnd();
        drain(g);
        return groupFail(rc, why);
    }
    g->flip ^= 1;
    *d_frame_out = G > 1 ? g->frame[b] : g->gathered[b];
    return RTX_OK;
}

void *rtx_group_root_stream(rtx_group *g) { return g && !g->xstream.empty() ? g->xstream[g->comm_of[0]] : nullptr; }

int rtx_group_wait(rtx_group *g) {
    if (!g) return groupFail(RTX_E_INVALID, "null group");
    for (int r = 0; r < g->n; ++r) {
        GROUP_HIP(hipSetDevice(g->device[r]));
        GROUP_HIP(hipStreamSynchronize(g->stream[r]));
    }
    for (size_t k = 0; k < g->xstream.size(); ++k) {
        GROUP_HIP(hipSetDevice(g->comm_devices[k]));
        GROUP_HIP(hipStreamSynchronize(g->xstream[k]));
    }
    return RTX_OK;
}

int rtx_group_render(rtx_group *g, const rtx_camera *cam, int width, int height, int depth, unsigned char *bgr_out, rtx_stats *stats) {
    if (!g || !cam) return groupFail(RTX_E_INVALID, "null argument");
    if (!bgr_out) return groupFail(RTX_E_INVALID, "null output buffer");
    void *d_frame = nullptr;
    if (int e = rtx_group_render_device(g, cam, width, height, depth, &d_frame)) return e;
    GROUP_HIP(hipSetDevice(g->device[0]));
    std::string why;
    if (!g->stage.toHost(bgr_out, d_frame, (size_t)height * width * 3, g->xstream[g->comm_of[0]], &why)) {
        drain(g);
        return groupFail(RTX_E_DEVICE, why);
    }
    if (int e = rtx_group_wait(g)) return e;  // (the senders are done once the root has received, but their streams may still be draining)
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int r = 0; r < g->n; ++r) {
            rtx_stats s;
            if (int e = rtx_get_stats(g->ctx[r], &s)) return groupFail(e, rtx_last_error());
            uint64_t *acc = reinterpret_cast<uint64_t *>(stats);
            const uint64_t *add = reinterpret_cast<const uint64_t *>(&s);
            for (size_t k = 0; k < offsetof(rtx_stats, kernel_ms) / sizeof(uint64_t); ++k) acc[k] += add[k];
            stats->kernel_ms = s.kernel_ms > stats->kernel_ms ? s.kernel_ms : stats->kernel_ms;
        }
    }
    return RTX_OK;
}

}  // extern "C"
